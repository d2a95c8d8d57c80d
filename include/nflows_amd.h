/*
 * nflows_amd.h -- C ABI of libnflows_amd.so, the MI355X (gfx950) implementation of the
 * bayesiains/nflows coupling-layer hot path (forward / inverse + log|det J|).
 *
 * The reference is pure Python and has no FFI; the "interface" each entry point replaces is the
 * Python functional / method cited next to it (paths relative to the reference repo).  The
 * reference-side binding a maintainer would add (a ctypes stub inside nflows/transforms) is
 * shown in INTEGRATION.md; nflows_amd/_native.py is that same binding used by this repo's
 * drop-in `nflows_amd.transforms` classes.
 *
 * Conventions
 *   - every pointer except `spec` is a DEVICE pointer (HIP, same device as `stream`);
 *     row-major, densely packed unless a stride argument says otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream);
 *     all entry points are asynchronous: they enqueue kernels and return;
 *   - inputs are borrowed and never written; outputs are caller-allocated;
 *   - the return value is NFA_OK or an NFA_ERR_* code (argument errors are detected before any
 *     launch; nothing is enqueued on error);
 *   - data-dependent errors (the reference's InputOutsideDomain / discriminant assertion) are
 *     OR-ed into the caller-provided device word `status` (may be NULL = not recorded); the
 *     caller decides when to read it back.  Kernels never clear it.
 *   - no global mutable state on the data path: entry points are re-entrant across host threads
 *     and streams (the optional nfa_profile_* measurement aid at the end is the one exception).
 */
#ifndef NFLOWS_AMD_H
#define NFLOWS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFA_ABI_VERSION 14 /* bumped whenever a packed layout, a flag set or an entry point changes (round 3: 3 .. 7; round 4: 8,
                              9: whole-layer kernels for 2 .. 16 bins, nfa_resnet_backward_f32, W_f^T in K14's backward stream;
                              round 5: 10: `bin_idx` outputs of the spline kernels, nfa_searchsorted_f32; 11: NFA_FLAG_RESIDUAL_BLOCKS;
                              round 6: 12: nfa_rqs_flow_resnet_f16x3_f32 (K8x), the *_logits_f32 diagnostic entries,
                              NFA_FLAG_ALL_PRODUCTS, nfa_weights_checksum_*) */

/* return codes */
#define NFA_OK 0
#define NFA_ERR_INVALID_ARGUMENT 1 /* NULL pointer, negative size, bad enum */
#define NFA_ERR_UNSUPPORTED 2      /* valid request this build has no kernel for */
#define NFA_ERR_MIN_BIN_WIDTH 3    /* min_bin_width * num_bins > 1  (rational_quadratic.py:86-87) */
#define NFA_ERR_MIN_BIN_HEIGHT 4   /* min_bin_height * num_bins > 1 (rational_quadratic.py:88-89) */
#define NFA_ERR_HIP 5              /* a HIP runtime call failed; see nfa_last_hip_error() */

/* bits OR-ed into *status by the kernels */
#define NFA_STATUS_OUTSIDE_DOMAIN 1   /* transforms/base.py:16 InputOutsideDomain; rational_quadratic.py:81-82 */
#define NFA_STATUS_NEG_DISCRIMINANT 2 /* rational_quadratic.py:142 assert (discriminant >= 0).all() */
#define NFA_STATUS_BAD_INDEX 4        /* a transform_idx / perm entry outside [0, features) */

/* `flags` argument of the coupling entry points */
#define NFA_FLAG_INVERSE 1                /* inverse pass (coupling.py:102-130) */
#define NFA_FLAG_ACCUMULATE_LOGABSDET 2   /* logabsdet[b] += layer sum: the `total_logabsdet +=`
                                             of CompositeTransform._cascade (base.py:48-51) folded
                                             into the layer; logabsdet must then hold the running
                                             total on entry */
#define NFA_FLAG_WEIGHTS_BF16X3 4         /* nfa_rqs_coupling_fused_linear_f32 only: weight_packed
                                             holds split-bf16 triples (layout below); the GEMM
                                             then runs on the bf16 matrix pipe, fp32-accurate */

#define NFA_FLAG_LOGITS_LOG2E 8           /* nfa_rqs_coupling_resnet_f32 only: the width / height
                                             rows of the packed final layer carry an extra factor
                                             log2(e), the kernel takes 2^x of their differences */

#define NFA_FLAG_STANDARD_NORMAL_LOG_PROB 16 /* nfa_rqs_flow_resnet_*: the run of layers ends a flow whose base
                                             density is the standard normal (flows/base.py:42-49 +
                                             distributions/normal.py:27-33): the `logabsdet` output
                                             receives log_prob = (-0.5 sum_j z_j^2 - 0.5 D log 2 pi) +
                                             logabsdet, computed from the row tile before it leaves
                                             the chip.  Forward pass only */
#define NFA_FLAG_SKIP_OUTPUTS 32          /* with NFA_FLAG_STANDARD_NORMAL_LOG_PROB: `outputs` (z) is not
                                             written (may be null): Flow.log_prob never looks at it */

#define NFA_FLAG_RESIDUAL_BLOCKS 64        /* nfa_affine_flow_mlp_f32 only (ABI 11): the conditioner is a ResidualNet (nn/nets/resnet.py:55-100,
                                           * SimpleRealNVP's conditioner, flows/realnvp.py:44-71): the num_hidden_layers (even) hidden Linears are the
                                           * linear_layers of num_hidden_layers / 2 residual blocks (resnet.py:39-52, ReLU, no batch norm), no
                                           * activation behind the initial layer or in front of the output layer; same packed layout. */
#define NFA_FLAG_PAD_COLUMNS_SHIFT 8       /* with NFA_FLAG_STANDARD_NORMAL_LOG_PROB: bits 8-10 = number of trailing */
#define NFA_FLAG_PAD_COLUMNS_MASK 0x700   /* columns (0-7) that are the host's padding of the row (they pass through
                                             every layer, see nflows_amd/ops.py: fused_geometry), not features: the
                                             density sums over the other `features - n` columns, D = features - n */
#define NFA_FLAG_PAD_COLUMNS(n) ((n) << NFA_FLAG_PAD_COLUMNS_SHIFT)
#define NFA_FLAG_ACTIVATION_SHIFT 12      /* nfa_rqs_flow_resnet_* / nfa_rqs_coupling_resnet_f32 (ABI 9): bits 12-14 = the */
#define NFA_FLAG_ACTIVATION_MASK 0x7000   /* activation of the conditioner's residual blocks (nn/nets/resnet.py:27, :44, :47): */
#define NFA_ACTIVATION_RELU 0             /* F.relu (the reference's default) */
#define NFA_ACTIVATION_LEAKY_RELU 1       /* F.leaky_relu, negative_slope 0.01 */
#define NFA_ACTIVATION_ELU 2              /* F.elu, alpha 1 */
#define NFA_ACTIVATION_TANH 3             /* torch.tanh / F.tanh.  Other than ReLU: 8 or 10 bins (round 5: also with a context), not K8s. */
#define NFA_FLAG_ACTIVATION(a) ((a) << NFA_FLAG_ACTIVATION_SHIFT)

/* tails */
#define NFA_TAILS_NONE 0   /* rational_quadratic_spline: K+1 derivative logits per element */
#define NFA_TAILS_LINEAR 1 /* unconstrained_rational_quadratic_spline(tails="linear"): K-1 */

/* affine scale activations (coupling.py:224-225, :263-269) */
#define NFA_SCALE_DEFAULT 0  /* sigmoid(u + 2) + 1e-3 */
#define NFA_SCALE_GENERAL 1  /* clamp(softplus(u) + 1e-3, 0, 3) */
#define NFA_SCALE_ADDITIVE 2 /* AdditiveCouplingTransform: scale == 1, logabsdet == 0 */
#define NFA_SCALE_GIVEN 3    /* caller evaluated an arbitrary scale_activation into `scale` */
#define NFA_SCALE_SOFTPLUS 4 /* softplus(u) + 1e-3 (autoregressive.py:101) */

/*
 * Spline hyper-parameters: the keyword arguments of
 *   nflows/transforms/splines/rational_quadratic.py:13-25 (unconstrained_...) and :66-80.
 * Python floats stay doubles here; the kernels round them to fp32 exactly where aten does.
 */
typedef struct nfa_rqs_spec {
    int32_t num_bins;      /* K >= 1 */
    int32_t tails;         /* NFA_TAILS_* */
    double left, right;    /* for linear tails: -tail_bound, +tail_bound */
    double bottom, top;    /* for linear tails: -tail_bound, +tail_bound */
    double min_bin_width;  /* DEFAULT_MIN_BIN_WIDTH = 1e-3 */
    double min_bin_height; /* DEFAULT_MIN_BIN_HEIGHT = 1e-3 */
    double min_derivative; /* DEFAULT_MIN_DERIVATIVE = 1e-3 */
    double softplus_beta;  /* 1, or ln2/(1-min_derivative) if enable_identity_init (:100-103) */
    double tail_logit;     /* log(exp(1-min_derivative)-1), the padded boundary logit (:33-36) */
    double wh_divisor;     /* sqrt(hidden_features) applied to width/height logits
                              (coupling.py:554-559); 0 = no scaling */
} nfa_rqs_spec;

int nfa_abi_version(void);
const char *nfa_build_arch(void);      /* "gfx950" */
const char *nfa_strerror(int code);
int nfa_last_hip_error(void);          /* hipError_t of the last NFA_ERR_HIP on this thread */

/*
 * K1.  Fused rational-quadratic coupling layer given the conditioner output.
 * Replaces, in one launch:
 *   CouplingTransform.forward/inverse split + scatter      coupling.py:82-83, 96-98, 111-112, 126-128
 *   PiecewiseCouplingTransform._coupling_transform         coupling.py:279-293 (reshape, row-sum)
 *   PiecewiseRationalQuadraticCouplingTransform._piecewise_cdf   coupling.py:549-582
 *   (un)constrained rational_quadratic_spline               splines/rational_quadratic.py:13-181
 *   torchutils.searchsorted / sum_except_batch              utils/torchutils.py:134-136, 19-24
 * and optionally the Permutation that precedes the layer    permutations.py:27-39
 *
 *   inputs        [batch, features]
 *   params        [batch, num_transform * P],  P = 3K-1 (linear tails) or 3K+1; per feature
 *                 [w_0..w_{K-1}, h_0..h_{K-1}, d...]  (coupling.py:289, 550-552)
 *   transform_idx [num_transform] int64, the `transform_features` buffer (coupling.py:47-49)
 *   in_perm       [features] int64 or NULL: the layer sees inputs[:, in_perm]  (a Permutation
 *                 placed before the layer, fused into the gather)
 *   out_scatter   [features] int64 or NULL: layer column c is stored at outputs[:, out_scatter[c]],
 *                 i.e. outputs = index_select(y, 1, argsort(out_scatter)) (a Permutation.inverse
 *                 placed after the layer, permutations.py:22-24, :44-45, fused into the scatter)
 *   outputs       [batch, features]; columns not in transform_idx are copied bit-exactly
 *   logabsdet     [batch]
 *   bin_idx       [batch, num_transform] int32 or NULL: the bin the kernel's search chose for every spline --
 *                 the value of `bin_idx` at rational_quadratic.py:115-118, i.e. what torchutils.searchsorted
 *                 (utils/torchutils.py:134-136) returns on the kernel's own knots: 0 .. K-1 (K when the input
 *                 reaches the nudged last knot: the reference's gather then fails, here OUTSIDE_DOMAIN), and -1
 *                 for elements the reference never searches (linear tails, outside the domain, NaN).  The same
 *                 kernels, the same arithmetic; one more store.  Column j belongs to transform_idx[j].
 *   flags         NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET
 */
int nfa_rqs_coupling_f32(const float *inputs, const float *params, const int64_t *transform_idx,
                         const int64_t *in_perm, const int64_t *out_scatter, float *outputs,
                         float *logabsdet, int32_t *bin_idx, int32_t *status,
                         int64_t batch, int32_t features, int32_t num_transform,
                         const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * K1-backward.  Gradient of nfa_rqs_coupling_f32 (same arguments, same flags & NFA_FLAG_INVERSE):
 * what torch.autograd computes through the ~520 eager ops of one reference layer when a user
 * trains with `loss = -flow.log_prob(x).mean(); loss.backward()` (examples/moons.ipynb cell 3).
 * The spline is recomputed from (inputs, params); nothing else is saved by the forward pass.
 *   grad_outputs   [batch, features]        d loss / d outputs
 *   grad_logabsdet [batch] or NULL (= 0)    d loss / d logabsdet
 *   grad_inputs    [batch, features]        d loss / d inputs through the layer itself (the path
 *                  through the conditioner is the caller's: it owns `params`)
 *   grad_params    [batch, num_transform*P] d loss / d params
 */
int nfa_rqs_coupling_backward_f32(const float *inputs, const float *params,
                                  const int64_t *transform_idx, const int64_t *in_perm,
                                  const int64_t *out_scatter, const float *grad_outputs,
                                  const float *grad_logabsdet, float *grad_inputs,
                                  float *grad_params, int32_t *status, int64_t batch,
                                  int32_t features, int32_t num_transform,
                                  const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * K7.  K1 with the conditioner's output layer folded in: params^T = W @ hidden^T + b is computed
 * tile by tile on the matrix cores inside the kernel and consumed from the accumulator
 * registers, so the [batch, d_t*P] parameter tensor never goes through HBM
 * (ResidualNet.final_layer, nn/nets/resnet.py:90, :99, followed by everything
 * nfa_rqs_coupling_f32 replaces).
 *   hidden        [batch, hidden_features]  input of the final Linear
 *   weight_packed the Linear's weight [d_t*P, hidden_features] re-tiled for the MFMA A operand.
 *                 Rows: each feature's 23 rows padded to 24 (zero row); row i of 32-row tile t
 *                 (group g = t/3) is logit (idx % 24) of feature 4g + 2*hf + idx/24, where
 *                 hf = (i>>2)&1, q = 4*(i>>3) + (i&3), idx = 16*(t%3) + q  -- i.e. the 48
 *                 accumulator values a lane-half receives from a group's three tiles are the
 *                 logits of its two features in order (nflows_amd/ops.py:_k7_row_order).
 *                 Default (fp32 MFMA): float [tiles][16][64 lanes][4], lane l element (j4, c) =
 *                 Wrows[tile*32 + (l & 31)][(l >> 5)*64 + j4*4 + c].
 *                 NFA_FLAG_WEIGHTS_BF16X3: bf16 [tiles][3 pieces][8][64 lanes][8], lane l
 *                 element (piece, ks, j) = piece of Wrows[tile*32 + (l & 31)][(l >> 5)*64 +
 *                 ks*8 + j], pieces hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid)
 *                 (round to nearest even).
 *   bias_padded   [tiles][2 lane-halves][16]: the bias of the rows above in accumulator order
 * Supported here: num_bins = 8, linear tails, hidden_features = 128, d_t % 4 == 0, d_t <= 64,
 * features <= 128, batch % 32 == 0 (% 128 with NFA_FLAG_WEIGHTS_BF16X3); anything else returns
 * NFA_ERR_UNSUPPORTED (callers then run the GEMM and nfa_rqs_coupling_f32).
 */
int nfa_rqs_coupling_fused_linear_f32(const float *inputs, const float *hidden,
                                      const float *weight_packed, const float *bias_padded,
                                      const int64_t *transform_idx, const int64_t *in_perm,
                                      const int64_t *out_scatter, float *outputs, float *logabsdet,
                                      int32_t *status, int64_t batch, int32_t features,
                                      int32_t num_transform, int32_t hidden_features,
                                      const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * K8.  A whole neural-spline coupling layer: the ResidualNet conditioner (nn/nets/resnet.py:55-100:
 * initial_layer, num_blocks x ResidualBlock (ReLU, no batch norm / dropout / context),
 * final_layer) computed inside the kernel on the bf16 matrix pipe at fp32 accuracy (split-bf16
 * operands, see NFA_FLAG_WEIGHTS_BF16X3), followed by everything nfa_rqs_coupling_f32 replaces.
 * Activations and spline parameters never leave the register file; HBM traffic is one coalesced
 * read of inputs and one coalesced write of outputs + logabsdet.
 *   layer_tables   int32 [256], the layer's column bookkeeping with both neighbouring
 *                  permutations folded in (layer column c reads input column src[c] =
 *                  in_perm[c] and its output is stored at position dst[c] = out_scatter[c]).
 *                  The kernel keeps each sample's row in a tile whose slot j holds input column j;
 *                  features are transformed in place:
 *                  [0, 64): slot of identity feature i (identity_features, coupling.py:44-56, in
 *                  the conditioner's input order) = src[identity_features[i]];
 *                  [64, 128): slot of transformed feature f = src[transform_features[f]];
 *                  [128, 256): for every output position p the slot stored there
 *                  (= src[c] for the column c with dst[c] = p).
 *                  Entries outside [0, features) set NFA_STATUS_BAD_INDEX.
 *   weights_packed bf16, [stages][768 x 8]: 12 KB stages in consumption order --
 *                  initial_layer, 2 stages (d_i <= 32) or 4 (d_i <= 64), k-step ks = 0, 1, ..:
 *                    [4 tiles][3 pieces][64 lanes][8],
 *                    lane l element j = piece of W[32*tile + (l&31)][16*ks + 8*(l>>5) + j]
 *                    (columns >= d_i zero);
 *                  with col(ks, hf, j) = 32*(ks/2) + 16*(ks%2) + 8*(j/4) + 4*hf + j%4:
 *                  for every block, linear_layers[0] then [1]: 8 stages (ks = 0..7) of
 *                    [4 tiles][3 pieces][64 lanes][8], element j = piece of
 *                    W[32*tile + (l&31)][col(ks, l>>5, j)];
 *                  final_layer: two stages per 32-row tile (k-steps 4*s .. 4*s+3),
 *                    [3 pieces][4 k-steps][64 lanes][8], same element rule;
 *                  final_layer's rows are padded / reordered as in K7 (num_bins = 10: every
 *                    feature's 29 rows padded to 32, two tiles per group of two features, row i of
 *                    tile t = logit 16*(t%2) + 4*(i/8) + i%4 of feature 2*(t/2) + (i/4)%2; in general,
 *                    ABI 9, for every num_bins other than 8: the feature's 3*num_bins - 1 rows padded
 *                    to 16*T, T = ceil((3*num_bins - 1) / 16) tiles per group of two features, row i of
 *                    tile t = logit 16*(t%T) + 4*(i/8) + i%4 of feature 2*(t/T) + (i/4)%2);
 *                    its width and height
 *                    rows (and their biases) are multiplied by 1/sqrt(hidden_features)
 *                    (coupling.py:554-556; spec->wh_divisor is ignored here), and by log2(e)
 *                    with NFA_FLAG_LOGITS_LOG2E.
 *   bias_packed    float: initial_layer [4 tiles][2 lane-halves][16], every hidden Linear the
 *                  same, final_layer [tiles][2][16] (rows as in K7)
 * Numerics: the GEMMs are fp32-accurate but sum in another order than the reference's, so results
 * agree with nfa_rqs_coupling_f32 on the reference's own parameters to ~1e-6 relative, not bit for
 * bit; for num_bins = 8 the spline evaluation therefore uses a shorter rounding sequence than the
 * other kernels (same error class as the reference's fp32 path; environment NFA_K8_PIPE=1 selects
 * the reference's exact sequence, =0 additionally the unwoven loop).
 * Supported: num_bins = 8 or 10 (the reference's default; not with NFA_FLAG_LOGITS_LOG2E) and, ABI 9,
 * any other num_bins from 2 to 16 and 20, 24, 32 (plain final-layer loop on the spline kernel's own evaluator), linear
 * tails, hidden_features = 128, d_i <= 64, d_t % 4 == 0, d_t <= 64, features % 4 == 0,
 * features <= 128, batch % 128 == 0; otherwise NFA_ERR_UNSUPPORTED.
 * ABI 14 (round 6): spec->tails = NFA_TAILS_NONE -- the constructor default of the reference's coupling
 * (coupling.py:503-515, :543-547, :565-570: the constrained spline on [left, right] x [bottom, top]) -- at every bin count
 * above, ReLU blocks, no context, no NFA_FLAG_LOGITS_LOG2E: P = 3 K + 1 logits per feature (K + 1 derivative logits), the
 * final layer's rows padded to 16 ceil((3 K + 1) / 16) per feature in the general row order, the plain loop; an input
 * outside the box leaves NFA_STATUS_OUTSIDE_DOMAIN in `status` (rational_quadratic.py:81-82).  Table slots must not repeat a
 * column the layer transforms (no spare columns: a pad feature would lie inside the box).
 */
int nfa_rqs_coupling_resnet_f32(const float *inputs, const void *weights_packed,
                                const float *bias_packed, const int32_t *layer_tables,
                                float *outputs, float *logabsdet, int32_t *status, int64_t batch,
                                int32_t features, int32_t num_transform, int32_t num_identity,
                                int32_t hidden_features, int32_t num_blocks,
                                const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * K8 over a run of layers.  Rows of a flow are independent, so num_layers coupling layers of the
 * K8 shape family (same features / d_t / d_i / num_blocks / spec; CompositeTransform._cascade,
 * transforms/base.py:45-52, over [Permutation, coupling] pairs) run back to back on the same rows
 * in ONE launch: the row tile stays in LDS, inputs are read once and outputs written once for the
 * whole run, logabsdet is the sum over the layers.
 *   weights_packed / bias_packed  the layers' K8 blobs concatenated in execution order
 *   flow_tables  int32 [(num_layers + 1) * 128]: per layer [0, 64) identity slots and [64, 128)
 *                transformed slots as above, but relative to the FIRST layer's input columns (slot
 *                j = input column j of the run; every permutation between the layers is composed
 *                into the tables, the tile itself never moves); the last 128 entries give the
 *                slot stored at every output position of the run.
 */
int nfa_rqs_flow_resnet_f32(const float *inputs, const void *weights_packed,
                            const float *bias_packed, const int32_t *flow_tables,
                            int32_t num_layers, float *outputs, float *logabsdet, int32_t *status,
                            int64_t batch, int32_t features, int32_t num_transform,
                            int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                            const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * K8h: the same run of whole coupling layers with the conditioner's GEMMs on the f16 matrix pipe
 * from TWO f16 pieces per fp32 operand (x = hi + lo, three cross products per k-step; fp32-accurate:
 * measured max / rms error against float64 equal to a sequential fp32 fma chain's,
 * profiles/r2/f16x2_probe.txt) -- half the matrix-pipe work and two thirds of the weight bytes
 * of the three-piece bf16 scheme above.  Replaces the same reference code (coupling.py:73-130,
 * 549-582, nn/nets/resnet.py:92-100, transforms/base.py:45-52).
 *   stream_packed  16 KB stages in consumption order, layer after layer; everything a layer needs
 *                  reaches the kernel through this one LDS-DMA stream.  Per layer:
 *                  (1) `param_stages` PARAMETER stages, each carrying 2048 32-bit words in its first
 *                      8 KB (the rest unused), together 2048 * param_stages words:
 *                      words [0, 64)  int32 slots of the identity features, [64, 128) of the
 *                      transformed features (the per-layer rows of `flow_tables` above, relative
 *                      to the first layer's input columns); then per GEMM -- initial_layer, every
 *                      block's two Linears, final_layer -- a 4-float header {out_scale, skip_scale,
 *                      0, 0} followed by the biases in accumulator order ([tiles][2 lane-halves][16],
 *                      final_layer rows as in K7) times the scale their accumulators carry.  With the
 *                      pieces of hidden activations kept at scale S and every GEMM's weights
 *                      multiplied by a power of two T before the split (max |w T| in [2^13, 2^14),
 *                      which keeps the low pieces in the normal f16 range): initial_layer biases x T,
 *                      out_scale = S / T; a block's first Linear biases x S T, out_scale = 1 / T; its
 *                      second the same and skip_scale = S T / (scale of the fp32 residual stream,
 *                      i.e. of the GEMM that wrote it last); final_layer biases x S T, header
 *                      {kappa = 1 / (S T), 1 / kappa, 0, 0}: the spline evaluation reads logits =
 *                      accumulators x kappa.  Zero-padded to whole stages.
 *                  (2) WEIGHT stages, f16 [1024 x 8]: eight (hi, lo) fragment pairs of
 *                      [64 lanes][8]; lane l element j of a pair for (tile, k-step ks) = piece of
 *                      W'[32 tile + (l & 31)][column(ks, l >> 5, j)], W' = W x T.  initial_layer
 *                      (column = 16 ks + 8 (l >> 5) + j, columns >= d_i zero; 2 k-steps for d_i <= 32,
 *                      4 otherwise) and hidden Linears (column rule col(ks, hf, j) of K8; 8 k-steps):
 *                      one stage per TWO k-steps, pair 4 (ks % 2) + t = output tile t.  final_layer: one
 *                      stage per 32-row tile, pair ks = k-step ks; rows ordered / padded / pre-divided
 *                      by sqrt(hidden_features) as for K8.
 *   final_positions int32 [128]: the slot stored at every output position of the run (the last
 *                  128 entries of `flow_tables`).
 *   redo_blocks    int32 [batch / 128], written by the kernel: 0 = the 128-row block is done, 1 = it
 *                  produced a non-finite value (an activation beyond the f16 range, or non-finite
 *                  inputs) and NOTHING of it was written (outputs, logabsdet, status): the caller
 *                  runs nfa_rqs_flow_resnet_redo_f32 -- the K8 kernel above restricted to the
 *                  flagged blocks, same tables, its own (bf16) weight / bias blobs -- right behind
 *                  it on the same stream; no host synchronisation in between.  (K8s on 64-row blocks sets
 *                  bit 1 / bit 2 instead: only the lower / upper 64 rows of the block are open, the other
 *                  half is written; the redo entry points honour the bits.)
 * Supported: num_bins = 8 or 10, and (ABI 9; with a context: round 5) any other num_bins from 2 to 16 and 20, 24, 32 -- final-layer rows as
 * K8's general rule above, 16 ceil((3 num_bins - 1) / 16) per feature --, linear tails, hidden_features = 128 (narrower conditioners: zero-padded by the packer), d_i <= 64, d_t % 4 == 0,
 * d_t <= 64, features % 4 == 0, features <= 128, batch % 128 == 0; otherwise NFA_ERR_UNSUPPORTED.
 * Table slots may repeat a column (d_t + d_i may exceed features): the host side pads other shapes into
 * this family with constant columns outside the spline's box (nflows_amd/ops.py: fused_geometry).
 */
int nfa_rqs_flow_resnet_f16x2_f32(const float *inputs, const void *stream_packed, int32_t param_stages,
                                  const int32_t *final_positions, int32_t num_layers, float *outputs,
                                  float *logabsdet, int32_t *redo_blocks, int32_t *status, int64_t batch,
                                  int32_t features, int32_t num_transform, int32_t num_identity,
                                  int32_t hidden_features, int32_t num_blocks,
                                  const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * K8x (ABI 12): the same run of whole coupling layers with the conditioner's GEMMs on the f16 matrix pipe from THREE f16
 * pieces per fp32 operand -- x s = hi + lo + r' 2^-8 (s a power of two; the last piece kept x 2^8), 33 significand bits:
 * every fp32 operand is carried EXACTLY while its last piece stays above f16's smallest subnormal, i.e. at the reference's
 * own operand width (nn/nets/resnet.py:92-100: F.linear on fp32).  Products per multiply-add: hi hi, hi lo, lo hi as three
 * v_mfma_f32_32x32x16_f16 per k-step; hi r and r hi -- 2^-22-level terms: a last piece is one bit, its partner needs a few --
 * as ONE v_mfma_scale_f32_32x32x64_f8f6f4 (bf8 x bf8, block scale 2^-8) per two k-steps; dropped: lo lo and below
 * (<= 2^-24 relative).  4 f16-MFMA times per k-step and tile against K8's 6.  Replaces the same reference code as K8
 * (coupling.py:73-130, :549-582, nn/nets/resnet.py:39-52, :92-100, transforms/base.py:45-52).
 *   weights_packed raw bytes (declared f16), [stages][12 KB]: K8's stage COUNT per layer and its column / row rules (above);
 *                  a stage holds twelve 1 KB fragments of [64 lanes] x 16 B, lane l = 32 (lane-half) + row.  With W' = W x T
 *                  (T a power of two per GEMM, max |w T| in [2^13, 2^14); the final layer's width / height rows pre-divided
 *                  by sqrt(hidden_features) before T is chosen), H / L = the f16 pieces hi / lo of the lane's 8 values of a
 *                  k-step, X = the lane's 32 bf8 (OCP e5m2) bytes of a PAIR of k-steps ks0, ks1:
 *                  [bf8(hi) ks0 | bf8(256 r) ks0 | bf8(hi) ks1 | bf8(256 r) ks1], r = w T - hi - lo, split into X lo
 *                  (bytes 0 .. 15) and X hi (16 .. 31):
 *                    initial_layer and the blocks' Linears (k-major): per pair of k-steps TWO stages -- output tiles 0, 1 and
 *                    tiles 2, 3 --, each [2 tiles][H ks0, L ks0, H ks1, L ks1, X lo, X hi];
 *                    final_layer (tile-major): per 32-row tile two stages of four k-steps each,
 *                    [H0, L0, H1, L1][H2, L2, H3, L3][X01 lo, X01 hi, X23 lo, X23 hi]
 *                  (fragments 0 .. 3 of EVERY stage are the f16 fragments its first MFMAs need: the kernel reads them ahead,
 *                  right behind the previous stage's barrier).  Final-layer rows per feature: 24 for 8 bins (K7's row
 *                  order), otherwise K8's general rule, 16 ceil((3 num_bins - 1) / 16).
 *   bias_packed    K8's order; every GEMM's biases x S T (its own T), S = act_scale.
 *   scales         float [num_layers][2 + 2 num_blocks][2]: per GEMM in execution order {1 / T, T}; the final layer's
 *                  pair is {kappa = 1 / (S T), S T}: the spline evaluation reads logits = accumulators x kappa.
 *   act_scale      S, a power of two: the scale at which activations (identity features included) are split into
 *                  pieces.  A value keeps all 24 bits while |v S| >= 2^-9; below, its absolute error is <= 2^-33 / S;
 *                  |v S| >= 65520 overflows and poisons the row block (next line).
 *   redo_blocks    as for nfa_rqs_flow_resnet_f16x2_f32: 1 = the 128-row block produced a non-finite value and
 *                  nothing of it was written; run nfa_rqs_flow_resnet_redo_f32 (K8's blobs) behind it.
 * Supported: num_bins from 2 to 16 and 20, 24, 32, linear tails, ReLU blocks, no context, hidden_features = 128, d_i <= 64,
 * d_t % 4 == 0, d_t <= 64, features % 4 == 0, features <= 128, batch % 128 == 0; otherwise NFA_ERR_UNSUPPORTED (callers: K8).
 */
int nfa_rqs_flow_resnet_f16x3_f32(const float *inputs, const void *weights_packed, const float *bias_packed,
                                  const float *scales, const int32_t *flow_tables, int32_t num_layers,
                                  float *outputs, float *logabsdet, int32_t *redo_blocks, int32_t *status,
                                  int64_t batch, int32_t features, int32_t num_transform, int32_t num_identity,
                                  int32_t hidden_features, int32_t num_blocks, float act_scale,
                                  const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * Diagnostic twins (ABI 12; tests/test_gpu_logits.py): the same launches through instances that also store the LOGITS
 * of the run's LAST layer -- the final Linear's output as the spline evaluation reads it: what
 * `transform_net(identity_split)` returns in the reference (coupling.py:85), its width / height entries divided by
 * sqrt(hidden_features) (coupling.py:554-556) -- so that the conditioner's GEMM arithmetic can be held to an fp32
 * library GEMM's error against float64 on its own, in front of the spline.
 *   logits  float [batch, num_transform * 24], PACKED row order: entry 32 * tile + 16 * half + q of a row is
 *           accumulator register q of lane-half `half` of final-layer tile `tile`, i.e. packed row
 *           32 * tile + 8 * (q / 4) + 4 * half + q % 4 of K7's row order (above); rows of blocks flagged in
 *           `redo_blocks` carry the first pass's (discarded) values.
 * Served: 8 bins, ReLU blocks, no context (the bench's kernel family); K8h: d_i <= 32, also fills `bin_idx` as
 * nfa_rqs_flow_resnet_f16x2_bins_f32; K8: the plain final-layer loop (same products in the same order as the woven one).
 */
int nfa_rqs_flow_resnet_f16x3_logits_f32(const float *inputs, const void *weights_packed, const float *bias_packed,
                                         const float *scales, const int32_t *flow_tables, int32_t num_layers,
                                         float *outputs, float *logabsdet, int32_t *redo_blocks, int32_t *status,
                                         int64_t batch, int32_t features, int32_t num_transform, int32_t num_identity,
                                         int32_t hidden_features, int32_t num_blocks, float act_scale,
                                         const nfa_rqs_spec *spec, int32_t flags, void *stream, float *logits);
int nfa_rqs_flow_resnet_f16x2_logits_f32(const float *inputs, const void *stream_packed, int32_t param_stages,
                                         const int32_t *final_positions, int32_t num_layers, float *outputs,
                                         float *logabsdet, int32_t *redo_blocks, int32_t *status, int64_t batch,
                                         int32_t features, int32_t num_transform, int32_t num_identity,
                                         int32_t hidden_features, int32_t num_blocks,
                                         const nfa_rqs_spec *spec, int32_t flags, void *stream, int32_t *bin_idx,
                                         float *logits);
int nfa_rqs_flow_resnet_logits_f32(const float *inputs, const void *weights_packed,
                                   const float *bias_packed, const int32_t *flow_tables,
                                   int32_t num_layers, float *outputs, float *logabsdet, int32_t *status,
                                   int64_t batch, int32_t features, int32_t num_transform,
                                   int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                                   const nfa_rqs_spec *spec, int32_t flags, void *stream, float *logits);

/*
 * K8s.  nfa_rqs_flow_resnet_f16x2_f32 (same reference lines: nn/nets/resnet.py:55-100, coupling.py:73-130,
 * :549-582, a run of layers in one launch, optionally with the base density) on SIXTEEN-sample tiles
 * (v_mfma_f32_16x16x32_f16): a batch that gives a CU at most one 128-row block -- config 4's 32 768-row shard
 * of an 8-GPU run, interactive batches -- gets twice the waves with half the matrix work each; a batch with no
 * more 64-row blocks than CUs runs in four-wave workgroups of 64 rows (one wave per SIMD: a block's 32 layers
 * take 0.64 instead of 0.86 ms).  Same arguments, same semantics (`redo_blocks`: see there); `stream_packed` has the same stages and parameter words with the
 * orders of this tile shape (ops.pack_resnet_conditioner_f16(tile16=True)):
 *   lane l = (sample n = l % 16, lane group g = l / 16); a fragment is [64 lanes][8 halves] with lane l holding
 *   row 16 T + l % 16, MFMA k position 8 (l / 16) + j; GEMM inputs made of accumulator tiles use the column rule
 *   col(S, g, j) = 32 S + 16 (j / 4) + 4 g + j % 4; initial / hidden Linears: one stage per 32-wide k-step S,
 *   pair T = output tile T (16 rows); final Linear: one stage per TWO 16-row tiles, pair 4 (t % 2) + S; its rows
 *   ordered so that tile 6 G + tau, row 4 g + i = logit 4 tau + i of feature 4 G + g (padded to 24); every GEMM's
 *   biases in natural row order.
 * Supported: num_bins = 8, linear tails, no context, hidden_features = 128, d_i <= 64, d_t % 4 == 0, d_t <= 64,
 * features % 4 == 0, features <= 128, batch % 128 == 0; otherwise NFA_ERR_UNSUPPORTED.
 */
int nfa_rqs_flow_resnet_f16x2_tile16_f32(const float *inputs, const void *stream_packed, int32_t param_stages,
                                         const int32_t *final_positions, int32_t num_layers, float *outputs,
                                         float *logabsdet, int32_t *redo_blocks, int32_t *status, int64_t batch,
                                         int32_t features, int32_t num_transform, int32_t num_identity,
                                         int32_t hidden_features, int32_t num_blocks,
                                         const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * K8c (ABI 13, round 6): K8s's run of whole coupling layers with every GEMM split by COLUMNS over the four waves of a
 * 64-row workgroup (csrc/rqs_resnet_f16c.hip) -- the form for batches that give a CU at most one 64-row block
 * (`Flow.sample(n)` / `log_prob` of a few thousand rows: flows/base.py:51-75, distributions/base.py:69-84; replaces the
 * same reference code as nfa_rqs_flow_resnet_f16x2_f32).  Arguments, results and restrictions as for
 * nfa_rqs_flow_resnet_f16x2_tile16_f32; `redo_blocks`: workgroups are 32 rows (bits 3 .. 6 of a 128-row block's word = its
 * four quarters were not written; nfa_rqs_flow_resnet_redo_f32 honours them) or, with NFA_K8C_ROWS=64, 64 rows (bits 1 / 2
 * as for the tile16 entry); `stream_packed` differs in the FINAL layer's stages only: per round r of four
 * groups of four transformed features twelve stages, stage 2 i + s = for every wave w (fragment pairs 2 w, 2 w + 1)
 * k-steps 2 s, 2 s + 1 of 16-row tile i of group 4 r + w (K8s's row order within a group; zero fragments for groups
 * beyond d_t / 4): stages per layer = param_stages + (d_i > 32 ? 2 : 1) + 8 num_blocks + 12 ceil(d_t / 16).
 */
int nfa_rqs_flow_resnet_f16x2_colsplit_f32(const float *inputs, const void *stream_packed, int32_t param_stages,
                                           const int32_t *final_positions, int32_t num_layers, float *outputs,
                                           float *logabsdet, int32_t *redo_blocks, int32_t *status, int64_t batch,
                                           int32_t features, int32_t num_transform, int32_t num_identity,
                                           int32_t hidden_features, int32_t num_blocks,
                                           const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * Diagnostic twins of the two launches above (round 5): the SAME kernels compiled with one more store, for the bench's
 * kernel family only (8 bins, ReLU blocks, no context, d_i <= 32; otherwise NFA_ERR_UNSUPPORTED).
 *   bin_idx [batch, num_transform] int32: the bin every spline evaluation of the run's LAST layer chose -- the value
 *   of `bin_idx` at rational_quadratic.py:115-118 (torchutils.searchsorted, utils/torchutils.py:134-136) on the knots
 *   THIS kernel built (fp32 running sums of its own logits, csrc/rqs_fused8.hpp); -1 for inputs outside the box.
 *   Column j belongs to the j-th transformed feature of the layer's tables (surplus padded features: -1).
 * K8h / K8s keep no bin index on the data path; these entry points exist so that a test can compare the chosen bin with
 * the reference's and hold the elements where they differ to the output tolerances (tests/test_gpu_bin_index.py).  Rows
 * of blocks flagged in `redo_blocks` carry the first pass's (discarded) choice.
 */
int nfa_rqs_flow_resnet_f16x2_bins_f32(const float *inputs, const void *stream_packed, int32_t param_stages,
                                       const int32_t *final_positions, int32_t num_layers, float *outputs,
                                       float *logabsdet, int32_t *redo_blocks, int32_t *status, int64_t batch,
                                       int32_t features, int32_t num_transform, int32_t num_identity,
                                       int32_t hidden_features, int32_t num_blocks,
                                       const nfa_rqs_spec *spec, int32_t flags, void *stream, int32_t *bin_idx);
int nfa_rqs_flow_resnet_f16x2_tile16_bins_f32(const float *inputs, const void *stream_packed, int32_t param_stages,
                                              const int32_t *final_positions, int32_t num_layers, float *outputs,
                                              float *logabsdet, int32_t *redo_blocks, int32_t *status, int64_t batch,
                                              int32_t features, int32_t num_transform, int32_t num_identity,
                                              int32_t hidden_features, int32_t num_blocks,
                                              const nfa_rqs_spec *spec, int32_t flags, void *stream, int32_t *bin_idx);

/*
 * nfa_rqs_flow_resnet_f32 for conditioners that take a context (nn/nets/resnet.py:9-52, :92-100):
 *   context        [batch, context_features] fp32, the rows handed to every conditioner of the run
 *                  (Flow._log_prob's embedded context, flows/base.py:42-49).
 *   weights_packed as for nfa_rqs_coupling_resnet_f32, with (a) the initial layer's columns =
 *                  [identity features | context] (num_identity + context_features <= 64; 2 k-steps up to 32
 *                  columns, else 4) and (b) per block, behind its two Linears, `context_layer`: with
 *                  context_features <= 16 ONE stage laid out like a k-step of the other Linears ([4 tiles]
 *                  [3 pieces][64 lanes][8], column = 8 (l >> 5) + j); otherwise four stages, tile t's 32 rows
 *                  each ([3 pieces][4 k-steps][64 lanes][8], column = 16 ks + 8 (l >> 5) + j); columns >=
 *                  context_features zero.
 *   bias_packed    per block 384 floats: linear_layers[0], linear_layers[1], context_layer (accumulator order).
 * The block computes h + (W_1 relu(W_0 relu(h) + b_0) + b_1) * sigmoid(W_c context + b_c) (F.glu of the
 * concatenation, resnet.py:46-52).  Supported: as nfa_rqs_flow_resnet_f32 without NFA_FLAG_LOGITS_LOG2E -- 8 or 10 bins
 * (woven final layer with ReLU blocks, the plain loop with NFA_FLAG_ACTIVATION leaky ReLU / ELU / tanh) and, round 5,
 * every other served bin count with ReLU blocks (csrc/rqs_resnet_ctx.hip) --; otherwise NFA_ERR_UNSUPPORTED.
 */
int nfa_rqs_flow_resnet_context_f32(const float *inputs, const float *context, int32_t context_features,
                                    const void *weights_packed, const float *bias_packed,
                                    const int32_t *flow_tables, int32_t num_layers, float *outputs,
                                    float *logabsdet, int32_t *status, int64_t batch, int32_t features,
                                    int32_t num_transform, int32_t num_identity, int32_t hidden_features,
                                    int32_t num_blocks, const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * nfa_rqs_flow_resnet_f16x2_f32 for conditioners that take a context (K8h with a context; same reference
 * lines as nfa_rqs_flow_resnet_context_f32).  context_features <= 32 and d_i <= 32: the initial layer always
 * has four k-steps, columns = [identity features, zero-padded to 32 | context, zero-padded to 32].
 *   stream_packed  as for nfa_rqs_flow_resnet_f16x2_f32, with per block, behind its two Linears, ONE more
 *                  weight stage: `context_layer` x T_c as two k-steps laid out like the initial layer's
 *                  (column = 16 ks + 8 (l >> 5) + j), and behind the block's two parameter groups a third one:
 *                  header {1 / T_c, 0, 0, 0} + the gate's 128 biases x T_c (accumulator order).  The second
 *                  Linear's header keeps {out_scale, skip_scale}: the residual stream is multiplied by
 *                  skip_scale when the gated product is added to it.
 * Served (round 5: csrc/rqs_resnet_f16_ctx_{a,b}.hip): every bin count of nfa_rqs_flow_resnet_f16x2_f32 with ReLU
 * blocks, 8 / 10 bins with the other NFA_FLAG_ACTIVATION codes; reference vectors: tests/golden/flows_context_more.npz.
 * The second pass on flagged row blocks is nfa_rqs_flow_resnet_context_redo_f32.
 */
int nfa_rqs_flow_resnet_context_f16x2_f32(const float *inputs, const float *context, int32_t context_features,
                                          const void *stream_packed, int32_t param_stages,
                                          const int32_t *final_positions, int32_t num_layers, float *outputs,
                                          float *logabsdet, int32_t *redo_blocks, int32_t *status, int64_t batch,
                                          int32_t features, int32_t num_transform, int32_t num_identity,
                                          int32_t hidden_features, int32_t num_blocks,
                                          const nfa_rqs_spec *spec, int32_t flags, void *stream);

/* nfa_rqs_flow_resnet_context_f32 on the row blocks with redo_blocks[block] != 0 only. */
int nfa_rqs_flow_resnet_context_redo_f32(const float *inputs, const float *context, int32_t context_features,
                                         const void *weights_packed, const float *bias_packed,
                                         const int32_t *flow_tables, int32_t num_layers, float *outputs,
                                         float *logabsdet, const int32_t *redo_blocks, int32_t *status,
                                         int64_t batch, int32_t features, int32_t num_transform,
                                         int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                                         const nfa_rqs_spec *spec, int32_t flags, void *stream);

/* nfa_rqs_flow_resnet_f32 on the row blocks with redo_blocks[block] != 0 only (second pass of K8h). */
int nfa_rqs_flow_resnet_redo_f32(const float *inputs, const void *weights_packed,
                                 const float *bias_packed, const int32_t *flow_tables,
                                 int32_t num_layers, float *outputs, float *logabsdet,
                                 const int32_t *redo_blocks, int32_t *status, int64_t batch,
                                 int32_t features, int32_t num_transform, int32_t num_identity,
                                 int32_t hidden_features, int32_t num_blocks,
                                 const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * K11.  A run of affine / additive coupling layers with MLP conditioners in ONE launch: per layer
 *   transform_net = MLP(hidden_sizes = [128] * (1 + num_hidden_layers)) (nn/nets/mlp.py:47-68) +
 *   CouplingTransform.forward / .inverse (coupling.py:73-130) + AffineCouplingTransform /
 *   AdditiveCouplingTransform._coupling_transform_* (coupling.py:212-269) + the neighbouring column
 *   permutations (permutations.py:27-39), and CompositeTransform._cascade's loop over the layers
 *   (transforms/base.py:45-52).  The GEMMs run on three bf16 pieces per operand (fp32-accurate, full
 *   fp32 range: no second pass); per element the arithmetic is K2's.
 *   weights_packed  bf16 stages of 12 KB (768 x 16 B), per layer in the order consumed:
 *                   _input_layer ([ks][4 tiles][3 pieces][64 lanes][8]; column = 16 ks + 8 (l >> 5) + j,
 *                   columns >= d_i zero; 2 k-steps for d_i <= 32, else 4), every _hidden_layers[i]
 *                   (8 k-steps, columns in K8's accumulator order), _output_layer tile-major
 *                   ([tile][2 half-stages][3 pieces][4 k-steps][64 lanes][8]) with rows ordered so that
 *                   accumulator register q of lane-half h of tile t is: affine -- q < 8: shift of
 *                   feature 16 t + 8 h + q, q >= 8: unconstrained scale of feature 16 t + 8 h + q - 8;
 *                   additive -- shift of feature 32 t + 16 h + q; rows of features >= d_t zero.
 *   bias_packed     fp32, all GEMMs' biases in accumulator order ([tile][2 lane-halves][16]), layer
 *                   after layer: 128 + 128 num_hidden_layers + 32 final_tiles floats per layer.
 *   tables          as for nfa_rqs_flow_resnet_f32 (int32 [(num_layers + 1) * 128]).
 *   scale_activation NFA_SCALE_DEFAULT | NFA_SCALE_GENERAL | NFA_SCALE_ADDITIVE.
 *   flags           NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET | NFA_FLAG_STANDARD_NORMAL_LOG_PROB
 *                   | NFA_FLAG_SKIP_OUTPUTS | NFA_FLAG_RESIDUAL_BLOCKS (ABI 11: ResidualNet conditioners -- initial_layer,
 *                   the blocks' linear_layers in order, final_layer in the places of _input_layer, _hidden_layers,
 *                   _output_layer; the reference's SimpleRealNVP, flows/realnvp.py:17-71).
 * Supported: hidden_features = 128, d_i <= 64, d_t <= 64, features % 4 == 0, features <= 128,
 * batch % 128 == 0; otherwise NFA_ERR_UNSUPPORTED (callers then run the conditioner's GEMMs and K2).
 */
int nfa_affine_flow_mlp_f32(const float *inputs, const void *weights_packed, const float *bias_packed,
                            const int32_t *tables, int32_t num_layers, float *outputs, float *logabsdet,
                            int32_t *status, int64_t batch, int32_t features, int32_t num_transform,
                            int32_t num_identity, int32_t hidden_features, int32_t num_hidden_layers,
                            int32_t scale_activation, int32_t flags, void *stream);

/*
 * K12.  The sequential part of AutoregressiveTransform.inverse (autoregressive.py:43-52) for
 *   MaskedPiecewiseRationalQuadraticAutoregressiveTransform (autoregressive.py:404-495) over a MADE
 *   conditioner (made.py:233-311) as one persistent kernel: for t = 0 .. sequential_steps - 1 the
 *   hidden units of degree t (made.py:60-63: they are functions of features < t), feature t's
 *   3K - 1 output rows, the spline inverse of column t.  Features >= sequential_steps (= the largest
 *   hidden degree) no longer change any hidden unit; the caller finishes them from `hidden_out` with
 *   one GEMM and one elementwise launch (K5 / K1).
 *   step_blocks / block_starts / layout: built by ops.pack_made_schedule.  step_blocks holds one
 *     contiguous fp32 block per step t = 0 .. sequential_steps (the last one only completes the hidden
 *     vector): [16 ints: number of hidden units of degree t, offset of the unit rows, of the output rows, of the
 *     output biases (in floats from the block's start)] [unit table, 8 words per unit in layer order: bias,
 *     unit index, padded columns, source vector (-1 = the features), destination vector (-1 = none),
 *     add_stream, set_stream, 0] [the units' masked weight rows, zero-padded to multiples of 16 columns]
 *     [feature t's 3K - 1 rows of the output layer, Hp columns] [its 3K - 1 biases], padded to 1 KB and at
 *     least 48 floats long; block_starts int32 [sequential_steps + 2] in
 *     1 KB grains; `layout` (host memory, int32):
 *       [num_linears, residual, final_src, stream_vec, num_vectors, Hp, Xp, max_block_floats] then per
 *       Linear [padded_columns, src_vector (-1 = the features), dst_vector (-1 = none), add_stream,
 *       set_stream].
 *   outputs     [batch, features]: columns < sequential_steps are written
 *   logabsdet   [batch]: sum of those columns' log-derivatives (the inverse's sign)
 *   hidden_out  [batch, hidden_features]: the output layer's input with every hidden unit final
 * Supported: 8 or 10 bins with linear tails, ReLU, no context / batch norm, per-sample state
 * (padded features + one vector per hidden Linear) x 16 samples + two step blocks within the LDS;
 * otherwise NFA_ERR_UNSUPPORTED (callers keep the column-wise host loop).
 */
int nfa_made_rqs_inverse_f32(const float *inputs, const float *step_blocks, const int32_t *block_starts,
                             const int32_t *layout, int32_t layout_len, float *outputs, float *logabsdet,
                             float *hidden_out, int32_t *status, int64_t batch, int32_t features,
                             int32_t hidden_features, int32_t sequential_steps, const nfa_rqs_spec *spec,
                             void *stream);

/*
 * K13.  The output layer of a MADE conditioner and the autoregressive spline layer behind it in one kernel:
 *   params = hidden @ (W * mask)^T + b   (MADE.final_layer, transforms/made.py:261-268, :282; MaskedLinear :71-72)
 *   outputs, logabsdet = the RQ functional per feature, logabsdet summed per sample
 *   (MaskedPiecewiseRationalQuadraticAutoregressiveTransform._elementwise, transforms/autoregressive.py:453-489)
 * -- the forward pass of the layer (autoregressive.py:38-41) behind the hidden layers, and the last pass of its
 * inverse (:43-52) for the features whose parameters no longer change.  The [batch, num_features * 23] parameter
 * tensor is never formed.
 *   inputs / outputs   [batch, row_stride] rows; the kernel reads / writes columns first_column ..
 *                      first_column + num_features - 1 (feature f of this call = column first_column + f)
 *   hidden             [batch, hidden_features] the output layer's input
 *   weight_packed      the num_features * 23 masked rows, per feature padded to 24 rows, features padded to a
 *                      multiple of 8 (an even number of four-feature groups), hidden width zero-padded to 256; rows in K7's order
 *                      (nfa_rqs_coupling_fused_linear_f32) as bf16 triples
 *                      [tiles][3 pieces][16 k-steps][64 lanes][8]: lane l, element (piece, ks, j) = piece of
 *                      Wrows[tile*32 + (l & 31)][(l >> 5)*128 + ks*8 + j]
 *   bias_padded        [tiles][2 lane-halves][16] in accumulator order
 *   logabsdet_partial  [chunks][batch], chunks = ceil(2 ceil(num_features / 8) / NFA_MADE_OUTPUT_GROUPS_PER_CHUNK):
 *                      chunk c holds the sum over features 4 * 26 c .. of every row; the caller adds the chunks
 *                      in order (deterministic; no atomics)
 * flags: NFA_FLAG_INVERSE.  Supported: num_bins = 8, linear tails, hidden_features <= 256 and % 4 == 0,
 * batch % 128 == 0; otherwise NFA_ERR_UNSUPPORTED (callers run the GEMM and nfa_rqs_coupling_f32).
 */
#define NFA_MADE_OUTPUT_GROUPS_PER_CHUNK 26
int nfa_rqs_made_output_f32(const float *inputs, int64_t row_stride, int32_t first_column, const float *hidden,
                            int32_t hidden_features, const void *weight_packed, const float *bias_padded,
                            float *outputs, float *logabsdet_partial, int32_t *status, int64_t batch,
                            int32_t num_features, const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * K5.  Elementwise rational-quadratic functional (no row-sum):
 *   unconstrained_rational_quadratic_spline / rational_quadratic_spline,
 *   splines/rational_quadratic.py:13-63 / :66-181, as called from
 *   autoregressive.py:453-489 and nonlinearities.py:431-467.
 *   inputs, outputs, logabsdet: [n];  logits: row i at uw + i*stride_w (K floats),
 *   uh + i*stride_h (K), ud + i*stride_d (num_derivatives floats); strides in elements.
 *   num_derivatives = K-1 (linear tails) or K+1; larger values are accepted like the reference
 *   accepts them (it pads and gathers by bin index, extra logits are never read).
 *   bin_idx [n] int32 or NULL: the searched bin of every element (see nfa_rqs_coupling_f32).
 */
int nfa_rqs_elementwise_f32(const float *inputs, const float *unnormalized_widths, int64_t stride_w,
                            const float *unnormalized_heights, int64_t stride_h,
                            const float *unnormalized_derivatives, int64_t stride_d,
                            int32_t num_derivatives, float *outputs, float *logabsdet, int32_t *bin_idx,
                            int32_t *status, int64_t n, const nfa_rqs_spec *spec, int32_t inverse, void *stream);

/*
 * K5 in float64: the same functional on double tensors (the reference is dtype-generic,
 * rational_quadratic.py:66-181) -- `.double()` flows and the float64 ground truth of parity checks run
 * on the device.  Same arguments and status bits as nfa_rqs_elementwise_f32; a plain kernel (one lane per
 * element, logits read from global memory), not a fast path.
 */
int nfa_rqs_elementwise_f64(const double *inputs, const double *unnormalized_widths, int64_t stride_w,
                            const double *unnormalized_heights, int64_t stride_h,
                            const double *unnormalized_derivatives, int64_t stride_d,
                            int32_t num_derivatives, double *outputs, double *logabsdet, int32_t *bin_idx,
                            int32_t *status, int64_t n, const nfa_rqs_spec *spec, int32_t inverse, void *stream);

/*
 * torchutils.searchsorted (utils/torchutils.py:134-136) on its own, as the reference's callers outside the
 * fused kernels use it (splines/linear.py, quadratic.py, cubic.py, rational_quadratic.py:115-118):
 *   bin_idx[i] = #{ j : inputs[i] >= knot(i, j) } - 1,   knot(i, num_knots - 1) = bin_locations[..] + eps
 * A COUNT over all knots, not a binary search (equal to one iff the knots are monotone; NaN inputs count 0
 * knots: -1), with `eps` added in the knots' own precision exactly like the reference's in-place `+=`
 * (which the reference leaves behind in the caller's tensor; bin_locations is NOT modified here).
 *   bin_locations  row i at bin_locations + i * row_stride (num_knots floats); row_stride 0 = one row of
 *                  knots shared by all inputs (the reference's broadcast `bin_locations[None, :]`,
 *                  tests/utils/torchutils_test.py:80-90)
 *   inputs [n], bin_idx [n] int64 (the reference's dtype)
 */
int nfa_searchsorted_f32(const float *bin_locations, int64_t row_stride, int32_t num_knots, const float *inputs,
                         int64_t *bin_idx, int64_t n, double eps, void *stream);

/*
 * K5d-backward.  Gradient of nfa_rqs_elementwise_f64 (same inputs, spec and direction) -- what the reference
 * obtains from autograd through rational_quadratic.py:13-181 on double tensors (`flow.double()` training,
 * torch.autograd.gradcheck of a spline layer).  grad_outputs / grad_logabsdet [n] are the upstream gradients
 * of the two results (either may be NULL = zeros); grad_inputs [n], grad_widths [n, K], grad_heights [n, K]
 * and grad_derivatives [n, num_derivatives] are dense and written in full.  Elements outside the box
 * (linear tails: the identity) or that the forward pass flagged get grad_inputs = grad_outputs and zero
 * logit gradients.  The inverse direction differentiates the root implicitly.  One lane per element.
 */
int nfa_rqs_elementwise_backward_f64(const double *inputs, const double *unnormalized_widths, int64_t stride_w,
                                     const double *unnormalized_heights, int64_t stride_h,
                                     const double *unnormalized_derivatives, int64_t stride_d,
                                     int32_t num_derivatives, const double *grad_outputs,
                                     const double *grad_logabsdet, double *grad_inputs, double *grad_widths,
                                     double *grad_heights, double *grad_derivatives, int64_t n,
                                     const nfa_rqs_spec *spec, int32_t inverse, void *stream);

/*
 * K14.  The hidden part of a ResidualNet conditioner under TRAINING (nn/nets/resnet.py:92-100 without the final
 * layer: initial Linear, then blocks h += W_1 relu(W_0 relu(h) + b_0) + b_1; the reference differentiates the eager
 * ops by autograd, examples/moons.ipynb cell 3) -- one kernel for the forward pass, one for the chain of input
 * gradients; the weight / bias gradients are nfa_linear_wgrad_f32's on the arrays these two leave behind.
 *
 *   forward:  identity_inputs [batch, num_identity] -> hidden [batch, 128];
 *             saved [2 num_blocks][batch][128]: saved[2k] = relu(h_k), saved[2k+1] = relu(a_k), the inputs of block
 *             k's two Linears (what the weight gradients need); BEHIND the planes (ABI 9) the packed ReLU masks of the
 *             backward pass: [2 num_blocks][batch / 32][64] 8-byte words (bit 16 t + q of a lane's word = its element q of
 *             tile t is > 0), i.e. the buffer is 2 num_blocks x batch x (512 + 16) bytes and the backward kernels read
 *             only that tail of it.
 *   backward: grad_hidden [batch, 128] (+ saved) -> grad_identity_inputs [batch, num_identity] and
 *             grads [2 num_blocks][batch][128]: grads[2k] = d loss / d h_k (= grad_outputs of the Linear that
 *             produced h_k: the initial layer for k = 0, block k-1's second Linear otherwise),
 *             grads[2k+1] = d loss / d a_k (grad_outputs of block k's first Linear).
 *
 * weights_packed: split-bf16 triples in 12 KB stages (layout of nfa_rqs_coupling_resnet_f32's hidden layers).
 *   forward stream: the initial layer ((num_identity > 32 ? 4 : 2) k-major stages), then per block W_0, W_1 (8
 *   k-major stages each, columns in accumulator order); bias_packed: accumulator-order biases, 128 + 256 per block.
 *   backward stream: per block from the LAST to the first W_1^T, W_0^T (8 k-major stages each, columns in accumulator
 *   order), then W_in^T as ceil(num_identity / 32) tile-major tiles of two stages (rows zero-padded to 32).
 * fp32-accurate products on the bf16 matrix pipe (three pieces per operand, six products), full fp32 range.
 * NFA_ERR_UNSUPPORTED (the caller keeps the eager path): hidden_features != 128 for the two kernels (the packer takes
 * 4 <= hidden_features <= 128 in multiples of 4 and pads), num_blocks > 3, num_identity > 64 or not a multiple of 4,
 * batch not a multiple of 128.
 */
/* Optionally the forward kernel also applies the net's final Linear (128 -> out_features, out_features % 4 == 0;
 * resnet.py:99) behind the blocks: its tile-major stages (ceil(out_features / 32) tiles of two stages, rows
 * zero-padded to 32) follow the hidden layers in the forward stream, final_bias_packed holds its bias in accumulator
 * order (32 per tile), params [batch, out_features] receives the conditioner's output; out_features = 0: hidden only.
 *
 * The packer of the two streams (one launch; the weights change with every optimiser step).  block_params: HOST array
 * of 4 num_blocks device pointers W_0, b_0, W_1, b_1 (fp32, contiguous, [H, H] / [H]); initial_weight
 * [H, num_identity]; final_weight [out_features, H] / final_bias (NULL, 0: none); H = hidden_features of the packer
 * call, 4 <= H <= 128, H % 4 == 0: a narrower net is zero-padded into the 128-wide streams (its surplus units stay 0
 * through both passes; the caller pads / slices the [.., 128] arrays at the two ends).  forward_stages:
 * ((num_identity > 32 ? 4 : 2) + 16 num_blocks + 2 ceil(out_features / 32)) x 12288 bytes, forward_bias:
 * 128 (1 + 2 num_blocks) floats, final_bias_packed: 32 ceil(out_features / 32) floats, backward_stages:
 * (16 num_blocks + 2 ceil(num_identity / 32)) x 12288 bytes. */
int nfa_pack_resnet_hidden_train_f32(const float *initial_weight, const float *initial_bias,
                                     const float *const *block_params, const float *final_weight,
                                     const float *final_bias, int32_t out_features, int32_t num_identity,
                                     int32_t hidden_features, int32_t num_blocks, void *forward_stages,
                                     float *forward_bias, float *final_bias_packed, void *backward_stages,
                                     void *stream);
int nfa_resnet_hidden_forward_f32(const float *identity_inputs, const void *weights_packed,
                                  const float *bias_packed, float *saved, float *hidden,
                                  const float *final_bias_packed, float *params, int32_t out_features, int64_t batch,
                                  int32_t num_identity, int32_t hidden_features, int32_t num_blocks, void *stream);
int nfa_resnet_hidden_backward_f32(const float *grad_hidden, const void *weights_packed, const float *saved,
                                   float *grads, float *grad_identity_inputs, int64_t batch, int32_t num_identity,
                                   int32_t hidden_features, int32_t num_blocks, void *stream);
/* The backward pass from the conditioner's OUTPUT gradient (round 4): grad_params [batch, out_features] (out_features
 * % 4 == 0) -- the kernel first forms d loss / d hidden = grad_params W_f (the final Linear's input gradient,
 * resnet.py:99 under autograd; a library GEMM before) as a k-major GEMM over the out_features columns, writes it to
 * grad_hidden [batch, 128] (the grad_outputs of the last block's second Linear, for nfa_linear_wgrad_f32) and continues
 * as nfa_resnet_hidden_backward_f32.  weights_packed: the packer's backward stream of a call WITH the final Linear:
 * behind W_in^T's stages it carries W_f^T as ceil(out_features / 16) k-major stages (k = column of grad_params in
 * natural order, zero past out_features, units past hidden_features zero), which this kernel consumes first;
 * backward_stages of the packer is then (16 num_blocks + 2 ceil(num_identity / 32) + ceil(out_features / 16)) x 12288
 * bytes, its prefix unchanged (nfa_resnet_hidden_backward_f32 reads only the prefix). */
int nfa_resnet_backward_f32(const float *grad_params, int32_t out_features, const void *weights_packed,
                            const float *saved, float *grads, float *grad_hidden, float *grad_identity_inputs,
                            int64_t batch, int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                            void *stream);

/*
 * K9.  The spline's siblings as elementwise functionals (no row-sum), same calling convention as
 * nfa_rqs_elementwise_f32; spec supplies num_bins, tails (NFA_TAILS_LINEAR: box = +-right,
 * elements outside pass through with logabsdet 0), the box, and for the quadratic spline
 * min_bin_width / min_bin_height / wh_divisor; the derivative fields are ignored.
 *
 *   linear_spline / unconstrained_linear_spline, splines/linear.py:40-105 / :9-37
 *     unnormalized_pdf  [n, K] rows `stride` floats apart
 *   quadratic_spline / unconstrained_quadratic_spline, splines/quadratic.py:55-159 / :11-52
 *     unnormalized_widths [n, K]; unnormalized_heights [n, num_heights], num_heights = K+1, or K-1
 *     (the two boundary heights are then derived so that they normalise to 1, :93-107)
 *   cubic_spline / unconstrained_cubic_spline, splines/cubic.py:63-267 / :15-60
 *     unnormalized_widths, unnormalized_heights [n, K]; unnorm_derivatives_left / _right [n, 1]
 *     (eps = 1e-5 and quadratic_threshold = 1e-3, the reference's defaults, are built in)
 * A constrained input outside [left, right] sets NFA_STATUS_OUTSIDE_DOMAIN.
 */
int nfa_linear_spline_f32(const float *inputs, const float *unnormalized_pdf, int64_t stride,
                          float *outputs, float *logabsdet, int32_t *status, int64_t n,
                          const nfa_rqs_spec *spec, int32_t inverse, void *stream);
int nfa_quadratic_spline_f32(const float *inputs, const float *unnormalized_widths, int64_t stride_w,
                             const float *unnormalized_heights, int64_t stride_h,
                             int32_t num_heights, float *outputs, float *logabsdet,
                             int32_t *status, int64_t n, const nfa_rqs_spec *spec,
                             int32_t inverse, void *stream);
int nfa_cubic_spline_f32(const float *inputs, const float *unnormalized_widths, int64_t stride_w,
                         const float *unnormalized_heights, int64_t stride_h,
                         const float *unnorm_derivatives_left, int64_t stride_l,
                         const float *unnorm_derivatives_right, int64_t stride_r, float *outputs,
                         float *logabsdet, int32_t *status, int64_t n, const nfa_rqs_spec *spec,
                         int32_t inverse, void *stream);

/*
 * K6.  Rational-quadratic CDF transform with parameters shared by the whole batch:
 *   PiecewiseRationalQuadraticCDF._spline, nonlinearities.py:431-467 (what the spline coupling
 *   layer applies to its identity half when apply_unconditional_transform=True, coupling.py:524-535).
 *   inputs, outputs [batch, features]; logits [features, K], [features, K], [features, K-1 | K+1];
 *   logabsdet [batch] = sum over features.  The per-feature knots are built once per workgroup in
 *   LDS instead of being broadcast to [batch, features, K] as the reference does (:226-227).
 *   flags: NFA_FLAG_INVERSE.
 */
int nfa_rqs_shared_f32(const float *inputs, const float *unnormalized_widths,
                       const float *unnormalized_heights, const float *unnormalized_derivatives,
                       float *outputs, float *logabsdet, int32_t *status, int64_t batch,
                       int32_t features, const nfa_rqs_spec *spec, int32_t flags, void *stream);

/*
 * K2.  Fused affine / additive coupling layer given the conditioner output.
 *   AffineCouplingTransform._scale_and_shift / _coupling_transform_forward / _inverse
 *   coupling.py:234-252; AdditiveCouplingTransform coupling.py:255-269; plus the split/scatter
 *   and optional preceding permutation as in K1.
 *   params [batch, 2*num_transform] = [shift | unconstrained_scale] (additive: [batch, num_transform])
 *   scale  [batch, num_transform], only for NFA_SCALE_GIVEN, else NULL
 */
int nfa_affine_coupling_f32(const float *inputs, const float *params, const float *scale,
                            const int64_t *transform_idx, const int64_t *in_perm,
                            const int64_t *out_scatter, float *outputs, float *logabsdet,
                            int32_t *status, int64_t batch, int32_t features, int32_t num_transform,
                            int32_t scale_activation, int32_t flags, void *stream);

/*
 * K2b. Elementwise affine transform with interleaved parameters, the autoregressive form:
 *   MaskedAffineAutoregressiveTransform._elementwise_forward/_inverse, autoregressive.py:96-128
 *   params [batch, features, 2]: [...,0] = unconstrained scale, [...,1] = shift;
 *   scale = softplus(u) + 1e-3.  Produces outputs [batch, features] and logabsdet [batch].
 */
int nfa_affine_autoregressive_f32(const float *inputs, const float *params, float *outputs,
                                  float *logabsdet, int64_t batch, int32_t features,
                                  int32_t inverse, void *stream);

/*
 * K4.  Column permutation, bit-exact for any 4-byte element type:
 *   Permutation._permute (dim=1) = torch.index_select(inputs, 1, perm), permutations.py:27-39.
 *   out[b, c] = in[b, perm[c]]
 */
int nfa_permute_cols_b32(const void *inputs, const int64_t *perm, void *outputs, int32_t *status,
                         int64_t batch, int32_t features, void *stream);

/*
 * K3.  Per-sample reduction: torchutils.sum_except_batch, utils/torchutils.py:19-24.
 *   out[b] = sum_c x[b, c]
 */
int nfa_rowsum_f32(const float *x, float *out, int64_t rows, int64_t cols, void *stream);

/*
 * StandardNormal._log_prob fused with the flow's final add (distributions/normal.py:23-33,
 * flows/base.py:49):  out[b] = -0.5 * sum_c z[b,c]^2 - 0.5*cols*log(2*pi) + (logabsdet ? logabsdet[b] : 0)
 */
int nfa_standard_normal_log_prob_f32(const float *z, const float *logabsdet, float *out,
                                     int64_t rows, int64_t cols, void *stream);

/*
 * The two numbers a rank contributes to the data-parallel log-likelihood (the reference's training /
 * evaluation loops call `log_prob(x).mean()` / `.sum()`, e.g. README.md:53-60): out[0] = sum_i values[i]
 * accumulated in float64 in a fixed order (the same bits every run), out[1] = n.  One launch instead
 * of a fill + a reduction.  `workspace`: nfa_sum_count_workspace_bytes() bytes, zero before the first
 * call, left zero by every call, not shared between streams that may run concurrently.
 */
size_t nfa_sum_count_workspace_bytes(void);
int nfa_sum_count_f64(const float *values, int64_t n, double *out, void *workspace, void *stream);

/*
 * Backward of the linear, quadratic and cubic spline functionals (the reference differentiates
 * splines/linear.py:40-105, splines/quadratic.py:55-159 and splines/cubic.py:63-267 by autograd
 * through their eager ops).  Dense rows: unnormalized_pdf / unnormalized_widths [n, num_bins],
 * unnormalized_heights [n, num_heights] (cubic: [n, num_bins]), boundary-derivative logits [n]; grad_outputs, grad_logabsdet (may be NULL = 0), grad_inputs [n]; the logit
 * gradients have the logits' shapes.  `inputs` are the inputs of the pass that is differentiated
 * (inverse != 0: of the inverse pass).  Elements in the linear tails / outside the box get
 * grad_inputs = grad_outputs and zero logit gradients.
 */
int nfa_linear_spline_backward_f32(const float *inputs, const float *unnormalized_pdf,
                                   const float *grad_outputs, const float *grad_logabsdet,
                                   float *grad_inputs, float *grad_unnormalized_pdf, int64_t n,
                                   const nfa_rqs_spec *spec, int32_t inverse, void *stream);
int nfa_quadratic_spline_backward_f32(const float *inputs, const float *unnormalized_widths,
                                      const float *unnormalized_heights, int32_t num_heights,
                                      const float *grad_outputs, const float *grad_logabsdet,
                                      float *grad_inputs, float *grad_unnormalized_widths,
                                      float *grad_unnormalized_heights, int64_t n,
                                      const nfa_rqs_spec *spec, int32_t inverse, void *stream);

int nfa_cubic_spline_backward_f32(const float *inputs, const float *unnormalized_widths,
                                  const float *unnormalized_heights,
                                  const float *unnorm_derivatives_left,
                                  const float *unnorm_derivatives_right, const float *grad_outputs,
                                  const float *grad_logabsdet, float *grad_inputs,
                                  float *grad_unnormalized_widths, float *grad_unnormalized_heights,
                                  float *grad_unnorm_derivatives_left,
                                  float *grad_unnorm_derivatives_right, int64_t n,
                                  const nfa_rqs_spec *spec, int32_t inverse, void *stream);

/*
 * K10.  Weight and bias gradient of a conditioner layer y = x W^T + b (torch.nn.Linear; the
 * reference's conditioners nn/nets/resnet.py:44,49,94,99 and nn/nets/mlp.py:47-68 reach it through
 * autograd when a flow is trained, examples/moons.ipynb cell 3):
 *   grad_weight[O, I] = grad_outputs[B, O]^T . inputs[B, I]      grad_bias[O] = sum_b grad_outputs[b, :]
 * The reduction runs over the batch: the batch is split over the chip (LDS-DMA ring; since round 4 the products of
 * layers with more than 32 inputs run on the bf16 matrix cores -- every fp32 operand as three bf16 pieces, six cross
 * products, fp32 accumulation: fp32-accurate, full fp32 range; the environment variable NFA_K10_ENGINE=f32 selects the
 * fp32 matrix instruction), partial results go to `workspace` (nfa_linear_wgrad_workspace_bytes(...) bytes of device
 * memory, contents undefined before and after) and are summed in a fixed order: results are
 * deterministic.  grad_bias may be NULL.  in_features and out_features must be multiples of 4 and
 * inputs / grad_outputs 16-byte aligned, otherwise NFA_ERR_UNSUPPORTED.  flags must be 0.
 */
size_t nfa_linear_wgrad_workspace_bytes(int64_t batch, int32_t in_features, int32_t out_features);
int nfa_linear_wgrad_f32(const float *inputs, const float *grad_outputs, float *grad_weight,
                         float *grad_bias, void *workspace, int64_t batch, int32_t in_features,
                         int32_t out_features, int32_t flags, void *stream);

/*
 * The same for `count` (1 .. 8) Linear layers of ONE shape in one launch pair (round 4: the four 128 x 128 layers of a
 * ResidualNet conditioner's two blocks -- resnet.py:44, :49 under autograd).  The problems share the chip: each gets
 * 1 / count of the batch slices a lone problem would get, i.e. `count` times fewer partial results to write and to
 * sum and a `count` times longer stream per workgroup (the ring's two-stage ramp is paid once per 32 stages instead
 * of once per 8 at B = 65 536).  inputs / grad_outputs / grad_weight / grad_bias: HOST arrays of `count` device
 * pointers (grad_bias may be NULL, or hold NULLs: no bias gradient for those layers); workspace:
 * nfa_linear_wgrad_batched_workspace_bytes(count, ...) bytes.  Results are bit-identical from call to call (fixed
 * summation order), not to the single-problem entry point (another number of slices).
 */
size_t nfa_linear_wgrad_batched_workspace_bytes(int32_t count, int64_t batch, int32_t in_features, int32_t out_features);
int nfa_linear_wgrad_batched_f32(int32_t count, const float *const *inputs, const float *const *grad_outputs,
                                 float *const *grad_weight, float *const *grad_bias, void *workspace, int64_t batch,
                                 int32_t in_features, int32_t out_features, int32_t flags, void *stream);

/*
 * Measurement aids (bench.py, tools/), not part of the data path; the library's only global state.
 *
 * While enabled, every launch of a coupling-layer kernel (nfa_rqs_coupling_f32, _fused_linear_f32,
 * _resnet_f32) carries its own start/stop HIP events attached to the dispatch
 * (hipExtLaunchKernelGGL), up to max_launches; nfa_profile_collect waits for them and returns the
 * kernels' own durations in launch order (what rocprofv3 --kernel-trace reports), then resets.
 * max_launches = 0 disables.
 */
int nfa_profile_enable(int32_t max_launches);
int nfa_profile_collect(float *durations_ms, int32_t capacity, int32_t *count);

/*
 * Name of the layer kernel the calling thread launched last (round 4), e.g.
 * "k8h::rqs_resnet_f16_kernel<inverse=0, init_ks=2, waves=8, K=8, ctx=0>": the launchers choose the instance from the
 * batch, the device's CU count and the LDS budget, so the host cannot know it otherwise.  bench.py reports it as
 * `roofline.kernel`; the engine-coverage parity tests assert it.  Writes a NUL-terminated string of at most
 * `capacity` bytes (truncated if longer) and returns its untruncated length; an empty string before any launch.
 */
int nfa_last_layer_kernel(char *buffer, int32_t capacity);

/*
 * Phase timeline of the fused kernels (tools/k7_trace.py, tools/k8_trace.py): device_buffer =
 * 512 uint64 on the device that lane 0 of wave 0 of workgroups 0 and 256 fills with
 * cycle-counter stamps at phase boundaries of the next launches; NULL switches it off.
 */
void nfa_debug_k7_trace(void *device_buffer);

#ifdef __cplusplus
}
#endif
#endif /* NFLOWS_AMD_H */
