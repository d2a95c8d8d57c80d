// K8x (round 6): the whole-layer kernel -- ResidualNet conditioner, nn/nets/resnet.py:55-100 (forward :92-100, blocks
// :39-52), + everything K1 replaces, coupling.py:73-130, :549-582, for a RUN of layers in one launch -- with its GEMMs
// on the matrix pipe from THREE f16 pieces per fp32 operand (f16x3_gemm.hpp): operands carried at the reference's width
// (nn/nets/resnet.py:92-100 is F.linear on fp32: 24-bit significands; the three pieces hold 33).  Five cross products
// per multiply-add: hi hi, hi lo, lo hi on v_mfma_f32_32x32x16_f16; the two at the 2^-22 level, hi r and r hi, need
// two significant bits of each factor and run on ONE v_mfma_scale_f32_32x32x64_f8f6f4 (bf8 x bf8) per two k-steps --
// the activation's bf8 bytes are the HIGH BYTES of its f16 pieces (one v_perm_b32 per register), the weight's are
// packed by the host, the last pieces are kept x 2^8 and the instruction's e8m0 block scale takes the factor out:
// 4 instruction times per k-step against the three-piece bf16 kernel's 6 (K8, rqs_resnet.hip), whose structure this
// kernel keeps:
//
//   * a wave owns 32 rows for the whole run; the rows live in an LDS tile by slot, every GEMM is computed transposed
//     and chained through the register file (the accumulator tiles of one GEMM are the B operand of the next, up to
//     a column permutation the host applies to the weights);
//   * the residual stream is NOT kept in fp32: its three pieces are exact, the skip connection rebuilds the value
//     from them (two v_fma_mix_f32 per value), the first ReLU of a block is a sign mask on the pieces; block order:
//     skip first (v = bias + T p), then u's pieces, then the second Linear (K8's order spilled 12 GB per launch here);
//   * weights: 12 KB stages of twelve 1 KB fragments ([64 lanes] x 16 B) -- k-major: two k-steps of two output tiles,
//     [H0, L0, H1, L1, X lo, X hi] per tile; tile-major final layer: four k-steps of a tile, [H0, L0, H1, L1][H2, L2,
//     H3, L3][X01 lo, X01 hi, X23 lo, X23 hi] -- through the three-slot LDS-DMA ring of bf16x3_gemm.hpp, four waves per
//     workgroup, two workgroups per CU; fragments 0 .. 3 of every stage are read right behind the PREVIOUS stage's
//     barrier (f16x3_gemm.hpp: Lead), the barrier stands in front of a stage's last MFMAs;
//   * the final layer is tile-major with the spline evaluation (rqs_fused8.hpp: one walk over fp32 running knot
//     sums, logits read at scale 1 / kappa straight from the accumulators) woven between its MFMAs: 32 time units per
//     tile (an f16 MFMA one, a bf8 MFMA two).
//
// What f16 pieces need that bf16 pieces did not -- range.  Every GEMM's weights are multiplied by a power of two T
// before the split (host: max |w T| in [2^13, 2^14)), activations -- the row tile's identity features included -- are
// split at scale S (a power of two, host: 16): a value keeps all its 24 bits while |v S| >= 2^-9 (the last piece,
// kept x 2^8, then is >= 2^-24, f16's smallest subnormal), below that the absolute error is <= 2^-33 / S; |v S| >=
// 65520 overflows.  Accumulators hold S T x (the reference's pre-activation); `scales` = per GEMM {1 / T, T}: 1 / T
// takes an accumulator to the next pieces' scale (exact), T takes the residual stream's pieces to the second Linear's
// accumulator scale; the final layer's pair is {kappa = 1 / (S T), S T} for the spline evaluation.  Overflow poisons
// (hi = inf, lo = -inf, r = NaN, and ReLU's sign mask keeps NaN): a row block with any non-finite result writes
// nothing and raises its entry of `redo`, the caller runs K8 (three bf16 pieces: full fp32 range) on the flagged
// blocks right behind (nfa_rqs_flow_resnet_redo_f32), as for K8h.
//
// Served: 2 .. 16, 20, 24 or 32 bins (8: the two-features-per-three-tiles final layer; the others K8h's general scheme:
// rqs_resnet_f16x3_bins_{a,b}.hip), linear tails, ReLU blocks, no context, hidden width 128 (narrower: zero-padded by
// the host), d_i <= 64, d_t % 4 == 0, d_t <= 64, D % 4 == 0, D <= 128, batch % 128 == 0.  DBG instances (tests, 8
// bins): the logits of the run's LAST layer (accumulators x kappa) are stored as well.

#include "rqs_resnet_f16x3_kernel.hpp"

using namespace nfa;

static int launch_f16x3(const float* inputs, const void* weights_packed, const float* bias_packed, const float* scales,
                        const int32_t* tables, int32_t num_layers, float* outputs, float* logabsdet, int32_t* redo_blocks,
                        int32_t* status, int64_t batch, int32_t features, int32_t num_transform, int32_t num_identity,
                        int32_t hidden_features, int32_t num_blocks, float act_scale, const nfa_rqs_spec* spec,
                        int32_t flags, void* stream, float* dbg_logits = nullptr) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET | NFA_FLAG_STANDARD_NORMAL_LOG_PROB |
                  NFA_FLAG_SKIP_OUTPUTS | NFA_FLAG_PAD_COLUMNS_MASK | NFA_FLAG_ACTIVATION_MASK))
        return NFA_ERR_INVALID_ARGUMENT;
    const int activation = (flags & NFA_FLAG_ACTIVATION_MASK) >> NFA_FLAG_ACTIVATION_SHIFT;
    if (activation > NFA_ACTIVATION_TANH) return NFA_ERR_INVALID_ARGUMENT;
    flags &= ~NFA_FLAG_ACTIVATION_MASK;
    if (!density_flags_valid(flags)) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1 || num_transform < 1 || num_identity < 1 || num_transform > features ||
        num_identity > features || num_blocks < 0 || num_layers < 1)
        return NFA_ERR_INVALID_ARGUMENT;
    // (a power of two: the pieces' scale must come out again exactly)
    int exponent = 0;
    if (!(act_scale > 0.0f) || frexpf(act_scale, &exponent) != 0.5f) return NFA_ERR_INVALID_ARGUMENT;
    k8x::Args a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (a.sp.beta != 1.0f || activation != NFA_ACTIVATION_RELU) return NFA_ERR_UNSUPPORTED;
    const bool bins_served = (a.sp.K >= 2 && a.sp.K <= 16) || a.sp.K == 20 || a.sp.K == 24 || a.sp.K == 32;
    if (!bins_served || (dbg_logits && a.sp.K != 8) || !a.sp.linear || hidden_features != 128 || (num_transform & 3) != 0 || num_transform > 64 ||
        num_identity > 64 || features > 128 || (features & 3) != 0 || (batch & 127) != 0 || num_blocks > 64 ||
        num_layers > 4096)
        return NFA_ERR_UNSUPPORTED;
    if (batch == 0) return NFA_OK;
    if (!inputs || !weights_packed || !bias_packed || !scales || !tables || !logabsdet || !redo_blocks ||
        (!outputs && !(flags & NFA_FLAG_SKIP_OUTPUTS)))
        return NFA_ERR_INVALID_ARGUMENT;
    a.normal = (flags & NFA_FLAG_STANDARD_NORMAL_LOG_PROB) ? 1 : 0;
    a.skip_out = (flags & NFA_FLAG_SKIP_OUTPUTS) ? 1 : 0;
    a.Ds = density_columns(flags, features);
    if (a.Ds < 1) return NFA_ERR_INVALID_ARGUMENT;
    a.log_z = standard_normal_log_z(a.Ds);
    a.act_scale = act_scale;
    a.x = inputs;
    a.w = reinterpret_cast<const vec4f*>(weights_packed);
    a.bias = bias_packed;
    a.scales = scales;
    a.tables = tables;
    a.out = outputs;
    a.lad = logabsdet;
    a.redo = redo_blocks;
    a.status = status;
    a.batch = batch;
    a.D = features;
    a.dt = num_transform;
    a.di = num_identity;
    a.num_blocks = num_blocks;
    a.num_layers = num_layers;
    a.dbg_logits = dbg_logits;
    const int init_ks = num_identity > 32 ? 4 : 2;
    const int rows_per_feature = a.sp.K == 8 ? 24 : 16 * ((3 * a.sp.K - 1 + 15) / 16);
    a.num_stages = init_ks + 16 * num_blocks + 2 * (num_transform * rows_per_feature / 32);
    a.bias_per_layer = 128 + 256 * num_blocks + num_transform * rows_per_feature;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    const size_t lds = (size_t)kRing * kStageVec4 * 16 + (size_t)(kBlock / kWave) * features * kRowPad * sizeof(float) +
                       (size_t)num_transform * rows_per_feature * sizeof(float);
    int64_t blocks = batch >> 7;
    const int64_t per_cu = lds + 2048 <= 80 * 1024 ? 2 : 1;
    const int64_t cap = (int64_t)device_cu_count() * per_cu;
    if (blocks > cap) blocks = cap;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    profile_next_launch(&e0, &e1);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(kBlock);
    const bool inv = (flags & NFA_FLAG_INVERSE) != 0;
    void (*kern)(const k8x::Args) = nullptr;
    int which = (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (dbg_logits ? 4 : 0);
    if (a.sp.K != 8) {
        which = 8 + (a.sp.K - 2) * 4 + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0);
        kern = k8x::bins_kernel_a(a.sp.K, inv, init_ks);
        if (!kern) kern = k8x::bins_kernel_b(a.sp.K, inv, init_ks);
        if (!kern) return NFA_ERR_UNSUPPORTED;
    } else switch (which) {
        case 0: kern = k8x::rqs_resnet_f16x3_kernel<false, 2>; break;
        case 1: kern = k8x::rqs_resnet_f16x3_kernel<true, 2>; break;
        case 2: kern = k8x::rqs_resnet_f16x3_kernel<false, 4>; break;
        case 3: kern = k8x::rqs_resnet_f16x3_kernel<true, 4>; break;
        case 4: kern = k8x::rqs_resnet_f16x3_kernel<false, 2, true>; break;
        case 5: kern = k8x::rqs_resnet_f16x3_kernel<true, 2, true>; break;
        case 6: kern = k8x::rqs_resnet_f16x3_kernel<false, 4, true>; break;
        default: kern = k8x::rqs_resnet_f16x3_kernel<true, 4, true>; break;
    }
    note_layer_kernel("k8x::rqs_resnet_f16x3_kernel<inverse=%d, init_ks=%d, K=%d, dbg=%d>", inv ? 1 : 0, init_ks, a.sp.K, dbg_logits ? 1 : 0);
    if (lds > 64 * 1024) {
        static unsigned long long raised[8 + 31 * 4] = {};   // device masks (raise_dynamic_lds)
        const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[which], 160 * 1024 - 2048);
        if (rc_lds != NFA_OK) return rc_lds;
    }
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_rqs_flow_resnet_f16x3_f32(const float* inputs, const void* weights_packed, const float* bias_packed,
                                             const float* scales, const int32_t* flow_tables, int32_t num_layers,
                                             float* outputs, float* logabsdet, int32_t* redo_blocks, int32_t* status,
                                             int64_t batch, int32_t features, int32_t num_transform,
                                             int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                                             float act_scale, const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    return launch_f16x3(inputs, weights_packed, bias_packed, scales, flow_tables, num_layers, outputs, logabsdet,
                        redo_blocks, status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                        act_scale, spec, flags, stream);
}

extern "C" int nfa_rqs_flow_resnet_f16x3_logits_f32(const float* inputs, const void* weights_packed,
                                                    const float* bias_packed, const float* scales,
                                                    const int32_t* flow_tables, int32_t num_layers, float* outputs,
                                                    float* logabsdet, int32_t* redo_blocks, int32_t* status,
                                                    int64_t batch, int32_t features, int32_t num_transform,
                                                    int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                                                    float act_scale, const nfa_rqs_spec* spec, int32_t flags,
                                                    void* stream, float* logits) {
    if (!logits) return NFA_ERR_INVALID_ARGUMENT;
    return launch_f16x3(inputs, weights_packed, bias_packed, scales, flow_tables, num_layers, outputs, logabsdet,
                        redo_blocks, status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                        act_scale, spec, flags, stream, logits);
}
