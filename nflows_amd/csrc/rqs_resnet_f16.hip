// K8h: the whole-layer kernel (ResidualNet conditioner, nn/nets/resnet.py:55-100, + everything K1
// replaces, coupling.py:73-130, :549-582, for a run of layers in one launch) with its GEMMs on the
// f16 matrix pipe from TWO pieces per fp32 operand.
//
//   x = hi + lo,  hi = RN16(x),  lo = RN16(x - hi)        |x - hi - lo| <= 2^-24 |x|
//   x * w ~= hi_x hi_w + hi_x lo_w + lo_x hi_w            dropped: lo_x lo_w <= 2^-24 |x w|
//
// Three v_mfma_f32_32x32x16_f16 per k-step and tile instead of the six bf16 products of the
// three-piece scheme (rqs_resnet.hip): half the matrix-pipe time, 4 bytes per weight instead of 6,
// a piece conversion of ~8 instead of 11+ VALU instructions per pair.  Measured on the GPU against
// float64 (tools/f16x2_probe.hip, K = 128): max 5.9e-7 / rms 5.2e-8 -- the figures of a sequential
// fp32 fma chain (5.2e-7 / 5.2e-8), better than the six-product bf16 scheme (6.6e-7 / 5.8e-8).
//
// What f16 needs that bf16 did not:
//   * range of the LOW pieces: a low piece below 2^-14 is subnormal (gfx950's f16 MFMA honours
//     subnormals, checked by the probe) and keeps only absolute precision 2^-25.  Weights are
//     therefore pre-scaled per GEMM by a power of two T (host: max |w T| in [2^13, 2^14)); the scale
//     comes back out when accumulators are converted to the next layer's pieces (a product with a
//     power of two: exact) or, for the final layer, inside the spline evaluation.  Activations stay
//     at scale S (1 by default): an activation below 0.25 / S carries an absolute error <= 2^-25 / S,
//     ~3e-8 absolute on a layer output (probe) -- invisible next to the fp32 rounding of the sum.
//   * range of the HIGH pieces: |activation| * S > 65504 overflows.  Every row block checks its
//     results: a block with any non-finite output writes nothing and raises its entries of `redo`;
//     the caller then runs the exact kernel (nfa_rqs_flow_resnet_redo_f32, three bf16 pieces: full
//     fp32 range) on the flagged blocks.  Overflow always poisons: an f16 infinity enters the
//     products, its low piece is x - inf = -inf, and inf - inf = NaN reaches every logit that depends
//     on it.  Rows with NaN / inf INPUTS take the same route (the reference's propagation rules).
//
// Structure (32 samples per wave, rows in an LDS tile by slot, GEMMs transposed and chained through
// the register file as in rqs_resnet.hip), and what is different here:
//   * ONE stream of 16 KB stages per layer feeds everything through LDS-DMA: first the layer's
//     PARAMETER stage(s) -- column tables, per-GEMM headers {out_scale, skip_scale}, all biases --
//     then the weights.  Inside the layer loop a wave issues no global load other than its LDS-DMA
//     requests: a `s_waitcnt vmcnt(n)` of the compiler for an ordinary load counts on in-order
//     return, which LDS-DMA requests sharing the counter do not give it.  (The stream is drained
//     before the ordinary loads / stores at the two ends of a row block.)
//   * The ring is deep (four 16 KB slots, three stages in flight) and shared by EIGHT waves (one workgroup
//     per CU) where the batch allows: a request lands 1-2 us after it was issued, so bytes per
//     second = bytes in flight / latency; eight waves per stream also halve the bytes per CU.
//   * Weight fragments are read from LDS one MFMA group ahead by asm reads with a counted lgkmcnt
//     (hipcc waits with lgkmcnt(0), i.e. also for the reads just issued for the next group).
//   * The final layer is tile-major (a 32-row output tile = 24 MFMAs = two stages) with the spline
//     evaluation woven between its MFMAs, one slice behind each MFMA (~5 independent VALU
//     instructions behind an MFMA are free, tools/weave_probe.hip); the hidden Linears are k-major
//     (four accumulators, every input piece read once).
//   * The residual stream h lives in fp32 accumulator registers across a block (hacc): the block's
//     second Linear accumulates straight into it (skip connection = one fma per value when the
//     accumulator is prepared), ReLU is applied when a tile is converted into pieces, not per k-step.
//   * No packed fp32 arithmetic (see split2).
//
// Restrictions: 2 .. 16, 20, 24 or 32 bins (8 and 10: tuned final-layer loops and conditioners with a context), linear tails, hidden width 128 (narrower: zero-padded by the host), ReLU
// blocks, d_i <= 64, d_t % 4 == 0, d_t <= 64, D % 4 == 0, D <= 128, batch % 128 == 0 (other feature counts
// and batches: padded by the host); with a context: up to 32 context features beside d_i <= 32.



#include "rqs_resnet_f16_kernel.hpp"

using namespace nfa;

static int launch_f16(const float* inputs, const float* context, int32_t context_features, const void* stream_packed,
                      int32_t param_stages, const int32_t* final_positions, int32_t num_layers, float* outputs,
                      float* logabsdet, int32_t* redo_blocks, int32_t* status, int64_t batch, int32_t features,
                      int32_t num_transform, int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                      const nfa_rqs_spec* spec, int32_t flags, void* stream, int32_t* dbg_bins = nullptr,
                      float* dbg_logits = nullptr) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET | NFA_FLAG_STANDARD_NORMAL_LOG_PROB |
                  NFA_FLAG_SKIP_OUTPUTS | NFA_FLAG_PAD_COLUMNS_MASK | NFA_FLAG_ACTIVATION_MASK))
        return NFA_ERR_INVALID_ARGUMENT;
    const int activation = (flags & NFA_FLAG_ACTIVATION_MASK) >> NFA_FLAG_ACTIVATION_SHIFT;
    if (activation > NFA_ACTIVATION_TANH) return NFA_ERR_INVALID_ARGUMENT;
    flags &= ~NFA_FLAG_ACTIVATION_MASK;
    if (!density_flags_valid(flags)) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1 || num_transform < 1 || num_identity < 1 ||
        num_transform > features || num_identity > features || num_blocks < 0 || num_layers < 1 || param_stages < 1)
        return NFA_ERR_INVALID_ARGUMENT;
    k8h::Args a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (a.sp.beta != 1.0f) return NFA_ERR_UNSUPPORTED;
    // bin counts: 8 and 10 have their own final-layer loops; 2 .. 16 and 20, 24, 32 otherwise
    const bool any_bins = a.sp.K != 8 && a.sp.K != 10;
    // activations other than ReLU: the two tuned bin counts (with or, round 5, without a context)
    if (activation != NFA_ACTIVATION_RELU && any_bins) return NFA_ERR_UNSUPPORTED;
    const bool bins_served = (a.sp.K >= 2 && a.sp.K <= 16) || a.sp.K == 20 || a.sp.K == 24 || a.sp.K == 32;
    if (!bins_served || !a.sp.linear || hidden_features != 128 || (num_transform & 3) != 0 || num_transform > 64 ||
        num_identity > 64 || features > 128 || (features & 3) != 0 || (batch & 127) != 0 || num_blocks > 64 ||
        num_layers > 4096)
        return NFA_ERR_UNSUPPORTED;
    const bool with_ctx = context_features > 0;
    if (context_features < 0) return NFA_ERR_INVALID_ARGUMENT;
    // with a context: two identity k-steps + two context k-steps in the initial layer
    if (with_ctx && (context_features > 32 || num_identity > 32)) return NFA_ERR_UNSUPPORTED;
    // rows of the final layer per transformed feature: 23 logits padded to 24 (8 bins: two features share three
    // tiles), otherwise 3 K - 1 padded to whole 16-row lane-half shares
    const int rows_per_feature = a.sp.K == 8 ? 24 : 16 * ((3 * a.sp.K - 1 + 15) / 16);
    const int param_words = k8h::kTabWords + (k8h::kHdr + 128) * (1 + (with_ctx ? 3 : 2) * num_blocks) + k8h::kHdr +
                            num_transform * rows_per_feature;
    if (param_stages * 2048 < param_words || param_stages > 4) return NFA_ERR_INVALID_ARGUMENT;
    if (batch == 0) return NFA_OK;
    if (!inputs || !stream_packed || !final_positions || !logabsdet || !redo_blocks ||
        (!outputs && !(flags & NFA_FLAG_SKIP_OUTPUTS)) || (with_ctx && !context))
        return NFA_ERR_INVALID_ARGUMENT;
    a.ctx = with_ctx ? context : nullptr;
    a.ce = context_features;
    a.dbg_bins = dbg_bins;
    a.dbg_logits = dbg_logits;
    // the diagnostic instances (nfa_rqs_flow_resnet_f16x2_bins_f32): the bench's kernel family only
    if (dbg_bins && (a.sp.K != 8 || with_ctx || activation != NFA_ACTIVATION_RELU)) return NFA_ERR_UNSUPPORTED;
    a.normal = (flags & NFA_FLAG_STANDARD_NORMAL_LOG_PROB) ? 1 : 0;
    a.skip_out = (flags & NFA_FLAG_SKIP_OUTPUTS) ? 1 : 0;
    a.Ds = density_columns(flags, features);
    if (a.Ds < 1) return NFA_ERR_INVALID_ARGUMENT;
    a.log_z = standard_normal_log_z(a.Ds);
    a.x = inputs;
    a.w = reinterpret_cast<const vec4f*>(stream_packed);
    a.final_tab = final_positions;
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
    a.param_stages = param_stages;
    a.param_words = param_words;
    const int init_ks = (with_ctx || num_identity > 32) ? 4 : 2;
    a.num_stages = param_stages + init_ks / 2 + (with_ctx ? 9 : 8) * num_blocks + num_transform * rows_per_feature / 32;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    a.trace = g_k7_trace;
    // workgroups of eight waves (256 rows, one per CU, one weight stream per CU) when the batch gives
    // every CU one; otherwise four waves (128 rows)
    const int cus = device_cu_count();
    static const int force_nw = getenv("NFA_K8H_WAVES") ? atoi(getenv("NFA_K8H_WAVES")) : 0;
    static const int force_ring = getenv("NFA_K8H_RING") ? atoi(getenv("NFA_K8H_RING")) : 0;   // 5: elastic stream (experiment, slower)
    int nw = ((batch & 255) == 0 && (batch >> 8) >= cus) ? 8 : 4;
    if (force_nw == 4 || (force_nw == 8 && (batch & 255) == 0)) nw = force_nw;
    const size_t lds_static = 1024;   // s_final, s_bad, s_sync (rounded up)
    const size_t lds_cap = 160 * 1024 - lds_static;
    auto lds_for = [&](int n, int ring) {
        return (size_t)ring * k8h::kStageVec4 * 16 + (size_t)n * features * k8h::kRowPad * sizeof(float) +
               (size_t)2 * ((param_words + 3) & ~3) * sizeof(float);
    };
    if (lds_for(nw, k8h::kRing) > lds_cap) nw = 4;
    if (lds_for(nw, k8h::kRing) > lds_cap) return NFA_ERR_UNSUPPORTED;
    // the elastic stream (five slots, counters instead of the per-stage barrier) where it fits: eight-wave
    // workgroups of the 8-bin kernel without a context (the bench's shape: 161 984 bytes at D = 64)
#ifdef NFA_K8H_ELASTIC
    const bool elastic = !dbg_bins && nw == 8 && !with_ctx && a.sp.K == 8 && force_ring == 5 && lds_for(8, k8h::kRingElastic) <= lds_cap;
#else
    const bool elastic = false;
    (void)force_ring;
#endif
    const size_t lds_launch = lds_for(nw, elastic ? k8h::kRingElastic : k8h::kRing);
    int64_t blocks = batch / (32 * nw);
    const int64_t per_cu = (nw == 4 && lds_launch + 2048 <= 80 * 1024) ? 2 : 1;
    const int64_t cap = (int64_t)cus * per_cu;
    if (blocks > cap) blocks = cap;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    profile_next_launch(&e0, &e1);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(nw * kWave);
    const bool inv = (flags & NFA_FLAG_INVERSE) != 0;
    void (*kern)(const k8h::Args) = nullptr;
    int which = with_ctx ? 16 + (inv ? 1 : 0) + (nw == 8 ? 2 : 0) + (a.sp.K == 10 ? 4 : 0)
                         : (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (nw == 8 ? 4 : 0) + (a.sp.K == 10 ? 8 : 0);
    if (elastic) which = 24 + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0);
    if (any_bins) which = 32 + (a.sp.K - 2) * 8 + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (nw == 8 ? 4 : 0);
    constexpr int kWhichContext = 32 + 31 * 8 + 3 * 16 + 8;   // (behind the diagnostic instances' eight)
    const bool more_ctx = with_ctx && (any_bins || activation != NFA_ACTIVATION_RELU);
    if (dbg_bins) {
        which = 32 + 31 * 8 + 3 * 16 + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (nw == 8 ? 4 : 0);
        kern = k8h::debug_kernel(inv, init_ks, nw);
        if (!kern) return NFA_ERR_UNSUPPORTED;
    } else if (more_ctx) {
        // a context beyond 8 / 10 bins with ReLU (round 5): rqs_resnet_f16_ctx_{a,b}.hip
        which = kWhichContext + (any_bins ? (a.sp.K - 2) * 4 : 31 * 4 + (activation - 1) * 8 + (a.sp.K == 10 ? 4 : 0)) +
                (inv ? 1 : 0) + (nw == 8 ? 2 : 0);
        kern = k8h::context_kernel_a(a.sp.K, activation, inv, nw);
        if (!kern) kern = k8h::context_kernel_b(a.sp.K, activation, inv, nw);
        if (!kern) return NFA_ERR_UNSUPPORTED;
    } else if (activation != NFA_ACTIVATION_RELU) {
        which = 32 + 31 * 8 + (activation - 1) * 16 + (a.sp.K == 10 ? 8 : 0) + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (nw == 8 ? 4 : 0);
        kern = k8h::activation_kernel(activation, a.sp.K, inv, init_ks, nw);
    } else if (any_bins) {
        kern = a.sp.K <= 9 ? k8h::bins_kernel_a(a.sp.K, inv, init_ks, nw)
               : a.sp.K <= 16 ? k8h::bins_kernel_b(a.sp.K, inv, init_ks, nw) : k8h::bins_kernel_c(a.sp.K, inv, init_ks, nw);
    }
    if (dbg_bins || more_ctx) {
    } else if (activation != NFA_ACTIVATION_RELU || any_bins) {
        if (!kern) return NFA_ERR_UNSUPPORTED;
    }
    else switch (which) {
#ifdef NFA_K8H_ELASTIC   // (experiment builds only: measured 3 % slower than the rigid stream, profiles/r3/k8h_elastic_stream.txt)
        case 24: kern = k8h::rqs_resnet_f16_kernel<false, 2, 8, 8, false, k8h::kRingElastic>; break;
        case 25: kern = k8h::rqs_resnet_f16_kernel<true, 2, 8, 8, false, k8h::kRingElastic>; break;
        case 26: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8, 8, false, k8h::kRingElastic>; break;
        case 27: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8, 8, false, k8h::kRingElastic>; break;
#endif
        case 16: kern = k8h::rqs_resnet_f16_kernel<false, 4, 4, 8, true>; break;
        case 17: kern = k8h::rqs_resnet_f16_kernel<true, 4, 4, 8, true>; break;
        case 18: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8, 8, true>; break;
        case 19: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8, 8, true>; break;
        case 20: kern = k8h::rqs_resnet_f16_kernel<false, 4, 4, 10, true>; break;
        case 21: kern = k8h::rqs_resnet_f16_kernel<true, 4, 4, 10, true>; break;
        case 22: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8, 10, true>; break;
        case 23: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8, 10, true>; break;
        case 0: kern = k8h::rqs_resnet_f16_kernel<false, 2, 4>; break;
        case 1: kern = k8h::rqs_resnet_f16_kernel<true, 2, 4>; break;
        case 2: kern = k8h::rqs_resnet_f16_kernel<false, 4, 4>; break;
        case 3: kern = k8h::rqs_resnet_f16_kernel<true, 4, 4>; break;
        case 4: kern = k8h::rqs_resnet_f16_kernel<false, 2, 8>; break;
        case 5: kern = k8h::rqs_resnet_f16_kernel<true, 2, 8>; break;
        case 6: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8>; break;
        case 7: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8>; break;
        case 8: kern = k8h::rqs_resnet_f16_kernel<false, 2, 4, 10>; break;
        case 9: kern = k8h::rqs_resnet_f16_kernel<true, 2, 4, 10>; break;
        case 10: kern = k8h::rqs_resnet_f16_kernel<false, 4, 4, 10>; break;
        case 11: kern = k8h::rqs_resnet_f16_kernel<true, 4, 4, 10>; break;
        case 12: kern = k8h::rqs_resnet_f16_kernel<false, 2, 8, 10>; break;
        case 13: kern = k8h::rqs_resnet_f16_kernel<true, 2, 8, 10>; break;
        case 14: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8, 10>; break;
        default: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8, 10>; break;
    }
    static const char* const act_names[] = {"relu", "leaky_relu", "elu", "tanh"};
    note_layer_kernel("k8h::rqs_resnet_f16_kernel<inverse=%d, init_ks=%d, waves=%d, K=%d, ctx=%d, ring=%d, act=%s>", inv ? 1 : 0,
                      init_ks, nw, a.sp.K, with_ctx ? 1 : 0, elastic ? k8h::kRingElastic : k8h::kRing, act_names[activation]);
    if (lds_launch > 64 * 1024) {
        static unsigned long long raised[32 + 31 * 8 + 3 * 16 + 8 + 31 * 4 + 3 * 8] = {};   // device masks (raise_dynamic_lds)
        {
            const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[which], (int)lds_cap);
            if (rc_lds != NFA_OK) return rc_lds;
        }
    }
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds_launch, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds_launch, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_rqs_flow_resnet_f16x2_f32(const float* inputs, const void* stream_packed, int32_t param_stages,
                                             const int32_t* final_positions, int32_t num_layers, float* outputs,
                                             float* logabsdet, int32_t* redo_blocks, int32_t* status, int64_t batch,
                                             int32_t features, int32_t num_transform, int32_t num_identity,
                                             int32_t hidden_features, int32_t num_blocks,
                                             const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    return launch_f16(inputs, nullptr, 0, stream_packed, param_stages, final_positions, num_layers, outputs, logabsdet,
                      redo_blocks, status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                      spec, flags, stream);
}

// the same launch through the diagnostic instances: bin_idx [batch, num_transform] receives the bin every evaluation of
// the LAST layer of the run chose (include/nflows_amd.h)
extern "C" int nfa_rqs_flow_resnet_f16x2_bins_f32(const float* inputs, const void* stream_packed, int32_t param_stages,
                                                  const int32_t* final_positions, int32_t num_layers, float* outputs,
                                                  float* logabsdet, int32_t* redo_blocks, int32_t* status, int64_t batch,
                                                  int32_t features, int32_t num_transform, int32_t num_identity,
                                                  int32_t hidden_features, int32_t num_blocks,
                                                  const nfa_rqs_spec* spec, int32_t flags, void* stream, int32_t* bin_idx) {
    if (!bin_idx) return NFA_ERR_INVALID_ARGUMENT;
    return launch_f16(inputs, nullptr, 0, stream_packed, param_stages, final_positions, num_layers, outputs, logabsdet,
                      redo_blocks, status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                      spec, flags, stream, bin_idx);
}

// the diagnostic instances once more, with the logits of the LAST layer (the final Linear's accumulators x kappa: the
// conditioner's output as the spline evaluation reads it) stored beside the bins (include/nflows_amd.h)
extern "C" int nfa_rqs_flow_resnet_f16x2_logits_f32(const float* inputs, const void* stream_packed, int32_t param_stages,
                                                    const int32_t* final_positions, int32_t num_layers, float* outputs,
                                                    float* logabsdet, int32_t* redo_blocks, int32_t* status, int64_t batch,
                                                    int32_t features, int32_t num_transform, int32_t num_identity,
                                                    int32_t hidden_features, int32_t num_blocks,
                                                    const nfa_rqs_spec* spec, int32_t flags, void* stream, int32_t* bin_idx,
                                                    float* logits) {
    if (!bin_idx || !logits) return NFA_ERR_INVALID_ARGUMENT;
    return launch_f16(inputs, nullptr, 0, stream_packed, param_stages, final_positions, num_layers, outputs, logabsdet,
                      redo_blocks, status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                      spec, flags, stream, bin_idx, logits);
}

extern "C" int nfa_rqs_flow_resnet_context_f16x2_f32(const float* inputs, const float* context,
                                                     int32_t context_features, const void* stream_packed,
                                                     int32_t param_stages, const int32_t* final_positions,
                                                     int32_t num_layers, float* outputs, float* logabsdet,
                                                     int32_t* redo_blocks, int32_t* status, int64_t batch,
                                                     int32_t features, int32_t num_transform, int32_t num_identity,
                                                     int32_t hidden_features, int32_t num_blocks,
                                                     const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    if (context_features < 1) return NFA_ERR_INVALID_ARGUMENT;
    return launch_f16(inputs, context, context_features, stream_packed, param_stages, final_positions, num_layers,
                      outputs, logabsdet, redo_blocks, status, batch, features, num_transform, num_identity,
                      hidden_features, num_blocks, spec, flags, stream);
}
