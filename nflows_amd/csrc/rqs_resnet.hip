// K8: a whole neural-spline coupling layer in ONE kernel -- the ResidualNet conditioner
// (nn/nets/resnet.py:55-100: Linear(d_i -> 128), num_blocks x [ReLU, Linear, ReLU, Linear, +skip],
// Linear(128 -> d_t*23)) followed by everything K1 replaces (coupling.py:73-130, :549-582).
//
//   * A wave owns 32 samples for the whole layer.  Their rows are read once, coalesced, into a
//     wave-private LDS tile laid out by OUTPUT position (both fused permutations applied); the
//     conditioner inputs and the spline inputs are picked from that tile, the spline results
//     overwrite their positions, and the tile is written out as whole rows: HBM sees one coalesced
//     read of the inputs and one coalesced write of the outputs, pass-through columns bit-exact.
//   * Activations never leave the register file: every GEMM is computed transposed
//     (out^T = W x act^T), so the 32x32 accumulator tiles a lane holds after one layer are -- up to
//     a fixed permutation of the k index that the host applies to the next layer's weight
//     columns -- exactly the MFMA B operand of the next layer.
//   * GEMMs run on the bf16 matrix pipe at fp32 accuracy: operands are split into three bf16
//     pieces (x = hi + mid + lo), six cross products per k-step (see K7b in rqs_fused_linear.hip).
//     Weights are split on the host, activations in registers right after each layer.
//   * The weights of the whole layer (984 KB as bf16 triples at the BASELINE shape) are streamed
//     by LDS-DMA (global_load_lds, no staging registers) through a ring of three 12 KB stages
//     shared by the four waves of the workgroup, two stages ahead of the MFMAs.
//   * The residual input is not kept in fp32: it is rebuilt from its three pieces (exact to
//     2^-25 |h|) when the skip connection is added, and the first ReLU of a block is applied to
//     the pieces on the fly (sign of the leading piece), which keeps the kernel inside 256 VGPRs.
//   * The last GEMM's accumulators are the spline logits of the lane's own two features per
//     group; they are evaluated straight from registers (as in K7).  The 1/sqrt(hidden) scale of
//     the width / height logits (coupling.py:554-556) is folded into those weight rows by the host.
//
// Restrictions (the host falls back to PyTorch GEMMs + K7/K1 otherwise): K = 8 or 10 bins, linear
// tails, hidden width 128, ReLU, no context / batch norm / active dropout, d_i <= 64,
// d_t % 4 == 0, d_t <= 64, D % 4 == 0, D <= 128, batch % 128 == 0 here (leftover rows: other path).

#include "rqs_resnet_kernel.hpp"

using namespace nfa;

static int launch_resnet_layers(const float* inputs, const void* weights_packed, const float* bias_packed,
                                const int32_t* tables, int32_t num_layers, float* outputs, float* logabsdet,
                                int32_t* status, int64_t batch, int32_t features, int32_t num_transform,
                                int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                                const nfa_rqs_spec* spec, int32_t flags, void* stream,
                                const int32_t* redo = nullptr, const float* context = nullptr,
                                int32_t context_features = 0, float* dbg_logits = nullptr) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET | NFA_FLAG_LOGITS_LOG2E |
                  NFA_FLAG_STANDARD_NORMAL_LOG_PROB | NFA_FLAG_SKIP_OUTPUTS | NFA_FLAG_PAD_COLUMNS_MASK |
                  NFA_FLAG_ACTIVATION_MASK))
        return NFA_ERR_INVALID_ARGUMENT;
    const int activation = (flags & NFA_FLAG_ACTIVATION_MASK) >> NFA_FLAG_ACTIVATION_SHIFT;
    if (activation > NFA_ACTIVATION_TANH) return NFA_ERR_INVALID_ARGUMENT;
    flags &= ~NFA_FLAG_ACTIVATION_MASK;
    if (!density_flags_valid(flags)) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1 || num_transform < 1 || num_identity < 1 ||
        num_transform > features || num_identity > features || num_blocks < 0 || num_layers < 1)
        return NFA_ERR_INVALID_ARGUMENT;
    ResnetArgs a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    const bool any_bins = a.sp.K != 8 && a.sp.K != 10;   // 2 .. 16, 20, 24, 32 bins: the plain loop, no log2(e) fold
    // activations other than ReLU: 8 or 10 bins, the plain loop, no log2(e) fold
    if (activation != NFA_ACTIVATION_RELU && (any_bins || (flags & NFA_FLAG_LOGITS_LOG2E)))
        return NFA_ERR_UNSUPPORTED;
    // tails=None (round 6): 3 K + 1 logits per feature, the plain loop, ReLU, no context, no log2(e) fold, no redo role
    const bool no_tails = !a.sp.linear;
    if (no_tails && (activation != NFA_ACTIVATION_RELU || (flags & NFA_FLAG_LOGITS_LOG2E) || context_features > 0 || dbg_logits || redo))
        return NFA_ERR_UNSUPPORTED;
    const int rows_per_feature = no_tails ? 16 * ((3 * a.sp.K + 1 + 15) / 16) : a.sp.K == 8 ? 24 : 16 * ((3 * a.sp.K - 1 + 15) / 16);
    if (a.sp.beta != 1.0f) return NFA_ERR_UNSUPPORTED;  // (identity initialisation: functional callers only)
    const bool bins_served = (a.sp.K >= 2 && a.sp.K <= 16) || a.sp.K == 20 || a.sp.K == 24 || a.sp.K == 32;
    if (!bins_served || (a.sp.K != 8 && (flags & NFA_FLAG_LOGITS_LOG2E)) ||
        hidden_features != 128 || (num_transform & 3) != 0 ||
        num_transform > 64 || num_identity > 64 || features > 128 || (features & 3) != 0 ||
        (batch & 127) != 0 || num_blocks > 64 || num_layers > 4096)
        return NFA_ERR_UNSUPPORTED;
    const bool with_ctx = context_features > 0;
    if (context_features < 0) return NFA_ERR_INVALID_ARGUMENT;
    // with a context: 8 bins, the default evaluation, identity features + context within the initial
    // layer's 64 input columns
    if (with_ctx && ((flags & NFA_FLAG_LOGITS_LOG2E) || num_identity + context_features > 64))
        return NFA_ERR_UNSUPPORTED;
    if (batch == 0) return NFA_OK;
    if (!inputs || !weights_packed || !bias_packed || !tables || !logabsdet ||
        (!outputs && !(flags & NFA_FLAG_SKIP_OUTPUTS)) || (with_ctx && !context))
        return NFA_ERR_INVALID_ARGUMENT;
    a.ctx = with_ctx ? context : nullptr;
    a.ce = context_features;
    a.dbg_logits = dbg_logits;
    // the diagnostic instances (nfa_rqs_flow_resnet_logits_f32): the bench's kernel family only
    if (dbg_logits && (a.sp.K != 8 || with_ctx || activation != NFA_ACTIVATION_RELU || (flags & NFA_FLAG_LOGITS_LOG2E) || redo))
        return NFA_ERR_UNSUPPORTED;
    a.normal = (flags & NFA_FLAG_STANDARD_NORMAL_LOG_PROB) ? 1 : 0;
    a.skip_out = (flags & NFA_FLAG_SKIP_OUTPUTS) ? 1 : 0;
    a.Ds = density_columns(flags, features);
    if (a.Ds < 1) return NFA_ERR_INVALID_ARGUMENT;
    a.log_z = standard_normal_log_z(a.Ds);
    a.x = inputs;
    a.w = reinterpret_cast<const vec4f*>(weights_packed);
    a.bias = bias_packed;
    a.tables = tables;
    a.out = outputs;
    a.lad = logabsdet;
    a.status = status;
    a.batch = batch;
    a.D = features;
    a.dt = num_transform;
    a.di = num_identity;
    a.num_blocks = num_blocks;
    a.num_layers = num_layers;
    const int init_ks = num_identity + context_features > 32 ? 4 : 2;
    a.num_stages = init_ks + (with_ctx ? (context_features <= 16 ? 17 : 20) : 16) * num_blocks +
                   2 * (num_transform * rows_per_feature / 32);
    a.bias_per_layer = 128 + (with_ctx ? 384 : 256) * num_blocks + num_transform * rows_per_feature;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    a.trace = g_k7_trace;
    a.redo = redo;
    // final layer with the spline evaluation woven into its MFMAs (not with the log2(e) fold):
    //   2 (default)  woven, FlatSteps<FAST>: cheaper rounding sequence, same error class
    //   1            woven, same results bit for bit as the plain loop
    //   0            the plain loop
    static const int use_pipe = [] {
        const char* e = getenv("NFA_K8_PIPE");
        return e ? atoi(e) : 2;
    }();
    // (with the log2(e) fold only the default woven form exists)
    const bool pipe = !no_tails && !dbg_logits && !any_bins && activation == NFA_ACTIVATION_RELU && use_pipe && ((a.sp.K == 8 && (!(flags & NFA_FLAG_LOGITS_LOG2E) || use_pipe == 2)) ||
                                  (a.sp.K == 10 && use_pipe == 2));
    // a context: the woven default form at 8 / 10 bins with ReLU; the plain loop for the other bin counts and
    // activations (round 5: rqs_resnet_ctx.hip)
    const bool more_ctx = with_ctx && (any_bins || activation != NFA_ACTIVATION_RELU);
    if (with_ctx && !more_ctx && !(pipe && use_pipe == 2)) return NFA_ERR_UNSUPPORTED;
    const size_t lds = (size_t)kRing * kStageVec4 * 16 + (size_t)(kBlock / kWave) * features * kRowPad * sizeof(float) +
                       (pipe ? (size_t)num_transform * rows_per_feature * sizeof(float) : 0) +
                       (size_t)(kBlock / kWave) * context_features * kRowPad * sizeof(float) +
                       (with_ctx ? (size_t)(kBlock / kWave) * 64 * kWave * sizeof(float) : 0);
    int64_t blocks = batch >> 7;
    const int64_t per_cu = lds + 2048 <= 80 * 1024 ? 2 : 1;
    const int64_t cap = (int64_t)device_cu_count() * per_cu;
    if (blocks > cap) blocks = cap;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!redo) profile_next_launch(&e0, &e1);  // (the second pass is not the measured kernel)
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(kBlock);
    const bool inv = (flags & NFA_FLAG_INVERSE) != 0, l2e = (flags & NFA_FLAG_LOGITS_LOG2E) != 0;
    void (*kern)(const ResnetArgs) = nullptr;
#define NFA_K8_PICK(INV_, PRE_)                                                                    \
    kern = init_ks == 4 ? rqs_resnet_kernel<INV_, PRE_, 4> : rqs_resnet_kernel<INV_, PRE_, 2>
    if (inv) {
        if (l2e) NFA_K8_PICK(true, 2);
        else NFA_K8_PICK(true, 1);
    } else {
        if (l2e) NFA_K8_PICK(false, 2);
        else NFA_K8_PICK(false, 1);
    }
#undef NFA_K8_PICK
    if (a.sp.K == 10 && pipe) {
        if (init_ks == 4) kern = inv ? rqs_resnet_kernel<true, 1, 4, 2, 10> : rqs_resnet_kernel<false, 1, 4, 2, 10>;
        else kern = inv ? rqs_resnet_kernel<true, 1, 2, 2, 10> : rqs_resnet_kernel<false, 1, 2, 2, 10>;
    } else if (a.sp.K == 10) {
        if (init_ks == 4) kern = inv ? rqs_resnet_kernel<true, 1, 4, 0, 10> : rqs_resnet_kernel<false, 1, 4, 0, 10>;
        else kern = inv ? rqs_resnet_kernel<true, 1, 2, 0, 10> : rqs_resnet_kernel<false, 1, 2, 0, 10>;
    } else if (pipe && use_pipe == 2 && l2e) {
        if (init_ks == 4) kern = inv ? rqs_resnet_kernel<true, 2, 4, 2> : rqs_resnet_kernel<false, 2, 4, 2>;
        else kern = inv ? rqs_resnet_kernel<true, 2, 2, 2> : rqs_resnet_kernel<false, 2, 2, 2>;
    } else if (pipe && use_pipe == 2) {
        if (init_ks == 4) kern = inv ? rqs_resnet_kernel<true, 1, 4, 2> : rqs_resnet_kernel<false, 1, 4, 2>;
        else kern = inv ? rqs_resnet_kernel<true, 1, 2, 2> : rqs_resnet_kernel<false, 1, 2, 2>;
    } else if (pipe) {
        if (init_ks == 4) kern = inv ? rqs_resnet_kernel<true, 1, 4, 1> : rqs_resnet_kernel<false, 1, 4, 1>;
        else kern = inv ? rqs_resnet_kernel<true, 1, 2, 1> : rqs_resnet_kernel<false, 1, 2, 1>;
    }
    if (activation != NFA_ACTIVATION_RELU) kern = resnet_activation_kernel(activation, a.sp.K, inv, init_ks);
    else if (any_bins) kern = resnet_bins_kernel(a.sp.K, inv, init_ks);
    if ((activation != NFA_ACTIVATION_RELU || any_bins) && !kern) return NFA_ERR_UNSUPPORTED;
    if (more_ctx) {
        kern = resnet_context_kernel(a.sp.K, activation, inv, init_ks);
        if (!kern) return NFA_ERR_UNSUPPORTED;
    } else if (with_ctx && a.sp.K == 10) {
        if (init_ks == 4) kern = inv ? rqs_resnet_kernel<true, 1, 4, 2, 10, true> : rqs_resnet_kernel<false, 1, 4, 2, 10, true>;
        else kern = inv ? rqs_resnet_kernel<true, 1, 2, 2, 10, true> : rqs_resnet_kernel<false, 1, 2, 2, 10, true>;
    } else if (with_ctx) {
        if (init_ks == 4) kern = inv ? rqs_resnet_kernel<true, 1, 4, 2, 8, true> : rqs_resnet_kernel<false, 1, 4, 2, 8, true>;
        else kern = inv ? rqs_resnet_kernel<true, 1, 2, 2, 8, true> : rqs_resnet_kernel<false, 1, 2, 2, 8, true>;
    }
    if (dbg_logits) kern = resnet_debug_kernel(inv, init_ks);
    if (no_tails) {
        kern = resnet_tails_kernel(a.sp.K, inv, init_ks);
        if (!kern) return NFA_ERR_UNSUPPORTED;
    }
    if (!redo)
        note_layer_kernel("rqs_resnet_kernel<inverse=%d, init_ks=%d, pipe=%d, K=%d, ctx=%d, act=%d%s>", inv ? 1 : 0, init_ks,
                          pipe ? use_pipe : 0, a.sp.K, with_ctx ? 1 : 0, activation, no_tails ? ", tails=none" : "");
    if (with_ctx && lds > 64 * 1024) {
        static unsigned long long raised_ctx[8 + 31 * 4 + 3 * 8] = {};   // device masks (raise_dynamic_lds)
        const int which = (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) +
                          (!more_ctx ? (a.sp.K == 10 ? 4 : 0)
                                     : 8 + (any_bins ? (a.sp.K - 2) * 4 : 31 * 4 + (activation - 1) * 8 + (a.sp.K == 10 ? 4 : 0)));
        {
            const int rc_lds = raise_dynamic_lds((const void*)kern, &raised_ctx[which], 160 * 1024 - 2048);
            if (rc_lds != NFA_OK) return rc_lds;
        }
    } else if (no_tails && lds > 64 * 1024) {
        static unsigned long long raised_tails[31 * 4] = {};
        const int rc_lds = raise_dynamic_lds((const void*)kern, &raised_tails[(a.sp.K - 2) * 4 + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0)], 160 * 1024 - 2048);
        if (rc_lds != NFA_OK) return rc_lds;
    } else if (dbg_logits && lds > 64 * 1024) {
        static unsigned long long raised_dbg[4] = {};
        const int rc_lds = raise_dynamic_lds((const void*)kern, &raised_dbg[(inv ? 1 : 0) + (init_ks == 4 ? 2 : 0)], 160 * 1024 - 2048);
        if (rc_lds != NFA_OK) return rc_lds;
    } else if (lds > 64 * 1024) {
        static unsigned long long raised[32 + 31 * 4 + 3 * 8] = {};   // device masks (raise_dynamic_lds)  // opt in to > 64 KB of dynamic LDS once per kernel
        const int which = activation != NFA_ACTIVATION_RELU ? 32 + 31 * 4 + (activation - 1) * 8 + (a.sp.K == 10 ? 4 : 0) + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0)
                          : any_bins ? 32 + (a.sp.K - 2) * 4 + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) : a.sp.K == 10 ? (pipe ? 24 : 12) + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0)
                          : (pipe && use_pipe == 2) ? 16 + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (l2e ? 4 : 0)
                          : pipe ? 8 + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) : (inv ? 1 : 0) + (l2e ? 2 : 0) + (init_ks == 4 ? 4 : 0);
        {
            const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[which], 160 * 1024 - 2048);
            if (rc_lds != NFA_OK) return rc_lds;
        }
    }
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_rqs_coupling_resnet_f32(const float* inputs, const void* weights_packed,
                                           const float* bias_packed, const int32_t* layer_tables,
                                           float* outputs, float* logabsdet, int32_t* status,
                                           int64_t batch, int32_t features, int32_t num_transform,
                                           int32_t num_identity, int32_t hidden_features,
                                           int32_t num_blocks, const nfa_rqs_spec* spec, int32_t flags,
                                           void* stream) {
    return launch_resnet_layers(inputs, weights_packed, bias_packed, layer_tables, 1, outputs, logabsdet, status,
                                batch, features, num_transform, num_identity, hidden_features, num_blocks, spec,
                                flags, stream);
}

extern "C" int nfa_rqs_flow_resnet_f32(const float* inputs, const void* weights_packed,
                                       const float* bias_packed, const int32_t* flow_tables,
                                       int32_t num_layers, float* outputs, float* logabsdet,
                                       int32_t* status, int64_t batch, int32_t features,
                                       int32_t num_transform, int32_t num_identity,
                                       int32_t hidden_features, int32_t num_blocks,
                                       const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    return launch_resnet_layers(inputs, weights_packed, bias_packed, flow_tables, num_layers, outputs, logabsdet,
                                status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                                spec, flags, stream);
}

// the diagnostic instances: the same launch with the LAST layer's logits stored (include/nflows_amd.h)
extern "C" int nfa_rqs_flow_resnet_logits_f32(const float* inputs, const void* weights_packed,
                                              const float* bias_packed, const int32_t* flow_tables,
                                              int32_t num_layers, float* outputs, float* logabsdet,
                                              int32_t* status, int64_t batch, int32_t features,
                                              int32_t num_transform, int32_t num_identity,
                                              int32_t hidden_features, int32_t num_blocks,
                                              const nfa_rqs_spec* spec, int32_t flags, void* stream, float* logits) {
    if (!logits) return NFA_ERR_INVALID_ARGUMENT;
    return launch_resnet_layers(inputs, weights_packed, bias_packed, flow_tables, num_layers, outputs, logabsdet,
                                status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                                spec, flags, stream, nullptr, nullptr, 0, logits);
}

extern "C" int nfa_rqs_flow_resnet_redo_f32(const float* inputs, const void* weights_packed,
                                            const float* bias_packed, const int32_t* flow_tables,
                                            int32_t num_layers, float* outputs, float* logabsdet,
                                            const int32_t* redo_blocks, int32_t* status, int64_t batch,
                                            int32_t features, int32_t num_transform, int32_t num_identity,
                                            int32_t hidden_features, int32_t num_blocks,
                                            const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    if (!redo_blocks) return NFA_ERR_INVALID_ARGUMENT;
    return launch_resnet_layers(inputs, weights_packed, bias_packed, flow_tables, num_layers, outputs, logabsdet,
                                status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                                spec, flags, stream, redo_blocks);
}

extern "C" int nfa_rqs_flow_resnet_context_f32(const float* inputs, const float* context, int32_t context_features,
                                               const void* weights_packed, const float* bias_packed,
                                               const int32_t* flow_tables, int32_t num_layers, float* outputs,
                                               float* logabsdet, int32_t* status, int64_t batch, int32_t features,
                                               int32_t num_transform, int32_t num_identity,
                                               int32_t hidden_features, int32_t num_blocks,
                                               const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    if (context_features < 1) return NFA_ERR_INVALID_ARGUMENT;
    return launch_resnet_layers(inputs, weights_packed, bias_packed, flow_tables, num_layers, outputs, logabsdet,
                                status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                                spec, flags, stream, nullptr, context, context_features);
}

extern "C" int nfa_rqs_flow_resnet_context_redo_f32(const float* inputs, const float* context,
                                                    int32_t context_features, const void* weights_packed,
                                                    const float* bias_packed, const int32_t* flow_tables,
                                                    int32_t num_layers, float* outputs, float* logabsdet,
                                                    const int32_t* redo_blocks, int32_t* status, int64_t batch,
                                                    int32_t features, int32_t num_transform, int32_t num_identity,
                                                    int32_t hidden_features, int32_t num_blocks,
                                                    const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    if (context_features < 1 || !redo_blocks) return NFA_ERR_INVALID_ARGUMENT;
    return launch_resnet_layers(inputs, weights_packed, bias_packed, flow_tables, num_layers, outputs, logabsdet,
                                status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                                spec, flags, stream, redo_blocks, context, context_features);
}
