// K11: a whole affine / additive coupling layer in ONE kernel, a run of such layers in one launch --
// the MLP conditioner (nn/nets/mlp.py:47-68: Linear(d_i -> 128), [ReLU, Linear(128 -> 128)] x n,
// ReLU, Linear(128 -> 2 d_t)) followed by everything K2 replaces (coupling.py:73-130, :212-269):
// split, scale activation, x * scale + shift (or its inverse), the log-determinant row sum, scatter,
// and the neighbouring permutations.
//
// Same skeleton as K8 (rqs_resnet.hip), same GEMM machinery (bf16x3_gemm.hpp): a wave owns 32 samples
// for the whole run, their rows live in a wave-private LDS tile indexed by slot, activations stay in
// registers as three bf16 pieces (fp32-accurate products on the bf16 matrix pipe, full fp32 range: no
// second pass needed), weights arrive through the LDS-DMA ring.  What differs:
//   * ReLU sits between the Linears (no skip connections): every GEMM applies it to its input pieces
//     on the fly, the accumulators are converted to pieces unchanged.
//   * The last Linear has 2 d_t rows (d_t for the additive layer).  The host orders them so that the
//     16 accumulator values of a lane-half are [shift of 8 features | unconstrained scale of the same 8
//     features] (affine) or [shift of 16 features] (additive): tile t covers features 16 t .. 16 t + 15
//     (32 t .. for additive), padded with zero rows.
//   * Per element the arithmetic is K2's, instruction for instruction (`scale_of`, logf, the IEEE
//     division of the inverse), so the only difference to "MLP by GEMMs, then K2" is the rounding of
//     the GEMM sums.
//
// Restrictions (the host takes GEMMs + K2 otherwise): hidden width 128 in every hidden layer, ReLU,
// d_i <= 64, d_t <= 64, D % 4 == 0, D <= 128, batch % 128 == 0 here (leftover rows: other path),
// scale activation default / general / additive.

// Round 6: every GEMM keeps its leading product in an accumulator of its own (NFA_MFMA6_SPLIT, fused_common.hpp): the additive
// coupling has no scale and no logarithm, its whole error is the rounding of the conditioners' GEMM sums, and with one
// accumulator for all six products that error was 2.4 x the reference CPU path's (tests/test_gpu_realnvp.py held it to 3 x
// in round 5).  The second set of accumulators is why the kernel asks for one workgroup per CU (512 registers per wave).
#define NFA_BF16X3_SPLIT_ACC
#include "bf16x3_gemm.hpp"

#include <hip/hip_ext.h>
#include <math.h>
#include <stdlib.h>

namespace nfa {

struct AffineMlpArgs {
    const float* x;         // [B, D]
    const vec4f* w;         // [num_layers * stages_per_layer][768] x 16 bytes
    const float* bias;      // accumulator-order biases of all GEMMs, layer after layer
    const int32_t* tables;  // [num_layers][128] slots of the identity / transformed features, then [128] final
    float* out;
    float* lad;
    int32_t* status;
    int64_t batch;          // multiple of 128
    int D, dt, di, num_hidden, num_layers, num_stages, bias_per_layer, accumulate, activation;
    int final_tiles;
    int normal, skip_out;   // NFA_FLAG_STANDARD_NORMAL_LOG_PROB / NFA_FLAG_SKIP_OUTPUTS
    float log_z;
    int Ds;                // columns the density sums over (features minus NFA_FLAG_PAD_COLUMNS)
};

// RESNET (round 5): the conditioner is a ResidualNet (nn/nets/resnet.py:55-100, the conditioner of the reference's own
// SimpleRealNVP, flows/realnvp.py:44-71) -- the same stream of stages (initial Linear, 2 x num_blocks hidden Linears,
// output tiles), other arithmetic between them: no activation behind the initial layer, every block computes
// h + W_1 relu(W_0 relu(h) + b_0) + b_1 (resnet.py:39-52) as in K8 (rqs_resnet_kernel.hpp), the output layer takes h itself.
template <bool INVERSE, int INIT_KS, bool ADDITIVE, bool RESNET = false>
__global__ void __launch_bounds__(kBlock, 1) affine_mlp_kernel(const AffineMlpArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ int s_tab[2][kTabLayer];
    __shared__ int s_final[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, dt = a.dt;
    int my_status = 0;
    auto checked = [&](int v, bool used) {
        if (used && (v < 0 || v >= D)) my_status |= NFA_STATUS_BAD_INDEX;
        return v < 0 ? 0 : (v >= D ? D - 1 : v);
    };
    if (tid < kTabLayer) {
        s_tab[0][tid] = checked(a.tables[tid], tid < kTabTr ? tid < a.di : tid - kTabTr < dt);
        s_final[tid] = checked(a.tables[a.num_layers * kTabLayer + tid], tid < D);
    }

    WeightStream sm;
    sm.w = a.w;
    sm.ring = reinterpret_cast<vec4f*>(lds_dyn);
    sm.slot = 1;
    sm.fetch = 0;
    sm.num_stages = a.num_stages * a.num_layers;
    sm.tid = tid;
    stream_request(sm);  // stage 0 -> slot 0
    sm.slot = 2;
    stream_request(sm);  // stage 1 -> slot 1
    sm.slot = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float* s_row = lds_dyn + kRing * kStageVec4 * 4 + wave * D * kRowPad;
    const int64_t num_quads = a.batch >> 7;
    int tb = 0;
    constexpr int kPerTile = ADDITIVE ? 16 : 8;   // features of a lane-half per final tile

    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = (quad << 7) + (wave << 5);
        int lane_here = lane, di = a.di;
        asm volatile("" : "+v"(lane_here), "+s"(di));
        const int half = lane_here >> 5, r = lane_here & 31;
        // ---- the wave's 32 rows: one coalesced read; slot j of the tile = input column j
        {
            const vec4f* xv = reinterpret_cast<const vec4f*>(a.x + row0 * D);
            const int nvec = D * 8;
            for (int e0 = lane; e0 < nvec; e0 += kWave * 4) {
                vec4f v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * kWave;
                    v[u] = xv[e < nvec ? e : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * kWave;
                    if (e < nvec) {
                        const int rr = (e * 4) / D, c0 = e * 4 - rr * D;
                        s_row[(c0 + 0) * kRowPad + rr] = v[u].x;
                        s_row[(c0 + 1) * kRowPad + rr] = v[u].y;
                        s_row[(c0 + 2) * kRowPad + rr] = v[u].z;
                        s_row[(c0 + 3) * kRowPad + rr] = v[u].w;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        float lad_acc = 0.0f;
        for (int layer = 0; layer < a.num_layers; ++layer) {
            if ((layer + (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            const int* tab = s_tab[tb];
            // the next layer's table goes to the other half now (read after this layer's stage barriers).
            // A SCALAR branch (the first two waves, whole): with `tid < kTabLayer` as a per-lane condition hipcc
            // (ROCm 7.2) put the spill store of `lad_acc` into the join block IN FRONT of the exec restore -- waves 2
            // and 3 arrive there with exec = 0, never stored it and lost the log-determinants of all layers but the
            // last (RESNET instances, round 5; tests/test_host_logic.py::test_no_spill_between_a_join_and_its_exec_restore).
            static_assert(kTabLayer % kWave == 0, "the table is copied by whole waves");
            if (__builtin_amdgcn_readfirstlane(wave) < kTabLayer / kWave) {
                const int nl = layer + 1 < a.num_layers ? layer + 1 : 0;
                s_tab[tb ^ 1][tid] = checked(a.tables[nl * kTabLayer + tid], tid < kTabTr ? tid < a.di : tid - kTabTr < dt);
            }
            const float* bias = a.bias + (size_t)layer * a.bias_per_layer + half * 16;  // + 32 per tile
            bf16x8 ph[8], pm[8], pl[8];  // the current activations (128 k per sample) as bf16 pieces

            // ---- identity features: k = ks*16 + half*8 + j
#pragma unroll
            for (int ks = 0; ks < INIT_KS; ++ks) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = ks * 16 + half * 8 + j;
                    const float xv = s_row[tab[kTabId + i] * kRowPad + r];
                    v[j] = i < di ? xv : 0.0f;
                }
                bf16x2 hh[4], mm[4], ll[4];
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) split3(vec2f{v[j2 * 2], v[j2 * 2 + 1]}, hh[j2], mm[j2], ll[j2]);
                ph[ks] = join4(hh[0], hh[1], hh[2], hh[3]);
                pm[ks] = join4(mm[0], mm[1], mm[2], mm[3]);
                pl[ks] = join4(ll[0], ll[1], ll[2], ll[3]);
            }

            // ---- input layer: h = W_0 x + b_0 (its ReLU is applied by the next GEMM)
            {
                f32x16 h[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) load_bias_tile(h[t], bias + t * 32);
                gemm_kmajor<false, INIT_KS>(h, ph, pm, pl, sm, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    tile_to_pieces<false>(h[t], ph[2 * t], pm[2 * t], pl[2 * t], ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
            }
            bias += 128;

            if constexpr (RESNET) {
                // ---- residual blocks: h += W_1 relu(W_0 relu(h) + b_0) + b_1 (the register budget of K8's blocks: the
                //      pieces of h survive the first Linear for the skip connection, which goes into the second
                //      Linear's accumulators tile by tile)
                for (int blk = 0; blk < (a.num_hidden >> 1); ++blk) {
                    bf16x8 qh[8], qm[8], ql[8];
                    {
                        f32x16 u[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) load_bias_tile(u[t], bias + t * 32);
                        gemm_kmajor<true, 8>(u, ph, pm, pl, sm, lane);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            tile_to_pieces<true>(u[t], qh[2 * t], qm[2 * t], ql[2 * t], qh[2 * t + 1], qm[2 * t + 1], ql[2 * t + 1]);
                    }
                    f32x16 v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        load_bias_tile(v[t], bias + 128 + t * 32);
                        add_pieces(v[t], 0, ph[2 * t], pm[2 * t], pl[2 * t]);
                        add_pieces(v[t], 8, ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
                    }
                    gemm_kmajor<false, 8>(v, qh, qm, ql, sm, lane);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        tile_to_pieces<false>(v[t], ph[2 * t], pm[2 * t], pl[2 * t], ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
                    bias += 256;
                }
            } else {
            // ---- hidden layers: h = W relu(h) + b
            for (int hl = 0; hl < a.num_hidden; ++hl) {
                f32x16 u[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) load_bias_tile(u[t], bias + t * 32);
                gemm_kmajor<true, 8>(u, ph, pm, pl, sm, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    tile_to_pieces<false>(u[t], ph[2 * t], pm[2 * t], pl[2 * t], ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
                bias += 128;
            }
            }

            // ---- output layer, one 32-row tile at a time, and the affine map of the tile's features;
            //      the results replace the inputs in their slots
            for (int t = 0; t < a.final_tiles; ++t) {
                f32x16 acc;
                load_bias_tile(acc, bias + t * 32);
                gemm_tile<!RESNET>(acc, ph, pm, pl, sm, lane);   // (MLP: ReLU in front of the output layer; ResidualNet: h itself)
#pragma unroll
                for (int j = 0; j < kPerTile; ++j) {
                    const int f = (t * 2 + half) * kPerTile + j;
                    if (f < dt) {
                        float* slot = s_row + tab[kTabTr + f] * kRowPad + r;
                        const float xin = *slot;
                        const float shift = acc[j];
                        float y;
                        if (ADDITIVE) {
                            y = INVERSE ? xin - shift : xin + shift;   // scale == 1: exact, logabsdet 0
                        } else {
                            float l;
                            affine_element<INVERSE>(xin, shift, scale_of(acc[8 + j], a.activation), y, l);
                            lad_acc += l;
                        }
                        *slot = y;
                    }
                }
            }
            tb ^= 1;
            // this wave's results must be visible to its own gathers of the next layer
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // ---- output rows: position p of a row comes from slot final[p]; 16 bytes per lane per store
        if (!a.skip_out) {
            vec4f* ov = reinterpret_cast<vec4f*>(a.out + row0 * D);
            const int nvec = D * 8;
            for (int e = lane; e < nvec; e += kWave) {
                const int rr = (e * 4) / D, c0 = e * 4 - rr * D;
                vec4f v;
                v.x = s_row[s_final[c0 + 0] * kRowPad + rr];
                v.y = s_row[s_final[c0 + 1] * kRowPad + rr];
                v.z = s_row[s_final[c0 + 2] * kRowPad + rr];
                v.w = s_row[s_final[c0 + 3] * kRowPad + rr];
                ov[e] = v;
            }
        }
        lad_acc += __shfl_xor(lad_acc, 32, kWave);
        float sumsq = 0.0f;
        if (a.normal) sumsq = tile_row_sumsq(s_row, a.Ds, half, r);
        if (half == 0) {
            float* dst = a.lad + row0 + r;
            float v = a.accumulate ? *dst + lad_acc : lad_acc;
            if (a.normal) v = (-0.5f * sumsq - a.log_z) + v;   // normal.py:31-33, flows/base.py:49
            *dst = v;
        }
        // stores and LDS-DMA requests complete out of order with each other: drain before the next
        // row block counts outstanding requests again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two stages requested past the end
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_affine_flow_mlp_f32(const float* inputs, const void* weights_packed, const float* bias_packed,
                                       const int32_t* tables, int32_t num_layers, float* outputs,
                                       float* logabsdet, int32_t* status, int64_t batch, int32_t features,
                                       int32_t num_transform, int32_t num_identity, int32_t hidden_features,
                                       int32_t num_hidden_layers, int32_t scale_activation, int32_t flags,
                                       void* stream) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET | NFA_FLAG_STANDARD_NORMAL_LOG_PROB |
                  NFA_FLAG_SKIP_OUTPUTS | NFA_FLAG_PAD_COLUMNS_MASK | NFA_FLAG_RESIDUAL_BLOCKS))
        return NFA_ERR_INVALID_ARGUMENT;
    const bool resnet = (flags & NFA_FLAG_RESIDUAL_BLOCKS) != 0;
    flags &= ~NFA_FLAG_RESIDUAL_BLOCKS;
    if (!density_flags_valid(flags)) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1 || num_transform < 1 || num_identity < 1 ||
        num_transform + num_identity > features || num_hidden_layers < 0 || num_layers < 1 ||
        (resnet && (num_hidden_layers & 1)))   // (residual blocks: two Linears each)
        return NFA_ERR_INVALID_ARGUMENT;
    if (scale_activation != NFA_SCALE_DEFAULT && scale_activation != NFA_SCALE_GENERAL &&
        scale_activation != NFA_SCALE_ADDITIVE)
        return NFA_ERR_UNSUPPORTED;
    if (hidden_features != 128 || num_transform > 64 || num_identity > 64 || features > 128 || (features & 3) != 0 ||
        (batch & 127) != 0 || num_hidden_layers > 64 || num_layers > 4096)
        return NFA_ERR_UNSUPPORTED;
    if (batch == 0) return NFA_OK;
    if (!inputs || !weights_packed || !bias_packed || !tables || !logabsdet ||
        (!outputs && !(flags & NFA_FLAG_SKIP_OUTPUTS)))
        return NFA_ERR_INVALID_ARGUMENT;
    const bool additive = scale_activation == NFA_SCALE_ADDITIVE;
    AffineMlpArgs a;
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
    a.num_hidden = num_hidden_layers;
    a.num_layers = num_layers;
    a.activation = scale_activation;
    a.final_tiles = additive ? (num_transform + 31) / 32 : (num_transform + 15) / 16;
    const int init_ks = num_identity > 32 ? 4 : 2;
    a.num_stages = init_ks + 8 * num_hidden_layers + 2 * a.final_tiles;
    a.bias_per_layer = 128 + 128 * num_hidden_layers + 32 * a.final_tiles;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    a.normal = (flags & NFA_FLAG_STANDARD_NORMAL_LOG_PROB) ? 1 : 0;
    a.skip_out = (flags & NFA_FLAG_SKIP_OUTPUTS) ? 1 : 0;
    a.Ds = density_columns(flags, features);
    if (a.Ds < 1) return NFA_ERR_INVALID_ARGUMENT;
    a.log_z = standard_normal_log_z(a.Ds);
    const size_t lds = (size_t)kRing * kStageVec4 * 16 + (size_t)(kBlock / kWave) * features * kRowPad * sizeof(float);
    int64_t blocks = batch >> 7;
    const int64_t cap = (int64_t)device_cu_count();   // (one workgroup per CU: 512 registers per wave)
    if (blocks > cap) blocks = cap;
    const bool inv = (flags & NFA_FLAG_INVERSE) != 0;
    void (*kern)(const AffineMlpArgs) = nullptr;
    const int which = (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (additive ? 4 : 0) + (resnet ? 8 : 0);
    switch (which) {
        case 8: kern = affine_mlp_kernel<false, 2, false, true>; break;
        case 9: kern = affine_mlp_kernel<true, 2, false, true>; break;
        case 10: kern = affine_mlp_kernel<false, 4, false, true>; break;
        case 11: kern = affine_mlp_kernel<true, 4, false, true>; break;
        case 12: kern = affine_mlp_kernel<false, 2, true, true>; break;
        case 13: kern = affine_mlp_kernel<true, 2, true, true>; break;
        case 14: kern = affine_mlp_kernel<false, 4, true, true>; break;
        case 15: kern = affine_mlp_kernel<true, 4, true, true>; break;
        case 0: kern = affine_mlp_kernel<false, 2, false>; break;
        case 1: kern = affine_mlp_kernel<true, 2, false>; break;
        case 2: kern = affine_mlp_kernel<false, 4, false>; break;
        case 3: kern = affine_mlp_kernel<true, 4, false>; break;
        case 4: kern = affine_mlp_kernel<false, 2, true>; break;
        case 5: kern = affine_mlp_kernel<true, 2, true>; break;
        case 6: kern = affine_mlp_kernel<false, 4, true>; break;
        default: kern = affine_mlp_kernel<true, 4, true>; break;
    }
    if (resnet) note_layer_kernel("affine_mlp_kernel<inverse=%d, init_ks=%d, additive=%d, resnet=1>", inv ? 1 : 0, init_ks, additive ? 1 : 0);
    else note_layer_kernel("affine_mlp_kernel<inverse=%d, init_ks=%d, additive=%d>", inv ? 1 : 0, init_ks, additive ? 1 : 0);
    if (lds > 64 * 1024) {
        static unsigned long long raised[16] = {};   // device masks (raise_dynamic_lds)
        {
            const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[which], 160 * 1024 - 2048);
            if (rc_lds != NFA_OK) return rc_lds;
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    profile_next_launch(&e0, &e1);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(kBlock);
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
