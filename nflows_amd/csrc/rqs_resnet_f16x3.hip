// K8x (round 6): the whole-layer kernel -- ResidualNet conditioner, nn/nets/resnet.py:55-100 (forward :92-100, blocks
// :39-52), + everything K1 replaces, coupling.py:73-130, :549-582, for a RUN of layers in one launch -- with its GEMMs
// on the f16 matrix pipe from THREE f16 pieces per fp32 operand, five cross products per multiply-add (f16x3_gemm.hpp):
// operands carried at the reference's width (nn/nets/resnet.py:92-100 is F.linear on fp32: 24-bit significands; the
// three pieces hold 33), at 5/6 of the matrix time of the three-piece bf16 kernel (K8, rqs_resnet.hip) whose
// structure this kernel keeps:
//
//   * a wave owns 32 rows for the whole run; the rows live in an LDS tile by slot, every GEMM is computed transposed
//     and chained through the register file (the accumulator tiles of one GEMM are the B operand of the next, up to
//     a column permutation the host applies to the weights);
//   * the residual stream is NOT kept in fp32: its three pieces are exact, the skip connection rebuilds the value
//     from them (two v_fma_mix_f32 per value), the first ReLU of a block is a sign mask on the pieces;
//   * weights: 12 KB stages of [4 tiles][3 pieces][64 lanes] x 16 B (k-major) / [3 pieces][4 k-steps][64 lanes]
//     (tile-major final layer) through the three-slot LDS-DMA ring of bf16x3_gemm.hpp, four waves per workgroup, two
//     workgroups per CU;
//   * the final layer is tile-major with the spline evaluation (rqs_fused8.hpp: one walk over fp32 running knot
//     sums, logits read at scale 1 / kappa straight from the accumulators) woven between its MFMAs, 40 slots per tile.
//
// What f16 pieces need that bf16 pieces did not -- range.  Every GEMM's weights are multiplied by a power of two T
// before the split (host: max |w T| in [2^13, 2^14)), activations -- the row tile's identity features included -- are
// split at scale S (a power of two, host: 16): a value keeps all its 24 bits while |v S| >= 2^-1 (the last piece then
// is >= 2^-24, f16's smallest subnormal), below that the absolute error is <= 2^-25 / S; |v S| >= 65520 overflows.
// Accumulators hold S T x (the reference's pre-activation); `scales` = per GEMM {1 / T, T}: 1 / T takes an accumulator
// to the next pieces' scale (exact), T takes the residual stream's pieces to the second Linear's accumulator scale;
// the final layer's pair is {kappa = 1 / (S T), S T} for the spline evaluation.  Overflow poisons (hi = inf, lo = -inf,
// r = NaN, and ReLU's sign mask keeps NaN): a row block with any non-finite result writes nothing and raises its entry
// of `redo`, the caller runs K8 (three bf16 pieces: full fp32 range) on the flagged blocks right behind
// (nfa_rqs_flow_resnet_redo_f32), as for K8h.
//
// Served: 8 bins, linear tails, ReLU blocks, no context, hidden width 128 (narrower: zero-padded by the host),
// d_i <= 64, d_t % 4 == 0, d_t <= 64, D % 4 == 0, D <= 128, batch % 128 == 0.  DBG instances (tests): the logits of
// the run's LAST layer (accumulators x kappa) are stored as well.

#include "f16x3_gemm.hpp"
#include "rqs_resnet_f16_kernel.hpp"   // FusedSteps' unit slicing (k8h::spline_unit_range)

#include <hip/hip_ext.h>
#include <stdlib.h>

namespace nfa {
namespace k8x {

struct Args {
    const float* x;          // [B, D]
    const vec4f* w;          // [num_layers * num_stages][768] x 16 bytes
    const float* bias;       // accumulator-order biases of all GEMMs (x the scale of their accumulators), layer after layer
    const float* scales;     // [num_layers][1 + 2 num_blocks + 1][2]
    const int32_t* tables;   // [num_layers][128] slots of the identity / transformed features, then [128] final
    float* out;
    float* lad;
    int32_t* redo;           // [batch / 128]
    int32_t* status;
    int64_t batch;
    int D, dt, di, num_blocks, num_layers, num_stages, bias_per_layer, accumulate;
    RqsDev sp;
    int normal, skip_out;
    float log_z, act_scale;
    int Ds;
    float* dbg_logits;       // DBG: [batch][dt * 24], packed row order (tile, lane-half, register)
};

// slices of an evaluation unit behind the MFMAs of a tile: the tile's 24 f16 MFMAs count one time unit each, its four bf8
// MFMAs two (64 against 32 cycles): 32 units per tile, the unit's slices spread evenly over them
template <int UNIT, int U0, int U1, class Steps>
__device__ __forceinline__ void unit_span(Steps& fa, Steps& fb, const RqsDev& sp) {
    if constexpr (UNIT != 0) {
        constexpr int N = k8h::spline_unit_slices<UNIT, Steps>();
        k8h::spline_unit_range<UNIT, (U0 * N) / 32, (U1 * N) / 32>(fa, fb, sp);
    }
}

#define NFA_K8X_PUMP_F16(U, A_, B_)                                         \
    acc = NFA_K8X_F16(A_, B_, acc);                                         \
    __builtin_amdgcn_sched_barrier(0);                                      \
    unit_span<UNIT, (U), (U) + 1>(fa, fb, sp);                              \
    __builtin_amdgcn_sched_barrier(0)
#define NFA_K8X_PUMP_BF8(U, A_, B_)                                         \
    acc = NFA_K8X_BF8(A_, B_, acc);                                         \
    __builtin_amdgcn_sched_barrier(0);                                      \
    unit_span<UNIT, (U), (U) + 2>(fa, fb, sp);                              \
    __builtin_amdgcn_sched_barrier(0)

// one stage of the final layer = four k-steps of the tile: two pairs of k-steps, each three f16 products per k-step and one
// bf8 instruction, a slice of the evaluation behind every MFMA.  Stage layout: fragments H0..H3 | L0..L3 | X01 lo, X01 hi,
// X23 lo, X23 hi.  The second pair's fragments are requested behind the MFMAs that free the first pair's registers.
template <int UNIT, int HS, class Steps>
__device__ __forceinline__ void stage_pumped(f32x16& acc, const Pieces (&p)[8], const i32x8 (&bx)[4], WeightStream& sm, int lane,
                                             Steps& fa, Steps& fb, const RqsDev& sp) {
    stream_request(sm);
    const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
    vec4f fh0 = cur[0 * 64], fl0 = cur[4 * 64], fh1 = cur[1 * 64], fl1 = cur[5 * 64];
    vec4f xa = cur[8 * 64], xb = cur[9 * 64];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const Pieces& b0 = p[HS * 4 + 2 * j];
        const Pieces& b1 = p[HS * 4 + 2 * j + 1];
        const f16x8 bh0 = __builtin_bit_cast(f16x8, b0.h), bl0 = __builtin_bit_cast(f16x8, b0.l);
        const f16x8 bh1 = __builtin_bit_cast(f16x8, b1.h), bl1 = __builtin_bit_cast(f16x8, b1.l);
        const f16x8 ah0 = __builtin_bit_cast(f16x8, fh0), al0 = __builtin_bit_cast(f16x8, fl0);
        const f16x8 ah1 = __builtin_bit_cast(f16x8, fh1), al1 = __builtin_bit_cast(f16x8, fl1);
        const i32x8 ax = join_x(xa, xb);
        const int u = HS * 16 + j * 8;   // (constant after unrolling)
        if (j == 0) {
            NFA_K8X_PUMP_F16(HS * 16 + 0, ah0, bl0);
            NFA_K8X_PUMP_F16(HS * 16 + 1, al0, bh0);
            NFA_K8X_PUMP_F16(HS * 16 + 2, ah0, bh0);
            fh0 = cur[2 * 64];
            fl0 = cur[6 * 64];
            NFA_K8X_PUMP_F16(HS * 16 + 3, ah1, bl1);
            NFA_K8X_PUMP_F16(HS * 16 + 4, al1, bh1);
            NFA_K8X_PUMP_F16(HS * 16 + 5, ah1, bh1);
            fh1 = cur[3 * 64];
            fl1 = cur[7 * 64];
            NFA_K8X_PUMP_BF8(HS * 16 + 6, ax, bx[HS * 2 + 0]);
            xa = cur[10 * 64];
            xb = cur[11 * 64];
        } else {
            NFA_K8X_PUMP_F16(HS * 16 + 8, ah0, bl0);
            NFA_K8X_PUMP_F16(HS * 16 + 9, al0, bh0);
            NFA_K8X_PUMP_F16(HS * 16 + 10, ah0, bh0);
            NFA_K8X_PUMP_F16(HS * 16 + 11, ah1, bl1);
            NFA_K8X_PUMP_F16(HS * 16 + 12, al1, bh1);
            NFA_K8X_PUMP_F16(HS * 16 + 13, ah1, bh1);
            NFA_K8X_PUMP_BF8(HS * 16 + 14, ax, bx[HS * 2 + 1]);
        }
        (void)u;
    }
    stream_advance(sm);
}
#undef NFA_K8X_PUMP_F16
#undef NFA_K8X_PUMP_BF8

template <int UNIT, class Steps>
__device__ __forceinline__ void gemm_tile_pumped(f32x16& acc, const Pieces (&p)[8], const i32x8 (&bx)[4], WeightStream& sm, int lane,
                                                 Steps& fa, Steps& fb, const RqsDev& sp) {
    stage_pumped<UNIT, 0>(acc, p, bx, sm, lane, fa, fb, sp);
    stage_pumped<UNIT, 1>(acc, p, bx, sm, lane, fa, fb, sp);
}

__device__ __forceinline__ bool not_finite(float v) { return !(__builtin_fabsf(v) < INFINITY); }

template <bool INVERSE, int INIT_KS, bool DBG = false>
__global__ void __launch_bounds__(kBlock, 2) rqs_resnet_f16x3_kernel(const Args a) {
    constexpr int kFinalRows = 24;   // 23 logits padded to 24: two features share three 32-row tiles
    constexpr int NW = kBlock / kWave;
    // dynamic LDS: the weight ring, per wave a [D][33] row tile, the final layer's biases of the current layer
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ int s_tab[2][kTabLayer];   // tables of the current and the next layer
    __shared__ int s_final[128];
    __shared__ int s_bad[NW];
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
    sm.slot = 1;  // so that the first two requests go to slots 0 and 1
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
    float* s_fbias = lds_dyn + kRing * kStageVec4 * 4 + NW * D * kRowPad;
    const int groups = dt >> 2;
    const int64_t num_quads = a.batch >> 7;
    const int gemms = 2 + 2 * a.num_blocks;   // per layer
    const float S = a.act_scale;
    int tb = 0;  // which half of s_tab holds the current layer's table

    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = (quad << 7) + (wave << 5);
        // (lane-derived values are made opaque per iteration: hoisted out of this loop they would stay live through
        //  the whole kernel and push the register allocation into scratch)
        int lane_here = lane, di = a.di;
        asm volatile("" : "+v"(lane_here), "+s"(di));
        const int half = lane_here >> 5, r = lane_here & 31;
        // ---- the wave's 32 rows: one coalesced read; slot j of the tile = input column j
        {
            const vec4f* xv = reinterpret_cast<const vec4f*>(a.x + row0 * D);
            const int nvec = D * 8;  // 32 * D / 4
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
        int quad_status = 0;
        for (int layer = 0; layer < a.num_layers; ++layer) {
            // the two workgroups resident on a CU take turns at the higher issue priority (see rqs_resnet_kernel.hpp)
            if ((layer + (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            const int* tab = s_tab[tb];
            // the next layer's table (the first one again after the last: next row block) goes to the other half
            // now; it is read only after this layer's many stage barriers
            if (tid < kTabLayer) {
                const int nl = layer + 1 < a.num_layers ? layer + 1 : 0;
                s_tab[tb ^ 1][tid] = checked(a.tables[nl * kTabLayer + tid], tid < kTabTr ? tid < a.di : tid - kTabTr < dt);
            }
            const float* bias = a.bias + (size_t)layer * a.bias_per_layer + half * 16;  // + 32 per tile
            const float* sc = a.scales + (size_t)layer * gemms * 2;                      // {1 / T, T} per GEMM (uniform)
            Pieces p[8];   // the current activations (128 k per sample) as f16 pieces at scale S

            // ---- identity features: k = ks*16 + half*8 + j
#pragma unroll
            for (int ks = 0; ks < INIT_KS; ++ks) {
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const int i0 = ks * 16 + half * 8 + j2 * 2;
                    float v0 = s_row[tab[kTabId + i0] * kRowPad + r], v1 = s_row[tab[kTabId + i0 + 1] * kRowPad + r];
                    v0 = i0 < di ? v0 : 0.0f;
                    v1 = i0 + 1 < di ? v1 : 0.0f;
                    unsigned hi, lo, rr;
                    split3_scaled(v0, v1, S, hi, lo, rr);
                    p[ks].h[j2] = hi;
                    p[ks].l[j2] = lo;
                    p[ks].r[j2] = rr;
                }
            }

            // ---- initial layer: h = W_i x + b_i   (accumulators: S T_0 h)
            {
                f32x16 h[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) load_bias_tile(h[t], bias + t * 32);
                gemm_kmajor<false, INIT_KS>(h, p, sm, lane);
                const float inv_t = sc[0];
#pragma unroll
                for (int t = 0; t < 4; ++t) tile_to_pieces<false>(h[t], inv_t, p[2 * t], p[2 * t + 1]);
            }
            bias += 128;
            sc += 2;
            {
                // (every wave has passed a stage barrier of this layer: nobody reads the previous layer's biases any
                //  more; the blocks' barriers come before the first use)
                const float* fbias = a.bias + (size_t)layer * a.bias_per_layer + 128 + 256 * a.num_blocks;
                for (int i = tid; i < dt * kFinalRows; i += kBlock) s_fbias[i] = fbias[i];
                if (a.num_blocks == 0) __syncthreads();   // without blocks the final layer follows at once
            }

            // ---- residual blocks: h += W_1 relu(W_0 relu(h) + b_0) + b_1, both Linears k-major.  The h pieces (96
            //      registers) survive the first Linear for the skip connection; u (64 accumulators) turns into the
            //      relu(u) pieces tile by tile; the skip is added into the second Linear's accumulators tile by tile.
            for (int blk = 0; blk < a.num_blocks; ++blk) {
                // Register budget (what keeps this kernel out of scratch: K8's order -- u into pieces FIRST, then the skip
                // connection -- holds the pieces of h, the pieces of relu(u) and the second Linear's accumulators at the same
                // time, 96 + 96 + 64 registers, and spilled 74 x the kernel's algorithmic bytes through HBM, profiles/r6):
                //   first Linear    pieces of h (96) + u (64)
                //   skip            v = b_1 + T h from the pieces, which die tile by tile: u (64) + v (64) + at most 96
                //   u -> pieces     relu(u) / T into q (96), u dies tile by tile: v (64) + q + what is left of u
                //   second Linear   v (64) + q (96)
                f32x16 v[4];
                Pieces q[8];
                {
                    f32x16 u[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) load_bias_tile(u[t], bias + t * 32);
                    gemm_kmajor<true, 8>(u, p, sm, lane);
                    const float t1 = sc[3];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        load_bias_tile(v[t], bias + 128 + t * 32);
                        add_pieces(v[t], 0, p[2 * t], t1);
                        add_pieces(v[t], 8, p[2 * t + 1], t1);
                    }
                    const float inv_t = sc[0];
#pragma unroll
                    for (int t = 0; t < 4; ++t) tile_to_pieces<true>(u[t], inv_t, q[2 * t], q[2 * t + 1]);
                }
                const float inv_t1 = sc[2];
                gemm_kmajor<false, 8>(v, q, sm, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t) tile_to_pieces<false>(v[t], inv_t1, p[2 * t], p[2 * t + 1]);
                bias += 256;
                sc += 4;
            }

            // ---- final layer with the spline evaluation woven into the MFMAs: the three tiles of a group hold the
            //      logits of this lane's two features A, B (A = T0 + T1[0:8], B = T1[8:16] + T2), at scale 1 / kappa
            {
                using Steps = FusedSteps<INVERSE, 8>;
                Steps fa, fb;
                const float kappa = sc[0];
                fa.kappa = fb.kappa = kappa;
                fa.kl2e = fb.kl2e = 1.44269502162933349609375f * kappa;
                fa.tail_s = fb.tail_s = a.sp.tail_logit * sc[1];
                float* slot_b = nullptr;
                const float* fbias = s_fbias + half * 16;
                // the bf8 B operands of the four k-step pairs, made once for the layer's 24 tiles (the r' pieces are read by
                // nothing else from here on: their registers are these)
                i32x8 bx[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bx[j] = bf8_operand(p[2 * j], p[2 * j + 1]);
                f32x16 acc[3];
                auto commit = [&](Steps& f, float* slot) {
                    *slot = f.y;
                    lad_acc += f.lad;
                    quad_status |= f.status;
                };
                [[maybe_unused]] auto store_logits = [&](const f32x16& t, int tile) {
                    if constexpr (DBG) {
                        if (layer == a.num_layers - 1) {
                            float* dst = a.dbg_logits + (size_t)(row0 + r) * (dt * kFinalRows) + tile * 32 + half * 16;
#pragma unroll
                            for (int q_ = 0; q_ < 16; ++q_) dst[q_] = t[q_] * kappa;
                            // (stores and LDS-DMA requests share vmcnt and complete out of order with each other: the
                            //  counted waits of the stream must not see them)
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                    }
                };
                for (int g = 0; g < groups; ++g) {
                    float* slot0 = s_row + tab[kTabTr + g * 4 + half * 2] * kRowPad + r;
                    float* slot1 = s_row + tab[kTabTr + g * 4 + half * 2 + 1] * kRowPad + r;
                    load_bias_tile(acc[0], fbias + (g * 3 + 0) * 32);
                    if (g > 0) {
                        gemm_tile_pumped<k8h::kUnitFinishB>(acc[0], p, bx, sm, lane, fa, fb, a.sp);
                        commit(fb, slot_b);
                    } else {
                        gemm_tile_pumped<0>(acc[0], p, bx, sm, lane, fa, fb, a.sp);
                    }
                    store_logits(acc[0], g * 3 + 0);
                    fa.x = *slot0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fa.ew[j] = acc[0][j];
                        fa.eh[j] = acc[0][8 + j];
                    }
                    load_bias_tile(acc[1], fbias + (g * 3 + 1) * 32);
                    gemm_tile_pumped<k8h::kUnitNumA>(acc[1], p, bx, sm, lane, fa, fb, a.sp);
                    store_logits(acc[1], g * 3 + 1);
                    fb.x = *slot1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (j < 7) fa.sd[j] = acc[1][j];
                        fb.ew[j] = acc[1][8 + j];
                    }
                    load_bias_tile(acc[2], fbias + (g * 3 + 2) * 32);
                    gemm_tile_pumped<k8h::kUnitFinishA>(acc[2], p, bx, sm, lane, fa, fb, a.sp);
                    store_logits(acc[2], g * 3 + 2);
                    commit(fa, slot0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fb.eh[j] = acc[2][j];
                        if (j < 7) fb.sd[j] = acc[2][8 + j];
                    }
                    slot_b = slot1;
                }
                k8h::spline_unit_range<k8h::kUnitFinishB, 0, k8h::spline_unit_slices<k8h::kUnitFinishB, Steps>()>(fa, fb, a.sp);
                commit(fb, slot_b);
            }
            tb ^= 1;
            // this wave's spline results must be visible to its own gathers of the next layer
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // ---- results: position p of a row comes from slot final[p].  A block with any non-finite value (f16 range
        //      exceeded somewhere, or non-finite inputs) is not written at all: the exact kernel redoes it.
        lad_acc += __shfl_xor(lad_acc, 32, kWave);
        const float sumsq = tile_row_sumsq(s_row, a.Ds, half, r);
        const bool bad = not_finite(lad_acc) || not_finite(sumsq);
        const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) s_bad[wave] = wave_bad ? 1 : 0;
        // stores and LDS-DMA requests complete out of order with each other: drain before the ordinary stores, and
        // before the next row block counts outstanding requests again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int any_bad = 0;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) any_bad |= s_bad[w_];
        if (!any_bad) {
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
            if (half == 0) {
                float* dst = a.lad + row0 + r;
                float v = a.accumulate ? *dst + lad_acc : lad_acc;
                if (a.normal) v = (-0.5f * sumsq - a.log_z) + v;   // normal.py:31-33, flows/base.py:49
                *dst = v;
            }
            my_status |= quad_status;
        }
        if (tid == 0) a.redo[quad] = any_bad ? 1 : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // s_bad is rewritten by the next row block
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two stages requested past the end
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace k8x
}  // namespace nfa

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
    if (a.sp.K != 8 || !a.sp.linear || hidden_features != 128 || (num_transform & 3) != 0 || num_transform > 64 ||
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
    a.num_stages = init_ks + 16 * num_blocks + 2 * (num_transform * 24 / 32);
    a.bias_per_layer = 128 + 256 * num_blocks + num_transform * 24;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    const size_t lds = (size_t)kRing * kStageVec4 * 16 + (size_t)(kBlock / kWave) * features * kRowPad * sizeof(float) +
                       (size_t)num_transform * 24 * sizeof(float);
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
    const int which = (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (dbg_logits ? 4 : 0);
    switch (which) {
        case 0: kern = k8x::rqs_resnet_f16x3_kernel<false, 2>; break;
        case 1: kern = k8x::rqs_resnet_f16x3_kernel<true, 2>; break;
        case 2: kern = k8x::rqs_resnet_f16x3_kernel<false, 4>; break;
        case 3: kern = k8x::rqs_resnet_f16x3_kernel<true, 4>; break;
        case 4: kern = k8x::rqs_resnet_f16x3_kernel<false, 2, true>; break;
        case 5: kern = k8x::rqs_resnet_f16x3_kernel<true, 2, true>; break;
        case 6: kern = k8x::rqs_resnet_f16x3_kernel<false, 4, true>; break;
        default: kern = k8x::rqs_resnet_f16x3_kernel<true, 4, true>; break;
    }
    note_layer_kernel("k8x::rqs_resnet_f16x3_kernel<inverse=%d, init_ks=%d, K=8, dbg=%d>", inv ? 1 : 0, init_ks, dbg_logits ? 1 : 0);
    if (lds > 64 * 1024) {
        static unsigned long long raised[8] = {};   // device masks (raise_dynamic_lds)
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
