// The kernel of rqs_resnet_f16x3.hip (K8x) as a header: the template is instantiated in several translation units
// (rqs_resnet_f16x3.hip: 8 bins and the diagnostic instances; rqs_resnet_f16x3_bins_{a,b}.hip: the other bin counts) so that
// the instances compile side by side.  Design notes: rqs_resnet_f16x3.hip, f16x3_gemm.hpp.
#pragma once

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

// What runs behind the MFMAs of a final-layer tile: a tile's 24 f16 MFMAs count one time unit each, its four bf8 MFMAs two
// (64 against 32 cycles): 32 units per tile, a weave's slices spread evenly over them.
struct NoWeave {
    template <int U0, int U1>
    __device__ __forceinline__ void span() {}
};
// 8 bins: the three evaluation units of K8h's two-features-per-three-tiles scheme (rqs_resnet_f16_kernel.hpp)
template <int UNIT, class Steps>
struct UnitWeave {
    Steps &fa, &fb;
    const RqsDev& sp;
    template <int U0, int U1>
    __device__ __forceinline__ void span() {
        constexpr int N = k8h::spline_unit_slices<UNIT, Steps>();
        k8h::spline_unit_range<UNIT, (U0 * N) / 32, (U1 * N) / 32>(fa, fb, sp);
    }
};
// any other bin count: the slice sequence MASK names (numerators of the widths / heights, the rest), one feature per lane-half
template <int MASK, class Steps>
struct SeqWeave {
    Steps& f;
    const RqsDev& sp;
    template <int U0, int U1>
    __device__ __forceinline__ void span() {
        constexpr int N = k8h::spline_seq_count<MASK, Steps>();
        k8h::spline_seq_range<MASK, (U0 * N) / 32, (U1 * N) / 32>(f, sp);
    }
};

#define NFA_K8X_PUMP_F16(U, A_, B_)                                         \
    acc = NFA_K8X_F16(A_, B_, acc);                                         \
    __builtin_amdgcn_sched_barrier(0);                                      \
    w.template span<(U), (U) + 1>();                                        \
    __builtin_amdgcn_sched_barrier(0)
#define NFA_K8X_PUMP_BF8(U, A_, B_)                                         \
    acc = NFA_K8X_BF8(A_, B_, acc);                                         \
    __builtin_amdgcn_sched_barrier(0);                                      \
    w.template span<(U), (U) + 2>();                                        \
    __builtin_amdgcn_sched_barrier(0)

// one stage of the final layer = four k-steps of the tile: two pairs of k-steps, each three f16 products per k-step and one
// bf8 instruction, a slice of the evaluation behind every MFMA.  Stage layout: fragments [H0, L0, H1, L1][H2, L2, H3, L3]
// [X01 lo, X01 hi, X23 lo, X23 hi].  The first pair's f16 fragments arrive in `lead` (read behind the previous stage's
// barrier), the second pair's are requested behind the MFMAs that free the first pair's registers, and the stage's own
// barrier stands in front of its last bf8 instruction: the next stage's lead fragments land while that one runs.
template <int HS, class W>
__device__ __forceinline__ void stage_pumped(f32x16& acc, const Pieces (&p)[8], const i32x8 (&bx)[4], WeightStream& sm, int lane,
                                             Lead& lead, W& w) {
    stream_request(sm);
    const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
    vec4f xa = cur[8 * 64], xb = cur[9 * 64];
    vec4f fh0 = lead.h0, fl0 = lead.l0, fh1 = lead.h1, fl1 = lead.l1;
    {
        const Pieces& b0 = p[HS * 4 + 0];
        const Pieces& b1 = p[HS * 4 + 1];
        const f16x8 bh0 = __builtin_bit_cast(f16x8, b0.h), bl0 = __builtin_bit_cast(f16x8, b0.l);
        const f16x8 bh1 = __builtin_bit_cast(f16x8, b1.h), bl1 = __builtin_bit_cast(f16x8, b1.l);
        const f16x8 ah0 = __builtin_bit_cast(f16x8, fh0), al0 = __builtin_bit_cast(f16x8, fl0);
        const f16x8 ah1 = __builtin_bit_cast(f16x8, fh1), al1 = __builtin_bit_cast(f16x8, fl1);
        const i32x8 ax = join_x(xa, xb);
        NFA_K8X_PUMP_F16(HS * 16 + 0, ah0, bl0);
        NFA_K8X_PUMP_F16(HS * 16 + 1, al0, bh0);
        NFA_K8X_PUMP_F16(HS * 16 + 2, ah0, bh0);
        fh0 = cur[4 * 64];
        fl0 = cur[5 * 64];
        NFA_K8X_PUMP_F16(HS * 16 + 3, ah1, bl1);
        NFA_K8X_PUMP_F16(HS * 16 + 4, al1, bh1);
        NFA_K8X_PUMP_F16(HS * 16 + 5, ah1, bh1);
        fh1 = cur[6 * 64];
        fl1 = cur[7 * 64];
        NFA_K8X_PUMP_BF8(HS * 16 + 6, ax, bx[HS * 2 + 0]);
        xa = cur[10 * 64];
        xb = cur[11 * 64];
    }
    {
        const Pieces& b0 = p[HS * 4 + 2];
        const Pieces& b1 = p[HS * 4 + 3];
        const f16x8 bh0 = __builtin_bit_cast(f16x8, b0.h), bl0 = __builtin_bit_cast(f16x8, b0.l);
        const f16x8 bh1 = __builtin_bit_cast(f16x8, b1.h), bl1 = __builtin_bit_cast(f16x8, b1.l);
        const f16x8 ah0 = __builtin_bit_cast(f16x8, fh0), al0 = __builtin_bit_cast(f16x8, fl0);
        const f16x8 ah1 = __builtin_bit_cast(f16x8, fh1), al1 = __builtin_bit_cast(f16x8, fl1);
        const i32x8 ax = join_x(xa, xb);
        NFA_K8X_PUMP_F16(HS * 16 + 8, ah0, bl0);
        NFA_K8X_PUMP_F16(HS * 16 + 9, al0, bh0);
        NFA_K8X_PUMP_F16(HS * 16 + 10, ah0, bh0);
        NFA_K8X_PUMP_F16(HS * 16 + 11, ah1, bl1);
        NFA_K8X_PUMP_F16(HS * 16 + 12, al1, bh1);
        NFA_K8X_PUMP_F16(HS * 16 + 13, ah1, bh1);
        stream_advance(sm);      // every read of this stage has landed in registers; the next stage is complete
        lead = read_lead(sm.ring + sm.slot * kStageVec4 + lane);
        __builtin_amdgcn_sched_barrier(0);
        NFA_K8X_PUMP_BF8(HS * 16 + 14, ax, bx[HS * 2 + 1]);
    }
}
#undef NFA_K8X_PUMP_F16
#undef NFA_K8X_PUMP_BF8

template <class W>
__device__ __forceinline__ void gemm_tile_pumped(f32x16& acc, const Pieces (&p)[8], const i32x8 (&bx)[4], WeightStream& sm, int lane,
                                                 Lead& lead, W&& w) {
    stage_pumped<0>(acc, p, bx, sm, lane, lead, w);
    stage_pumped<1>(acc, p, bx, sm, lane, lead, w);
}

// tiles TI .. T - 1 of a group of the general final layer (K8h's any_group_tiles on this kernel's GEMM): biases, the tile's
// MFMAs with their share of the evaluation, the tile's sixteen logits into the evaluation's arrays
template <int TI, int T, int KB, class Steps>
__device__ __forceinline__ void any_group_tiles(f32x16& acc, Steps& f, const float* gb, const Pieces (&p)[8], const i32x8 (&bx)[4],
                                                WeightStream& sm, int lane, Lead& lead, const RqsDev& sp) {
    if constexpr (TI < T) {
        constexpr int kMask = k8h::any_tile_mask(KB, TI);
        load_bias_tile(acc, gb + TI * 32);
        if constexpr (kMask != 0) gemm_tile_pumped(acc, p, bx, sm, lane, lead, SeqWeave<kMask, Steps>{f, sp});
        else gemm_tile_pumped(acc, p, bx, sm, lane, lead, NoWeave{});
        k8h::take_chunk<TI, KB>(f, acc);
        any_group_tiles<TI + 1, T, KB, Steps>(acc, f, gb, p, bx, sm, lane, lead, sp);
    }
}

__device__ __forceinline__ bool not_finite(float v) { return !(__builtin_fabsf(v) < INFINITY); }

template <bool INVERSE, int INIT_KS, bool DBG = false, int KB = 8>
__global__ void __launch_bounds__(kBlock, 2) rqs_resnet_f16x3_kernel(const Args a) {
    static_assert(!DBG || KB == 8, "the diagnostic instances: 8 bins");
    // rows of the final layer per transformed feature: 8 bins: 23 logits padded to 24, two features share three 32-row tiles;
    // otherwise 3 K - 1 padded to whole 16-row lane-half shares (one feature per lane-half and group of T tiles)
    constexpr int kFinalRows = KB == 8 ? 24 : 16 * ((3 * KB - 1 + 15) / 16);
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
        my_status |= (used && (v < 0 || v >= D)) ? NFA_STATUS_BAD_INDEX : 0;
        return v < 0 ? 0 : (v >= D ? D - 1 : v);
    };
    // (every thread takes part -- entry tid mod 128, the two halves of the workgroup write the same values: a divergent
    //  region here or at the head of the layer loop is where hipcc put this kernel's spill stores in FRONT of the region's
    //  exec restore -- the d_i > 32 instances lost the running log-determinant and status of waves 2, 3 that way, round 6;
    //  tests/test_host_logic.py scans every translation unit's assembly for the pattern)
    const int te = tid & (kTabLayer - 1);
    s_tab[0][te] = checked(a.tables[te], te < kTabTr ? te < a.di : te - kTabTr < dt);
    s_final[te] = checked(a.tables[a.num_layers * kTabLayer + te], te < D);

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
            {
                const int nl = layer + 1 < a.num_layers ? layer + 1 : 0;
                s_tab[tb ^ 1][te] = checked(a.tables[nl * kTabLayer + te], te < kTabTr ? te < a.di : te - kTabTr < dt);
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
            if constexpr (KB == 8) {
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
                // (the first stage's lead fragments: the final layer's 48 stages read ahead of one another from here on; the
                //  last one reads the next layer's first stage, which nobody uses -- the next GEMM reads its own)
                Lead lead = read_lead(sm.ring + sm.slot * kStageVec4 + lane);
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
                UnitWeave<k8h::kUnitNumA, Steps> w0{fa, fb, a.sp};
                UnitWeave<k8h::kUnitFinishA, Steps> w1{fa, fb, a.sp};
                UnitWeave<k8h::kUnitFinishB, Steps> w2{fa, fb, a.sp};
                for (int g = 0; g < groups; ++g) {
                    float* slot0 = s_row + tab[kTabTr + g * 4 + half * 2] * kRowPad + r;
                    float* slot1 = s_row + tab[kTabTr + g * 4 + half * 2 + 1] * kRowPad + r;
                    load_bias_tile(acc[0], fbias + (g * 3 + 0) * 32);
                    if (g > 0) {
                        gemm_tile_pumped(acc[0], p, bx, sm, lane, lead, w2);
                        commit(fb, slot_b);
                    } else {
                        gemm_tile_pumped(acc[0], p, bx, sm, lane, lead, NoWeave{});
                    }
                    store_logits(acc[0], g * 3 + 0);
                    fa.x = *slot0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fa.ew[j] = acc[0][j];
                        fa.eh[j] = acc[0][8 + j];
                    }
                    load_bias_tile(acc[1], fbias + (g * 3 + 1) * 32);
                    gemm_tile_pumped(acc[1], p, bx, sm, lane, lead, w0);
                    store_logits(acc[1], g * 3 + 1);
                    fb.x = *slot1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (j < 7) fa.sd[j] = acc[1][j];
                        fb.ew[j] = acc[1][8 + j];
                    }
                    load_bias_tile(acc[2], fbias + (g * 3 + 2) * 32);
                    gemm_tile_pumped(acc[2], p, bx, sm, lane, lead, w1);
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
            else {
                // ---- any other bin count (K8h's general scheme): T tiles per group of two features, one per lane-half; the
                //      lane-half's 16 T accumulator values are the feature's 3 K - 1 logits, then padding
                constexpr int T = kFinalRows / 16;
                static_assert(T >= 1 && T <= 6, "2 .. 32 bins");
                using Steps = FusedSteps<INVERSE, KB>;
                constexpr int kRest = k8h::any_rest_mask(KB);   // what is left for the next group's first tile
                Steps f;
                const float kappa = sc[0];
                f.kappa = kappa;
                f.kl2e = 1.44269502162933349609375f * kappa;
                f.tail_s = a.sp.tail_logit * sc[1];
                const float* fbias = s_fbias + half * 16;
                i32x8 bx[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bx[j] = bf8_operand(p[2 * j], p[2 * j + 1]);
                Lead lead = read_lead(sm.ring + sm.slot * kStageVec4 + lane);
                const int groups_any = dt >> 1;
                f32x16 acc;
                float* slot = s_row + tab[kTabTr + half] * kRowPad + r;
                load_bias_tile(acc, fbias);
                gemm_tile_pumped(acc, p, bx, sm, lane, lead, NoWeave{});
                for (int g = 0; g < groups_any; ++g) {
                    const float* gb = fbias + g * T * 32;
                    f.x = *slot;
                    k8h::take_chunk<0, KB>(f, acc);
                    any_group_tiles<1, T, KB, Steps>(acc, f, gb, p, bx, sm, lane, lead, a.sp);
                    if (g + 1 < groups_any) {
                        float* next_slot = s_row + tab[kTabTr + (g + 1) * 2 + half] * kRowPad + r;
                        load_bias_tile(acc, gb + T * 32);
                        gemm_tile_pumped(acc, p, bx, sm, lane, lead, SeqWeave<kRest, Steps>{f, a.sp});
                        *slot = f.y;
                        slot = next_slot;
                    } else {
                        k8h::spline_seq_range<kRest, 0, k8h::spline_seq_count<kRest, Steps>()>(f, a.sp);
                        *slot = f.y;
                    }
                    lad_acc += f.lad;
                    quad_status |= f.status;
                }
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


namespace nfa {
namespace k8x {
typedef void (*KernelFn)(const Args);
// the instances of the other translation units: nullptr when the unit does not hold the combination
KernelFn bins_kernel_a(int K, bool inverse, int init_ks);   // 2 .. 7, 9 .. 12 bins
KernelFn bins_kernel_b(int K, bool inverse, int init_ks);   // 13 .. 16, 20, 24, 32 bins
}  // namespace k8x
}  // namespace nfa
