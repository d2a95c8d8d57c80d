// The kernel of rqs_resnet.hip (K8) as a header: the template is instantiated in two translation units
// (rqs_resnet.hip: 8 and 10 bins, contexts; rqs_resnet_bins.hip: the other bin counts and the other block
// activations, round 4).  Design notes: rqs_resnet.hip.
#pragma once

#include "bf16x3_gemm.hpp"

#include <hip/hip_ext.h>
#include <stdlib.h>

namespace nfa {

struct ResnetArgs {
    const float* x;      // [B, D]
    const vec4f* w;      // [num_layers * stages_per_layer][768] x 16 bytes, layout in include/nflows_amd.h
    const float* bias;   // accumulator-order biases of all GEMMs, layer after layer
    const int32_t* tables;  // [num_layers][128] slots of the identity / transformed features, then [128] final
    float* out;
    float* lad;
    int32_t* status;
    int64_t batch;  // multiple of 128
    int D, dt, di, num_blocks, num_layers, num_stages, bias_per_layer, accumulate;
    RqsDev sp;
    unsigned long long* trace;
    const int32_t* redo;  // optional [batch / 128]: only row blocks with a non-zero entry are processed (bits: see the kernel)
    int normal, skip_out;  // NFA_FLAG_STANDARD_NORMAL_LOG_PROB / NFA_FLAG_SKIP_OUTPUTS
    float log_z;           // 0.5 D log(2 pi)
    int Ds;                // columns the density sums over (features minus NFA_FLAG_PAD_COLUMNS)
    const float* ctx;      // [B, ce] context rows of the conditioners (resnet.py:92-100), or null
    int ce;                // context features (columns of ctx)
    float* dbg_logits;     // the DBG instances only (round 6): [B, dt * 24] the LAST layer's logits, packed row order
};

// ---- the final layer with the spline evaluation woven into its MFMAs ------------------------
// A lane's two features of a group (A, B) draw their 24 + 24 logits from the group's three
// accumulator tiles: A = T0[0:16] (widths, heights) + T1[0:7] (derivatives), B = T1[8:16] +
// T2[0:16].  The evaluation is cut into three units, each started once its tiles are complete
// and executed one FlatSteps piece per MFMA of the NEXT tile (same 48 accumulator registers as
// the plain loop: a tile's registers are recycled when its unit is done):
//     U0 = numerators of A                  during T1's MFMAs
//     U1 = finish A, width numerators of B  during T2's MFMAs
//     U2 = height numerators of B, finish B during T0's MFMAs of the next group
// `sched_barrier(0)` around every piece keeps hipcc from regrouping MFMAs and VALU work.
enum { kUnitNone = 0, kUnitNumA = 1, kUnitFinishA = 2, kUnitFinishB = 3,
       kUnitNumW10 = 4, kUnitRest10 = 5 };  // 10 bins: one feature per lane-half, widths | heights + finish

template <int UNIT, class Steps>
constexpr int spline_unit_slices() {
    return UNIT == kUnitNumA ? 2 * Steps::kNumSlices
           : UNIT == kUnitNumW10 ? Steps::kNumSlices
                                 : (UNIT == kUnitNone ? 0 : Steps::kNumSlices + Steps::kFinishSlices);
}

// Slice I of a unit.  Where two parts of a unit do not depend on each other their slices alternate,
// so that the VALU stream between two MFMAs holds two independent dependency chains:
//   U0: width / height numerators of A alternate
//   U1: finish A alternates with the width numerators of B (3 : 1)
//   U2: height numerators of B, interleaved with B's first walk when that walk reads the widths
//       (forward direction; the inverse searches the heights first)
template <int UNIT, int I, class Steps>
__device__ __forceinline__ void spline_unit_slice(Steps& fa, Steps& fb, const RqsDev& sp) {
    constexpr int N = Steps::kNumSlices;
    if constexpr (UNIT == kUnitNumW10) {
        fa.template num_w<I>();
    } else if constexpr (UNIT == kUnitRest10) {
        if constexpr (I < N) fa.template num_h<I>();
        else fa.template finish<I - N>(sp);
    } else if constexpr (UNIT == kUnitNumA) {
        if constexpr ((I & 1) == 0) fa.template num_w<(I >> 1)>();
        else fa.template num_h<(I >> 1)>();
    } else if constexpr (UNIT == kUnitFinishA) {
        // positions 3, 7, 11, ... (the first N of them) carry B's numerators
        if constexpr ((I & 3) == 3 && (I >> 2) < N) fb.template num_w<(I >> 2)>();
        else fa.template finish<I - ((I >> 2) < N ? (I >> 2) : N)>(sp);
    } else if constexpr (UNIT == kUnitFinishB) {
        static_assert(N <= Steps::kFirstWalkSlices, "the alternating part stays inside the first walk");
        if constexpr (Steps::kInverse) {
            if constexpr (I < N) fb.template num_h<I>();
            else fb.template finish<I - N>(sp);
        } else if constexpr (I < 2 * N) {
            if constexpr ((I & 1) == 0) fb.template num_h<(I >> 1)>();
            else fb.template finish<(I >> 1)>(sp);
        } else {
            fb.template finish<I - N>(sp);
        }
    }
}

template <int UNIT, int I, int END, class Steps>
__device__ __forceinline__ void spline_unit_range(Steps& fa, Steps& fb, const RqsDev& sp) {
    if constexpr (I < END) {
        spline_unit_slice<UNIT, I>(fa, fb, sp);
        spline_unit_range<UNIT, I + 1, END>(fa, fb, sp);
    }
}

// the slices of a unit spread evenly over the 48 MFMA slots of a tile
template <int UNIT, int SLOT, class Steps>
__device__ __forceinline__ void spline_unit_step(Steps& fa, Steps& fb, const RqsDev& sp) {
    constexpr int N = spline_unit_slices<UNIT, Steps>();
#ifdef NFA_NO_WEAVE
    // measurement aid: same pipeline, but a unit runs as one block behind the tile's last MFMA
    if constexpr (SLOT == 47) spline_unit_range<UNIT, 0, N>(fa, fb, sp);
#else
    spline_unit_range<UNIT, (SLOT * N) / 48, ((SLOT + 1) * N) / 48>(fa, fb, sp);
#endif
}

#define NFA_PUMP(SLOT, A_, B_)                                           \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, acc, 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                   \
    spline_unit_step<UNIT, SLOT>(fa, fb, sp);                            \
    __builtin_amdgcn_sched_barrier(0)

// one k-step (six MFMAs, one unit piece behind each); fh / fm / fl hold this k-step's weight
// fragments on entry and the next k-step's on exit (requested five MFMAs ahead of their use)
template <int UNIT, int KS, class Steps>
__device__ __forceinline__ void kstep_pumped(f32x16& acc, bf16x8 bh, bf16x8 bm, bf16x8 bl, vec4f& fh, vec4f& fm,
                                             vec4f& fl, const vec4f* cur, Steps& fa, Steps& fb, const RqsDev& sp) {
    constexpr int K4 = KS & 3;
    const bf16x8 ah = __builtin_bit_cast(bf16x8, fh), am = __builtin_bit_cast(bf16x8, fm),
                 al = __builtin_bit_cast(bf16x8, fl);
    NFA_PUMP(KS * 6 + 0, al, bh);
    if (K4 < 3) {
        fh = cur[(0 * 4 + K4 + 1) * 64];
        fm = cur[(1 * 4 + K4 + 1) * 64];
        fl = cur[(2 * 4 + K4 + 1) * 64];
    }
    NFA_PUMP(KS * 6 + 1, ah, bl);
    NFA_PUMP(KS * 6 + 2, am, bm);
    NFA_PUMP(KS * 6 + 3, am, bh);
    NFA_PUMP(KS * 6 + 4, ah, bm);
    NFA_PUMP(KS * 6 + 5, ah, bh);
}
#undef NFA_PUMP

template <int UNIT, int HS, class Steps>
__device__ __forceinline__ void stage_pumped(f32x16& acc, const bf16x8 (&ph)[8], const bf16x8 (&pm)[8],
                                             const bf16x8 (&pl)[8], WeightStream& sm, int lane, Steps& fa,
                                             Steps& fb, const RqsDev& sp) {
    stream_request(sm);
    const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
    vec4f fh = cur[0 * 4 * 64], fm = cur[1 * 4 * 64], fl = cur[2 * 4 * 64];
    kstep_pumped<UNIT, HS * 4 + 0>(acc, ph[HS * 4 + 0], pm[HS * 4 + 0], pl[HS * 4 + 0], fh, fm, fl, cur, fa, fb, sp);
    kstep_pumped<UNIT, HS * 4 + 1>(acc, ph[HS * 4 + 1], pm[HS * 4 + 1], pl[HS * 4 + 1], fh, fm, fl, cur, fa, fb, sp);
    kstep_pumped<UNIT, HS * 4 + 2>(acc, ph[HS * 4 + 2], pm[HS * 4 + 2], pl[HS * 4 + 2], fh, fm, fl, cur, fa, fb, sp);
    kstep_pumped<UNIT, HS * 4 + 3>(acc, ph[HS * 4 + 3], pm[HS * 4 + 3], pl[HS * 4 + 3], fh, fm, fl, cur, fa, fb, sp);
    stream_advance(sm);
}

template <int UNIT, class Steps>
__device__ __forceinline__ void gemm_tile_pumped(f32x16& acc, const bf16x8 (&ph)[8], const bf16x8 (&pm)[8],
                                                 const bf16x8 (&pl)[8], WeightStream& sm, int lane, Steps& fa,
                                                 Steps& fb, const RqsDev& sp) {
    stage_pumped<UNIT, 0>(acc, ph, pm, pl, sm, lane, fa, fb, sp);
    stage_pumped<UNIT, 1>(acc, ph, pm, pl, sm, lane, fa, fb, sp);
}

// bf16 pieces of k-step k4 of the wave's context tile in LDS (k = k4*16 + half*8 + j; columns >= ce are zero)
__device__ __forceinline__ void context_pieces(const float* s_ctx, int ce, int k4, int half, int r, bf16x8& bh,
                                               bf16x8& bm, bf16x8& bl) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = k4 * 16 + half * 8 + j;
        const float cv = s_ctx[(i < ce ? i : 0) * kRowPad + r];
        v[j] = i < ce ? cv : 0.0f;
    }
    bf16x2 hh[4], mm[4], ll[4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) split3(vec2f{v[j2 * 2], v[j2 * 2 + 1]}, hh[j2], mm[j2], ll[j2]);
    bh = join4(hh[0], hh[1], hh[2], hh[3]);
    bm = join4(mm[0], mm[1], mm[2], mm[3]);
    bl = join4(ll[0], ll[1], ll[2], ll[3]);
}

// one 32-row tile of a block's context layer with more than 16 context features: acc += W_c_tile[32 x ce] x
// context^T.  ONE stage ([3 pieces][4 k-steps][64 lanes] x 16 bytes, k-steps beyond ce zero) per tile; the
// context pieces are made on the spot.  (Up to 16 context features the four tiles share one k-major stage:
// see the kernel.)
__device__ __forceinline__ void gemm_context_tile(f32x16& acc, const float* s_ctx, int ce, int half, int r,
                                                  WeightStream& sm, int lane) {
    stream_request(sm);
    const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
    const int nks = (ce + 15) >> 4;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        if (k4 < nks) {
            bf16x8 bh, bm, bl;
            context_pieces(s_ctx, ce, k4, half, r, bh, bm, bl);
            const bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(0 * 4 + k4) * 64]);
            const bf16x8 am = __builtin_bit_cast(bf16x8, cur[(1 * 4 + k4) * 64]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, cur[(2 * 4 + k4) * 64]);
            NFA_MFMA6(acc, ah, am, al, bh, bm, bl);
        }
    }
    stream_advance(sm);
}

// PRESCALED: 1 = the host folded 1/sqrt(hidden) into the width / height rows of the final layer,
// 2 = 1/sqrt(hidden) and log2(e) (NFA_FLAG_LOGITS_LOG2E: softmax numerators are then one v_exp_f32)
// INIT_KS: k-steps of the initial layer, 2 (d_i <= 32) or 4 (d_i <= 64)
//
// The kernel runs num_layers coupling layers back to back on the same 32 rows per wave: rows of a
// flow are independent, so a workgroup can take its 128 rows through every layer without meeting
// the others.  The row tile never moves between layers: slot j of the tile is input column j of
// the first layer; every layer reads its identity / transformed features from, and writes its
// spline results back to, fixed slots given by its table (the host composes all the permutations
// between the layers into these tables), and the last table says which slot ends up at which
// output position.  Weights and biases of all layers form one stream in execution order.
// CTX: the conditioners take a context (resnet.py:9-52, :92-100): its `ce` columns follow the identity
// features in the initial layer's input, and every residual block's result is multiplied by
// sigmoid(context_layer(context)) before the skip connection (F.glu of the concatenation).
// DBG (rqs_resnet_dbg.hip, tests): the plain 8-bin loop with the last layer's logits stored as well.
// LINEAR = false (round 6): tails=None couplings (coupling.py:565-570: the constrained spline on [left, right] x [bottom, top],
// K + 1 derivative logits per feature, an input outside the box raises NFA_STATUS_OUTSIDE_DOMAIN) -- the plain loop on K1's
// register evaluator at every bin count, 3 K + 1 logits per feature padded to whole 16-row lane-half shares
template <bool INVERSE, int PRESCALED, int INIT_KS, int PIPE = 0, int KB = 8, bool CTX = false, int ACT = kActRelu, bool DBG = false,
          bool LINEAR = true>  // PIPE: 0 plain loop, 1 woven, 2 woven with FlatSteps<FAST>; ACT: the blocks' activation
__global__ void __launch_bounds__(kBlock, 2) rqs_resnet_kernel(const ResnetArgs a) {
    static_assert(LINEAR || (PIPE == 0 && PRESCALED == 1 && !CTX && !DBG && ACT == kActRelu), "tails=None: the plain loop, ReLU, no context");
    static_assert(!DBG || (KB == 8 && PIPE == 0 && !CTX), "the diagnostic instances: 8 bins, the plain loop");
    static_assert(ACT == kActRelu || (PIPE == 0 && PRESCALED == 1 && ACT >= kActLeakyRelu && ACT <= kActTanh),
                  "other activations: the plain loop");
    static_assert(KB == 8 || (KB == 10 && PIPE != 1 && PRESCALED == 1) || (KB >= 2 && KB <= 32 && PIPE == 0 && PRESCALED == 1),
                  "10 bins: plain loop, or woven with the shorter sequence; other bin counts (2 .. 16, 20, 24, 32): plain loop");
    // rows of the final layer per transformed feature (8 bins: 23 logits padded to 24, two features share three tiles;
    // otherwise 3 K - 1 padded to whole 16-row lane-half shares)
    constexpr int kFinalRows = !LINEAR ? 16 * ((3 * KB + 1 + 15) / 16) : KB == 8 ? 24 : 16 * ((3 * KB - 1 + 15) / 16);
    // dynamic LDS: the weight ring, then per wave a [D][33] row tile
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ int s_tab[2][kTabLayer];   // tables of the current and the next layer
    __shared__ int s_final[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, dt = a.dt;
    int my_status = 0;
    auto checked = [&](int v, bool used) {
        if (used && (v < 0 || v >= D)) my_status |= NFA_STATUS_BAD_INDEX;
        return v < 0 ? 0 : (v >= D ? D - 1 : v);
    };
    // Second pass behind K8h / K8s (round 5): in the common step NO block is flagged and this launch used to cost a
    // weight-ring prologue (two 12 KB stages per workgroup) before its workgroups found that out quad by quad.  Now
    // the workgroup looks at the flags of ITS quads first and leaves when none is set.
    if (a.redo) {
        int any = 0;
        const int64_t quads = a.batch >> 7;
        for (int64_t q = blockIdx.x + (int64_t)gridDim.x * tid; q < quads; q += (int64_t)gridDim.x * kBlock) any |= a.redo[q];
        if (__syncthreads_or(any) == 0) return;
    }
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
    // PIPE: the final layer's biases of the current layer, staged once per layer (the woven loop
    // cannot afford an L2 round trip in front of every tile)
    float* s_fbias = lds_dyn + kRing * kStageVec4 * 4 + (kBlock / kWave) * D * kRowPad;
    // CTX: per wave the [ce][33] context tile of its 32 rows, behind the final-layer biases
    float* s_ctx = s_fbias + (PIPE != 0 ? dt * kFinalRows : 0) + wave * a.ce * kRowPad;
    // CTX: per wave the fp32 residual stream h, [4 tiles x 16 registers][64 lanes]
    [[maybe_unused]] float* s_hacc = s_fbias + (PIPE != 0 ? dt * kFinalRows : 0) + (kBlock / kWave) * a.ce * kRowPad +
                                     wave * (64 * kWave);
    const int groups = dt >> 2;
    const int64_t num_quads = a.batch >> 7;
    int tb = 0;  // which half of s_tab holds the current layer's table

    unsigned long long* tr = nullptr;
    int ti = 0;
    if (a.trace && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 256))
        tr = a.trace + (blockIdx.x ? 256 : 0);
    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        // second pass behind the f16 kernel (rqs_resnet_f16.hip): only the row blocks it gave up on
        // second pass of K8h / K8s: 0 = block done; bit 0 = the whole block is open; bits 1 / 2 = its lower / upper
        // 64 rows are (K8s's four-wave workgroups: the other half was written, log-determinant accumulated, by them)
        const int redo_flag = a.redo ? a.redo[quad] : 1;
        if (redo_flag == 0) continue;
        // (bit 0: the whole block; bits 1, 2: its lower / upper 64 rows -- K8s's and K8c's 64-row workgroups --; bits 3 .. 6: its
        //  four 32-row quarters -- K8c's 32-row workgroups, round 6: wave w of this kernel owns quarter w)
        const bool write_rows = (redo_flag & 1) || ((redo_flag >> (1 + (wave >> 1))) & 1) || ((redo_flag >> (3 + wave)) & 1);
        const int64_t row0 = (quad << 7) + (wave << 5);
        // (lane-derived values are made opaque per iteration: hoisted out of this loop they would
        // stay live through the whole kernel and push the register allocation into scratch)
        int lane_here = lane, di = a.di;
        asm volatile("" : "+v"(lane_here), "+s"(di));
        const int half = lane_here >> 5, r = lane_here & 31;
        NFA_STAMP()
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
        if (CTX) {
            const float* crow = a.ctx + row0 * a.ce;
            const int nctx = 32 * a.ce;
            for (int e0 = lane; e0 < nctx; e0 += kWave * 4) {   // (four loads in flight per lane, like the rows)
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * kWave;
                    v[u] = crow[e < nctx ? e : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * kWave;
                    if (e < nctx) {
                        const int rr = e / a.ce, c = e - rr * a.ce;
                        s_ctx[c * kRowPad + rr] = v[u];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        float lad_acc = 0.0f;
        for (int layer = 0; layer < a.num_layers; ++layer) {
            // the two workgroups resident on a CU take turns at the higher issue priority: issue
            // arbitration on a SIMD is strictly by priority, then age, so without this the younger
            // workgroup loses every slot and finishes the run alone (measured: -3 % run time)
            if ((layer + (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            const int* tab = s_tab[tb];
            // the next layer's table (the first one again after the last: next row block) goes to
            // the other half now; it is read only after this layer's many stage barriers
            if (tid < kTabLayer) {
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
                    if (CTX) {   // input of the initial layer = [identity features | context] (resnet.py:93-94)
                        const int ic = i - di;
                        const float xv = i < di ? s_row[tab[kTabId + (i < 64 ? i : 0)] * kRowPad + r]
                                                : s_ctx[(ic < a.ce ? ic : 0) * kRowPad + r];
                        v[j] = i < di + a.ce ? xv : 0.0f;
                    } else {
                        const float xv = s_row[tab[kTabId + i] * kRowPad + r];
                        v[j] = i < di ? xv : 0.0f;
                    }
                }
                bf16x2 hh[4], mm[4], ll[4];
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) split3(vec2f{v[j2 * 2], v[j2 * 2 + 1]}, hh[j2], mm[j2], ll[j2]);
                ph[ks] = join4(hh[0], hh[1], hh[2], hh[3]);
                pm[ks] = join4(mm[0], mm[1], mm[2], mm[3]);
                pl[ks] = join4(ll[0], ll[1], ll[2], ll[3]);
            }
            NFA_STAMP()

            // ---- initial layer: h = W_i x + b_i
            // (CTX: the residual stream h is also kept in fp32 in the wave's LDS scratch -- the gated block
            //  needs h after its second GEMM, and neither its pieces nor its accumulators fit the register
            //  file next to that GEMM's operands; one tile at a time comes back when the gate is applied)
            {
                f32x16 h[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) load_bias_tile(h[t], bias + t * 32);
                gemm_kmajor<false, INIT_KS>(h, ph, pm, pl, sm, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    tile_to_pieces<false>(h[t], ph[2 * t], pm[2 * t], pl[2 * t], ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
                    if constexpr (CTX) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) s_hacc[(t * 16 + q) * kWave + lane_here] = h[t][q];
                    }
                }
            }
            bias += 128;
            if (PIPE != 0) {
                // (every wave has passed a stage barrier of this layer: nobody reads the previous
                // layer's biases any more; the blocks' barriers come before the first use)
                const float* fbias = a.bias + (size_t)layer * a.bias_per_layer + 128 + (CTX ? 384 : 256) * a.num_blocks;
                for (int i = tid; i < dt * kFinalRows; i += kBlock) s_fbias[i] = fbias[i];
                // without residual blocks the final layer follows at once: no stage barrier in between
                if (a.num_blocks == 0) __syncthreads();
            }
            NFA_STAMP()

            // ---- residual blocks: h += W_1 relu(W_0 relu(h) + b_0) + b_1, both Linears k-major.
            //      Register budget: the h pieces (96) must survive the first Linear for the skip
            //      connection; u (64 accumulators) turns into the relu(u) pieces (96) tile by tile,
            //      then the skip is added into the second Linear's accumulators tile by tile (the h
            //      pieces die), whose input pieces die k-step by k-step.
            for (int blk = 0; blk < a.num_blocks; ++blk) {
                bf16x8 qh[8], qm[8], ql[8];
                {
                    f32x16 u[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) load_bias_tile(u[t], bias + t * 32);
                    gemm_kmajor<ACT, 8>(u, ph, pm, pl, sm, lane);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        tile_to_pieces<ACT>(u[t], qh[2 * t], qm[2 * t], ql[2 * t], qh[2 * t + 1], qm[2 * t + 1], ql[2 * t + 1]);
                }
                NFA_STAMP()
                f32x16 v[4];
                if constexpr (CTX) {
                    // temps = W_1 relu(u) + b_1; h += temps * sigmoid(W_c context + b_c)   (resnet.py:46-52)
#pragma unroll
                    for (int t = 0; t < 4; ++t) load_bias_tile(v[t], bias + 128 + t * 32);
                    gemm_kmajor<false, 8>(v, qh, qm, ql, sm, lane);
                    NFA_STAMP()
                    // up to 16 context features: the gate's four tiles share ONE k-major stage ([4 tiles][3 pieces]
                    // [64 lanes] x 16 bytes) and one set of context pieces
                    const bool one_stage = a.ce <= 16;
                    const vec4f* gcur = nullptr;
                    bf16x8 cbh, cbm, cbl;
                    if (one_stage) {
                        stream_request(sm);
                        gcur = sm.ring + sm.slot * kStageVec4 + lane;
                        context_pieces(s_ctx, a.ce, 0, half, r, cbh, cbm, cbl);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        f32x16 gate;
                        load_bias_tile(gate, bias + 256 + t * 32);
                        if (one_stage) {
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, gcur[(t * 3 + 0) * 64]);
                            const bf16x8 am = __builtin_bit_cast(bf16x8, gcur[(t * 3 + 1) * 64]);
                            const bf16x8 al = __builtin_bit_cast(bf16x8, gcur[(t * 3 + 2) * 64]);
                            NFA_MFMA6(gate, ah, am, al, cbh, cbm, cbl);
                        } else {
                            gemm_context_tile(gate, s_ctx, a.ce, half, r, sm, lane);
                        }
                        f32x16 hn;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            // sigmoid on v_exp_f32 / v_rcp_f32 (1 ulp each) with one residual correction of the
                            // reciprocal; the exponent is capped so that 1 + 2^t stays finite (sigmoid < 2^-126 there)
                            float tg = gate[q] * -1.44269502162933349609375f;
                            tg = tg > 126.0f ? 126.0f : tg;   // (a comparison, not fminf: NaN stays NaN)
                            const float e2 = __builtin_amdgcn_exp2f(tg);
                            const float dn = 1.0f + e2;
                            const float r0 = __builtin_amdgcn_rcpf(dn);
                            const float sg = __builtin_fmaf(__builtin_fmaf(-dn, r0, 1.0f), r0, r0);
                            float* hp = s_hacc + (t * 16 + q) * kWave + lane_here;
                            hn[q] = *hp + v[t][q] * sg;
                            *hp = hn[q];
                        }
                        tile_to_pieces<false>(hn, ph[2 * t], pm[2 * t], pl[2 * t], ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
                    }
                    if (one_stage) stream_advance(sm);
                    bias += 384;
                } else {
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
                NFA_STAMP()
            }

            if constexpr (KB == 10 && PIPE == 2) {
                // ---- 10 bins, woven: the feature's width numerators (tile 0 holds the ten width
                //      logits) run between the MFMAs of tile 1, everything else between those of the
                //      next group's tile 0; one FlatSteps object, 32 accumulator registers
                using Steps = FlatSteps<INVERSE, 1, true, 10>;
                Steps f;
                const float* fbias = s_fbias + half * 16;
                const int groups10 = dt >> 1;
                f32x16 acc0, acc1;
                float hrest[6];
                float* slot = s_row + tab[kTabTr + half] * kRowPad + r;
                load_bias_tile(acc0, fbias);
                gemm_tile<false>(acc0, ph, pm, pl, sm, lane);
                for (int g = 0; g < groups10; ++g) {
                    f.x = *slot;
#pragma unroll
                    for (int j = 0; j < 10; ++j) f.ew[j] = acc0[j];
#pragma unroll
                    for (int j = 0; j < 6; ++j) hrest[j] = acc0[10 + j];
                    load_bias_tile(acc1, fbias + (g * 2 + 1) * 32);
                    gemm_tile_pumped<kUnitNumW10>(acc1, ph, pm, pl, sm, lane, f, f, a.sp);
#pragma unroll
                    for (int j = 0; j < 6; ++j) f.eh[j] = hrest[j];
#pragma unroll
                    for (int j = 0; j < 4; ++j) f.eh[6 + j] = acc1[j];
#pragma unroll
                    for (int j = 0; j < 9; ++j) f.sd[j] = acc1[4 + j];
                    if (g + 1 < groups10) {
                        float* next_slot = s_row + tab[kTabTr + (g + 1) * 2 + half] * kRowPad + r;
                        load_bias_tile(acc0, fbias + (g * 2 + 2) * 32);
                        gemm_tile_pumped<kUnitRest10>(acc0, ph, pm, pl, sm, lane, f, f, a.sp);
                        *slot = f.y;
                        slot = next_slot;
                    } else {
                        spline_unit_range<kUnitRest10, 0, spline_unit_slices<kUnitRest10, Steps>()>(f, f, a.sp);
                        *slot = f.y;
                    }
                    lad_acc += f.lad;
                    my_status |= f.status;
                }
                NFA_STAMP()
            } else if constexpr (!LINEAR || (KB != 8 && KB != 10)) {
                // ---- any other bin count (round 4: the second pass behind K8h's instances for 2 .. 16 bins): 3 K - 1
                //      logits per feature padded to T tiles' lane-half shares (16 T rows), the rows ordered so that the
                //      16 T accumulator values of lane-half h are the logits of feature 2g + h; evaluated by the
                //      register instance of K1's function (rqs_math.hpp: rqs_eval<K, ., linear tails, REGS>)
                constexpr int T = kFinalRows / 16;
                RqsDev sp0 = a.sp;
                sp0.divisor = 0.0f;  // 1/sqrt(hidden) is folded into the weight rows
                for (int g = 0; g < (dt >> 1); ++g) {
                    float* slot = s_row + tab[kTabTr + g * 2 + half] * kRowPad + r;
                    const float xin = *slot;
                    float p[16 * T];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        f32x16 acc;
                        load_bias_tile(acc, bias + (g * T + t) * 32);
                        gemm_tile<false>(acc, ph, pm, pl, sm, lane);
#pragma unroll
                        for (int q = 0; q < 16; ++q) p[16 * t + q] = acc[q];
                    }
                    float y, l;
                    my_status |= rqs_eval<KB, INVERSE, LINEAR, true>(xin, p, sp0, y, l);
                    *slot = y;
                    lad_acc += l;
                }
                NFA_STAMP()
            } else if constexpr (KB == 10) {
                // ---- final layer for 10 bins (the reference's default): 29 logits per feature padded
                //      to 32 rows, two 32-row tiles per group; the rows are ordered so that the 32
                //      accumulator values of lane-half h are the logits of feature 2g + h
                RqsDev sp10 = a.sp;
                sp10.divisor = 0.0f;  // 1/sqrt(hidden) is folded into the weight rows
                for (int g = 0; g < (dt >> 1); ++g) {
                    float* slot = s_row + tab[kTabTr + g * 2 + half] * kRowPad + r;
                    const float xin = *slot;
                    f32x16 acc[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        load_bias_tile(acc[t], bias + (g * 2 + t) * 32);
                        gemm_tile<false>(acc[t], ph, pm, pl, sm, lane);
                    }
                    float p[32];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        p[q] = acc[0][q];
                        p[16 + q] = acc[1][q];
                    }
                    // (the shorter rounding sequence, as for 8 bins: see FlatSteps<FAST>)
                    FlatSteps<INVERSE, 1, true, 10> f;
                    f.x = xin;
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        f.ew[j] = p[j];
                        f.eh[j] = p[10 + j];
                        if (j < 9) f.sd[j] = p[20 + j];
                    }
                    flat_steps_all(f, sp10);
                    *slot = f.y;
                    lad_acc += f.lad;
                    my_status |= f.status;
                }
                NFA_STAMP()
            } else if constexpr (PIPE != 0) {
                // ---- final layer with the spline evaluation woven into the MFMAs (see gemm_tile_pumped)
                using Steps = FlatSteps<INVERSE, PRESCALED, PIPE == 2>;
                Steps fa, fb;
                float* slot_b = nullptr;
                const float* fbias = s_fbias + half * 16;
                f32x16 acc[3];
                auto commit = [&](Steps& f, float* slot) {
                    *slot = f.y;
                    lad_acc += f.lad;
                    my_status |= f.status;
                };
                for (int g = 0; g < groups; ++g) {
                    float* slot0 = s_row + tab[kTabTr + g * 4 + half * 2] * kRowPad + r;
                    float* slot1 = s_row + tab[kTabTr + g * 4 + half * 2 + 1] * kRowPad + r;
                    load_bias_tile(acc[0], fbias + (g * 3 + 0) * 32);
                    if (g > 0) {
                        gemm_tile_pumped<kUnitFinishB>(acc[0], ph, pm, pl, sm, lane, fa, fb, a.sp);
                        commit(fb, slot_b);
                    } else {
                        gemm_tile<false>(acc[0], ph, pm, pl, sm, lane);
                    }
                    fa.x = *slot0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fa.ew[j] = acc[0][j];
                        fa.eh[j] = acc[0][8 + j];
                    }
                    load_bias_tile(acc[1], fbias + (g * 3 + 1) * 32);
                    gemm_tile_pumped<kUnitNumA>(acc[1], ph, pm, pl, sm, lane, fa, fb, a.sp);
                    fb.x = *slot1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (j < 7) fa.sd[j] = acc[1][j];
                        fb.ew[j] = acc[1][8 + j];
                    }
                    load_bias_tile(acc[2], fbias + (g * 3 + 2) * 32);
                    gemm_tile_pumped<kUnitFinishA>(acc[2], ph, pm, pl, sm, lane, fa, fb, a.sp);
                    commit(fa, slot0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fb.eh[j] = acc[2][j];
                        if (j < 7) fb.sd[j] = acc[2][8 + j];
                    }
                    slot_b = slot1;
                }
                spline_unit_range<kUnitFinishB, 0, spline_unit_slices<kUnitFinishB, Steps>()>(fa, fb, a.sp);
                commit(fb, slot_b);
                NFA_STAMP()
            } else {
                // ---- final layer, three 32-row tiles (= 4 features) at a time, and the splines; the
                //      results replace the inputs in their slots
                for (int g = 0; g < groups; ++g) {
                    float* slot0 = s_row + tab[kTabTr + g * 4 + half * 2] * kRowPad + r;
                    float* slot1 = s_row + tab[kTabTr + g * 4 + half * 2 + 1] * kRowPad + r;
                    const float xin0 = *slot0, xin1 = *slot1;
                    f32x16 acc[3];
    #pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        load_bias_tile(acc[t], bias + (g * 3 + t) * 32);
                        gemm_tile<false>(acc[t], ph, pm, pl, sm, lane);
                        if constexpr (DBG) {
                            if (layer == a.num_layers - 1) {
                                float* dst = a.dbg_logits + (size_t)(row0 + r) * (dt * 24) + (g * 3 + t) * 32 + half * 16;
#pragma unroll
                                for (int q_ = 0; q_ < 16; ++q_) dst[q_] = acc[t][q_];
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (not under the stream's counted waits)
                            }
                        }
                    }
                    NFA_STAMP()
                    {
                        NFA_K7_FEATURE_A(pa, acc[0], acc[1]);
                        NFA_K7_FEATURE_B(pb, acc[1], acc[2]);
                        float y0, l0, y1, l1;
                        my_status |= rqs_eval_flat8<INVERSE, PRESCALED>(xin0, pa, a.sp, y0, l0);
                        my_status |= rqs_eval_flat8<INVERSE, PRESCALED>(xin1, pb, a.sp, y1, l1);
                        *slot0 = y0;
                        *slot1 = y1;
                        lad_acc += l0;
                        lad_acc += l1;
                    }
                    NFA_STAMP()
                }
            }
            tb ^= 1;
            // this wave's spline results must be visible to its own gathers of the next layer
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // ---- output rows: position p of a row comes from slot final[p]; 16 bytes per lane per store
        if (!a.skip_out && write_rows) {
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
        if (half == 0 && write_rows) {
            float* dst = a.lad + row0 + r;
            float v = a.accumulate ? *dst + lad_acc : lad_acc;
            if (a.normal) v = (-0.5f * sumsq - a.log_z) + v;   // normal.py:31-33, flows/base.py:49
            *dst = v;
        }
        NFA_STAMP()
        // stores and LDS-DMA requests complete out of order with each other: drain before the next
        // row block counts outstanding requests again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two stages requested past the end
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace nfa

namespace nfa {
typedef void (*ResnetKernelFn)(const ResnetArgs);
// instances of rqs_resnet_bins.hip (plain final-layer loop): nullptr when that unit does not hold the combination
ResnetKernelFn resnet_bins_kernel(int K, bool inverse, int init_ks);                      // 2 .. 16 except 8 / 10, 20, 24, 32
ResnetKernelFn resnet_context_kernel(int K, int activation, bool inverse, int init_ks);     // with a context (round 5): those of the two lines above
ResnetKernelFn resnet_activation_kernel(int activation, int K, bool inverse, int init_ks);  // NFA_ACTIVATION_* > 0, 8 / 10 bins
ResnetKernelFn resnet_tails_kernel(int K, bool inverse, int init_ks);                     // rqs_resnet_tails.hip: tails=None, every bin count
ResnetKernelFn resnet_debug_kernel(bool inverse, int init_ks);                              // rqs_resnet_dbg.hip: 8 bins, ReLU, logits stored
}  // namespace nfa
