// K8s: the whole-layer kernel of rqs_resnet_f16.hip (K8h: ResidualNet conditioner, nn/nets/resnet.py:55-100, +
// everything K1 replaces, coupling.py:73-130, :549-582, for a run of layers in one launch; GEMMs on two f16
// pieces per fp32 operand) on SIXTEEN-sample tiles: v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16.
//
// Why a second tile shape.  K8h gives a wave 32 rows; a batch of 32 768 rows (config 4's share of one GPU at
// N = 8) is then one wave per SIMD, and 16 384 rows leave half the CUs idle.  With 16 rows per wave the same batch
// is twice the waves, each with half the matrix work per layer.  Measured (profiles/r3/k8s_*.txt): 32 768 rows
// 1.00-1.04 ms against K8h's 1.06-1.11, 16 384 rows 0.85-0.87 against 0.90-0.91 -- 4-7 %, not the factor the
// halved matrix work suggests: a small batch is bound by the rate at which ONE CU can pull the layer's 672 KB of
// weights through LDS-DMA (16 KB per ~0.62 us = 26 GB/s per CU: the guide's "ldsdma-fill" cadence), 27 us per layer
// and 0.86 ms per 32-layer pass whatever the tile shape.  At large batches K8s is 4-7 % slower than K8h (twice the
// fragment reads per row), so it serves only the batches that give a CU at most one 128-row block.
//
// Same stream format as K8h (16 KB stages of eight (hi, lo) fragment pairs, one parameter stage per layer, the
// same number of stages per GEMM), same parameter words, same tables, same piece conversion, same spline
// evaluation (FusedSteps8); what differs is who holds what:
//   lane l of a wave: sample n = l % 16, lane group g = l / 16 (0 .. 3)
//   A fragment (weights, 16 features x 32 k): lane (m = l % 16, k = 8 g + j), j = 0 .. 7      [one 16-byte read]
//   B fragment (activations, 32 k x 16 samples): lane (n, k = 8 g + j)                      [one uvec4 per piece]
//   accumulator tile (16 features x 16 samples): lane (n, features 4 g + i), i = 0 .. 3      [four registers]
// Chaining: k-step S of the next GEMM (32 k) is made of accumulator tiles 2 S and 2 S + 1 -- the eight values a
// lane holds of them -- i.e. MFMA k position 8 g + j <-> feature 32 S + 16 (j / 4) + 4 g + j % 4; the host orders
// the next weight's columns accordingly (ops._k8s_column_order).  The final layer's rows are ordered so that
// the six tiles of a group of four features give lane group g the 24 logits of feature 4 G + g
// (ops._k8s_row_order): one spline evaluation per lane and group, straight from the accumulators.
//
// Restrictions: K = 8 bins, linear tails, no context, hidden width 128 (narrower: zero-padded by the host), ReLU
// blocks, d_i <= 64, d_t % 4 == 0, d_t <= 64, D % 4 == 0, D <= 128, batch % 128 == 0.  Workgroups of eight waves
// (128 rows: the granularity of `redo_blocks`).

#include "k8h_common.hpp"

namespace nfa {
namespace k8s {

using namespace k8h;

constexpr int kRowPad16 = 17;
constexpr int kWavesPerGroup = 8;   // waves per workgroup
typedef vec4f f32x4;

#define NFA_K8S_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

// one fragment pair = the three products of one (tile, k-step) on accumulator `acc`, a weave slice behind each
template <int G, class W>
__device__ __forceinline__ void cell(f32x4& acc, uvec4 bhw, uvec4 blw, Frags& fr, unsigned cur, unsigned nxt, W& w, int slot0) {
    const f16x8 bh = __builtin_bit_cast(f16x8, bhw), bl = __builtin_bit_cast(f16x8, blw);
    const Frags nf = next_frags<G>(cur, nxt);
    await_frags(fr);
    const f16x8 ah = __builtin_bit_cast(f16x8, fr.h), al = __builtin_bit_cast(f16x8, fr.l);
    fr = nf;
    // (smallest terms first)
    acc = NFA_K8S_MFMA(al, bh, acc);
    __builtin_amdgcn_sched_barrier(0);
    w.step(slot0 + 0);
    __builtin_amdgcn_sched_barrier(0);
    acc = NFA_K8S_MFMA(ah, bl, acc);
    __builtin_amdgcn_sched_barrier(0);
    w.step(slot0 + 1);
    __builtin_amdgcn_sched_barrier(0);
    acc = NFA_K8S_MFMA(ah, bh, acc);
    __builtin_amdgcn_sched_barrier(0);
    // The operands stay live past the last product: hipcc renames the four-register accumulator from MFMA to MFMA
    // and, left alone, puts a result on the registers of an A operand that has just had its last use
    // (`v_mfma v[6:9], v[6:9], ...`) -- observed as one wave in a few thousand 1e-4 off, not reproducible per
    // launch: the matrix pipe may still be reading the fragment when the first result rows are written.
    asm volatile("" ::"v"(ah), "v"(al), "v"(bh), "v"(bl));
    w.step(slot0 + 2);
    __builtin_amdgcn_sched_barrier(0);
}

struct NoWeave16 {
    __device__ __forceinline__ void step(int) {}
};

// k-major stage: the eight output tiles of one k-step (pair T = tile T)
template <class SM, class W>
__device__ __forceinline__ void kstep_stage(f32x4 (&acc)[8], uvec4 bh, uvec4 bl, SM& sm, Frags& fr, int lane, W&& w) {
    unsigned cur, nxt;
    stage_begin(sm, cur, nxt, lane);
    cell<0>(acc[0], bh, bl, fr, cur, nxt, w, 0);
    cell<1>(acc[1], bh, bl, fr, cur, nxt, w, 3);
    cell<2>(acc[2], bh, bl, fr, cur, nxt, w, 6);
    cell<3>(acc[3], bh, bl, fr, cur, nxt, w, 9);
    cell<4>(acc[4], bh, bl, fr, cur, nxt, w, 12);
    cell<5>(acc[5], bh, bl, fr, cur, nxt, w, 15);
    cell<6>(acc[6], bh, bl, fr, cur, nxt, w, 18);
    cell<7>(acc[7], bh, bl, fr, cur, nxt, w, 21);
    stream_advance(sm);
}

// tile-major stage of the final layer: tiles `a0`, `a1` over the four k-steps each (pairs 0..3 / 4..7)
template <class SM, class W>
__device__ __forceinline__ void tile_pair_stage(f32x4& a0, f32x4& a1, const uvec4 (&ph)[4], const uvec4 (&pl)[4], SM& sm,
                                                Frags& fr, int lane, W&& w) {
    unsigned cur, nxt;
    stage_begin(sm, cur, nxt, lane);
    cell<0>(a0, ph[0], pl[0], fr, cur, nxt, w, 0);
    cell<1>(a0, ph[1], pl[1], fr, cur, nxt, w, 3);
    cell<2>(a0, ph[2], pl[2], fr, cur, nxt, w, 6);
    cell<3>(a0, ph[3], pl[3], fr, cur, nxt, w, 9);
    cell<4>(a1, ph[0], pl[0], fr, cur, nxt, w, 12);
    cell<5>(a1, ph[1], pl[1], fr, cur, nxt, w, 15);
    cell<6>(a1, ph[2], pl[2], fr, cur, nxt, w, 18);
    cell<7>(a1, ph[3], pl[3], fr, cur, nxt, w, 21);
    stream_advance(sm);
}

// pieces of k-step S of the next GEMM from accumulator tiles t0 = 2 S, t1 = 2 S + 1 (ReLU'd when RELU, x scale)
template <bool RELU>
__device__ __forceinline__ void convert_kstep(uvec4& h, uvec4& l, const f32x4& t0, const f32x4& t1, float scale, float& peak) {
    unsigned hi, lo;
    convert_pair<RELU, true>(t0[0], t0[1], scale, peak, hi, lo);
    h[0] = hi;
    l[0] = lo;
    convert_pair<RELU, true>(t0[2], t0[3], scale, peak, hi, lo);
    h[1] = hi;
    l[1] = lo;
    convert_pair<RELU, true>(t1[0], t1[1], scale, peak, hi, lo);
    h[2] = hi;
    l[2] = lo;
    convert_pair<RELU, true>(t1[2], t1[3], scale, peak, hi, lo);
    h[3] = hi;
    l[3] = lo;
}

// conversion of k-step S + 1 behind the 24 MFMAs of the stage that consumes k-step S: one pair per six slots
// (three phases two slots apart)
template <bool RELU>
struct ConvWeave16 {
    const f32x4 &t0, &t1;
    uvec4 &h, &l;
    float scale;
    float& peak;
    __device__ __forceinline__ void step(int slot) {
        // (slot is a compile-time constant at every call site after inlining)
        if (slot == 1) pair(0, t0[0], t0[1]);
        else if (slot == 7) pair(1, t0[2], t0[3]);
        else if (slot == 13) pair(2, t1[0], t1[1]);
        else if (slot == 19) pair(3, t1[2], t1[3]);
    }
    __device__ __forceinline__ void pair(int j, float v0, float v1) {
        unsigned hi, lo;
        convert_pair<RELU, true>(v0, v1, scale, peak, hi, lo);
        if (j == 0) { h[0] = hi; l[0] = lo; }
        else if (j == 1) { h[1] = hi; l[1] = lo; }
        else if (j == 2) { h[2] = hi; l[2] = lo; }
        else { h[3] = hi; l[3] = lo; }
    }
};

// k-major 128 -> 128 GEMM whose input pieces are made on the way from the accumulator tiles `src` of the previous
// GEMM: k-step 0 up front, k-step S + 1 behind the MFMAs of k-step S
template <class SM>
__device__ __forceinline__ void gemm_converting(f32x4 (&acc)[8], uvec4 (&ph)[4], uvec4 (&pl)[4], const f32x4 (&src)[8],
                                                float scale, float& worst, SM& sm, Frags& fr, int lane) {
    float peak = 0.0f;
    convert_kstep<true>(ph[0], pl[0], src[0], src[1], scale, peak);
    kstep_stage(acc, ph[0], pl[0], sm, fr, lane, ConvWeave16<true>{src[2], src[3], ph[1], pl[1], scale, peak});
    kstep_stage(acc, ph[1], pl[1], sm, fr, lane, ConvWeave16<true>{src[4], src[5], ph[2], pl[2], scale, peak});
    kstep_stage(acc, ph[2], pl[2], sm, fr, lane, ConvWeave16<true>{src[6], src[7], ph[3], pl[3], scale, peak});
    kstep_stage(acc, ph[3], pl[3], sm, fr, lane, NoWeave16{});
    worst = __builtin_fmaxf(worst, peak * scale);
}

__device__ __forceinline__ void load_bias4(f32x4& acc, const float* p) { acc = *reinterpret_cast<const vec4f*>(p); }

// slice I of an evaluation: 0 .. N - 1 width numerators, N .. 2 N - 1 height numerators, then finish
template <class Steps, int I>
__device__ __forceinline__ void run_slice(Steps& f, const RqsDev& sp) {
    constexpr int N = Steps::kNumSlices;
    if constexpr (I < N) f.template num_w<I>();
    else if constexpr (I < 2 * N) f.template num_h<I - N>();
    else f.template finish<I - 2 * N>(sp);
}

template <class Steps, int I, int END>
__device__ __forceinline__ void run_range(Steps& f, const RqsDev& sp) {
    if constexpr (I < END) {
        run_slice<Steps, I>(f, sp);
        run_range<Steps, I + 1, END>(f, sp);
    }
}

// Slices of the spline evaluation behind the 24 MFMAs of a stage (`step(slot)` with a slot that is a constant
// after inlining).  FinishWeave: slices [FIRST, FIRST + COUNT) of finish; NumWeave: the two numerator sets
// alternating (two independent chains).
#define NFA_K8S_SLOT_SWITCH(CALL)                                                                          \
    switch (slot) {                                                                                        \
        case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break;    \
        case 4: CALL(4); break; case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break;    \
        case 8: CALL(8); break; case 9: CALL(9); break; case 10: CALL(10); break; case 11: CALL(11); break; \
        case 12: CALL(12); break; case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; \
        case 16: CALL(16); break; case 17: CALL(17); break; case 18: CALL(18); break; case 19: CALL(19); break; \
        case 20: CALL(20); break; case 21: CALL(21); break; case 22: CALL(22); break; default: CALL(23); break; \
    }

template <class Steps, int FIRST, int COUNT>
struct FinishWeave {
    Steps& f;
    const RqsDev& sp;
    template <int SLOT>
    __device__ __forceinline__ void at() {
        constexpr int N2 = 2 * Steps::kNumSlices;
        run_range<Steps, N2 + FIRST + (SLOT * COUNT) / 24, N2 + FIRST + ((SLOT + 1) * COUNT) / 24>(f, sp);
    }
    __device__ __forceinline__ void step(int slot) {
#define NFA_K8S_AT(S) at<S>()
        NFA_K8S_SLOT_SWITCH(NFA_K8S_AT)
#undef NFA_K8S_AT
    }
};

template <class Steps, int I, int END>
__device__ __forceinline__ void num_range(Steps& f) {   // slice I: even = width numerators, odd = height numerators
    if constexpr (I < END) {
        if constexpr ((I & 1) == 0) f.template num_w<(I >> 1)>();
        else f.template num_h<(I >> 1)>();
        num_range<Steps, I + 1, END>(f);
    }
}

template <class Steps>
struct NumWeave {
    Steps& f;
    template <int SLOT>
    __device__ __forceinline__ void at() {
        constexpr int COUNT = 2 * Steps::kNumSlices;
        num_range<Steps, (SLOT * COUNT) / 24, ((SLOT + 1) * COUNT) / 24>(f);
    }
    __device__ __forceinline__ void step(int slot) {
#define NFA_K8S_AT(S) at<S>()
        NFA_K8S_SLOT_SWITCH(NFA_K8S_AT)
#undef NFA_K8S_AT
    }
};

// RING: slots of the weight ring (RING - 1 stages in flight).  A small batch is bound by the bytes one CU has in
// flight from L2: with 16 rows per wave the row tiles are small enough for seven slots (96 KB in flight instead of 48).
// DBG (rqs_resnet_f16_dbg.hip): the same kernel with the last layer's chosen bins stored to a.dbg_bins (FusedSteps' kbin).
template <bool INVERSE, int INIT_KS, int RING, int NW_ = kWavesPerGroup, bool DBG = false>
__global__ void __launch_bounds__(NW_* kWave, 2) rqs_resnet_f16s_kernel(const Args a) {
    constexpr int NW = NW_, kThreads = NW * kWave;
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ int s_final[128];
    __shared__ int s_bad[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, dt = a.dt;
    int my_status = 0;
    if (tid < 128) {
        const int v = a.final_tab[tid];
        if (tid < D && (v < 0 || v >= D)) my_status |= NFA_STATUS_BAD_INDEX;
        s_final[tid] = v < 0 ? 0 : (v >= D ? D - 1 : v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no ordinary load in flight once the stream starts)

    using Stream = WeightStream<NW, RING>;
    Stream sm;
    sm.w = a.w;
    sm.ring = reinterpret_cast<vec4f*>(lds_dyn);
    sm.fetch = 0;
    sm.num_stages = a.num_stages * a.num_layers;
    sm.tid = tid;
    sm.sync = 0;
    sm.gen = NW;
    sm.peek = 0;
#pragma unroll
    for (int j = 0; j < RING - 1; ++j) {   // stages 0 .. RING - 2 -> slots 0 .. RING - 2
        sm.slot = ring_next<Stream>(j, 1);
        stream_request(sm);
    }
    sm.slot = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Frags fr;   // the weight fragments the next MFMAs need (carried across stages, layers and row blocks)
    fr.h = sm.ring[lane];
    fr.l = sm.ring[64 + lane];

    const int pblock = (a.param_words + 3) & ~3;
    float* s_row = lds_dyn + RING * kStageVec4 * 4 + wave * D * kRowPad16;
    float* s_param = lds_dyn + RING * kStageVec4 * 4 + NW * D * kRowPad16;   // [2][pblock]
    const int groups = dt >> 2;
    const int64_t num_quads = a.batch / (16 * NW);
    int pb = 0;

    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = quad * (16 * NW) + (wave << 4);
        int lane_here = lane, di = a.di;
        asm volatile("" : "+v"(lane_here), "+s"(di));
        const int g = lane_here >> 4, n = lane_here & 15;
        // ---- the wave's 16 rows: one coalesced read; slot j of the tile = input column j
        {
            const vec4f* xv = reinterpret_cast<const vec4f*>(a.x + row0 * D);
            const int nvec = D * 4;
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
                        s_row[(c0 + 0) * kRowPad16 + rr] = v[u].x;
                        s_row[(c0 + 1) * kRowPad16 + rr] = v[u].y;
                        s_row[(c0 + 2) * kRowPad16 + rr] = v[u].z;
                        s_row[(c0 + 3) * kRowPad16 + rr] = v[u].w;
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        float lad_acc = 0.0f;
        float worst = 0.0f;
        int quad_status = 0;
        for (int layer = 0; layer < a.num_layers; ++layer) {
            // (the two waves of a SIMD alternate the higher issue priority layer by layer)
            if ((layer + (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            // ---- the layer's parameter stage(s): ring -> parameter block `pb`
            float* prm = s_param + pb * pblock;
            for (int p = 0; p < a.param_stages; ++p) {
                unsigned cur, nxt;
                stage_begin(sm, cur, nxt, lane);
                const vec4f* src = sm.ring + sm.slot * kStageVec4;
                vec4f* dst = reinterpret_cast<vec4f*>(prm) + p * kParamVec4;
                const int used = (pblock >> 2) - p * kParamVec4;
                for (int i = tid; i < (used < kParamVec4 ? used : kParamVec4); i += kThreads) {
                    vec4f v = src[i];
                    if (p == 0 && i < kTabWords / 4) {
                        uvec4 u = __builtin_bit_cast(uvec4, v);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int idx = i * 4 + c;
                            const int e = (int)u[c];
                            const bool used_entry = idx < kTabTr ? idx < a.di : idx - kTabTr < dt;
                            if (used_entry && (e < 0 || e >= D)) my_status |= NFA_STATUS_BAD_INDEX;
                            u[c] = (unsigned)(e < 0 ? 0 : (e >= D ? D - 1 : e));
                        }
                        v = __builtin_bit_cast(vec4f, u);
                    }
                    dst[i] = v;
                }
                fr = next_frags<kPairs - 1>(cur, nxt);   // pair 0 of the stage behind this one
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr.h), "+v"(fr.l));
                stream_advance(sm, true);
            }
            const int* tab = reinterpret_cast<const int*>(prm);
            const float* gemm = prm + kTabWords;   // header + biases of the next GEMM
            pb ^= 1;

            uvec4 ph[4], pl[4];   // the current activations (128 k per sample) as f16 pieces (8 per register quad)
            f32x4 hacc[8];        // the residual stream h in fp32 (x the scale of the GEMM that wrote it)

            // ---- identity features (scale 1): k = 32 S + 8 g + j
#pragma unroll
            for (int S = 0; S < INIT_KS; ++S) {
                uvec4 hw, lw;
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const int i0 = S * 32 + g * 8 + j2 * 2;
                    float v0 = s_row[tab[kTabId + i0] * kRowPad16 + n], v1 = s_row[tab[kTabId + i0 + 1] * kRowPad16 + n];
                    v0 = i0 < di ? v0 : 0.0f;
                    v1 = i0 + 1 < di ? v1 : 0.0f;
                    unsigned hi, lo;
                    split2(v0, v1, hi, lo);
                    hw[j2] = hi;
                    lw[j2] = lo;
                }
                ph[S] = hw;
                pl[S] = lw;
            }
            // ---- initial layer
            {
                const float* bias = gemm + kHdr + g * 4;
#pragma unroll
                for (int t = 0; t < 8; ++t) load_bias4(hacc[t], bias + t * 16);
                kstep_stage(hacc, ph[0], pl[0], sm, fr, lane, NoWeave16{});
                if constexpr (INIT_KS == 2) kstep_stage(hacc, ph[1], pl[1], sm, fr, lane, NoWeave16{});
            }
            float conv_scale = gemm[0];
            gemm += kHdr + 128;

            // ---- residual blocks: h += W_1 relu(W_0 relu(h) + b_0) + b_1
            for (int blk = 0; blk < a.num_blocks; ++blk) {
                uvec4 qh[4], ql[4];   // pieces of relu(u)
                f32x4 u[8];
                {
                    const float* bias = gemm + kHdr + g * 4;
#pragma unroll
                    for (int t = 0; t < 8; ++t) load_bias4(u[t], bias + t * 16);
                    gemm_converting(u, ph, pl, hacc, conv_scale, worst, sm, fr, lane);
                    conv_scale = gemm[0];
                }
                gemm += kHdr + 128;
                {
                    // second Linear accumulates into the residual stream itself: hacc = hacc * ratio + bias, then + W_1 relu(u)
                    const float* bias = gemm + kHdr + g * 4;
                    const float ratio = gemm[1];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const vec4f b = *reinterpret_cast<const vec4f*>(bias + t * 16);
                        hacc[t][0] = __builtin_fmaf(hacc[t][0], ratio, b.x);
                        hacc[t][1] = __builtin_fmaf(hacc[t][1], ratio, b.y);
                        hacc[t][2] = __builtin_fmaf(hacc[t][2], ratio, b.z);
                        hacc[t][3] = __builtin_fmaf(hacc[t][3], ratio, b.w);
                    }
                    gemm_converting(hacc, qh, ql, u, conv_scale, worst, sm, fr, lane);
                    conv_scale = gemm[0];
                }
                gemm += kHdr + 128;
            }
            // pieces of h itself for the final layer (no ReLU in front of it: resnet.py:99-100)
            {
                float peak = 0.0f;
                convert_kstep<false>(ph[0], pl[0], hacc[0], hacc[1], conv_scale, peak);
                convert_kstep<false>(ph[1], pl[1], hacc[2], hacc[3], conv_scale, peak);
                convert_kstep<false>(ph[2], pl[2], hacc[4], hacc[5], conv_scale, peak);
                convert_kstep<false>(ph[3], pl[3], hacc[6], hacc[7], conv_scale, peak);
                worst = __builtin_fmaxf(worst, peak * conv_scale);
            }

            // ---- final layer: the six tiles of a group hold the 24 logits of this lane's feature 4 G + g
            {
                using Steps = FusedSteps<INVERSE, 8, DBG>;
                Steps f;
                const float kappa = gemm[0];
                f.kappa = kappa;
                f.kl2e = 1.44269502162933349609375f * kappa;
                f.tail_s = a.sp.tail_logit * gemm[1];   // gemm[1] = 1 / kappa
                const float* fbias = gemm + kHdr + g * 4;
                // The evaluation of group G - 1 rides behind the MFMAs of group G: finish in two halves behind
                // stages 0 and 1 (tiles 0 .. 3 wait in their accumulators meanwhile), both numerator sets of group G
                // behind stage 2.  One slice or less per MFMA.
                constexpr int FIN = Steps::kFinishSlices, FIN0 = FIN / 2;
                NoWeave16 none;
                FinishWeave<Steps, 0, FIN0> fin0{f, a.sp};
                FinishWeave<Steps, FIN0, FIN - FIN0> fin1{f, a.sp};
                NumWeave<Steps> nums{f};
                float* slot_prev = nullptr;
                for (int G = 0; G < groups; ++G) {
                    float* slot = s_row + tab[kTabTr + G * 4 + g] * kRowPad16 + n;
                    f32x4 t[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) load_bias4(t[i], fbias + (G * 6 + i) * 16);
                    if (G > 0) {
                        tile_pair_stage(t[0], t[1], ph, pl, sm, fr, lane, fin0);
                        tile_pair_stage(t[2], t[3], ph, pl, sm, fr, lane, fin1);
                        *slot_prev = f.y;
                        lad_acc += f.lad;
                        quad_status |= f.status;
                        if constexpr (DBG) {
                            if (layer == a.num_layers - 1) a.dbg_bins[(row0 + n) * dt + (G - 1) * 4 + g] = f.kbin;
                        }
                    } else {
                        tile_pair_stage(t[0], t[1], ph, pl, sm, fr, lane, none);
                        tile_pair_stage(t[2], t[3], ph, pl, sm, fr, lane, none);
                    }
                    f.x = *slot;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f.ew[j] = t[0][j];
                        f.ew[4 + j] = t[1][j];
                        f.eh[j] = t[2][j];
                        f.eh[4 + j] = t[3][j];
                    }
                    tile_pair_stage(t[4], t[5], ph, pl, sm, fr, lane, nums);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f.sd[j] = t[4][j];
                        if (j < 3) f.sd[4 + j] = t[5][j];
                    }
                    slot_prev = slot;
                }
                run_range<Steps, 2 * Steps::kNumSlices, 2 * Steps::kNumSlices + FIN>(f, a.sp);
                *slot_prev = f.y;
                lad_acc += f.lad;
                quad_status |= f.status;
                if constexpr (DBG) {
                    if (layer == a.num_layers - 1) a.dbg_bins[(row0 + n) * dt + (groups - 1) * 4 + g] = f.kbin;
                }
            }
            // this wave's spline results must be visible to its own gathers of the next layer
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // ---- results: position p of a row comes from slot final[p]; a block with any non-finite value or an
        //      activation beyond the f16 range is not written: the exact kernel redoes it from the inputs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the stream: ordinary stores / loads follow
        lad_acc += __shfl_xor(lad_acc, 16, kWave);
        lad_acc += __shfl_xor(lad_acc, 32, kWave);
        float sumsq = 0.0f;
        for (int j = g; j < a.Ds; j += 4) {
            const float v = s_row[j * kRowPad16 + n];
            sumsq = __builtin_fmaf(v, v, sumsq);
        }
        sumsq += __shfl_xor(sumsq, 16, kWave);
        sumsq += __shfl_xor(sumsq, 32, kWave);
        const bool bad = !(__builtin_fabsf(lad_acc) < INFINITY) || !(__builtin_fabsf(sumsq) < INFINITY) || !(worst < kF16Overflow);
        const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) s_bad[wave] = wave_bad ? 1 : 0;
        __syncthreads();
        int any_bad = 0;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) any_bad |= s_bad[w_];
        const bool quad_bad = any_bad != 0;
        if (!quad_bad) {
            if (!a.skip_out) {
                const int nvec = D * 4;
                vec4f* ov = reinterpret_cast<vec4f*>(a.out + row0 * D);
                for (int e = lane; e < nvec; e += kWave) {
                    const int rr = (e * 4) / D, c0 = e * 4 - rr * D;
                    vec4f v;
                    v.x = s_row[s_final[c0 + 0] * kRowPad16 + rr];
                    v.y = s_row[s_final[c0 + 1] * kRowPad16 + rr];
                    v.z = s_row[s_final[c0 + 2] * kRowPad16 + rr];
                    v.w = s_row[s_final[c0 + 3] * kRowPad16 + rr];
                    ov[e] = v;
                }
            }
            if (g == 0) {
                float* dst = a.lad + row0 + n;
                float v = a.accumulate ? *dst + lad_acc : lad_acc;
                if (a.normal) v = (-0.5f * sumsq - a.log_z) + v;   // normal.py:31-33, flows/base.py:49
                *dst = v;
            }
            my_status |= quad_status;
        }
        // one flag per 128 rows: 1 = the whole block is open; four-wave workgroups share a flag (zeroed by the
        // launcher) and set bit 1 / bit 2 for its lower / upper 64 rows -- the half of the OTHER workgroup is written,
        // its log-determinant accumulated, and must not be written again by the second pass
        if constexpr (NW == 8) {
            if (tid == 0) a.redo[quad] = quad_bad ? 1 : 0;
        } else {
            if (tid == 0 && quad_bad) atomicOr(a.redo + (quad >> 1), 2 << (quad & 1));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // s_bad is rewritten by the next row block
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (my_status && a.status) atomicOr(a.status, my_status);
}

// the 64-row form's `redo` words start at zero (each half block ORs its bit in)
__global__ void zero_words_kernel(int32_t* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

}  // namespace k8s
}  // namespace nfa

using namespace nfa;

static int launch_tile16(const float* inputs, const void* stream_packed, int32_t param_stages,
                         const int32_t* final_positions, int32_t num_layers, float* outputs, float* logabsdet,
                         int32_t* redo_blocks, int32_t* status, int64_t batch, int32_t features, int32_t num_transform,
                         int32_t num_identity, int32_t hidden_features, int32_t num_blocks, const nfa_rqs_spec* spec,
                         int32_t flags, void* stream, int32_t* dbg_bins) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET | NFA_FLAG_STANDARD_NORMAL_LOG_PROB |
                  NFA_FLAG_SKIP_OUTPUTS | NFA_FLAG_PAD_COLUMNS_MASK))
        return NFA_ERR_INVALID_ARGUMENT;
    if (!density_flags_valid(flags)) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1 || num_transform < 1 || num_identity < 1 || num_transform > features ||
        num_identity > features || num_blocks < 0 || num_layers < 1 || param_stages < 1)
        return NFA_ERR_INVALID_ARGUMENT;
    k8h::Args a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (a.sp.beta != 1.0f) return NFA_ERR_UNSUPPORTED;
    if (a.sp.K != 8 || !a.sp.linear || hidden_features != 128 || (num_transform & 3) != 0 || num_transform > 64 ||
        num_identity > 64 || features > 128 || (features & 3) != 0 || (batch & 127) != 0 || num_blocks > 64 ||
        num_layers > 4096)
        return NFA_ERR_UNSUPPORTED;
    const int param_words = k8h::kTabWords + (k8h::kHdr + 128) * (1 + 2 * num_blocks) + k8h::kHdr + num_transform * 24;
    if (param_stages * 2048 < param_words || param_stages > 4) return NFA_ERR_INVALID_ARGUMENT;
    if (batch == 0) return NFA_OK;
    if (!inputs || !stream_packed || !final_positions || !logabsdet || !redo_blocks ||
        (!outputs && !(flags & NFA_FLAG_SKIP_OUTPUTS)))
        return NFA_ERR_INVALID_ARGUMENT;
    a.ctx = nullptr;
    a.ce = 0;
    a.dbg_bins = dbg_bins;
    a.dbg_logits = nullptr;
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
    const int init_ks = num_identity > 32 ? 2 : 1;
    // stages per layer: parameters, one per k-step of the initial layer, four per hidden Linear, three per group
    // of four transformed features -- the same count as K8h's stream
    a.num_stages = param_stages + (init_ks == 2 ? 2 : 1) + 8 * num_blocks + num_transform * 24 / 32;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    a.trace = nullptr;
    const size_t lds_cap = 160 * 1024 - 1024;
    const int cus = device_cu_count();
    // fewer 64-row blocks than CUs: four-wave workgroups (one wave per SIMD) spread the batch over twice the CUs
    static const int half_env = getenv("NFA_K8S_HALF") ? atoi(getenv("NFA_K8S_HALF")) : 1;
    const bool half = half_env == 2 || (half_env == 1 && batch / 64 <= cus);
    const int nw = half ? 4 : k8s::kWavesPerGroup;
    auto lds_for = [&](int ring) {
        return (size_t)ring * k8h::kStageVec4 * 16 + (size_t)nw * features * k8s::kRowPad16 * sizeof(float) +
               (size_t)2 * ((param_words + 3) & ~3) * sizeof(float);
    };
    static const int ring_env = getenv("NFA_K8S_RING") ? atoi(getenv("NFA_K8S_RING")) : 7;
    const int ring = (!half && ring_env == 7 && lds_for(7) <= lds_cap) ? 7 : 4;
    const size_t lds_launch = lds_for(ring);
    if (lds_launch > lds_cap) return NFA_ERR_UNSUPPORTED;
    int64_t blocks = batch / (16 * nw);
    if (blocks > cus) blocks = cus;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    profile_next_launch(&e0, &e1);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(nw * kWave);
    const bool inv = (flags & NFA_FLAG_INVERSE) != 0;
    void (*kern)(const k8h::Args) = nullptr;
    int which = half ? 8 + (inv ? 1 : 0) + (init_ks == 2 ? 2 : 0) : (inv ? 1 : 0) + (init_ks == 2 ? 2 : 0) + (ring == 7 ? 4 : 0);
    if (dbg_bins) {   // the diagnostic instances: d_i <= 32, the two workgroup forms the launcher picks by itself
        if (init_ks != 1 || (!half && ring != 7)) return NFA_ERR_UNSUPPORTED;
        which = 12 + (inv ? 1 : 0) + (half ? 2 : 0);
        kern = half ? (inv ? k8s::rqs_resnet_f16s_kernel<true, 1, 4, 4, true> : k8s::rqs_resnet_f16s_kernel<false, 1, 4, 4, true>)
                    : (inv ? k8s::rqs_resnet_f16s_kernel<true, 1, 7, k8s::kWavesPerGroup, true>
                           : k8s::rqs_resnet_f16s_kernel<false, 1, 7, k8s::kWavesPerGroup, true>);
    } else switch (which) {
        case 0: kern = k8s::rqs_resnet_f16s_kernel<false, 1, 4>; break;
        case 1: kern = k8s::rqs_resnet_f16s_kernel<true, 1, 4>; break;
        case 2: kern = k8s::rqs_resnet_f16s_kernel<false, 2, 4>; break;
        case 3: kern = k8s::rqs_resnet_f16s_kernel<true, 2, 4>; break;
        case 4: kern = k8s::rqs_resnet_f16s_kernel<false, 1, 7>; break;
        case 5: kern = k8s::rqs_resnet_f16s_kernel<true, 1, 7>; break;
        case 6: kern = k8s::rqs_resnet_f16s_kernel<false, 2, 7>; break;
        case 7: kern = k8s::rqs_resnet_f16s_kernel<true, 2, 7>; break;
        case 8: kern = k8s::rqs_resnet_f16s_kernel<false, 1, 4, 4>; break;
        case 9: kern = k8s::rqs_resnet_f16s_kernel<true, 1, 4, 4>; break;
        case 10: kern = k8s::rqs_resnet_f16s_kernel<false, 2, 4, 4>; break;
        default: kern = k8s::rqs_resnet_f16s_kernel<true, 2, 4, 4>; break;
    }
    note_layer_kernel("k8s::rqs_resnet_f16s_kernel<inverse=%d, init_ks=%d, waves=%d, K=8, ring=%d>", inv ? 1 : 0, init_ks, nw, ring);
    // (a kernel, not hipMemsetAsync: captured into a HIP graph the memset NODE cost ~2 ms per replay -- GraphedLogProb at
    //  <= 16 384 rows took 2.8 ms where the launches themselves take 0.6, tools/small_batch_probe.py)
    if (half) hipLaunchKernelGGL(k8s::zero_words_kernel, dim3((unsigned)((batch / 128 + 255) / 256)), dim3(256), 0, st,
                                 redo_blocks, (int)(batch / 128));
    if (lds_launch > 64 * 1024) {
        static unsigned long long raised[16] = {};   // device masks (raise_dynamic_lds)
        const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[which], (int)lds_cap);
        if (rc_lds != NFA_OK) return rc_lds;
    }
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds_launch, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds_launch, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_rqs_flow_resnet_f16x2_tile16_f32(const float* inputs, const void* stream_packed, int32_t param_stages,
                                                    const int32_t* final_positions, int32_t num_layers, float* outputs,
                                                    float* logabsdet, int32_t* redo_blocks, int32_t* status, int64_t batch,
                                                    int32_t features, int32_t num_transform, int32_t num_identity,
                                                    int32_t hidden_features, int32_t num_blocks, const nfa_rqs_spec* spec,
                                                    int32_t flags, void* stream) {
    return launch_tile16(inputs, stream_packed, param_stages, final_positions, num_layers, outputs, logabsdet, redo_blocks,
                         status, batch, features, num_transform, num_identity, hidden_features, num_blocks, spec, flags,
                         stream, nullptr);
}

// the same launch through the diagnostic instances (see nfa_rqs_flow_resnet_f16x2_bins_f32)
extern "C" int nfa_rqs_flow_resnet_f16x2_tile16_bins_f32(const float* inputs, const void* stream_packed, int32_t param_stages,
                                                         const int32_t* final_positions, int32_t num_layers, float* outputs,
                                                         float* logabsdet, int32_t* redo_blocks, int32_t* status,
                                                         int64_t batch, int32_t features, int32_t num_transform,
                                                         int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                                                         const nfa_rqs_spec* spec, int32_t flags, void* stream,
                                                         int32_t* bin_idx) {
    if (!bin_idx) return NFA_ERR_INVALID_ARGUMENT;
    return launch_tile16(inputs, stream_packed, param_stages, final_positions, num_layers, outputs, logabsdet, redo_blocks,
                         status, batch, features, num_transform, num_identity, hidden_features, num_blocks, spec, flags,
                         stream, bin_idx);
}
