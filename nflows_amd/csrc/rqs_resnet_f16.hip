// K8h: the whole-layer kernel of rqs_resnet.hip (ResidualNet conditioner, nn/nets/resnet.py:55-100,
// + everything K1 replaces, coupling.py:73-130, :549-582, for a run of layers in one launch) with
// its GEMMs on the f16 matrix pipe from TWO pieces per fp32 operand.
//
//   x = hi + lo,  hi = RN16(x),  lo = RN16(x - hi)        |x - hi - lo| <= 2^-24 |x|
//   x * w ~= hi_x hi_w + hi_x lo_w + lo_x hi_w            dropped: lo_x lo_w <= 2^-24 |x w|
//
// Three v_mfma_f32_32x32x16_f16 per k-step and tile instead of the six bf16 products of the
// three-piece scheme: half the matrix-pipe time, 4 bytes per weight instead of 6, and a piece
// conversion of 5-6 instead of 11 VALU instructions per pair.  Measured on the GPU against float64
// (tools/f16x2_probe.hip, K = 128, N(0,1)-like activations, U(-1/sqrt(128), 1/sqrt(128)) weights):
// max 5.9e-7 / rms 5.2e-8 -- the figures of a sequential fp32 fma chain (5.2e-7 / 5.2e-8) and
// better than the six-product bf16 scheme (6.6e-7 / 5.8e-8).
//
// What f16 needs that bf16 did not:
//   * range of the LOW pieces: a low piece below 2^-14 is subnormal (gfx950's f16 MFMA honours
//     subnormals, checked by the probe) and keeps only absolute precision 2^-25.  Weights are
//     therefore pre-scaled per GEMM by a power of two T (host: max |w T| in [2^13, 2^14)), which
//     makes every weight's low piece exact relative to the largest weight (2^-38); the scale comes
//     back out when the accumulators are converted to the next layer's pieces (one multiplication
//     by a power of two: exact) or, for the final layer, inside the spline evaluation
//     (FlatSteps<.., SCALED>).  Activations stay at scale S (1 by default): an activation below
//     0.25 / S carries an absolute error <= 2^-25 / S, which after the product with a weight
//     (|w| ~ 0.1) and the sum over 128 terms is ~3e-8 absolute on a layer output (probe: x 1e-2 and
//     x 1e-4 rows) -- half an ulp of 1.0, invisible next to the fp32 rounding of the sum itself.
//   * range of the HIGH pieces: |activation| * S > 65504 overflows.  Every row block checks its
//     results: a block with a non-finite output where ... (any non-finite output at all) writes
//     nothing and raises its entry of `redo`; the caller then runs the exact kernel
//     (nfa_rqs_flow_resnet_redo_f32, three bf16 pieces: full fp32 range) on the flagged blocks.
//     Overflow always poisons: an f16 infinity enters the products, its low piece is
//     x - inf = -inf, and inf - inf = NaN reaches every logit that depends on it.  Rows with
//     NaN / inf INPUTS take the same route, which keeps the reference's propagation rules for them.
//
// Everything else is the structure of rqs_resnet.hip: 32 samples per wave, row tile in LDS by
// slot, transposed GEMMs chained through the register file, weights streamed by LDS-DMA through a
// three-slot ring (8 KB stages here), final layer tile-major with the spline evaluation woven into
// its MFMAs (three-unit pipeline on 48 accumulator registers).
//
// Restrictions: K = 8 bins, linear tails, hidden width 128, ReLU blocks, d_i <= 64, d_t % 4 == 0,
// d_t <= 64, D % 4 == 0, D <= 128, batch % 128 == 0.

#include "fused_common.hpp"
#include "rqs_fused8.hpp"

#include <hip/hip_ext.h>
#include <stdlib.h>

namespace nfa {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace k8h {

constexpr int kStageVec4 = 512;    // 8 KB: [4 tiles][2 pieces][64 lanes] or [2 pieces][4 k-steps][64 lanes] x 16 B
constexpr int kRing = 6;           // five stages in flight behind the one being consumed
constexpr int kRowPad = 33;
constexpr int kTabId = 0, kTabTr = 64, kTabLayer = 128;
constexpr int kHdr = 4;            // floats in front of every GEMM's biases: {out_scale, skip_scale, 0, 0}

struct Args {
    const float* x;
    const vec4f* w;         // [num_layers * stages_per_layer][512] x 16 bytes
    const float* bias;      // per layer, per GEMM: header + accumulator-order biases (pre-scaled)
    const int32_t* tables;
    float* out;
    float* lad;
    int32_t* redo;          // [batch / 128]: 1 = block not written, run the exact kernel on it
    int32_t* status;
    int64_t batch;
    int D, dt, di, num_blocks, num_layers, num_stages, bias_per_layer, accumulate;
    RqsDev sp;
    unsigned long long* trace;  // debug: [gridDim.x][64] cycle stamps of wave 0 (first row block), null = off
};

#define NFA_HSTAMP() if (tr && ti < 32) tr[ti++] = __builtin_readcyclecounter();

// NW = waves per workgroup (4 or 8) sharing the ring
template <int NW_>
struct WeightStream {
    static constexpr int NW = NW_;
    const vec4f* w;
    vec4f* ring;
    int slot, fetch, num_stages, tid;
};

// Why the ring is deep: an LDS-DMA request lands 1-2 microseconds after it was issued when every CU
// streams, and the bytes a CU receives per second are (bytes in flight) / (that latency).  With two
// 8 KB stages in flight per workgroup the kernel ran at the speed of this stream (24 GB/s per CU,
// every stage barrier waiting for its data: tools/k8h_trace.py); five stages in flight cover it.
// Eight waves sharing one ring (one workgroup per CU) also halve the bytes a CU has to pull.
template <class SM>
__device__ __forceinline__ void stream_request(SM& sm) {
    constexpr int NW = SM::NW, kThreads = NW * kWave;
    const int dst_slot = sm.slot >= 1 ? sm.slot - 1 : kRing - 1;  // (slot + kRing - 1) % kRing
    const char* stage = reinterpret_cast<const char*>(sm.w) + (size_t)sm.fetch * (kStageVec4 * 16);
    const int wave = __builtin_amdgcn_readfirstlane(sm.tid >> 6);
    char* slot = reinterpret_cast<char*>(sm.ring) + dst_slot * (kStageVec4 * 16) + wave * (kWave * 16);
    const unsigned lane_off = (unsigned)sm.tid * 16u;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)((stage + i * kThreads * 16) + lane_off),
            (__attribute__((address_space(3))) void*)(slot + i * kThreads * 16), 16, 0, 0);
    sm.fetch = (sm.fetch + 1 == sm.num_stages) ? 0 : sm.fetch + 1;
}

// end of stage s: this wave's requests of stage s + 2 have landed (those of the three stages after
// it may still be in flight), every wave is done reading stage s.  Stage s + 1 was complete one
// barrier earlier, which is what lets a wave read the first weight fragments of the NEXT stage while
// it still issues the MFMAs of the current one (no LDS latency in front of any MFMA).
template <class SM>
__device__ __forceinline__ void stream_advance(SM& sm) {
    if constexpr (SM::NW == 8) asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    sm.slot = (sm.slot + 1 == kRing) ? 0 : sm.slot + 1;
}

// A stage is four fragment pairs (hi, lo pieces of a 32 x 16 weight block): pair g at vec4 offsets
// g * 128 (hi) and g * 128 + 64 (lo).  k-major stages: pair g = output tile g of one k-step; final
// layer: pair g = k-step g of one half tile.  `fr` always holds the pair the next MFMAs need; its
// successor -- the next pair of this stage or pair 0 of the next stage -- is requested from LDS
// before those MFMAs are issued.
struct Frags {
    vec4f h, l;
};

__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}

// (cur / nxt: LDS byte addresses of this lane's 16 bytes in the current / the next stage)
template <class SM>
__device__ __forceinline__ void stage_begin(SM& sm, unsigned& cur, unsigned& nxt, int lane) {
    stream_request(sm);
    const unsigned base = lds_address(sm.ring) + (unsigned)lane * 16u;
    cur = base + (unsigned)sm.slot * (kStageVec4 * 16);
    nxt = base + (unsigned)(sm.slot + 1 == kRing ? 0 : sm.slot + 1) * (kStageVec4 * 16);
}

// The fragment reads are written as asm: hipcc waits for every LDS read it knows about with
// lgkmcnt(0), i.e. also for the pair requested a moment ago for the NEXT group, which puts the full
// LDS latency in front of every second MFMA group (measured: 850-1200 cycles per k-step of twelve
// MFMAs instead of ~450).  Here the pair in `fr` is awaited with a counted lgkmcnt(2): LDS reads
// return in order, so with the two reads of the following pair as the only younger requests `fr` has
// landed (other LDS / scalar-memory traffic can only make the wait stricter, never weaker).
template <int G>
__device__ __forceinline__ Frags next_frags(unsigned cur, unsigned nxt) {
    Frags f;
    if constexpr (G < 3) {
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                     : "=v"(f.h), "=v"(f.l)
                     : "v"(cur), "i"((G + 1) * 2048), "i"((G + 1) * 2048 + 1024));
    } else {
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=v"(f.h), "=v"(f.l) : "v"(nxt));
    }
    return f;
}

// `fr` has landed (its successor's two reads are the only younger requests of this wave)
__device__ __forceinline__ void await_frags(Frags& fr) {
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fr.h), "+v"(fr.l));
}

// smallest terms first
#define NFA_MFMA3(acc, ah, al, bh, bl)                                            \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);           \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);           \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0)

typedef unsigned uvec4 __attribute__((ext_vector_type(4)));

// ReLU on a value given as f16 pieces: both are cleared where the leading piece is negative and
// not a NaN (f16 patterns 0x8000..0xFC00 = int16 <= -1024), so NaNs keep propagating like
// torch.relu's.
__device__ __forceinline__ void relu_pieces(f16x8& h, f16x8& l) {
    uvec4 hw = __builtin_bit_cast(uvec4, h), lw = __builtin_bit_cast(uvec4, l);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned keep;
#ifdef NFA_DBG_C
        {
            const unsigned a_ = hw[i];
            const unsigned lo_neg = ((a_ & 0x8000u) && (a_ & 0x7FFFu) <= 0x7C00u) ? 0xFFFFu : 0u;
            const unsigned hi_neg = ((a_ & 0x80000000u) && ((a_ >> 16) & 0x7FFFu) <= 0x7C00u) ? 0xFFFF0000u : 0u;
            keep = ~(lo_neg | hi_neg);
            hw[i] &= keep;
            lw[i] &= keep;
            continue;
        }
#endif
        asm volatile("v_pk_min_i16 %0, %1, 0\n\t"
            "v_pk_add_i16 %0, %0, %2\n\t"
            "v_pk_ashrrev_i16 %0, %3, %0\n\t"
            "v_not_b32 %0, %0"
            : "=&v"(keep)
            : "v"(hw[i]), "s"(0x03FF03FFu), "s"(0x000F000Fu));
        hw[i] &= keep;
        lw[i] &= keep;
    }
    h = __builtin_bit_cast(f16x8, hw);
    l = __builtin_bit_cast(f16x8, lw);
}

// NO PACKED fp32 ARITHMETIC IN THIS FILE (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): beside a
// co-resident wave that issues MFMAs, packed fp32 results came out wrong in lanes 16-31 / 48-63
// (measured: nondeterministic 1e-3 relative errors in the pieces of samples 16..31 of a wave,
// only with two workgroups per CU; gone with the packed forms removed -- DESIGN.md section 4).
// The residuals are therefore computed per element and pinned against re-vectorisation, and the
// file is compiled with -fno-slp-vectorize.
__device__ __forceinline__ void split2(vec2f v, f16x2& hi, f16x2& lo) {
    hi = __builtin_convertvector(v, f16x2);
    float r0 = v[0] - (float)hi[0], r1 = v[1] - (float)hi[1];
    asm volatile("" : "+v"(r0));
    asm volatile("" : "+v"(r1));
    lo = __builtin_convertvector(vec2f{r0, r1}, f16x2);
}

__device__ __forceinline__ f16x8 join4(f16x2 a, f16x2 b, f16x2 c, f16x2 d) {
    return f16x8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

// debug aid: a compiler-visible read of an accumulator register -- the compiler pads it with the
// MFMA -> VALU wait states, so every MFMA issued before has completed when it executes
__device__ __forceinline__ void mfma_drain(const f32x16& a) {
#if defined(NFA_DBG_G) || defined(NFA_DBG_H)
    const int t = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a[0]));
    asm volatile("" ::"s"(t));
#endif
}
__device__ __forceinline__ void mfma_drain_h(const f32x16& a) {
#ifdef NFA_DBG_H
    __builtin_amdgcn_sched_barrier(0);
    const int t = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a[0]));
    asm volatile("" ::"s"(t));
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ void mfma_drain_t(const f32x16& a) {
    const int t = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a[0]));
    asm volatile("" ::"s"(t));
}

// k-major GEMM: out^T[128 x 32 samples] += W[128 x 16*NKS] x act^T; one 8 KB stage per k-step
template <bool RELU, int NKS, class SM>
__device__ __forceinline__ void gemm_kmajor(f32x16 (&acc)[4], const f16x8 (&ph)[8], const f16x8 (&pl)[8],
                                            SM& sm, Frags& fr, int lane, unsigned long long* ft = nullptr) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        unsigned cur, nxt;
#ifdef NFA_K8H_FINE_TRACE
        if (ft) ft[3 * ks] = __builtin_readcyclecounter();
#endif
        stage_begin(sm, cur, nxt, lane);
        f16x8 bh = ph[ks], bl = pl[ks];
        if (RELU) relu_pieces(bh, bl);  // (the input pieces themselves stay: skip connection)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            Frags nf;
            if (t == 0) nf = next_frags<0>(cur, nxt);
            else if (t == 1) nf = next_frags<1>(cur, nxt);
            else if (t == 2) nf = next_frags<2>(cur, nxt);
            else nf = next_frags<3>(cur, nxt);
            await_frags(fr);
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 ah = __builtin_bit_cast(f16x8, fr.h), al = __builtin_bit_cast(f16x8, fr.l);
            NFA_MFMA3(acc[t], ah, al, bh, bl);
            mfma_drain_h(acc[t]);
            __builtin_amdgcn_sched_barrier(0);
            fr = nf;
        }
        mfma_drain(acc[3]);
#ifdef NFA_K8H_FINE_TRACE
        if (ft) { mfma_drain_t(acc[3]); ft[3 * ks + 1] = __builtin_readcyclecounter(); }
#endif
        stream_advance(sm);
#ifdef NFA_K8H_FINE_TRACE
        if (ft) ft[3 * ks + 2] = __builtin_readcyclecounter();
#endif
    }
}

// one 32-row output tile of the final layer without anything woven in (the first tile of a layer):
// two stages of [2 pieces][4 k-steps][64 lanes] x 16 bytes
template <class SM>
__device__ __forceinline__ void gemm_tile(f32x16& acc, const f16x8 (&ph)[8], const f16x8 (&pl)[8],
                                          SM& sm, Frags& fr, int lane) {
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        unsigned cur, nxt;
        stage_begin(sm, cur, nxt, lane);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int ks = hs * 4 + k4;
            Frags nf;
            if (k4 == 0) nf = next_frags<0>(cur, nxt);
            else if (k4 == 1) nf = next_frags<1>(cur, nxt);
            else if (k4 == 2) nf = next_frags<2>(cur, nxt);
            else nf = next_frags<3>(cur, nxt);
            await_frags(fr);
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 ah = __builtin_bit_cast(f16x8, fr.h), al = __builtin_bit_cast(f16x8, fr.l);
            NFA_MFMA3(acc, ah, al, ph[ks], pl[ks]);
            mfma_drain_h(acc);
            __builtin_amdgcn_sched_barrier(0);
            fr = nf;
        }
        mfma_drain(acc);
        stream_advance(sm);
    }
}

// ---- the final layer with the spline evaluation woven into its MFMAs (see rqs_resnet.hip) -----
enum { kUnitNone = 0, kUnitNumA = 1, kUnitFinishA = 2, kUnitFinishB = 3 };
constexpr int kSlots = 24;  // MFMAs of one tile

template <int UNIT, class Steps>
constexpr int spline_unit_slices() {
    return UNIT == kUnitNumA ? 2 * Steps::kNumSlices
                             : (UNIT == kUnitNone ? 0 : Steps::kNumSlices + Steps::kFinishSlices);
}

// Slice I of a unit.  U0: width / height numerators of A alternate (two independent chains);
// U1: finish A with the width numerators of B on every third position; U2: height numerators of
// B, then finish B (its single walk needs both numerator sets).
template <int UNIT, int I, class Steps>
__device__ __forceinline__ void spline_unit_slice(Steps& fa, Steps& fb, const RqsDev& sp) {
    constexpr int N = Steps::kNumSlices;
    if constexpr (UNIT == kUnitNumA) {
        if constexpr ((I & 1) == 0) fa.template num_w<(I >> 1)>();
        else fa.template num_h<(I >> 1)>();
    } else if constexpr (UNIT == kUnitFinishA) {
        static_assert(Steps::kFinishSlices >= 2 * N, "one numerator slice behind every two finish slices");
        if constexpr (I % 3 == 2 && I / 3 < N) fb.template num_w<I / 3>();
        else fa.template finish<I - ((I + 1) / 3 < N ? (I + 1) / 3 : N)>(sp);
    } else if constexpr (UNIT == kUnitFinishB) {
        if constexpr (I < N) fb.template num_h<I>();
        else fb.template finish<I - N>(sp);
    }
}

template <int UNIT, int I, int END, class Steps>
__device__ __forceinline__ void spline_unit_range(Steps& fa, Steps& fb, const RqsDev& sp) {
    if constexpr (I < END) {
        spline_unit_slice<UNIT, I>(fa, fb, sp);
        spline_unit_range<UNIT, I + 1, END>(fa, fb, sp);
    }
}

template <int UNIT, int SLOT, class Steps>
__device__ __forceinline__ void spline_unit_step(Steps& fa, Steps& fb, const RqsDev& sp) {
    constexpr int N = spline_unit_slices<UNIT, Steps>();
    spline_unit_range<UNIT, (SLOT * N) / kSlots, ((SLOT + 1) * N) / kSlots>(fa, fb, sp);
}

#define NFA_PUMP(SLOT, A_, B_)                                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, acc, 0, 0, 0); \
    mfma_drain_h(acc);                                                  \
    __builtin_amdgcn_sched_barrier(0);                                  \
    spline_unit_step<UNIT, SLOT>(fa, fb, sp);                           \
    __builtin_amdgcn_sched_barrier(0)

template <int UNIT, int KS, class Steps>
__device__ __forceinline__ void kstep_pumped(f32x16& acc, f16x8 bh, f16x8 bl, Frags& fr, unsigned cur,
                                             unsigned nxt, Steps& fa, Steps& fb, const RqsDev& sp) {
    const Frags nf = next_frags<(KS & 3)>(cur, nxt);   // the next k-step's fragments, three MFMAs ahead of their use
    await_frags(fr);
    const f16x8 ah = __builtin_bit_cast(f16x8, fr.h), al = __builtin_bit_cast(f16x8, fr.l);
    fr = nf;
    NFA_PUMP(KS * 3 + 0, al, bh);
    NFA_PUMP(KS * 3 + 1, ah, bl);
    NFA_PUMP(KS * 3 + 2, ah, bh);
    mfma_drain(acc);
}
#undef NFA_PUMP

template <int UNIT, int HS, class Steps, class SM>
__device__ __forceinline__ void stage_pumped(f32x16& acc, const f16x8 (&ph)[8], const f16x8 (&pl)[8],
                                             SM& sm, Frags& fr, int lane, Steps& fa, Steps& fb, const RqsDev& sp) {
    unsigned cur, nxt;
    stage_begin(sm, cur, nxt, lane);
    kstep_pumped<UNIT, HS * 4 + 0>(acc, ph[HS * 4 + 0], pl[HS * 4 + 0], fr, cur, nxt, fa, fb, sp);
    kstep_pumped<UNIT, HS * 4 + 1>(acc, ph[HS * 4 + 1], pl[HS * 4 + 1], fr, cur, nxt, fa, fb, sp);
    kstep_pumped<UNIT, HS * 4 + 2>(acc, ph[HS * 4 + 2], pl[HS * 4 + 2], fr, cur, nxt, fa, fb, sp);
    kstep_pumped<UNIT, HS * 4 + 3>(acc, ph[HS * 4 + 3], pl[HS * 4 + 3], fr, cur, nxt, fa, fb, sp);
    stream_advance(sm);
}

template <int UNIT, class Steps, class SM>
__device__ __forceinline__ void gemm_tile_pumped(f32x16& acc, const f16x8 (&ph)[8], const f16x8 (&pl)[8],
                                                 SM& sm, Frags& fr, int lane, Steps& fa, Steps& fb,
                                                 const RqsDev& sp) {
    stage_pumped<UNIT, 0>(acc, ph, pl, sm, fr, lane, fa, fb, sp);
    stage_pumped<UNIT, 1>(acc, ph, pl, sm, fr, lane, fa, fb, sp);
}

// accumulator tile t (times `scale`, a power of two), registers 8*hk .. 8*hk+7 -> pieces of k-step 2t + hk
template <bool RELU>
__device__ __forceinline__ void tile_to_pieces(const f32x16& a, float scale, f16x8& h0, f16x8& l0, f16x8& h1,
                                               f16x8& l1) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        v[q] = a[q] * scale;
        if (RELU) v[q] = (v[q] < 0.0f) ? 0.0f : v[q];  // NaN stays NaN
    }
    f16x2 hh[8], ll[8];
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) split2(vec2f{v[q2 * 2], v[q2 * 2 + 1]}, hh[q2], ll[q2]);
    h0 = join4(hh[0], hh[1], hh[2], hh[3]);
    l0 = join4(ll[0], ll[1], ll[2], ll[3]);
    h1 = join4(hh[4], hh[5], hh[6], hh[7]);
    l1 = join4(ll[4], ll[5], ll[6], ll[7]);
}

// skip connection: value of the pieces of one k-step times `scale`, added to 8 accumulator registers
__device__ __forceinline__ void add_pieces(f32x16& a, int q0, const f16x8& h, const f16x8& l, float scale) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[q0 + j] = __builtin_fmaf((float)h[j], scale, a[q0 + j]);
        a[q0 + j] = __builtin_fmaf((float)l[j], scale, a[q0 + j]);
    }
}

__device__ __forceinline__ void load_bias_tile(f32x16& acc, const float* bias_tile_half) {
    const vec4f* bp = reinterpret_cast<const vec4f*>(bias_tile_half);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const vec4f b = bp[q4];
        acc[q4 * 4 + 0] = b.x;
        acc[q4 * 4 + 1] = b.y;
        acc[q4 * 4 + 2] = b.z;
        acc[q4 * 4 + 3] = b.w;
    }
}

__device__ __forceinline__ bool not_finite(float v) { return !(__builtin_fabsf(v) < INFINITY); }

template <bool INVERSE, int INIT_KS, int NW>
__global__ void __launch_bounds__(NW * kWave, 2) rqs_resnet_f16_kernel(const Args a) {
    constexpr int kThreads = NW * kWave;
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ int s_tab[2][kTabLayer];
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

    WeightStream<NW> sm;
    sm.w = a.w;
    sm.ring = reinterpret_cast<vec4f*>(lds_dyn);
    sm.fetch = 0;
    sm.num_stages = a.num_stages * a.num_layers;
    sm.tid = tid;
#pragma unroll
    for (int j = 0; j < kRing - 1; ++j) {   // stages 0 .. kRing-2 -> slots 0 .. kRing-2
        sm.slot = j + 1 == kRing ? 0 : j + 1;
        stream_request(sm);
    }
    sm.slot = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Frags fr;   // the weight fragments the next MFMAs need (carried across stages, layers and row blocks)
    fr.h = sm.ring[lane];
    fr.l = sm.ring[64 + lane];

    float* s_row = lds_dyn + kRing * kStageVec4 * 4 + wave * D * kRowPad;
    float* s_fbias = lds_dyn + kRing * kStageVec4 * 4 + NW * D * kRowPad;
    const int groups = dt >> 2;
    const int64_t num_quads = a.batch / (32 * NW);   // row blocks of this workgroup size
    int tb = 0;

    unsigned long long* tr = nullptr;
    int ti = 1;
    if (a.trace && lane == 0 && wave == 0) {
        tr = a.trace + (size_t)blockIdx.x * 64;
        tr[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // HW_ID, XCC_ID
    }
    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = quad * (32 * NW) + (wave << 5);
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
#ifdef NFA_DBG_A
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif

        float lad_acc = 0.0f;
        int quad_status = 0;
        for (int layer = 0; layer < a.num_layers; ++layer) {
            // (the two workgroups of a CU alternate the higher issue priority, see rqs_resnet.hip)
#ifndef NFA_DBG_D
            if ((layer + (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#endif
            const int* tab = s_tab[tb];
            if (tid < kTabLayer) {
                const int nl = layer + 1 < a.num_layers ? layer + 1 : 0;
                s_tab[tb ^ 1][tid] = checked(a.tables[nl * kTabLayer + tid], tid < kTabTr ? tid < a.di : tid - kTabTr < dt);
            }
            const float* gemm = a.bias + (size_t)layer * a.bias_per_layer;   // header + biases of the next GEMM
            f16x8 ph[8], pl[8];  // the current activations (128 k per sample) as f16 pieces
            NFA_HSTAMP()

            // ---- identity features (scale 1): k = ks*16 + half*8 + j
#pragma unroll
            for (int ks = 0; ks < INIT_KS; ++ks) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = ks * 16 + half * 8 + j;
                    const float xv = s_row[tab[kTabId + i] * kRowPad + r];
                    v[j] = i < di ? xv : 0.0f;
                }
                f16x2 hh[4], ll[4];
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) split2(vec2f{v[j2 * 2], v[j2 * 2 + 1]}, hh[j2], ll[j2]);
                ph[ks] = join4(hh[0], hh[1], hh[2], hh[3]);
                pl[ks] = join4(ll[0], ll[1], ll[2], ll[3]);
            }

            // ---- initial layer
            {
                const float out_scale = gemm[0];
                const float* bias = gemm + kHdr + half * 16;
                f32x16 h[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) load_bias_tile(h[t], bias + t * 32);
                gemm_kmajor<false, INIT_KS>(h, ph, pl, sm, fr, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    tile_to_pieces<false>(h[t], out_scale, ph[2 * t], pl[2 * t], ph[2 * t + 1], pl[2 * t + 1]);
            }
            gemm += kHdr + 128;
            NFA_HSTAMP()
            {
                // the final layer's header and biases of this layer go to LDS once (every wave has
                // passed a stage barrier of this layer: nobody reads the previous layer's any more)
                const float* fb = a.bias + (size_t)layer * a.bias_per_layer + (kHdr + 128) * (1 + 2 * a.num_blocks);
                for (int i = tid; i < kHdr + dt * 24; i += kThreads) s_fbias[i] = fb[i];
                if (a.num_blocks == 0) __syncthreads();
            }

            // ---- residual blocks
            for (int blk = 0; blk < a.num_blocks; ++blk) {
                f16x8 qh[8], ql[8];
                {
                    const float out_scale = gemm[0];
                    const float* bias = gemm + kHdr + half * 16;
                    f32x16 u[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) load_bias_tile(u[t], bias + t * 32);
                    gemm_kmajor<true, 8>(u, ph, pl, sm, fr, lane);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        tile_to_pieces<true>(u[t], out_scale, qh[2 * t], ql[2 * t], qh[2 * t + 1], ql[2 * t + 1]);
                }
                gemm += kHdr + 128;
                NFA_HSTAMP()
                const float out_scale = gemm[0], skip_scale = gemm[1];
                const float* bias = gemm + kHdr + half * 16;
                f32x16 v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    load_bias_tile(v[t], bias + t * 32);
                    add_pieces(v[t], 0, ph[2 * t], pl[2 * t], skip_scale);
                    add_pieces(v[t], 8, ph[2 * t + 1], pl[2 * t + 1], skip_scale);
                }
                gemm_kmajor<false, 8>(v, qh, ql, sm, fr, lane, (tr && layer == 0 && blk == 0) ? tr + 36 : nullptr);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    tile_to_pieces<false>(v[t], out_scale, ph[2 * t], pl[2 * t], ph[2 * t + 1], pl[2 * t + 1]);
                gemm += kHdr + 128;
                NFA_HSTAMP()
            }

            // ---- final layer with the spline evaluation woven into the MFMAs
            {
#ifdef NFA_K8H_FLATSTEPS
                using Steps = FlatSteps<INVERSE, 1, true, 8, true>;   // (rqs_resnet.hip's evaluation, for A/B runs)
#else
                using Steps = FusedSteps8<INVERSE>;
#endif
                Steps fa, fb;
                const float kappa = s_fbias[0];
                fa.kappa = fb.kappa = kappa;
                fa.kl2e = fb.kl2e = 1.44269502162933349609375f * kappa;
                fa.tail_s = fb.tail_s = a.sp.tail_logit * s_fbias[1];  // s_fbias[1] = 1 / kappa
                float* slot_b = nullptr;
                const float* fbias = s_fbias + kHdr + half * 16;
                f32x16 acc[3];
                auto commit = [&](Steps& f, float* slot) {
                    *slot = f.y;
                    lad_acc += f.lad;
                    quad_status |= f.status;
                };
#ifdef NFA_DBG_E
                for (int g = 0; g < groups; ++g) {
                    float* slot0 = s_row + tab[kTabTr + g * 4 + half * 2] * kRowPad + r;
                    float* slot1 = s_row + tab[kTabTr + g * 4 + half * 2 + 1] * kRowPad + r;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        load_bias_tile(acc[t], fbias + (g * 3 + t) * 32);
                        gemm_tile(acc[t], ph, pl, sm, fr, lane);
                    }
                    fa.x = *slot0;
                    fb.x = *slot1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fa.ew[j] = acc[0][j];
                        fa.eh[j] = acc[0][8 + j];
                        if (j < 7) fa.sd[j] = acc[1][j];
                        fb.ew[j] = acc[1][8 + j];
                        fb.eh[j] = acc[2][j];
                        if (j < 7) fb.sd[j] = acc[2][8 + j];
                    }
                    flat_steps_all(fa, a.sp);
                    flat_steps_all(fb, a.sp);
                    commit(fa, slot0);
                    slot_b = slot1;
                    if (g + 1 < groups) commit(fb, slot_b);
                }
#else
                for (int g = 0; g < groups; ++g) {
                    float* slot0 = s_row + tab[kTabTr + g * 4 + half * 2] * kRowPad + r;
                    float* slot1 = s_row + tab[kTabTr + g * 4 + half * 2 + 1] * kRowPad + r;
                    load_bias_tile(acc[0], fbias + (g * 3 + 0) * 32);
                    if (g > 0) {
                        gemm_tile_pumped<kUnitFinishB>(acc[0], ph, pl, sm, fr, lane, fa, fb, a.sp);
                        commit(fb, slot_b);
                    } else {
                        gemm_tile(acc[0], ph, pl, sm, fr, lane);
                    }
                    fa.x = *slot0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fa.ew[j] = acc[0][j];
                        fa.eh[j] = acc[0][8 + j];
                    }
                    load_bias_tile(acc[1], fbias + (g * 3 + 1) * 32);
                    gemm_tile_pumped<kUnitNumA>(acc[1], ph, pl, sm, fr, lane, fa, fb, a.sp);
                    fb.x = *slot1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (j < 7) fa.sd[j] = acc[1][j];
                        fb.ew[j] = acc[1][8 + j];
                    }
                    load_bias_tile(acc[2], fbias + (g * 3 + 2) * 32);
                    gemm_tile_pumped<kUnitFinishA>(acc[2], ph, pl, sm, fr, lane, fa, fb, a.sp);
                    commit(fa, slot0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fb.eh[j] = acc[2][j];
                        if (j < 7) fb.sd[j] = acc[2][8 + j];
                    }
                    slot_b = slot1;
                    NFA_HSTAMP()
                }
                spline_unit_range<kUnitFinishB, 0, spline_unit_slices<kUnitFinishB, Steps>()>(fa, fb, a.sp);
#endif
                commit(fb, slot_b);
            }
            tb ^= 1;
            NFA_HSTAMP()
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef NFA_DBG_B
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        }

        // ---- results: position p of a row comes from slot final[p].  A block with any non-finite
        //      value (f16 range exceeded somewhere, or non-finite inputs) is not written at all:
        //      the exact kernel redoes it from the inputs.
        lad_acc += __shfl_xor(lad_acc, 32, kWave);
        const int nvec = D * 8;
        bool bad = not_finite(lad_acc);
        for (int e = lane; e < nvec; e += kWave) {
            const int rr = (e * 4) / D, c0 = e * 4 - rr * D;
#pragma unroll
            for (int c = 0; c < 4; ++c) bad |= not_finite(s_row[s_final[c0 + c] * kRowPad + rr]);
        }
        bool quad_bad = false;
        if (!(a.accumulate & 2)) {
        const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) s_bad[wave] = wave_bad ? 1 : 0;
        __syncthreads();
        int any_bad = 0;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) any_bad |= s_bad[w_];
        quad_bad = any_bad != 0;
        }
        if (!quad_bad) {
            vec4f* ov = reinterpret_cast<vec4f*>(a.out + row0 * D);
            for (int e = lane; e < nvec; e += kWave) {
                const int rr = (e * 4) / D, c0 = e * 4 - rr * D;
                vec4f v;
                v.x = s_row[s_final[c0 + 0] * kRowPad + rr];
                v.y = s_row[s_final[c0 + 1] * kRowPad + rr];
                v.z = s_row[s_final[c0 + 2] * kRowPad + rr];
                v.w = s_row[s_final[c0 + 3] * kRowPad + rr];
                ov[e] = v;
            }
            if (half == 0) {
                float* dst = a.lad + row0 + r;
                *dst = (a.accumulate & 1) ? *dst + lad_acc : lad_acc;
            }
            my_status |= quad_status;
        }
        if (tid < NW / 4) a.redo[quad * (NW / 4) + tid] = quad_bad ? 1 : 0;   // one flag per 128 rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // s_bad is rewritten by the next row block
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two stages requested past the end
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace k8h
}  // namespace nfa

using namespace nfa;

extern "C" int nfa_rqs_flow_resnet_f16x2_f32(const float* inputs, const void* weights_packed,
                                             const float* bias_packed, const int32_t* flow_tables,
                                             int32_t num_layers, float* outputs, float* logabsdet,
                                             int32_t* redo_blocks, int32_t* status, int64_t batch,
                                             int32_t features, int32_t num_transform, int32_t num_identity,
                                             int32_t hidden_features, int32_t num_blocks,
                                             const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET)) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1 || num_transform < 1 || num_identity < 1 ||
        num_transform + num_identity > features || num_blocks < 0 || num_layers < 1)
        return NFA_ERR_INVALID_ARGUMENT;
    k8h::Args a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (a.sp.beta != 1.0f) return NFA_ERR_UNSUPPORTED;
    if (a.sp.K != 8 || !a.sp.linear || hidden_features != 128 || (num_transform & 3) != 0 || num_transform > 64 ||
        num_identity > 64 || features > 128 || (features & 3) != 0 || (batch & 127) != 0 || num_blocks > 64 ||
        num_layers > 4096)
        return NFA_ERR_UNSUPPORTED;
    if (batch == 0) return NFA_OK;
    if (!inputs || !weights_packed || !bias_packed || !flow_tables || !outputs || !logabsdet || !redo_blocks)
        return NFA_ERR_INVALID_ARGUMENT;
    a.x = inputs;
    a.w = reinterpret_cast<const vec4f*>(weights_packed);
    a.bias = bias_packed;
    a.tables = flow_tables;
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
    const int init_ks = num_identity > 32 ? 4 : 2;
    a.num_stages = init_ks + 16 * num_blocks + 2 * (num_transform * 24 / 32);
    a.bias_per_layer = (k8h::kHdr + 128) * (1 + 2 * num_blocks) + k8h::kHdr + num_transform * 24;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    a.trace = g_k7_trace;
    // workgroups of eight waves (256 rows, one per CU, one weight stream per CU) when the batch gives
    // every CU one; otherwise four waves (128 rows)
    const int cus = device_cu_count();
    static const int force_nw = getenv("NFA_K8H_WAVES") ? atoi(getenv("NFA_K8H_WAVES")) : 0;
    int nw = ((batch & 255) == 0 && (batch >> 8) >= cus) ? 8 : 4;
    if (force_nw == 4 || (force_nw == 8 && (batch & 255) == 0)) nw = force_nw;
    const size_t lds = (size_t)k8h::kRing * k8h::kStageVec4 * 16 +
                       (size_t)nw * features * k8h::kRowPad * sizeof(float) +
                       (size_t)(k8h::kHdr + num_transform * 24) * sizeof(float);
    if (lds + 2048 > 160 * 1024) nw = 4;
    const size_t lds_launch = (size_t)k8h::kRing * k8h::kStageVec4 * 16 +
                              (size_t)nw * features * k8h::kRowPad * sizeof(float) +
                              (size_t)(k8h::kHdr + num_transform * 24) * sizeof(float);
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
    const int which = (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (nw == 8 ? 4 : 0);
    switch (which) {
        case 0: kern = k8h::rqs_resnet_f16_kernel<false, 2, 4>; break;
        case 1: kern = k8h::rqs_resnet_f16_kernel<true, 2, 4>; break;
        case 2: kern = k8h::rqs_resnet_f16_kernel<false, 4, 4>; break;
        case 3: kern = k8h::rqs_resnet_f16_kernel<true, 4, 4>; break;
        case 4: kern = k8h::rqs_resnet_f16_kernel<false, 2, 8>; break;
        case 5: kern = k8h::rqs_resnet_f16_kernel<true, 2, 8>; break;
        case 6: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8>; break;
        default: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8>; break;
    }
    if (lds_launch > 64 * 1024) {
        static bool raised[8] = {false, false, false, false, false, false, false, false};
        if (!raised[which]) {
            NFA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            raised[which] = true;
        }
    }
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds_launch, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds_launch, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
