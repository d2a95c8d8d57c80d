// Shared by the whole-layer kernels on the f16 matrix pipe (rqs_resnet_f16.hip: K8h, 32 samples per wave on
// v_mfma_f32_32x32x16_f16; rqs_resnet_f16s.hip: K8s, 16 samples per wave on v_mfma_f32_16x16x32_f16): kernel arguments,
// the LDS-DMA weight stream and its ring, weight-fragment reads with counted waits, the piece conversion.
#pragma once

#include "fused_common.hpp"
#include "rqs_fused8.hpp"

#include <hip/hip_ext.h>
#include <stdlib.h>

namespace nfa {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace k8h {

// Timing ablations (tools/k8h_ablation.sh; results are garbage, never part of the product build):
//   -DNFA_ABL_NO_MFMA     the matrix instructions are left out (fragment reads, barriers, VALU work stay)
//   -DNFA_ABL_NO_WEAVE    no spline evaluation and no piece conversion behind the MFMAs
//   -DNFA_ABL_NO_FRAGS    the weight fragments are not re-read from LDS (same registers for every MFMA)
//   -DNFA_ABL_CONST_FRAGS as NO_FRAGS, but the registers keep REAL weights (the first fragment pair of the layer):
//                         the matrix instructions see a constant, non-trivial A operand
//   -DNFA_ABL_NO_BARRIER  the stage barriers are left out (the counted waits stay)
//   -DNFA_ABL_NO_DMA      no LDS-DMA requests (the ring keeps whatever it holds)
#ifdef NFA_ABL_CONST_FRAGS
#define NFA_ABL_NO_FRAGS
#endif
#ifdef NFA_ABL_NO_MFMA
#define NFA_K8H_MFMA(a, b, c, x, y, z) (c)
#else
#define NFA_K8H_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#endif
#ifdef NFA_ABL_CONST_FRAGS
#define NFA_K8H_KEEP_FRAGS(fr, nf) (void)nf;
#else
#define NFA_K8H_KEEP_FRAGS(fr, nf) fr = nf;
#endif
#ifdef NFA_ABL_NO_WEAVE
#define NFA_K8H_WEAVE(call)
#else
#define NFA_K8H_WEAVE(call) call
#endif

constexpr int kStageVec4 = 1024;   // 16 KB: eight (hi, lo) fragment pairs of [64 lanes] x 16 B = two k-steps of a
                                   // k-major GEMM or one 32-row tile of the final layer (one barrier each)
constexpr int kPairs = 8;          // fragment pairs per stage
constexpr int kParamVec4 = 512;    // a parameter stage carries 2048 words in its first 8 KB
constexpr int kRing = 4;           // rigid ring: three stages in flight behind the one being consumed
constexpr int kRingElastic = 5;    // elastic ring (below): one more slot for the waves that are a stage behind
constexpr int kRowPad = 33;
constexpr int kTabId = 0, kTabTr = 64, kTabWords = 128;   // parameter words [0, 128): slots of identity / transformed features
constexpr int kHdr = 4;            // floats in front of every GEMM's biases: {out_scale, skip_scale, 0, 0}
constexpr int kSlots = 24;         // MFMAs of one tile

struct Args {
    const float* x;
    const vec4f* w;            // the stream: per layer `param_stages` parameter stages, then the weight stages
    const int32_t* final_tab;  // [128]: slot stored at every output position of the run
    float* out;
    float* lad;
    int32_t* redo;             // [batch / 128]: 1 = block not written, run the exact kernel on it
    int32_t* status;
    int64_t batch;
    int D, dt, di, num_blocks, num_layers, num_stages, param_stages, param_words, accumulate;
    RqsDev sp;
    unsigned long long* trace;  // debug: [gridDim.x][64] cycle stamps of wave 0 (first row block), null = off
    int normal, skip_out;       // NFA_FLAG_STANDARD_NORMAL_LOG_PROB / NFA_FLAG_SKIP_OUTPUTS
    float log_z;                // 0.5 D log(2 pi)
    int Ds;                // columns the density sums over (features minus NFA_FLAG_PAD_COLUMNS)
    const float* ctx;           // CTX: [batch, ce] context rows (nn/nets/resnet.py:9-52, :92-100)
    int ce;
    int32_t* dbg_bins;          // the DBG instances only: [batch, dt] bin chosen by the LAST layer's evaluations
    float* dbg_logits;          // the DBG instances only (round 6), optional: [batch, dt * 24] the LAST layer's logits
                                // (accumulators x kappa), packed row order (tile, lane-half, register)
};

// (debug stamps: the switch and the index are wave-uniform -- scalar registers -- and the pointer is rebuilt from the
//  kernel arguments at every stamp: a per-lane pointer kept for the whole kernel cost three vector registers)
#define NFA_HSTAMP()                                                                                   \
    if (tracing && ti < 63) {                                                                          \
        if (lane == 0) a.trace[(size_t)blockIdx.x * 64 + ti] = __builtin_readcyclecounter();           \
        ++ti;                                                                                          \
    }

// NW = waves per workgroup (4 or 8) sharing the ring; RING = slots of 16 KB.
//
// RING == kRing (4), the rigid stream: one workgroup barrier at the end of every stage.
//
// RING == kRingElastic (5), the elastic stream (round 3).  The per-stage barrier is what keeps the eight waves
// of a workgroup in lock step -- every stage all of them wait for the slowest, and the two waves of a SIMD meet
// the same phase (fragment waits, VALU-heavy slices, DMA issue) at the same time (timing ablation: the kernel
// without its stage barriers runs 26 % faster, profiles/r3/k8h_ablation.txt).  Here a wave may be ONE stage
// ahead of the slowest one.  sync[slot] counts, per use of the slot, the waves that have (a) finished the
// stage two before the one the slot holds and (b) seen their own share of the slot's stage land:
//   end of stage s     own share of s + 2 landed (vmcnt), own reads of s done (lgkmcnt) -> sync[s + 2] += 1
//   inside stage s     before the first read of stage s + 1 (the fragment pair prefetched behind the last MFMAs
//                      of s): wait for sync[s + 1] == NW x generation: stage s + 1 is complete, and every wave
//                      has left stage s - 1
//   begin of stage s   request stage s + 3 into the slot of stage s - 2 (free: see the previous line, one
//                      stage earlier).  Three stages in flight as in the rigid ring; slot s - 1 is the one a
//                      straggler may still be reading.
// The counter is read at the beginning of the stage (the value is awaited behind the fragment reads, no
// extra latency) and polled only if that early value was not enough.  Parameter stages end with a real barrier
// (their contents are copied by all threads for all waves).
template <int NW_, int RING_>
struct WeightStream {
    static constexpr int NW = NW_, RING = RING_;
    static constexpr bool ELASTIC = RING_ == kRingElastic;
    const vec4f* w;
    vec4f* ring;
    int slot, fetch, num_stages, tid;
    unsigned sync;       // ELASTIC: LDS byte address of the [RING] counters
    unsigned gen;        // ELASTIC: NW x (uses of the current stage's slot so far, this one included)
    unsigned peek;       // ELASTIC: sync[next slot] as read at the beginning of the stage
};

template <class SM>
__device__ __forceinline__ int ring_next(int slot, int by = 1) {
    const int t = slot + by;
    return t >= SM::RING ? t - SM::RING : t;
}

// cache policy of the weight stream's LDS-DMA loads (round 6, the bounded energy experiment: profiles/r6/
// k8h_energy_experiment.txt -- 0 = default, 2 = nt, 16 = sc1; nothing moved the launch's energy, the default stays)
#ifndef NFA_K8H_DMA_AUX
#define NFA_K8H_DMA_AUX 0
#endif

template <class SM>
__device__ __forceinline__ void stream_request(SM& sm) {
    constexpr int NW = SM::NW, kThreads = NW * kWave;
    // rigid: the slot of the stage just finished (slot - 1); elastic: the one before that (slot - 2)
    const int dst_slot = ring_next<SM>(sm.slot, SM::ELASTIC ? SM::RING - 2 : SM::RING - 1);
    const char* stage = reinterpret_cast<const char*>(sm.w) + (size_t)sm.fetch * (kStageVec4 * 16);
    const int wave = __builtin_amdgcn_readfirstlane(sm.tid >> 6);
    char* slot = reinterpret_cast<char*>(sm.ring) + dst_slot * (kStageVec4 * 16) + wave * (kWave * 16);
    const unsigned lane_off = (unsigned)sm.tid * 16u;
#ifndef NFA_ABL_NO_DMA
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)((stage + i * kThreads * 16) + lane_off),
            (__attribute__((address_space(3))) void*)(slot + i * kThreads * 16), 16, 0, NFA_K8H_DMA_AUX);
#endif
    sm.fetch = (sm.fetch + 1 == sm.num_stages) ? 0 : sm.fetch + 1;
}

__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}

// ELASTIC, inside stage s, in front of the first read of stage s + 1: that stage is complete and the slot the
// next request goes to is free.  Fast path: the counter value read at the beginning of the stage (it is older
// than every fragment read still in flight: two of them at the call sites) already says so.
template <class SM>
__device__ __forceinline__ void stream_ensure_next(SM& sm) {
    if constexpr (SM::ELASTIC) {
        const unsigned need = sm.gen + (sm.slot + 1 == SM::RING ? SM::NW : 0);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(sm.peek));
        unsigned seen = __builtin_amdgcn_readfirstlane(sm.peek);
        if (seen < need) {
            const unsigned c = sm.sync + 4u * (unsigned)ring_next<SM>(sm.slot);
            do {
                __builtin_amdgcn_s_sleep(1);
                unsigned v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(c) : "memory");
                seen = __builtin_amdgcn_readfirstlane(v);
            } while (seen < need);
        }
        asm volatile("" ::: "memory");
    }
}

// end of stage s.  Rigid: this wave's requests of stage s + 2 have landed (those of stage s + 3 may still be
// in flight: 16 / NW per wave), every wave is done reading stage s (barrier).  Stage s + 1 was complete one
// barrier earlier, which is what lets a wave read the first weight fragments of the NEXT stage while
// it still issues the MFMAs of the current one (no LDS latency in front of any MFMA).
// Elastic: the same two waits, then the wave's tick on the counter of stage s + 2; `barrier` (parameter
// stages) additionally brings the workgroup together.
template <class SM>
__device__ __forceinline__ void stream_advance(SM& sm, bool barrier = false) {
    if constexpr (SM::ELASTIC) {
        if constexpr (SM::NW == 8) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        if ((sm.tid & 63) == 0)
            asm volatile("ds_add_u32 %0, %1" ::"v"(sm.sync + 4u * (unsigned)ring_next<SM>(sm.slot, 2)), "v"(1u) : "memory");
        if (barrier) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (sm.slot + 1 == SM::RING) sm.gen += SM::NW;
    } else {
#ifdef NFA_ABL_NO_BARRIER
        if constexpr (SM::NW == 8) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#else
        // (a deeper rigid ring -- RING slots, RING - 1 stages in flight -- keeps the same rule: stage s + 2's own share
        //  has landed, the stages behind it, RING - 3 of them, may still be in flight)
        if constexpr (SM::RING == 7 && SM::NW == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if constexpr (SM::RING == 6 && SM::NW == 8) asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if constexpr (SM::NW == 8) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        static_assert(SM::RING == 4 || SM::RING == 5 || (SM::NW == 8 && (SM::RING == 6 || SM::RING == 7)), "ring depth");
#endif
    }
    sm.slot = ring_next<SM>(sm.slot);
}

// A weight stage is eight fragment pairs (hi, lo pieces of a 32 x 16 weight block): pair g at byte
// offsets g * 2048 (hi) and g * 2048 + 1024 (lo) (+ 16 * lane).  `fr` always holds the pair the next
// MFMAs need; its successor -- the next pair of this stage or pair 0 of the next stage -- is requested
// from LDS before those MFMAs are issued.
struct Frags {
    vec4f h, l;
};

// (cur / nxt: LDS byte addresses of this lane's 16 bytes in the current / the next stage)
template <class SM>
__device__ __forceinline__ void stage_begin(SM& sm, unsigned& cur, unsigned& nxt, int lane) {
    // (all waves request at the beginning of the stage; spreading the requests over the stage's MFMA cells, one
    //  requesting wave per cell, measured 3-4 % slower: profiles/r3/k8h_dma_stagger.txt)
    stream_request(sm);
    const unsigned base = lds_address(sm.ring) + (unsigned)lane * 16u;
    cur = base + (unsigned)sm.slot * (kStageVec4 * 16);
    nxt = base + (unsigned)ring_next<SM>(sm.slot) * (kStageVec4 * 16);
    if constexpr (SM::ELASTIC)   // the next stage's counter, awaited in stream_ensure_next
        asm volatile("ds_read_b32 %0, %1" : "=v"(sm.peek) : "v"(sm.sync + 4u * (unsigned)ring_next<SM>(sm.slot)));
}

// The fragment reads are written as asm: hipcc waits for every LDS read it knows about with
// lgkmcnt(0), i.e. also for the pair requested a moment ago for the NEXT group.  Here the pair in
// `fr` is awaited with a counted lgkmcnt(2): LDS reads return in order, so with the two reads of the
// following pair as the only younger requests `fr` has landed (other LDS / scalar-memory traffic can
// only make the wait stricter, never weaker).
template <int G>
__device__ __forceinline__ Frags next_frags(unsigned cur, unsigned nxt) {
    Frags f;
#ifdef NFA_ABL_NO_FRAGS
    asm volatile("" : "=v"(f.h), "=v"(f.l) : "v"(cur), "v"(nxt));
    return f;
#endif
#ifdef NFA_ABL_DOUBLE_FRAGS   // (energy probe: every fragment pair is read twice into the same registers)
    if constexpr (G < kPairs - 1) {
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\tds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                     : "=&v"(f.h), "=&v"(f.l)
                     : "v"(cur), "i"((G + 1) * 2048), "i"((G + 1) * 2048 + 1024));
    } else {
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(f.h), "=&v"(f.l) : "v"(nxt));
    }
    return f;
#endif
    // EARLY-CLOBBER on the first destination (round 5).  Two reads, one address register: without the `&` hipcc gives
    // f.h the address register wherever the address dies in this statement (`ds_read_b128 v[30:33], v32` /
    // `ds_read_b128 v[66:69], v32 offset:1024`: 91 of the 204 K8h instances had such a site, profiles/r5/
    // k8h_exposed_fragment_reads_before_fix.txt).  LDS data lands ~100 cycles after the issue, so the second read
    // normally issues long before the first one's data arrives -- unless the wave is held between the two
    // instructions: an instruction-cache miss on cold code (the first launch, or a second stream keeping the device busy).
    // Then the second read takes its address from fragment data and the wave's next tile of logits is off: the
    // "one wave in a few thousand, 1e-3 off, a different one every launch" of rounds 3 to 5.
    // tests/test_host_logic.py::test_no_mfma_result_lands_on_its_own_operands checks every asm block of the five files.
    if constexpr (G < kPairs - 1) {
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                     : "=&v"(f.h), "=v"(f.l)
                     : "v"(cur), "i"((G + 1) * 2048), "i"((G + 1) * 2048 + 1024));
    } else {
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(f.h), "=v"(f.l) : "v"(nxt));
    }
    return f;
}

// `fr` has landed (its successor's two reads are the only younger requests of this wave)
__device__ __forceinline__ void await_frags(Frags& fr) {
#ifdef NFA_ABL_NO_FRAGS
    asm volatile("" : "+v"(fr.h), "+v"(fr.l));
    return;
#endif
#ifdef NFA_ABL_DOUBLE_FRAGS
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fr.h), "+v"(fr.l));
    return;
#endif
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fr.h), "+v"(fr.l));
}

typedef unsigned uvec4 __attribute__((ext_vector_type(4)));

// NO PACKED fp32 ARITHMETIC IN THIS FILE (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): beside a
// co-resident wave that issues MFMAs, the piece conversion written with packed fp32 forms gave
// nondeterministic 1e-3 relative errors in the pieces of samples 16..31 of a wave (lanes 16-31 /
// 48-63), only with two waves per SIMD; gone with the packed forms removed.
// The file is compiled with -fno-slp-vectorize.
//
// Piece conversion (round 3): two values v0, v1 (x `scale`, a power of two) -> packed f16 pairs
//   hi = RN16(v * scale)            v_fma_mixlo_f16 / v_fma_mixhi_f16  (fp32 fma, result rounded to f16)
//   lo = RN16(v * scale - hi)       the same instructions with the f16 `hi` as negated addend: the
//                                   product and the difference are exact in fp32 (hi = RN16 of it)
// four instructions per pair where convert / convert back / subtract / convert took nine.
// (one asm block per group: hipcc puts an `s_nop 0` between two adjacent asm statements)
__device__ __forceinline__ void split2_scaled(float v0, float v1, float scale, unsigned& hi, unsigned& lo) {
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l)
        : "v"(v0), "v"(v1), "v"(scale));
    hi = h;
    lo = l;
}

// (inputs at scale 1: the high pieces are one v_cvt_pk_f16_f32)
__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo) {
    unsigned h, l;
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l)
        : "v"(v0), "v"(v1));
    hi = h;
    lo = l;
}

// ReLU as v_max_f32 (one instruction; compare + select is two and a hazard wait).  v_max_f32 returns the
// other operand for a NaN, so the f16-range check can no longer ride on NaNs surviving the ReLUs: every
// conversion tracks max |value| instead (`peak`, one v_max3_f32 per pair) and the row block is handed to
// the exact kernel when a value x scale reaches the f16 overflow threshold.  NaN / inf INPUTS reach the
// check through the pass-through columns and the final layer (no ReLU in front of it), see the epilogue.
constexpr float kF16Overflow = 65520.0f;   // RN16 of anything >= this is infinity

// one pair of accumulator values -> (ReLU) -> peak, high pieces, low pieces.
// GUARD: three wait states in front.  The block's temporaries and results are written by VALU instructions the
// compiler's hazard recogniser does not look into; placed right behind an MFMA they may land on registers that MFMA
// still reads as its SrcC (a 16x16 MFMA reads it for three more issue slots).  K8h accumulates in place -- the
// accumulator registers stay live and cannot be handed to the block --, K8s's four-register accumulators are renamed
// from MFMA to MFMA, the old ones are free at once: without the guard one wave in a few thousand came out 1e-5 off.
#define NFA_CONVERT_RELU(PRE)                                                          \
    asm(PRE "v_max_f32 %2, %5, 0\n\t"                                                   \
            "v_max_f32 %3, %6, 0\n\t"                                                   \
            "v_fma_mixlo_f16 %0, %2, %7, 0 op_sel_hi:[0,0,0]\n\t"                       \
            "v_fma_mixhi_f16 %0, %3, %7, 0 op_sel_hi:[0,0,0]\n\t"                       \
            "v_max3_f32 %4, %4, %2, %3\n\t"                                             \
            "v_fma_mixlo_f16 %1, %2, %7, -%0 op_sel_hi:[0,0,1]\n\t"                     \
            "v_fma_mixhi_f16 %1, %3, %7, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"          \
        : "=&v"(h), "=&v"(l), "=&v"(m0), "=&v"(m1), "+v"(peak)                         \
        : "v"(s0), "v"(s1), "v"(scale))
#define NFA_CONVERT_PLAIN(PRE)                                                         \
    asm(PRE "v_fma_mixlo_f16 %0, %3, %5, 0 op_sel_hi:[0,0,0]\n\t"                       \
            "v_fma_mixhi_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]\n\t"                       \
            "v_max3_f32 %2, %2, |%3|, |%4|\n\t"                                         \
            "v_fma_mixlo_f16 %1, %3, %5, -%0 op_sel_hi:[0,0,1]\n\t"                     \
            "v_fma_mixhi_f16 %1, %4, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"          \
        : "=&v"(h), "=&v"(l), "+v"(peak)                                               \
        : "v"(s0), "v"(s1), "v"(scale))
// ACT: kActNone / kActRelu (what `false` / `true` convert to) or one of the other activations (fused_common.hpp:
// round 4, K8h only) -- those are evaluated in front and the plain conversion takes |value| for the peak
template <int ACT, bool GUARD = false>
__device__ __forceinline__ void convert_pair(float s0, float s1, float scale, float& peak, unsigned& hi, unsigned& lo) {
    unsigned h, l;
    if constexpr (ACT == kActRelu) {
        float m0, m1;
        if constexpr (GUARD) NFA_CONVERT_RELU("s_nop 2\n\t");
        else NFA_CONVERT_RELU("");
    } else {
        if constexpr (ACT != kActNone && activation_is_homogeneous(ACT)) {
            s0 = activate<ACT>(s0);
            s1 = activate<ACT>(s1);
        } else if constexpr (ACT != kActNone) {   // ELU, tanh: of the value at its own scale; `peak` is then post-scale
            s0 = activate<ACT>(s0 * scale);
            s1 = activate<ACT>(s1 * scale);
            scale = 1.0f;
        }
        if constexpr (GUARD) NFA_CONVERT_PLAIN("s_nop 2\n\t");
        else NFA_CONVERT_PLAIN("");
    }
    hi = h;
    lo = l;
}
#undef NFA_CONVERT_RELU
#undef NFA_CONVERT_PLAIN

}  // namespace k8h
}  // namespace nfa
