// Rational-quadratic spline kernels for gfx950 (MI355X): K1 fused coupling layer, K5 elementwise.
//
// What is computed, per (sample, transformed feature) -- reference lines are
// nflows/transforms/splines/rational_quadratic.py unless another file is named:
//   inside-interval test / identity tails         :26, :38-39
//   w,h logits / sqrt(hidden)                     coupling.py:554-556
//   softmax -> min + (1-min*K)*p -> cumsum -> knots, forced end knots, bin sizes as knot
//   differences                                    :91-98, :106-113
//   bin search (count of x >= knot, last knot + 1e-6)   utils/torchutils.py:134-136
//   min_d + softplus(d logits) at the bin's two knots   :100-104, :127-128 (tail logit :33-36)
//   forward rational-quadratic map + log-derivative     :162-181
//   inverse (quadratic root) + log-derivative           :132-160
//   split / scatter / per-sample sum of logabsdet       coupling.py:82-83, :96-98, :293
//
// Design: HBM-bound streaming kernel.  One workgroup (4 wave64) owns a tile of
// R whole samples; the tile's conditioner output (R*d_t*P contiguous floats) and inputs are
// brought in with aligned 16-byte-per-lane loads into LDS, each lane then evaluates one spline
// from its own P consecutive LDS words (stride P is odd for even K: conflict-free ds_read_b32),
// results are scattered into an LDS output tile and leave with aligned 16-byte stores; the
// per-sample logabsdet is a fixed-order wave shuffle reduction.  No atomics on the data path.
//
// Arithmetic follows aten's fp32 CPU kernels step by step (fp contraction is off; division
// and sqrt are IEEE; cumsum and the softmax denominator accumulate in double exactly like
// aten's cumsum does), so the only differences from the reference are <= 1 ulp in exp/log.

#include "rqs_math.hpp"

#include <hip/hip_ext.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>


namespace nfa {

// ------------------------------------------------------------------------------------------
// K1: fused coupling layer
struct CouplingArgs {
    const float* x;
    const float* params;
    const int64_t* tidx;
    const int64_t* perm;     // in_perm, may be null
    const int64_t* scatter;  // out_scatter, may be null
    float* out;
    float* lad;
    int32_t* bins;    // may be null: [batch, dt] bin chosen for every spline (rqs_eval's `bin`)
    int32_t* status;  // may be null
    int64_t batch;
    int D;   // features
    int dt;  // transformed features
    int R;   // samples per tile
    int C;   // splines per LDS chunk; R*dt unless one sample's parameters exceed the LDS budget
    int accumulate;  // 1: logabsdet[b] += sum (caller's running total), 0: logabsdet[b] = sum
    FastDiv div_dt, div_D;
    RqsDev sp;
    // LDS carve-up (float offsets)
    int off_x, off_out, off_lad, off_idx;
};

template <int KT, bool INVERSE, int BLOCK>
__global__ void __launch_bounds__(BLOCK) rqs_coupling_kernel(const CouplingArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_p = lds;
    float* s_x = lds + a.off_x;
    float* s_out = lds + a.off_out;
    float* s_lad = lds + a.off_lad;
    int* s_tidx = reinterpret_cast<int*>(lds + a.off_idx);  // [dt]
    int* s_src = s_tidx + a.dt;                             // [D] source column of each output column
    int* s_dst = s_src + a.D;                               // [D] output position of each column
    unsigned char* s_ist = reinterpret_cast<unsigned char*>(s_dst + a.D);  // [D] 1 = transformed

    const int tid = threadIdx.x;
    const int D = a.D, dt = a.dt, P = a.sp.P;
    int my_status = 0;

    for (int c = tid; c < D; c += BLOCK) {
        int src = c;
        if (a.perm) {
            const int64_t p = a.perm[c];
            if (p < 0 || p >= D) my_status |= NFA_STATUS_BAD_INDEX;
            src = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
        }
        int dst = c;
        if (a.scatter) {
            const int64_t p = a.scatter[c];
            if (p < 0 || p >= D) my_status |= NFA_STATUS_BAD_INDEX;
            dst = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
        }
        s_src[c] = src;
        s_dst[c] = dst;
        s_ist[c] = 0;
    }
    __syncthreads();
    for (int j = tid; j < dt; j += BLOCK) {
        const int64_t t = a.tidx[j];
        if (t < 0 || t >= D) my_status |= NFA_STATUS_BAD_INDEX;
        const int col = (int)(t < 0 ? 0 : (t >= D ? D - 1 : t));
        s_tidx[j] = col;
        s_ist[col] = 1;
    }
    // (visibility of s_tidx / s_ist is covered by the first barrier inside the loop)

    const int64_t num_tiles = (a.batch + a.R - 1) / a.R;
    for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * a.R;
        const int rows = (int)((a.batch - row0) < a.R ? (a.batch - row0) : a.R);
        const int nitems = rows * dt;

        const int mx = tile_load<BLOCK>(a.x + row0 * D, rows * D, s_x, tid);
        // the output tile is laid out as the 16-byte aligned image of its global destination
        float* s_o = s_out + tile_store_offset(a.out + row0 * D);
        const bool chunked = a.C < nitems;  // only with R == 1: one very wide sample (d_t*P large)
        float acc = 0.0f;                   // chunked mode: this lane's share of the sample's sum
        if (nitems == 0) {
            // a mask without transformed features (num_transform == 0): the layer is the fused
            // permutation alone; the chunk loop below, which normally hosts this copy, never runs
            __syncthreads();
            for (int e = tid; e < rows * D; e += BLOCK) {
                const int r = (int)fastdiv((uint32_t)e, a.div_D);
                const int c = e - r * D;
                s_o[e - c + s_dst[c]] = s_x[mx + e - c + s_src[c]];
            }
            __syncthreads();
        }
        for (int c0 = 0; c0 < nitems; c0 += a.C) {
            const int cn = (nitems - c0) < a.C ? (nitems - c0) : a.C;
            const int mp = tile_load<BLOCK>(a.params + (row0 * dt + c0) * (int64_t)P, cn * P, s_p, tid);
            __syncthreads();
            if (c0 == 0) {
                // untouched columns: bit-exact copy (with the fused permutation)
                for (int e = tid; e < rows * D; e += BLOCK) {
                    const int r = (int)fastdiv((uint32_t)e, a.div_D);
                    const int c = e - r * D;
                    if (!s_ist[c]) s_o[e - c + s_dst[c]] = s_x[mx + e - c + s_src[c]];
                }
            }
            for (int ii = tid; ii < cn; ii += BLOCK) {
                const int i = c0 + ii;
                const int r = (int)fastdiv((uint32_t)i, a.div_dt);
                const int j = i - r * dt;
                const int col = s_tidx[j];
                const float xin = s_x[mx + r * D + s_src[col]];
                float y, l;
                int* bin = a.bins ? a.bins + (row0 * dt + i) : nullptr;
                my_status |= a.sp.linear ? rqs_eval<KT, INVERSE, true>(xin, s_p + mp + ii * P, a.sp, y, l, bin)
                                         : rqs_eval<KT, INVERSE, false>(xin, s_p + mp + ii * P, a.sp, y, l, bin);
                s_o[r * D + s_dst[col]] = y;
                if (chunked)
                    acc += l;
                else
                    s_lad[i] = l;
            }
            __syncthreads();  // s_p is overwritten by the next chunk / tile
        }

        tile_store<BLOCK>(a.out + row0 * D, rows * D, s_out, tid);
        const int wave = tid >> 6, lane = tid & 63;
        if (chunked) {
            // one sample: fixed-order reduction of the per-lane partial sums
            acc = wave_sum(acc);
            if (lane == 0) s_lad[wave] = acc;
            __syncthreads();
            if (tid == 0) {
                float v = 0.0f;
                for (int w = 0; w < BLOCK / kWave; ++w) v += s_lad[w];
                a.lad[row0] = a.accumulate ? a.lad[row0] + v : v;
            }
            __syncthreads();
        } else {
            // per-sample logabsdet: wave w reduces rows w, w+4, ...
            for (int r = wave; r < rows; r += BLOCK / kWave) {
                float v = 0.0f;
                for (int m = lane; m < dt; m += kWave) v += s_lad[r * dt + m];
                v = wave_sum(v);
                if (lane == 0) a.lad[row0 + r] = a.accumulate ? a.lad[row0 + r] + v : v;
            }
        }
        // next iteration's loads only touch s_p / s_x, whose readers all passed the barrier above;
        // s_out / s_lad are rewritten only after the next iteration's first barrier.
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}


// ------------------------------------------------------------------------------------------
// K1, software-pipelined form for 16-byte aligned layouts (d_t*P % 4 == 0, D % 4 == 0, aligned
// base pointers; the BASELINE shape 32*23 = 736 and D = 64 qualifies).
//
// The generic kernel above keeps loads in flight only while a workgroup sits in its load phase.
// Here every lane carries the NEXT tile in registers (NV float4 of conditioner output + one
// float4 of inputs): the loads are issued right after the current tile has been written to LDS
// and stay in flight across the whole evaluation / store phase.  The two workgroup barriers are
// raw `s_waitcnt lgkmcnt(0); s_barrier` (LDS-only): __syncthreads() would also drain vmcnt and
// serialise the prefetch.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef float vec4 __attribute__((ext_vector_type(4)));  // native 16-byte vector
// conditioner output is read exactly once: optionally mark the loads non-temporal
#ifdef NFA_NT_LOADS
#define NFA_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define NFA_STREAM_LOAD(p) (*(p))
#endif
#ifndef NFA_PIPE_WAVES
#define NFA_PIPE_WAVES 4
#endif
template <int KT, bool INVERSE, bool LINEAR, bool BINS = false>   // BINS: the instance that also stores a.bins
__global__ void __launch_bounds__(kBlock, NFA_PIPE_WAVES) rqs_coupling_pipelined(const CouplingArgs a) {
    // float4 per lane per tile: enough for 256 splines of 3K+1 logits (lanes past the tile's end
    // re-read its last vector)
    constexpr int NV = (3 * KT + 1 + 3) / 4;
    static_assert(NV <= 8, "add prefetch registers");
    // Preconditions (checked by the host): every tile is full (a.batch % a.R == 0 here; the host
    // sends leftover rows to the generic kernel), R*dt <= 256 (one spline per lane),
    // R*D <= 512 (<= 2 pass-through slots per lane, <= 128 float4 of inputs per tile),
    // dt*P % 4 == 0, D % 4 == 0, 16-byte aligned params / inputs / outputs.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_p = lds;
    float* s_x = lds + a.off_x;
    float* s_out = lds + a.off_out;
    float* s_lad = lds + a.off_lad;
    int* s_tidx = reinterpret_cast<int*>(lds + a.off_idx);
    int* s_src = s_tidx + a.dt;
    int* s_dst = s_src + a.D;
    unsigned char* s_ist = reinterpret_cast<unsigned char*>(s_dst + a.D);

    const int tid = threadIdx.x;
    const int D = a.D, dt = a.dt, P = a.sp.P, R = a.R;
    int my_status = 0;

    for (int c = tid; c < D; c += kBlock) {
        int src = c, dst = c;
        if (a.perm) {
            const int64_t p = a.perm[c];
            if (p < 0 || p >= D) my_status |= NFA_STATUS_BAD_INDEX;
            src = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
        }
        if (a.scatter) {
            const int64_t p = a.scatter[c];
            if (p < 0 || p >= D) my_status |= NFA_STATUS_BAD_INDEX;
            dst = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
        }
        s_src[c] = src;
        s_dst[c] = dst;
        s_ist[c] = 0;
    }
    __syncthreads();
    for (int j = tid; j < dt; j += kBlock) {
        const int64_t t = a.tidx[j];
        if (t < 0 || t >= D) my_status |= NFA_STATUS_BAD_INDEX;
        const int col = (int)(t < 0 ? 0 : (t >= D ? D - 1 : t));
        s_tidx[j] = col;
        s_ist[col] = 1;
    }
    __syncthreads();

    // ---- per-lane tile-invariant state ------------------------------------------------------
    const int nitems = R * dt;            // <= 256
    const int nvp = (nitems * P) >> 2;    // float4 of conditioner output per tile
    const int nvx = (R * D) >> 2;         // float4 of inputs / outputs per tile (<= 128)
    const bool has_item = tid < nitems;
    int it_x = 0, it_y = 0;               // LDS word offsets of this lane's input / output
    {
        const int i = has_item ? tid : 0;
        const int r = (int)fastdiv((uint32_t)i, a.div_dt);
        const int col = s_tidx[i - r * dt];
        it_x = r * D + s_src[col];
        it_y = r * D + s_dst[col];
    }
    const float* it_p = s_p + (has_item ? tid : 0) * P;
    // pass-through copies: elements tid and tid + 256 of the [R, D] tile
    int cp_src0 = -1, cp_dst0 = 0, cp_src1 = -1, cp_dst1 = 0;
    {
        const int e0 = tid, e1 = tid + kBlock;
        if (e0 < R * D) {
            const int r = (int)fastdiv((uint32_t)e0, a.div_D), c = e0 - r * D;
            if (!s_ist[c]) { cp_src0 = e0 - c + s_src[c]; cp_dst0 = e0 - c + s_dst[c]; }
        }
        if (e1 < R * D) {
            const int r = (int)fastdiv((uint32_t)e1, a.div_D), c = e1 - r * D;
            if (!s_ist[c]) { cp_src1 = e1 - c + s_src[c]; cp_dst1 = e1 - c + s_dst[c]; }
        }
    }
    // logabsdet: when the d_t splines of a sample sit in d_t consecutive lanes of one wave
    // (d_t a power of two <= 64) the per-sample sum is a butterfly over those lanes
    const bool lad_shuffle = (dt & (dt - 1)) == 0 && dt <= kWave;
    const int64_t tile_stride_p = (int64_t)nitems * P;  // floats
    const int tile_stride_x = R * D;

    vec4 pr0, pr1, pr2, pr3, pr4, pr5, pr6, pr7;
    vec4 xr;
#define NFA_LD(k)                                                              \
    if (NV > k) {                                                              \
        const int v_ = k * kBlock + tid;                                       \
        pr##k = NFA_STREAM_LOAD(&gp_[v_ < nvp ? v_ : nvp - 1]);                \
    }
#define NFA_ST(k)                                                              \
    if (NV > k) {                                                              \
        const int v_ = k * kBlock + tid;                                       \
        if (v_ < nvp) reinterpret_cast<vec4*>(s_p)[v_] = pr##k;              \
    }
    // index-clamped, unconditional loads: the tile stays in VGPRs; past the last tile the lanes
    // re-read tile 0 (L2-resident by then), the data is never used
#define NFA_ISSUE_TILE(TILE)                                                                    \
    {                                                                                           \
        const int64_t t_ = (TILE) < num_tiles ? (TILE) : 0;                                     \
        const vec4* gp_ = reinterpret_cast<const vec4*>(a.params + t_ * tile_stride_p);     \
        const vec4* gx_ = reinterpret_cast<const vec4*>(a.x + t_ * tile_stride_x);          \
        NFA_LD(0) NFA_LD(1) NFA_LD(2) NFA_LD(3) NFA_LD(4) NFA_LD(5) NFA_LD(6) NFA_LD(7)         \
        xr = gx_[tid < nvx ? tid : nvx - 1];                                                    \
    }

    const int64_t num_tiles = a.batch / R;
    int64_t tile = blockIdx.x;
    NFA_ISSUE_TILE(tile)
    for (; tile < num_tiles; tile += gridDim.x) {
        NFA_ST(0) NFA_ST(1) NFA_ST(2) NFA_ST(3) NFA_ST(4) NFA_ST(5) NFA_ST(6) NFA_ST(7)
        if (tid < nvx) reinterpret_cast<vec4*>(s_x)[tid] = xr;
        const int64_t next = tile + gridDim.x;
        NFA_ISSUE_TILE(next)  // in flight until the next iteration's LDS writes
        lds_barrier();

        if (cp_src0 >= 0) s_out[cp_dst0] = s_x[cp_src0];
        if (cp_src1 >= 0) s_out[cp_dst1] = s_x[cp_src1];
        float l = 0.0f;
        if (has_item) {
            float y;
#ifdef NFA_K1_ABL_NO_EVAL   // (measurement: the kernel's memory structure without the spline arithmetic)
            y = s_x[it_x] + it_p[3];
            l = it_p[5];
#else
            int* bin = BINS ? a.bins + (tile * nitems + tid) : nullptr;
            my_status |= rqs_eval<KT, INVERSE, LINEAR>(s_x[it_x], const_cast<float*>(it_p), a.sp, y, l, bin);
#endif
            s_out[it_y] = y;
        }
        const int64_t row0 = tile * R;
        if (lad_shuffle) {
            for (int off = dt >> 1; off > 0; off >>= 1) l += __shfl_xor(l, off, kWave);
            if (has_item && (tid & (dt - 1)) == 0) {
                float* dst = a.lad + row0 + (tid >> __builtin_ctz(dt));
                *dst = a.accumulate ? *dst + l : l;
            }
        } else if (has_item) {
            s_lad[tid] = l;
        }
        lds_barrier();

        if (tid < nvx)
            reinterpret_cast<vec4*>(a.out + row0 * D)[tid] = reinterpret_cast<const vec4*>(s_out)[tid];
        if (!lad_shuffle) {
            const int wave = tid >> 6, lane = tid & 63;
            for (int r = wave; r < R; r += kBlock / kWave) {
                float v = 0.0f;
                for (int m = lane; m < dt; m += kWave) v += s_lad[r * dt + m];
                v = wave_sum(v);
                if (lane == 0) a.lad[row0 + r] = a.accumulate ? a.lad[row0 + r] + v : v;
            }
        }
    }
#undef NFA_ISSUE_TILE
#undef NFA_LD
#undef NFA_ST
    if (my_status && a.status) atomicOr(a.status, my_status);
}


// ------------------------------------------------------------------------------------------
// K5: elementwise functional over n independent elements
struct ElementwiseArgs {
    const float* x;
    const float* uw;
    const float* uh;
    const float* ud;
    int64_t sw, sh, sd;  // row strides (elements)
    float* y;
    float* lad;
    int32_t* bins;    // may be null: [n] bin chosen for every element (rqs_eval's `bin`)
    int32_t* status;
    int64_t n;
    int packed;  // 1: the three logit arrays are one [n, P] buffer starting at uw
    int nd;      // derivative logits per element
    int slot;    // LDS words per element when !packed (odd)
    int T;       // elements per tile (<= kBlock; smaller when K is large)
    RqsDev sp;
};

template <int KT, bool INVERSE>
__global__ void __launch_bounds__(kBlock) rqs_elementwise_kernel(const ElementwiseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int P = a.sp.P, K = a.sp.K;
    int my_status = 0;
    const int64_t num_tiles = (a.n + a.T - 1) / a.T;
    for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int64_t i0 = tile * a.T;
        const int cnt = (int)((a.n - i0) < a.T ? (a.n - i0) : a.T);
        float* mine;
        if (a.packed) {
            const int mp = tile_load(a.uw + i0 * P, cnt * P, lds, tid);
            __syncthreads();
            mine = lds + mp + tid * P;
        } else {
            mine = lds + tid * a.slot;
            if (tid < cnt) {
                const int64_t i = i0 + tid;
                for (int q = 0; q < K; ++q) mine[q] = a.uw[i * a.sw + q];
                for (int q = 0; q < K; ++q) mine[K + q] = a.uh[i * a.sh + q];
                for (int q = 0; q < a.nd; ++q) mine[2 * K + q] = a.ud[i * a.sd + q];
            }
        }
        if (tid < cnt) {
            float y, l;
            int* bin = a.bins ? a.bins + (i0 + tid) : nullptr;
            my_status |= a.sp.linear ? rqs_eval<KT, INVERSE, true>(a.x[i0 + tid], mine, a.sp, y, l, bin)
                                     : rqs_eval<KT, INVERSE, false>(a.x[i0 + tid], mine, a.sp, y, l, bin);
            a.y[i0 + tid] = y;
            a.lad[i0 + tid] = l;
        }
        if (a.packed) __syncthreads();  // before the next tile overwrites the LDS image
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}


// ------------------------------------------------------------------------------------------
// K1, wave-tile form (round 4): linear tails, K = 8 / 10, d_t a power of two <= 64, D = 2 d_t (alternating masks on
// power-of-two feature counts: the BASELINE shape d_t = 32, D = 64), 16-byte aligned arrays.
//
// The pipelined kernel above carries the next tile in 28 VGPRs per lane (101 in all: four waves per SIMD) and its
// four waves meet at three workgroup barriers per tile; its ~470 VALU instructions per tile then take as long to
// issue as the tile's bytes take to arrive, and what one phase waits for the other cannot use.  Here a WAVE owns its
// tiles: 64 splines = 64 / d_t whole samples, whose conditioner output is one contiguous 256 P-byte chunk.
//   * The chunk comes in by LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction, no staging registers) into the
//     wave's private LDS image; a lane copies its P consecutive words out (stride P words, P odd: conflict free)
//     and the DMA of the wave's NEXT tile is requested right behind those reads, into the same image -- in
//     flight for the whole evaluation.  One buffer per wave: 6 656 bytes (K = 8), so 5-6 waves per SIMD fit.
//   * No workgroup barrier anywhere: the only waits are the wave's own vmcnt(0) at the top of a tile and lgkmcnt
//     for its own LDS reads; the other waves of the SIMD fill every gap.
//   * The tile's 128 inputs are two coalesced dwords per lane; the spline input of a lane is picked from them with
//     ds_bpermute, pass-through columns and results meet in a 512-byte output image that leaves as one 16-byte
//     store per lane of the first half wave.  Same rqs_eval, same bits as the pipelined kernel.
// cache policy of the LDS-DMA requests: nt (aux = 2) -- the conditioner output is read exactly once; measured on the
// MI355X at 262 144 rows (profiles/r4/k1_wavetile.txt): memory structure alone 178 -> 157 us per layer, the kernel
// 204 -> 185-197 us
#ifndef NFA_WT_AUX
#define NFA_WT_AUX 2
#endif
#ifdef NFA_WT_NT_IO   // (experiment: the input / output rows non-temporal as well)
#define NFA_WT_LOAD(p) __builtin_nontemporal_load(p)
#define NFA_WT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define NFA_WT_LOAD(p) (*(p))
#define NFA_WT_STORE(v, p) (*(p) = (v))
#endif

template <int KT, bool INVERSE, bool BINS = false>   // BINS: the instance that also stores a.bins
__global__ void __launch_bounds__(kBlock) rqs_coupling_wavetile(const CouplingArgs a) {
    constexpr int P = 3 * KT - 1;
    constexpr int kTileBytes = 64 * P * 4;
    constexpr int NI = (kTileBytes + 1023) / 1024;   // DMA instructions per lane and tile
    constexpr int kWaveLds = NI * 1024 + 512;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* my = reinterpret_cast<char*>(lds) + wave * kWaveLds;
    float* s_par = reinterpret_cast<float*>(my);
    float* s_o = reinterpret_cast<float*>(my + NI * 1024);
    const int D = a.D, dt = a.dt;
    const int log_dt = __builtin_ctz(dt), log_D = log_dt + 1;
    const int RW = 64 >> log_dt;                     // samples per wave tile
    int my_status = 0;

    // ---- tables (once per wave; the parameter image doubles as scratch) -----------------------------
    int* t_src = reinterpret_cast<int*>(s_par);
    int* t_dst = t_src + D;
    int* t_ist = t_dst + D;
    int* t_inv = t_ist + D;
    int* t_col = t_inv + D;
    for (int c = lane; c < D; c += kWave) {
        int src = c, dst = c;
        if (a.perm) {
            const int64_t q = a.perm[c];
            if (q < 0 || q >= D) my_status |= NFA_STATUS_BAD_INDEX;
            src = (int)(q < 0 ? 0 : (q >= D ? D - 1 : q));
        }
        if (a.scatter) {
            const int64_t q = a.scatter[c];
            if (q < 0 || q >= D) my_status |= NFA_STATUS_BAD_INDEX;
            dst = (int)(q < 0 ? 0 : (q >= D ? D - 1 : q));
        }
        t_src[c] = src;
        t_dst[c] = dst;
        t_ist[c] = 0;
        t_inv[c] = 0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int j = lane; j < dt; j += kWave) {
        const int64_t t = a.tidx[j];
        if (t < 0 || t >= D) my_status |= NFA_STATUS_BAD_INDEX;
        const int col = (int)(t < 0 ? 0 : (t >= D ? D - 1 : t));
        t_col[j] = col;
        t_ist[col] = 1;
    }
    for (int c = lane; c < D; c += kWave) t_inv[t_src[c]] = c;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // this lane's spline: sample r_i of the tile, transformed feature j
    const int r_i = lane >> log_dt;
    const int col = t_col[lane & (dt - 1)];
    const int e_x = (r_i << log_D) + t_src[col];     // element of the [RW, D] input tile it reads
    const int x_sel = (e_x & 63) << 2;               // ds_bpermute address of the lane that loaded it
    const bool x_hi = e_x >= 64;
    const int y_off = (r_i << log_D) + t_dst[col];
    // the two input elements this lane loads (e = lane, lane + 64): where they go when their column passes through
    int pt_off0, pt_off1;
    bool pt_ok0, pt_ok1;
    {
        const int cs0 = lane & (D - 1), c0 = t_inv[cs0];
        pt_ok0 = !t_ist[c0];
        pt_off0 = ((lane >> log_D) << log_D) + t_dst[c0];
        const int e1 = lane + 64, cs1 = e1 & (D - 1), c1 = t_inv[cs1];
        pt_ok1 = !t_ist[c1];
        pt_off1 = ((e1 >> log_D) << log_D) + t_dst[c1];
    }
    const bool lad_lane = (lane & (dt - 1)) == 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tables are read: the image may be overwritten

    const int64_t num_tiles = a.batch >> (6 - log_dt);   // batch / RW (the host sends whole tiles)
    const int64_t total_waves = (int64_t)gridDim.x * (kBlock / kWave);
    int64_t tile = (int64_t)blockIdx.x * (kBlock / kWave) + wave;
    const unsigned lane_off = (unsigned)lane * 16u;

#ifdef NFA_K1_ABL_NO_MEM
#define NFA_WT_REQUEST(TILE) {}
#else
#define NFA_WT_REQUEST(TILE)                                                                                   \
    {                                                                                                          \
        const char* g_ = reinterpret_cast<const char*>(a.params) + (TILE) * (int64_t)kTileBytes;               \
        _Pragma("unroll") for (int k_ = 0; k_ < NI; ++k_) {                                                    \
            const unsigned o_ = (unsigned)k_ * 1024u + lane_off;                                               \
            /* (the last instruction of a 256 P-byte chunk is partial: the lanes past its end sit out) */      \
            if (k_ < NI - 1 || !(kTileBytes & 1023) || o_ < (unsigned)kTileBytes)                              \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g_ + o_),     \
                                                 (__attribute__((address_space(3))) void*)(my + k_ * 1024), 16, 0, NFA_WT_AUX); \
        }                                                                                                      \
    }
#endif
    float xa0 = 0.0f, xa1 = 0.0f, la = 0.0f;
    if (tile < num_tiles) {
        NFA_WT_REQUEST(tile)
        const float* gx = a.x + tile * 128;
        xa0 = NFA_WT_LOAD(gx + lane);
        xa1 = NFA_WT_LOAD(gx + lane + 64);
        if (a.accumulate && lad_lane) la = a.lad[tile * RW + r_i];
    }
    for (; tile < num_tiles; tile += total_waves) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this tile's image and inputs have landed
        float p[P];
#pragma unroll
        for (int q = 0; q < P; ++q) p[q] = s_par[lane * P + q];
        const float xv0 = xa0, xv1 = xa1, lacc = la;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the image is in registers: it may be refilled
        {
            int64_t next = tile + total_waves;
            next = next < num_tiles ? next : tile;                // (past the end: a harmless re-read)
            NFA_WT_REQUEST(next)
#ifndef NFA_K1_ABL_NO_MEM
            const float* gx = a.x + next * 128;
            xa0 = NFA_WT_LOAD(gx + lane);
            xa1 = NFA_WT_LOAD(gx + lane + 64);
            if (a.accumulate && lad_lane) la = a.lad[next * RW + r_i];
#else
            xa0 += 0.001f; xa1 -= 0.001f;
#endif
        }
        const float xs0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(x_sel, __builtin_bit_cast(int, xv0)));
        const float xs1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(x_sel, __builtin_bit_cast(int, xv1)));
        const float xs = x_hi ? xs1 : xs0;
        float y, l;
#ifdef NFA_K1_ABL_NO_EVAL   // (measurement: the kernel's memory structure without the spline arithmetic)
        y = xs + p[3] + p[P - 1];
        l = p[5] + p[11];
#else
        // (lane = sample r_i, transformed feature lane & (dt - 1) of the tile: element tile * 64 + lane of [batch, dt])
        my_status |= rqs_eval<KT, INVERSE, true, true>(xs, p, a.sp, y, l, BINS ? a.bins + (tile * 64 + lane) : nullptr);
#endif
        if (pt_ok0) s_o[pt_off0] = xv0;
        if (pt_ok1) s_o[pt_off1] = xv1;
        s_o[y_off] = y;
        for (int off = dt >> 1; off > 0; off >>= 1) l += __shfl_xor(l, off, kWave);
#ifdef NFA_K1_ABL_NO_MEM   // (measurement: the arithmetic + LDS traffic without any global memory access in the loop)
        if (lad_lane && l == 123.456f) a.lad[tile * RW + r_i] = lacc + l;
        if (lane < 32 && l == 123.456f) reinterpret_cast<vec4*>(a.out + tile * 128)[lane] = reinterpret_cast<const vec4*>(s_o)[lane];
#else
        if (lad_lane) a.lad[tile * RW + r_i] = a.accumulate ? lacc + l : l;
        if (lane < 32) NFA_WT_STORE(reinterpret_cast<const vec4*>(s_o)[lane], reinterpret_cast<vec4*>(a.out + tile * 128) + lane);
#endif
    }
#undef NFA_WT_REQUEST
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (my_status && a.status) atomicOr(a.status, my_status);
}

// ------------------------------------------------------------------------------------------
constexpr int kMaxDynLds = 64 * 1024;

// ---- optional measurement aid (bench.py): per-launch begin/end timestamps of the K1 kernels.
// hipExtLaunchKernelGGL attaches a start and a stop event to the dispatch itself, so
// hipEventElapsedTime(start, stop) is the kernel's own duration on its stream (what rocprofv3's
// kernel trace reports), free of launch gaps.  Off by default; the only global state in the library.
struct ProfileState {
    std::mutex mu;
    bool enabled = false;
    size_t capacity = 0;
    std::vector<hipEvent_t> start, stop;
};
static ProfileState g_profile;

// Hands out a start/stop event pair for the next profiled launch (null when profiling is off or
// the budget is used up).  Shared with the other translation units through common.hpp.
void profile_next_launch(hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    std::lock_guard<std::mutex> lock(g_profile.mu);
    if (g_profile.enabled && g_profile.start.size() < g_profile.capacity) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            g_profile.start.push_back(e0);
            g_profile.stop.push_back(e1);
            *start = e0;
            *stop = e1;
        }
    }
}

static thread_local char g_last_layer_kernel[192] = "";

void note_layer_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_layer_kernel, sizeof(g_last_layer_kernel), fmt, ap);
    va_end(ap);
}

template <typename Kernel>
static void launch_k1(Kernel kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, const CouplingArgs& a) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    profile_next_launch(&e0, &e1);
    if (e0)
        hipExtLaunchKernelGGL(kernel, grid, block, lds, st, e0, e1, 0, a);
    else
        hipLaunchKernelGGL(kernel, grid, block, lds, st, a);
}

template <int KT, int BLOCK>
static int launch_coupling(const CouplingArgs& a, int inverse, dim3 grid, size_t lds, hipStream_t st) {
    note_layer_kernel("rqs_coupling_kernel<K=%d, inverse=%d, block=%d>", KT, inverse ? 1 : 0, BLOCK);
    if (inverse)
        launch_k1(rqs_coupling_kernel<KT, true, BLOCK>, grid, dim3(BLOCK), lds, st, a);
    else
        launch_k1(rqs_coupling_kernel<KT, false, BLOCK>, grid, dim3(BLOCK), lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

template <int KT>
static int launch_wavetile(const CouplingArgs& a, int inverse, hipStream_t st) {
    constexpr int P = 3 * KT - 1;
    constexpr int kWaveLds = ((64 * P * 4 + 1023) / 1024) * 1024 + 512;
    const size_t lds = (size_t)(kBlock / kWave) * kWaveLds;
    void (*kern)(const CouplingArgs) =
        a.bins ? (inverse ? rqs_coupling_wavetile<KT, true, true> : rqs_coupling_wavetile<KT, false, true>)
               : (inverse ? rqs_coupling_wavetile<KT, true> : rqs_coupling_wavetile<KT, false>);
    // persistent: exactly the workgroups that are resident together (registers and LDS decide)
    // (per device: occupancy is a property of the device the launch goes to)
    static int per_cu_cache[64][4] = {};
    int dev = 0;
    NFA_HIP_CHECK(hipGetDevice(&dev));
    int& per_cu = per_cu_cache[dev & 63][(inverse ? 1 : 0) + (a.bins ? 2 : 0)];
    if (per_cu == 0) {
        int n = 0;
        NFA_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, kBlock, lds));
        per_cu = n > 0 ? n : 1;
        const char* e = getenv("NFA_K1_WT_PER_CU");   // (experiments: fewer resident workgroups)
        if (e && atoi(e) > 0 && atoi(e) < per_cu) per_cu = atoi(e);
    }
    const int log_dt = __builtin_ctz((unsigned)a.dt);
    const int64_t tiles = a.batch >> (6 - log_dt);
    int64_t g = (int64_t)device_cu_count() * per_cu;
    const int64_t need = (tiles + kBlock / kWave - 1) / (kBlock / kWave);
    if (g > need) g = need;
    note_layer_kernel("rqs_coupling_wavetile<K=%d, inverse=%d> (%d workgroups per CU)", KT, inverse ? 1 : 0, per_cu);
    launch_k1(kern, dim3((unsigned)g), dim3(kBlock), lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

template <int KT>
static int launch_pipelined(const CouplingArgs& a, int inverse, dim3 grid, size_t lds, hipStream_t st) {
    note_layer_kernel("rqs_coupling_pipelined<K=%d, inverse=%d, linear=%d>", KT, inverse ? 1 : 0, a.sp.linear ? 1 : 0);
    if (a.bins) {   // (the instances that store the chosen bins: same arithmetic, one more store)
        if (a.sp.linear) {
            if (inverse) launch_k1(rqs_coupling_pipelined<KT, true, true, true>, grid, dim3(kBlock), lds, st, a);
            else launch_k1(rqs_coupling_pipelined<KT, false, true, true>, grid, dim3(kBlock), lds, st, a);
        } else {
            if (inverse) launch_k1(rqs_coupling_pipelined<KT, true, false, true>, grid, dim3(kBlock), lds, st, a);
            else launch_k1(rqs_coupling_pipelined<KT, false, false, true>, grid, dim3(kBlock), lds, st, a);
        }
    } else if (a.sp.linear) {
        if (inverse)
            launch_k1(rqs_coupling_pipelined<KT, true, true>, grid, dim3(kBlock), lds, st, a);
        else
            launch_k1(rqs_coupling_pipelined<KT, false, true>, grid, dim3(kBlock), lds, st, a);
    } else {
        if (inverse)
            launch_k1(rqs_coupling_pipelined<KT, true, false>, grid, dim3(kBlock), lds, st, a);
        else
            launch_k1(rqs_coupling_pipelined<KT, false, false>, grid, dim3(kBlock), lds, st, a);
    }
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

template <int KT>
static int launch_elementwise(const ElementwiseArgs& a, int inverse, dim3 grid, size_t lds, hipStream_t st) {
    if (inverse)
        hipLaunchKernelGGL((rqs_elementwise_kernel<KT, true>), grid, dim3(kBlock), lds, st, a);
    else
        hipLaunchKernelGGL((rqs_elementwise_kernel<KT, false>), grid, dim3(kBlock), lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_rqs_coupling_f32(const float* inputs, const float* params,
                                    const int64_t* transform_idx, const int64_t* in_perm,
                                    const int64_t* out_scatter, float* outputs, float* logabsdet,
                                    int32_t* bin_idx, int32_t* status, int64_t batch,
                                    int32_t features, int32_t num_transform, const nfa_rqs_spec* spec,
                                    int32_t flags, void* stream) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET)) return NFA_ERR_INVALID_ARGUMENT;
    const int inverse = flags & NFA_FLAG_INVERSE;
    if (batch < 0 || features < 1 || num_transform < 0 || num_transform > features)
        return NFA_ERR_INVALID_ARGUMENT;
    CouplingArgs a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (batch == 0) return NFA_OK;
    if (!inputs || !outputs || !logabsdet || (num_transform > 0 && (!params || !transform_idx)))
        return NFA_ERR_INVALID_ARGUMENT;
    if (features > 65535) return NFA_ERR_UNSUPPORTED;

    const int P = a.sp.P, D = features, dt = num_transform;
    const int BT = kBlock;
    // samples per tile: aim at one item per lane, whole samples, LDS within budget
    int R = dt > 0 ? BT / dt : BT / (D < BT ? D : BT);
    if (R < 1) R = 1;
    if ((int64_t)R > batch) R = (int)batch;
    int C = 0;  // splines per chunk (0 = whole tile)
    auto lds_floats = [&](int r, int* ox, int* oo, int* ol, int* oi) {
        const int chunk_items = C > 0 ? C : r * dt;
        int o = round_up4(chunk_items * P) + 4;
        *ox = o;
        o += round_up4(r * D) + 4;
        *oo = o;
        o += round_up4(r * D) + 4;
        *ol = o;
        o += round_up4(C > 0 ? kBlock : (r * dt > kBlock / kWave ? r * dt : kBlock / kWave));
        *oi = o;
        o += dt + 2 * D + (D + 3) / 4;
        return o;
    };
    int ox, oo, ol, oi;
    while (R > 1 && (size_t)lds_floats(R, &ox, &oo, &ol, &oi) * 4 > (size_t)kMaxDynLds) R >>= 1;
    if (R == 1 && (size_t)lds_floats(1, &ox, &oo, &ol, &oi) * 4 > (size_t)kMaxDynLds) {
        // one sample does not fit: stream its parameters through LDS in chunks of C splines
        const size_t fixed = (size_t)(2 * (round_up4(D) + 4) + kBlock + dt + 2 * D + (D + 3) / 4 + 8) * 4;
        if (fixed + (size_t)kBlock * P * 4 > (size_t)kMaxDynLds) return NFA_ERR_UNSUPPORTED;
        C = (int)(((size_t)kMaxDynLds - fixed) / ((size_t)P * 4));
        C = (C / kBlock) * kBlock;
        if (C >= dt) C = 0;
    }
    const size_t lds = (size_t)lds_floats(R, &ox, &oo, &ol, &oi) * 4;
    if (lds > (size_t)kMaxDynLds || (int64_t)R * dt >= 65536 || (int64_t)R * D >= 65536)
        return NFA_ERR_UNSUPPORTED;
    a.C = C > 0 ? C : R * dt;

    a.x = inputs;
    a.params = params;
    a.tidx = transform_idx;
    a.perm = in_perm;
    a.scatter = out_scatter;
    a.out = outputs;
    a.lad = logabsdet;
    a.bins = bin_idx;
    a.status = status;
    a.batch = batch;
    a.D = D;
    a.dt = dt;
    a.R = R;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    a.div_dt = make_fastdiv((uint32_t)(dt > 0 ? dt : 1));
    a.div_D = make_fastdiv((uint32_t)D);
    a.off_x = ox;
    a.off_out = oo;
    a.off_lad = ol;
    a.off_idx = oi;

    const int64_t tiles = (batch + R - 1) / R;
    const int cus = device_cu_count();
    int per_cu = (int)((size_t)(160 * 1024) / (lds + 256));
    const int cap = 8;
    if (per_cu > cap) per_cu = cap;
    if (per_cu < 1) per_cu = 1;
    int64_t g = (int64_t)cus * per_cu;
    if (g > tiles) g = tiles;
    const dim3 grid((unsigned)g);
    hipStream_t st = (hipStream_t)stream;
    // aligned layouts take the software-pipelined kernel
    static const int use_pipe = [] {
        const char* e = getenv("NFA_K1_PIPELINE");
        return e ? atoi(e) : 1;
    }();
    const int nv = (int)(((int64_t)R * dt * P / 4 + kBlock - 1) / kBlock);
    const bool aligned = dt > 0 && (dt * P) % 4 == 0 && D % 4 == 0 && R * dt <= kBlock &&
                         R * D <= 2 * kBlock && (int64_t)R <= batch &&
                         (reinterpret_cast<uintptr_t>(params) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(inputs) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(outputs) & 15) == 0;
    const bool pipe_k = a.sp.K == 4 || a.sp.K == 8 || a.sp.K == 10;
    if (use_pipe && aligned && pipe_k && nv <= (3 * a.sp.K + 4) / 4) {
        const int64_t full_rows = (batch / R) * R;
        CouplingArgs f = a;
        f.batch = full_rows;
        // wave tiles (LDS-DMA, no workgroup barriers) where the layout allows: linear tails, 8 / 10 bins, d_t a
        // power of two with D = 2 d_t; NFA_K1_WAVETILE=0 keeps the register-pipelined kernel (A/B runs)
        // (read per launch on purpose: tests/test_gpu_steep.py and bench.py switch it between launches of one process)
        const char* wt_env = getenv("NFA_K1_WAVETILE");
        const bool wavetile = (!wt_env || atoi(wt_env) != 0) && a.sp.linear && (a.sp.K == 8 || a.sp.K == 10) &&
                              dt >= 4 && dt <= 64 && (dt & (dt - 1)) == 0 && D == 2 * dt && a.sp.P == 3 * a.sp.K - 1 &&
                              (reinterpret_cast<uintptr_t>(logabsdet) & 3) == 0;
        const int64_t wt_rows = wavetile ? (batch >> (6 - __builtin_ctz((unsigned)dt))) << (6 - __builtin_ctz((unsigned)dt)) : 0;
        if (wavetile && wt_rows > 0) {
            // whole wave tiles (64 / d_t rows each); the < 64 / d_t rows behind them go to the generic kernel below
            f.batch = wt_rows;
            const int wrc = a.sp.K == 8 ? launch_wavetile<8>(f, inverse, st) : launch_wavetile<10>(f, inverse, st);
            if (wrc != NFA_OK) return wrc;
            if (wt_rows == batch) return NFA_OK;
            a.x = inputs + wt_rows * D;
            a.params = params + wt_rows * (int64_t)dt * P;
            a.out = outputs + wt_rows * D;
            a.lad = logabsdet + wt_rows;
            if (bin_idx) a.bins = bin_idx + wt_rows * dt;
            a.batch = batch - wt_rows;
            return a.sp.K == 10 ? launch_coupling<10, kBlock>(a, inverse, dim3(1), lds, st)
                                : launch_coupling<8, kBlock>(a, inverse, dim3(1), lds, st);
        } else {
        // one block fewer per CU than LDS alone would allow: the prefetch registers cost occupancy
        // (K = 4 needs only ~76 VGPRs: 6 waves per SIMD)
        const int pipe_blocks = a.sp.K <= 4 ? 6 : NFA_PIPE_WAVES;
        int64_t gp = (int64_t)cus * (per_cu > pipe_blocks ? pipe_blocks : per_cu);
        if (gp > full_rows / R) gp = full_rows / R;
        const dim3 pgrid((unsigned)gp);
        int prc;
        switch (a.sp.K) {
            case 4: prc = launch_pipelined<4>(f, inverse, pgrid, lds, st); break;
            case 10: prc = launch_pipelined<10>(f, inverse, pgrid, lds, st); break;
            default: prc = launch_pipelined<8>(f, inverse, pgrid, lds, st); break;
        }
        if (prc != NFA_OK) return prc;
        }
        if (full_rows == batch) return NFA_OK;
        // leftover rows (< R): generic kernel on the tail of every array
        a.x = inputs + full_rows * D;
        a.params = params + full_rows * (int64_t)dt * P;
        a.out = outputs + full_rows * D;
        a.lad = logabsdet + full_rows;
        if (bin_idx) a.bins = bin_idx + full_rows * dt;
        a.batch = batch - full_rows;
        switch (a.sp.K) {
            case 4: return launch_coupling<4, kBlock>(a, inverse, dim3(1), lds, st);
            case 10: return launch_coupling<10, kBlock>(a, inverse, dim3(1), lds, st);
            default: return launch_coupling<8, kBlock>(a, inverse, dim3(1), lds, st);
        }
    }
    switch (a.sp.K) {
        case 4: return launch_coupling<4, kBlock>(a, inverse, grid, lds, st);
        case 8: return launch_coupling<8, kBlock>(a, inverse, grid, lds, st);
        case 10: return launch_coupling<10, kBlock>(a, inverse, grid, lds, st);
        default: return launch_coupling<0, kBlock>(a, inverse, grid, lds, st);
    }
}

extern "C" int nfa_rqs_elementwise_f32(const float* inputs, const float* uw, int64_t stride_w,
                                       const float* uh, int64_t stride_h, const float* ud,
                                       int64_t stride_d, int32_t num_derivatives, float* outputs,
                                       float* logabsdet, int32_t* bin_idx, int32_t* status, int64_t n,
                                       const nfa_rqs_spec* spec, int32_t inverse, void* stream) {
    if (n < 0) return NFA_ERR_INVALID_ARGUMENT;
    ElementwiseArgs a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (n == 0) return NFA_OK;
    const int K = a.sp.K;
    // the reference pads / gathers whatever width it is given (rational_quadratic.py:33-36,
    // :127-128): more logits than K-1 (K+1) are legal, the extras are never read
    if (num_derivatives < a.sp.nd || num_derivatives > 3 * K + 8) return NFA_ERR_INVALID_ARGUMENT;
    a.sp.nd = num_derivatives;
    a.sp.P = 2 * K + num_derivatives;
    const int P = a.sp.P;
    a.nd = num_derivatives;
    if (!inputs || !outputs || !logabsdet || !uw || !uh || (a.nd > 0 && !ud))
        return NFA_ERR_INVALID_ARGUMENT;
    a.x = inputs;
    a.uw = uw;
    a.uh = uh;
    a.ud = ud;
    a.sw = stride_w;
    a.sh = stride_h;
    a.sd = stride_d;
    a.y = outputs;
    a.lad = logabsdet;
    a.bins = bin_idx;
    a.status = status;
    a.n = n;
    a.packed = (uh == uw + K) && (a.nd == 0 || ud == uw + 2 * K) && stride_w == P && stride_h == P &&
               (a.nd == 0 || stride_d == P);
    a.slot = P | 1;
    int T = kBlock;
    auto lds_bytes = [&](int t) {
        return a.packed ? (size_t)(round_up4(t * P) + 8) * 4 : (size_t)t * a.slot * 4;
    };
    while (T > 1 && lds_bytes(T) > (size_t)kMaxDynLds) T >>= 1;
    const size_t lds = lds_bytes(T);
    if (lds > (size_t)kMaxDynLds) return NFA_ERR_UNSUPPORTED;
    a.T = T;
    const int64_t tiles = (n + T - 1) / T;
    const int cus = device_cu_count();
    int per_cu = (int)((size_t)(160 * 1024) / (lds + 256));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int64_t g = (int64_t)cus * per_cu;
    if (g > tiles) g = tiles;
    const dim3 grid((unsigned)g);
    hipStream_t st = (hipStream_t)stream;
    switch (K) {
        case 8: return launch_elementwise<8>(a, inverse, grid, lds, st);
        default: return launch_elementwise<0>(a, inverse, grid, lds, st);
    }
}


extern "C" int nfa_last_layer_kernel(char* buffer, int32_t capacity) {
    if (capacity < 0 || (capacity > 0 && !buffer)) return NFA_ERR_INVALID_ARGUMENT;
    if (capacity > 0) snprintf(buffer, (size_t)capacity, "%s", g_last_layer_kernel);
    return (int)strlen(g_last_layer_kernel);
}

extern "C" int nfa_profile_enable(int32_t max_launches) {
    if (max_launches < 0) return NFA_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(g_profile.mu);
    g_profile.enabled = max_launches > 0;
    g_profile.capacity = (size_t)max_launches;
    return NFA_OK;
}

extern "C" int nfa_profile_collect(float* durations_ms, int32_t capacity, int32_t* count) {
    if (!count || capacity < 0 || (capacity > 0 && !durations_ms)) return NFA_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(g_profile.mu);
    int32_t n = 0;
    for (size_t i = 0; i < g_profile.start.size(); ++i) {
        NFA_HIP_CHECK(hipEventSynchronize(g_profile.stop[i]));
        float ms = 0.0f;
        NFA_HIP_CHECK(hipEventElapsedTime(&ms, g_profile.start[i], g_profile.stop[i]));
        if (n < capacity) durations_ms[n++] = ms;
        (void)hipEventDestroy(g_profile.start[i]);
        (void)hipEventDestroy(g_profile.stop[i]);
    }
    g_profile.start.clear();
    g_profile.stop.clear();
    *count = n;
    return NFA_OK;
}
