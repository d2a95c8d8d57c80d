// K7: conditioner output layer + spline coupling layer in ONE kernel (SURVEY.md section 8f, row f3:
// "consume the final Linear's tiles from LDS instead of a [B, d_t*P] round trip through HBM").
//
//   params = hidden @ W^T + b            Linear(H -> d_t*P), nn/nets/resnet.py:90, :99
//   outputs, logabsdet = RQ coupling     coupling.py:73-130, :549-582 (exactly K1's arithmetic)
//
// The [B, d_t*P] parameter tensor (193 MB per layer at the BASELINE shape, written by the GEMM
// and read back by K1) never exists: a wave owns 32 samples, keeps their 128 hidden activations
// in 64 VGPRs as the A operand of v_mfma_f32_32x32x2_f32, streams the weights (pre-packed so that
// every load is a coalesced 16 bytes per lane; 376 KB, L2-resident) as the B operand, and drops
// each 32x32 accumulator tile (+ bias) into its private LDS slice.  After three tiles (96 columns
// = 4 features x 24: the packing pads each feature's 23 logits to 24) the wave evaluates those
// 4 x 32 splines from LDS and accumulates their log-derivatives.  The f32 MFMA is an exact fp32 FMA
// chain, so the parameters equal a plain fp32 GEMM up to summation order.
//
// Restrictions of this fast path (the host falls back to GEMM + K1 otherwise): K = 8 bins,
// linear tails (P = 23), hidden width 128, d_t a multiple of 4, batch a multiple of 32 handled
// here (leftover rows go through the unfused path).

#include "fused_common.hpp"

#include <hip/hip_ext.h>
#include <stdlib.h>

#ifndef NFA_K7_STAGGER
#define NFA_K7_STAGGER 1  // x 8128 cycles
#endif

namespace nfa {

constexpr int kH = 128;        // hidden width (GEMM K dimension)

struct FusedArgs {
    const float* x;       // [B, D]
    const float* hidden;  // [B, 128]
    const float* wpacked; // [(dt*24/32) tiles][16][64][4]
    const float* bpad;    // [dt*24]
    const int64_t* tidx;
    const int64_t* perm;
    const int64_t* scatter;
    float* out;
    float* lad;
    int32_t* status;
    int64_t batch;  // multiple of 32
    int D, dt, accumulate;
    FastDiv div_D;
    RqsDev sp;
    unsigned long long* trace;  // debug: per-phase timestamps of a few waves (null normally)
};

template <bool INVERSE>
__global__ void __launch_bounds__(kBlock, 2) rqs_fused_linear_kernel(const FusedArgs a) {
    // dynamic LDS: per wave a [32][dt|1] tile of transformed outputs
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const int ystride = a.dt | 1;
    __shared__ int s_dinv[128];   // layer column stored at output position p
    __shared__ int s_slot[128];   // index of a transformed column in transform_idx
    __shared__ int s_src[128], s_dst[128], s_tsrc[64];
    __shared__ unsigned char s_ist[128];  // 1: column is transformed (written by the spline lanes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, dt = a.dt;
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
        s_slot[c] = 0;
    }
    __syncthreads();
    for (int c = tid; c < D; c += kBlock) s_dinv[s_dst[c]] = c;
    if (tid < dt) {
        const int64_t t = a.tidx[tid];
        if (t < 0 || t >= D) my_status |= NFA_STATUS_BAD_INDEX;
        const int col = (int)(t < 0 ? 0 : (t >= D ? D - 1 : t));
        s_tsrc[tid] = s_src[col];
        s_ist[col] = 1;
        s_slot[col] = tid;
    }
    __syncthreads();

    float* s_y = lds_dyn + wave * 32 * ystride;
    const int half = lane >> 5, r = lane & 31;
    const int groups = dt >> 2;  // 4 features = 3 MFMA tiles per group
    const int64_t num_tiles = a.batch >> 5;
    const int64_t wave_global = (int64_t)blockIdx.x * (kBlock / kWave) + wave;
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / kWave);

    // debug trace: lane 0 of wave 0 of blocks 0 and 256 stamps s_memtime at phase boundaries
    unsigned long long* tr = nullptr;
    int ti = 0;
    if (a.trace && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 256))
        tr = a.trace + (blockIdx.x ? 256 : 0);
    for (int64_t tile = wave_global; tile < num_tiles; tile += nwaves) {
        const int64_t row0 = tile << 5;
        NFA_STAMP()
        // ---- hidden^T as the MFMA B operand: lane (sample r, half) holds hidden[row0 + r][half*64 ..+63]
        vec4f hv[16];
        const vec4f* hp = reinterpret_cast<const vec4f*>(a.hidden + (row0 + r) * kH + half * 64);
#pragma unroll
        for (int j4 = 0; j4 < 16; ++j4) hv[j4] = hp[j4];

        float lad_acc = 0.0f;
        // weights as the A operand, streamed in half tiles (8 x 16 bytes per lane), one ahead
        const vec4f* wbase = reinterpret_cast<const vec4f*>(a.wpacked) + lane;
        const vec4f* bias_lane = reinterpret_cast<const vec4f*>(a.bpad) + half * 4;
        vec4f wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = wbase[j * 64];
        const int num_half_tiles = groups * 6;

        NFA_STAMP()
        for (int g = 0; g < groups; ++g) {
            // this group's spline inputs: requested now, consumed after the three MFMA tiles
            const float xin0 = a.x[(row0 + r) * D + s_tsrc[g * 4 + half * 2]];
            const float xin1 = a.x[(row0 + r) * D + s_tsrc[g * 4 + half * 2 + 1]];
            f32x16 acc[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const vec4f* bp = bias_lane + (size_t)(g * 3 + t) * 8;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const vec4f b = bp[q4];
                    acc[t][q4 * 4 + 0] = b.x;
                    acc[t][q4 * 4 + 1] = b.y;
                    acc[t][q4 * 4 + 2] = b.z;
                    acc[t][q4 * 4 + 3] = b.w;
                }
            }
#pragma unroll
            for (int hh = 0; hh < 6; ++hh) {
                const int ht = g * 6 + hh;
                const int htn = (ht + 1 < num_half_tiles) ? ht + 1 : 0;
                const vec4f* wn = wbase + (size_t)htn * 8 * 64;
                vec4f wnext[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wnext[j] = wn[j * 64];
#ifndef NFA_K7_NOMFMA
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const vec4f h4 = hv[(hh & 1) * 8 + j];
                    acc[hh >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j].x, h4.x, acc[hh >> 1], 0, 0, 0);
                    acc[hh >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j].y, h4.y, acc[hh >> 1], 0, 0, 0);
                    acc[hh >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j].z, h4.z, acc[hh >> 1], 0, 0, 0);
                    acc[hh >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j].w, h4.w, acc[hh >> 1], 0, 0, 0);
                }
#else
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[hh >> 1][j] += wv[j].x + hv[(hh & 1) * 8 + j].y;
#endif
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = wnext[j];
            }
            NFA_STAMP()
            // ---- the lane's two features of this group, straight from the accumulators
            {
                NFA_K7_FEATURE_A(p, acc[0], acc[1]);
                float y, l;
#ifdef NFA_K7_NOSPLINE
                y = xin0 + p[0]; l = p[8];
#else
                my_status |= rqs_eval<8, INVERSE, true, true>(xin0, p, a.sp, y, l);
#endif
                s_y[r * ystride + g * 4 + half * 2] = y;
                lad_acc += l;
            }
            {
                NFA_K7_FEATURE_B(p, acc[1], acc[2]);
                float y, l;
#ifdef NFA_K7_NOSPLINE
                y = xin1 + p[0]; l = p[8];
#else
                my_status |= rqs_eval<8, INVERSE, true, true>(xin1, p, a.sp, y, l);
#endif
                s_y[r * ystride + g * 4 + half * 2 + 1] = y;
                lad_acc += l;
            }
            NFA_STAMP()
        }
        // ---- assemble the 32 output rows: position p holds layer column c = dinv[p]; transformed
        //      columns come from the LDS y tile, the others are copied bit-exactly from the inputs
        //      (gathered through the fused permutation).  Rows are written contiguously.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (int e = lane; e < 32 * D; e += kWave) {
            const int rr = (int)fastdiv((uint32_t)e, a.div_D);
            const int pcol = e - rr * D;
            const int c = s_dinv[pcol];
            a.out[(row0 + rr) * D + pcol] = s_ist[c] ? s_y[rr * ystride + s_slot[c]] : a.x[(row0 + rr) * D + s_src[c]];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lad_acc += __shfl_xor(lad_acc, 32, kWave);
        if (half == 0) {
            float* dst = a.lad + row0 + r;
            *dst = a.accumulate ? *dst + lad_acc : lad_acc;
        }
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

// ------------------------------------------------------------------------------------------
// K7b: the same layer with the GEMM on the bf16 matrix pipe at fp32 accuracy.  Every fp32 operand
// is the sum of three bf16 numbers (x = hi + mid + lo, exact to 2^-25 |x|); the six largest cross
// products (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 -- 6 x 8 passes per 16 k instead of 8 x 16 passes for the f32 MFMA, and
// unlike the f32 MFMA it does not occupy the VALU that the spline arithmetic needs.
// The weight pieces (host-split, 24 KB per 32-row tile) are shared by the four waves of a
// workgroup through a double-buffered LDS tile; the activations are split once per row tile.
template <bool INVERSE>
__global__ void __launch_bounds__(kBlock, 2) rqs_fused_linear_bf16_kernel(const FusedArgs a) {
    // dynamic LDS: two weight tiles, then per wave a [32][dt|1] tile of transformed outputs
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const int ystride = a.dt | 1;
    vec4f* s_w = reinterpret_cast<vec4f*>(lds_dyn);
    __shared__ int s_dinv[128], s_slot[128], s_src[128], s_dst[128], s_tsrc[64];
    __shared__ unsigned char s_ist[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, dt = a.dt;
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
        s_slot[c] = 0;
    }
    __syncthreads();
    for (int c = tid; c < D; c += kBlock) s_dinv[s_dst[c]] = c;
    if (tid < dt) {
        const int64_t t = a.tidx[tid];
        if (t < 0 || t >= D) my_status |= NFA_STATUS_BAD_INDEX;
        const int col = (int)(t < 0 ? 0 : (t >= D ? D - 1 : t));
        s_tsrc[tid] = s_src[col];
        s_ist[col] = 1;
        s_slot[col] = tid;
    }

    float* s_y = lds_dyn + 2 * kWTileVec4 * 4 + wave * 32 * ystride;
    const int half = lane >> 5, r = lane & 31;
    const int groups = dt >> 2;
    const int ntiles = groups * 3;
    const int64_t num_quads = a.batch >> 7;  // 4 waves x 32 samples
    const vec4f* wg = reinterpret_cast<const vec4f*>(a.wpacked);

    // weight tile 0 -> LDS buffer 0
    {
        vec4f w[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = wg[tid + i * kBlock];
#pragma unroll
        for (int i = 0; i < 6; ++i) s_w[tid + i * kBlock] = w[i];
    }
    __syncthreads();
#ifdef NFA_K7B_STAGGER
    // experiment: offset the second workgroup resident on a CU by about half a group period so
    // that its MFMA phases meet the other workgroup's spline (VALU) phases
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_sleep(NFA_K7B_STAGGER);
#endif
    int it = 0;  // running tile counter: parity selects the LDS buffer holding the current tile

    unsigned long long* tr = nullptr;  // debug trace (tools/k7_trace.py)
    int ti = 0;
    if (a.trace && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 256))
        tr = a.trace + (blockIdx.x ? 256 : 0);
    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = (quad << 7) + (wave << 5);
        NFA_STAMP()
        // ---- hidden^T, split into bf16 pieces: lane (sample r, half) covers k = half*64 + ks*8 + 0..7
        bf16x8 bh[8], bm[8], bl[8];
        {
            const vec4f* hp = reinterpret_cast<const vec4f*>(a.hidden + (row0 + r) * kH + half * 64);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const vec4f v0 = hp[ks * 2], v1 = hp[ks * 2 + 1];
                bf16x2 h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
                split3(vec2f{v0.x, v0.y}, h0, m0, l0);
                split3(vec2f{v0.z, v0.w}, h1, m1, l1);
                split3(vec2f{v1.x, v1.y}, h2, m2, l2);
                split3(vec2f{v1.z, v1.w}, h3, m3, l3);
                bh[ks] = join4(h0, h1, h2, h3);
                bm[ks] = join4(m0, m1, m2, m3);
                bl[ks] = join4(l0, l1, l2, l3);
            }
        }
        float lad_acc = 0.0f;
        const vec4f* bias_lane = reinterpret_cast<const vec4f*>(a.bpad) + half * 4;
        NFA_STAMP()

        for (int g = 0; g < groups; ++g) {
            const float xin0 = a.x[(row0 + r) * D + s_tsrc[g * 4 + half * 2]];
            const float xin1 = a.x[(row0 + r) * D + s_tsrc[g * 4 + half * 2 + 1]];
            f32x16 acc[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int nt = g * 3 + t;
                {
                    const vec4f* bp = bias_lane + (size_t)nt * 8;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const vec4f b = bp[q4];
                        acc[t][q4 * 4 + 0] = b.x;
                        acc[t][q4 * 4 + 1] = b.y;
                        acc[t][q4 * 4 + 2] = b.z;
                        acc[t][q4 * 4 + 3] = b.w;
                    }
                }
                // next tile (wrapping to tile 0 for the next quad): global -> registers now,
                // registers -> the other LDS buffer after this tile's MFMAs
                const int ntn = (nt + 1 < ntiles) ? nt + 1 : 0;
                const vec4f* wn = wg + (size_t)ntn * kWTileVec4;
                vec4f wnext[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) wnext[i] = wn[tid + i * kBlock];

                const vec4f* cur = s_w + (it & 1) * kWTileVec4 + lane;
#ifndef NFA_K7_NOMFMA
                bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(0 * 8) * 64]);
                bf16x8 am = __builtin_bit_cast(bf16x8, cur[(1 * 8) * 64]);
                bf16x8 al = __builtin_bit_cast(bf16x8, cur[(2 * 8) * 64]);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    // the next k-step's weight fragments are requested before this step's MFMAs
                    const int kn = ks < 7 ? ks + 1 : 7;
                    const bf16x8 nh = __builtin_bit_cast(bf16x8, cur[(0 * 8 + kn) * 64]);
                    const bf16x8 nm = __builtin_bit_cast(bf16x8, cur[(1 * 8 + kn) * 64]);
                    const bf16x8 nl = __builtin_bit_cast(bf16x8, cur[(2 * 8 + kn) * 64]);
                    // smallest products first
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[ks], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ks], acc[t], 0, 0, 0);
                    ah = nh;
                    am = nm;
                    al = nl;
                }
#else
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    acc[t][ks] += cur[ks * 64].x + __builtin_bit_cast(vec4f, bh[ks]).x + __builtin_bit_cast(vec4f, bm[ks]).y + __builtin_bit_cast(vec4f, bl[ks]).z;
#endif
                NFA_STAMP()
                vec4f* nxt = s_w + ((it + 1) & 1) * kWTileVec4;
#pragma unroll
                for (int i = 0; i < 6; ++i) nxt[tid + i * kBlock] = wnext[i];
                __syncthreads();
                ++it;
                NFA_STAMP()
            }
            {
                NFA_K7_FEATURE_A(pa, acc[0], acc[1]);
                NFA_K7_FEATURE_B(pb, acc[1], acc[2]);
                float y0, l0, y1, l1;
#ifdef NFA_K7_NOSPLINE
                y0 = xin0 + pa[0] + pa[23]; l0 = pa[8] + pa[16];
                y1 = xin1 + pb[0] + pb[23]; l1 = pb[8] + pb[16];
#else
                my_status |= rqs_eval_flat8<INVERSE>(xin0, pa, a.sp, y0, l0);
                my_status |= rqs_eval_flat8<INVERSE>(xin1, pb, a.sp, y1, l1);
#endif
                s_y[r * ystride + g * 4 + half * 2] = y0;
                s_y[r * ystride + g * 4 + half * 2 + 1] = y1;
                lad_acc += l0;
                lad_acc += l1;
            }
            NFA_STAMP()
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (int e0 = lane; e0 < 32 * D; e0 += kWave * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // all eight gathers in flight before the first store
                const int e = e0 + u * kWave;
                v[u] = 0.0f;
                if (e < 32 * D) {
                    const int rr = (int)fastdiv((uint32_t)e, a.div_D);
                    const int c = s_dinv[e - rr * D];
                    v[u] = s_ist[c] ? s_y[rr * ystride + s_slot[c]] : a.x[(row0 + rr) * D + s_src[c]];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * kWave;
                if (e < 32 * D) a.out[row0 * D + e] = v[u];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lad_acc += __shfl_xor(lad_acc, 32, kWave);
        if (half == 0) {
            float* dst = a.lad + row0 + r;
            *dst = a.accumulate ? *dst + lad_acc : lad_acc;
        }
        NFA_STAMP()
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace nfa

using namespace nfa;

unsigned long long* nfa::g_k7_trace = nullptr;
// debug aid (tools/k7_trace.py): device buffer of 512 uint64 receiving phase timestamps; NULL = off
extern "C" void nfa_debug_k7_trace(void* device_buffer) { nfa::g_k7_trace = (unsigned long long*)device_buffer; }

extern "C" int nfa_rqs_coupling_fused_linear_f32(const float* inputs, const float* hidden,
                                                 const float* weight_packed, const float* bias_padded,
                                                 const int64_t* transform_idx, const int64_t* in_perm,
                                                 const int64_t* out_scatter, float* outputs,
                                                 float* logabsdet, int32_t* status, int64_t batch,
                                                 int32_t features, int32_t num_transform,
                                                 int32_t hidden_features, const nfa_rqs_spec* spec,
                                                 int32_t flags, void* stream) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET | NFA_FLAG_WEIGHTS_BF16X3))
        return NFA_ERR_INVALID_ARGUMENT;
    const bool split_bf16 = (flags & NFA_FLAG_WEIGHTS_BF16X3) != 0;
    if (batch < 0 || features < 1 || num_transform < 1 || num_transform > features)
        return NFA_ERR_INVALID_ARGUMENT;
    FusedArgs a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (a.sp.K != 8 || !a.sp.linear || hidden_features != kH || (num_transform & 3) != 0 ||
        num_transform > 64 || features > 128 || (batch & (split_bf16 ? 127 : 31)) != 0)
        return NFA_ERR_UNSUPPORTED;
    if (batch == 0) return NFA_OK;
    if (!inputs || !hidden || !weight_packed || !bias_padded || !transform_idx || !outputs || !logabsdet)
        return NFA_ERR_INVALID_ARGUMENT;
    a.x = inputs;
    a.hidden = hidden;
    a.wpacked = weight_packed;
    a.bpad = bias_padded;
    a.tidx = transform_idx;
    a.perm = in_perm;
    a.scatter = out_scatter;
    a.out = outputs;
    a.lad = logabsdet;
    a.status = status;
    a.batch = batch;
    a.D = features;
    a.dt = num_transform;
    a.div_D = make_fastdiv((uint32_t)features);
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    a.trace = g_k7_trace;
    const int64_t cap = (int64_t)device_cu_count() * 2;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    profile_next_launch(&e0, &e1);
    hipStream_t st = (hipStream_t)stream;
    const dim3 block(kBlock);
    const bool inverse = (flags & NFA_FLAG_INVERSE) != 0;
    const size_t ybytes = (size_t)(kBlock / kWave) * 32 * (num_transform | 1) * sizeof(float);
    if (split_bf16) {
        int64_t blocks = batch >> 7;
        if (blocks > cap) blocks = cap;
        const dim3 grid((unsigned)blocks);
        const size_t lds = ybytes + 2 * kWTileVec4 * 16;
        auto kern = inverse ? rqs_fused_linear_bf16_kernel<true> : rqs_fused_linear_bf16_kernel<false>;
        if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, a);
        else hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    } else {
        int64_t blocks = ((batch >> 5) + 3) / 4;
        if (blocks > cap) blocks = cap;
        const dim3 grid((unsigned)blocks);
        auto kern = inverse ? rqs_fused_linear_kernel<true> : rqs_fused_linear_kernel<false>;
        if (e0) hipExtLaunchKernelGGL(kern, grid, block, ybytes, st, e0, e1, 0, a);
        else hipLaunchKernelGGL(kern, grid, block, ybytes, st, a);
    }
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
