// K7 / K7b: conditioner output layer + spline coupling layer in ONE kernel (SURVEY.md section 8f,
// row f3: "consume the final Linear's tiles ... instead of a [B, d_t*P] round trip through HBM").
//
//   params = hidden @ W^T + b            Linear(H -> d_t*P), nn/nets/resnet.py:90, :99
//   outputs, logabsdet = RQ coupling     coupling.py:73-130, :549-582 (exactly K1's arithmetic)
//
// The [B, d_t*P] parameter tensor (193 MB per layer at the BASELINE shape, written by the GEMM
// and read back by K1) never exists: a wave owns 32 samples and computes params^T = W x hidden^T
// tile by tile on the matrix cores (hidden^T, 128 activations per sample, is the B operand and
// stays in registers; the weights, re-tiled by the host, are the A operand).  The host orders the
// weight rows so that the 48 accumulator values a lane receives from the three tiles of a group
// are the 24 + 24 logits (23 + one pad row) of that lane's own two features: the spline is
// evaluated straight from the accumulators.  Outputs go through a small LDS tile and are written
// as whole rows.
//
// Restrictions of this fast path (the host falls back to GEMM + K1 otherwise): K = 8 bins,
// linear tails (P = 23), hidden width 128, d_t a multiple of 4, batch a multiple of 32 (K7) or
// 128 (K7b) handled here (leftover rows go through the unfused path).  K8 (rqs_resnet.hip) extends
// the same scheme to the whole conditioner.

#include "fused_common.hpp"

#include <hip/hip_ext.h>

namespace nfa {

constexpr int kH = 128;  // hidden width (GEMM K dimension)

struct FusedArgs {
    const float* x;       // [B, D]
    const float* hidden;  // [B, 128]
    const float* wpacked; // fp32 [(dt*24/32) tiles][16][64][4], or bf16 triples (see the header)
    const float* bpad;    // [tiles][2][16]
    const int64_t* tidx;
    const int64_t* perm;
    const int64_t* scatter;
    float* out;
    float* lad;
    int32_t* status;
    int64_t batch;  // multiple of 32 (K7) / 128 (K7b)
    int D, dt, accumulate;
    FastDiv div_D;
    RqsDev sp;
    unsigned long long* trace;  // debug: per-phase timestamps of a few waves (null normally)
};

// accumulators of a 32-row tile start from the bias (accumulator order [lane-half][16])
__device__ __forceinline__ void bias_into(f32x16& acc, const vec4f* bias_tile_half) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const vec4f b = bias_tile_half[q4];
        acc[q4 * 4 + 0] = b.x;
        acc[q4 * 4 + 1] = b.y;
        acc[q4 * 4 + 2] = b.z;
        acc[q4 * 4 + 3] = b.w;
    }
}

// the lane's two features of a group, straight from the three accumulators; results into the
// wave's LDS y tile, log-derivatives summed
template <bool INVERSE>
__device__ __forceinline__ int eval_group(const f32x16 (&acc)[3], float xin0, float xin1, const RqsDev& sp,
                                          float* y_slot, float& lad_acc) {
    NFA_K7_FEATURE_A(pa, acc[0], acc[1]);
    NFA_K7_FEATURE_B(pb, acc[1], acc[2]);
    float y0, l0, y1, l1;
    int st = rqs_eval_flat8<INVERSE>(xin0, pa, sp, y0, l0);
    st |= rqs_eval_flat8<INVERSE>(xin1, pb, sp, y1, l1);
    y_slot[0] = y0;
    y_slot[1] = y1;
    lad_acc += l0;
    lad_acc += l1;
    return st;
}

__device__ __forceinline__ void store_lad(float* lad, int64_t row0, int r, int half, float lad_acc, int accumulate) {
    lad_acc += __shfl_xor(lad_acc, 32, kWave);
    if (half == 0) {
        float* dst = lad + row0 + r;
        *dst = accumulate ? *dst + lad_acc : lad_acc;
    }
}

// ---- K7: fp32 MFMA, weights straight from L2, waves independent (no workgroup barriers) ------
template <bool INVERSE>
__global__ void __launch_bounds__(kBlock, 2) rqs_fused_linear_kernel(const FusedArgs a) {
    // dynamic LDS: per wave a [32][dt|1] tile of transformed outputs
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ LayerTables T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, dt = a.dt, ystride = dt | 1;
    int my_status = build_layer_tables(T, a.perm, a.scatter, a.tidx, nullptr, D, dt, 0, tid, kBlock);

    float* s_y = lds_dyn + wave * 32 * ystride;
    const int half = lane >> 5, r = lane & 31;
    const int groups = dt >> 2;  // 4 features = 3 MFMA tiles per group
    const int64_t num_tiles = a.batch >> 5;
    const int64_t wave_global = (int64_t)blockIdx.x * (kBlock / kWave) + wave;
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / kWave);

    unsigned long long* tr = nullptr;
    int ti = 0;
    if (a.trace && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 256))
        tr = a.trace + (blockIdx.x ? 256 : 0);
    for (int64_t tile = wave_global; tile < num_tiles; tile += nwaves) {
        const int64_t row0 = tile << 5;
        NFA_STAMP()
        // hidden^T as the MFMA B operand: lane (sample r, half) holds hidden[row0 + r][half*64 ..+63]
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
            const float xin0 = a.x[(row0 + r) * D + T.tsrc[g * 4 + half * 2]];
            const float xin1 = a.x[(row0 + r) * D + T.tsrc[g * 4 + half * 2 + 1]];
            f32x16 acc[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) bias_into(acc[t], bias_lane + (size_t)(g * 3 + t) * 8);
#pragma unroll
            for (int hh = 0; hh < 6; ++hh) {
                const int ht = g * 6 + hh;
                const int htn = (ht + 1 < num_half_tiles) ? ht + 1 : 0;
                const vec4f* wn = wbase + (size_t)htn * 8 * 64;
                vec4f wnext[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wnext[j] = wn[j * 64];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const vec4f h4 = hv[(hh & 1) * 8 + j];
                    acc[hh >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j].x, h4.x, acc[hh >> 1], 0, 0, 0);
                    acc[hh >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j].y, h4.y, acc[hh >> 1], 0, 0, 0);
                    acc[hh >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j].z, h4.z, acc[hh >> 1], 0, 0, 0);
                    acc[hh >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j].w, h4.w, acc[hh >> 1], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = wnext[j];
            }
            NFA_STAMP()
            my_status |= eval_group<INVERSE>(acc, xin0, xin1, a.sp, s_y + r * ystride + g * 4 + half * 2, lad_acc);
            NFA_STAMP()
        }
        assemble_rows(T, s_y, ystride, a.x, a.out, row0, D, a.div_D, lane);
        store_lad(a.lad, row0, r, half, lad_acc, a.accumulate);
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

// ---- K7b: the same layer with the GEMM on the bf16 matrix pipe at fp32 accuracy.  Every fp32
// operand is the sum of three bf16 numbers (x = hi + mid + lo, exact to 2^-25 |x|); the six largest
// cross products (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 -- 6 x 8 passes per 16 k instead of 8 x 16 passes for the f32 MFMA.
// The weight pieces (host-split, 24 KB per 32-row tile) are shared by the four waves of a
// workgroup through a double-buffered LDS tile; the activations are split once per row tile.
template <bool INVERSE>
__global__ void __launch_bounds__(kBlock, 2) rqs_fused_linear_bf16_kernel(const FusedArgs a) {
    // dynamic LDS: two weight tiles, then per wave a [32][dt|1] tile of transformed outputs
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ LayerTables T;
    vec4f* s_w = reinterpret_cast<vec4f*>(lds_dyn);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, dt = a.dt, ystride = dt | 1;
    int my_status = build_layer_tables(T, a.perm, a.scatter, a.tidx, nullptr, D, dt, 0, tid, kBlock);

    float* s_y = lds_dyn + 2 * kWTileVec4 * 4 + wave * 32 * ystride;
    const int half = lane >> 5, r = lane & 31;
    const int groups = dt >> 2;
    const int ntiles = groups * 3;
    const int64_t num_quads = a.batch >> 7;  // 4 waves x 32 samples
    const vec4f* wg = reinterpret_cast<const vec4f*>(a.wpacked);

    {  // weight tile 0 -> LDS buffer 0
        vec4f w[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = wg[tid + i * kBlock];
#pragma unroll
        for (int i = 0; i < 6; ++i) s_w[tid + i * kBlock] = w[i];
    }
    __syncthreads();
    int it = 0;  // running tile counter: parity selects the LDS buffer holding the current tile

    unsigned long long* tr = nullptr;
    int ti = 0;
    if (a.trace && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 256))
        tr = a.trace + (blockIdx.x ? 256 : 0);
    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = (quad << 7) + (wave << 5);
        NFA_STAMP()
        // hidden^T, split into bf16 pieces: lane (sample r, half) covers k = half*64 + ks*8 + 0..7
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
            const float xin0 = a.x[(row0 + r) * D + T.tsrc[g * 4 + half * 2]];
            const float xin1 = a.x[(row0 + r) * D + T.tsrc[g * 4 + half * 2 + 1]];
            f32x16 acc[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int nt = g * 3 + t;
                bias_into(acc[t], bias_lane + (size_t)nt * 8);
                // next tile (wrapping to tile 0 for the next quad): global -> registers now,
                // registers -> the other LDS buffer after this tile's MFMAs
                const int ntn = (nt + 1 < ntiles) ? nt + 1 : 0;
                const vec4f* wn = wg + (size_t)ntn * kWTileVec4;
                vec4f wnext[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) wnext[i] = wn[tid + i * kBlock];

                const vec4f* cur = s_w + (it & 1) * kWTileVec4 + lane;
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
                    NFA_MFMA6(acc[t], ah, am, al, bh[ks], bm[ks], bl[ks]);   // (bf16x3_gemm.hpp: product order)
                    ah = nh;
                    am = nm;
                    al = nl;
                }
                NFA_STAMP()
                vec4f* nxt = s_w + ((it + 1) & 1) * kWTileVec4;
#pragma unroll
                for (int i = 0; i < 6; ++i) nxt[tid + i * kBlock] = wnext[i];
                __syncthreads();
                ++it;
                NFA_STAMP()
            }
            my_status |= eval_group<INVERSE>(acc, xin0, xin1, a.sp, s_y + r * ystride + g * 4 + half * 2, lad_acc);
            NFA_STAMP()
        }
        assemble_rows(T, s_y, ystride, a.x, a.out, row0, D, a.div_D, lane);
        store_lad(a.lad, row0, r, half, lad_acc, a.accumulate);
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
    note_layer_kernel("%s<inverse=%d>", split_bf16 ? "rqs_fused_linear_bf16_kernel" : "rqs_fused_linear_kernel", inverse ? 1 : 0);
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
