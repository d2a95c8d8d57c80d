// K8: a whole neural-spline coupling layer in ONE kernel -- the ResidualNet conditioner
// (nn/nets/resnet.py:55-100: Linear(d_i -> 128), num_blocks x [ReLU, Linear, ReLU, Linear, +skip],
// Linear(128 -> d_t*23)) followed by everything K1 replaces (coupling.py:73-130, :549-582).
//
//   * A wave owns 32 samples for the whole layer.  Their activations never leave the register
//     file: every GEMM is computed transposed (out^T = W x act^T), so the 32x32 accumulator tiles
//     a lane holds after one layer are -- up to a fixed permutation of the k index that the host
//     applies to the next layer's weight columns -- exactly the MFMA B operand of the next layer.
//   * GEMMs run on the bf16 matrix pipe at fp32 accuracy: operands are split into three bf16
//     pieces (x = hi + mid + lo), six cross products per k-step (see K7b in rqs_fused_linear.hip).
//     Weights are split on the host, activations in registers right after each layer.
//   * The weights of the whole layer (984 KB as bf16 triples at the BASELINE shape) are streamed
//     through a double-buffered 24 KB LDS stage shared by the four waves of the workgroup.
//   * The residual input is not kept in fp32: it is rebuilt from its three pieces (exact to
//     2^-25 |h|) when the skip connection is added, and the first ReLU of a block is applied to
//     the pieces on the fly (sign of the leading piece), which keeps the kernel inside 256 VGPRs.
//   * The last GEMM's accumulators are the spline logits of the lane's own two features per
//     group; they are evaluated straight from registers (as in K7).
//
// Restrictions (the host falls back to PyTorch GEMMs + K7/K1 otherwise): K = 8 bins, linear
// tails, hidden width 128, ReLU, no context / batch norm / active dropout, d_i <= 32,
// d_t % 4 == 0, d_t <= 64, D <= 128, batch % 128 == 0 here (leftover rows: other path).

#include "fused_common.hpp"

#include <hip/hip_ext.h>

namespace nfa {

typedef short short2v __attribute__((ext_vector_type(2)));
typedef short short8v __attribute__((ext_vector_type(8)));

struct ResnetArgs {
    const float* x;      // [B, D]
    const vec4f* w;      // [num_stages][1536] x 16 bytes, layout in include/nflows_amd.h
    const float* bias;   // accumulator-order biases of all GEMMs
    const int64_t* tidx;
    const int64_t* iidx;
    const int64_t* perm;
    const int64_t* scatter;
    float* out;
    float* lad;
    int32_t* status;
    int64_t batch;  // multiple of 128
    int D, dt, di, num_blocks, num_stages, accumulate;
    FastDiv div_D;
    RqsDev sp;
};

// weight stream: the stage after the current one is fetched global -> registers while the current
// one is consumed from LDS, then stored into the other LDS buffer
struct WeightStream {
    vec4f* s_w;
    const vec4f* w;
    int it;          // stages consumed so far: parity selects the LDS buffer holding the current one
    int next_stage;  // index (in the layer's stage list) of the stage to fetch next
    int num_stages;
    int tid;
};

__device__ __forceinline__ void stage_fetch(const WeightStream& sm, vec4f (&wnext)[6]) {
    const vec4f* wn = sm.w + (size_t)sm.next_stage * kWTileVec4;
#pragma unroll
    for (int i = 0; i < 6; ++i) wnext[i] = wn[sm.tid + i * kBlock];
}

__device__ __forceinline__ void stage_commit(WeightStream& sm, const vec4f (&wnext)[6]) {
    vec4f* nxt = sm.s_w + ((sm.it + 1) & 1) * kWTileVec4;
#pragma unroll
    for (int i = 0; i < 6; ++i) nxt[sm.tid + i * kBlock] = wnext[i];
    __syncthreads();
    ++sm.it;
    sm.next_stage = (sm.next_stage + 1 == sm.num_stages) ? 0 : sm.next_stage + 1;
}

#define NFA_MFMA6(acc, ah, am, al, bh, bm, bl)                                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0)

// ReLU applied to a value given as bf16 pieces: all three are cleared where the leading piece is
// negative and not a NaN (bf16 bit patterns 0x8000..0xFF80 = int16 <= -128), so that NaNs keep
// propagating like torch.relu's.
__device__ __forceinline__ void relu_pieces(bf16x8& h, bf16x8& m, bf16x8& l) {
    const short8v hs = __builtin_bit_cast(short8v, h);
    const short8v zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const short8v neg = (__builtin_elementwise_min(hs, zero) + (short)127) >> 15;  // -1 where cleared
    h = __builtin_bit_cast(bf16x8, (short8v)(hs & ~neg));
    m = __builtin_bit_cast(bf16x8, (short8v)(__builtin_bit_cast(short8v, m) & ~neg));
    l = __builtin_bit_cast(bf16x8, (short8v)(__builtin_bit_cast(short8v, l) & ~neg));
}

// out^T[128 x 32 samples] += W[128 x (16*NKS)] x act^T, act given as pieces; NKS/2 stages of
// [2 k-steps][4 tiles][3 pieces][64 lanes] x 16 bytes
template <bool RELU, int NKS>
__device__ __forceinline__ void gemm_128_out(f32x16 (&acc)[4], const bf16x8 (&ph)[8], const bf16x8 (&pm)[8],
                                             const bf16x8 (&pl)[8], WeightStream& sm, int lane) {
#pragma unroll
    for (int st = 0; st < NKS / 2; ++st) {
        vec4f wnext[6];
        stage_fetch(sm, wnext);
        const vec4f* cur = sm.s_w + (sm.it & 1) * kWTileVec4 + lane;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 bh = ph[st * 2 + kk], bm = pm[st * 2 + kk], bl = pl[st * 2 + kk];
            if (RELU) relu_pieces(bh, bm, bl);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, cur[((kk * 4 + t) * 3 + 0) * 64]);
                const bf16x8 am = __builtin_bit_cast(bf16x8, cur[((kk * 4 + t) * 3 + 1) * 64]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, cur[((kk * 4 + t) * 3 + 2) * 64]);
                NFA_MFMA6(acc[t], ah, am, al, bh, bm, bl);
            }
        }
        stage_commit(sm, wnext);
    }
}

// accumulator tile t, registers 8*hk .. 8*hk+7  ->  pieces of k-step 2t + hk
template <bool RELU>
__device__ __forceinline__ void tile_to_pieces(const f32x16& a, bf16x8& h0, bf16x8& m0, bf16x8& l0,
                                               bf16x8& h1, bf16x8& m1, bf16x8& l1) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        v[q] = a[q];
        if (RELU) v[q] = (v[q] < 0.0f) ? 0.0f : v[q];  // NaN stays NaN
    }
    bf16x2 hh[8], mm[8], ll[8];
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) split3(vec2f{v[q2 * 2], v[q2 * 2 + 1]}, hh[q2], mm[q2], ll[q2]);
    h0 = join4(hh[0], hh[1], hh[2], hh[3]);
    m0 = join4(mm[0], mm[1], mm[2], mm[3]);
    l0 = join4(ll[0], ll[1], ll[2], ll[3]);
    h1 = join4(hh[4], hh[5], hh[6], hh[7]);
    m1 = join4(mm[4], mm[5], mm[6], mm[7]);
    l1 = join4(ll[4], ll[5], ll[6], ll[7]);
}

// value of the pieces of one k-step, added to 8 accumulator registers (the skip connection)
__device__ __forceinline__ void add_pieces(f32x16& a, int q0, const bf16x8& h, const bf16x8& m, const bf16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[q0 + j] += ((float)h[j] + (float)m[j]) + (float)l[j];
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

template <bool INVERSE>
__global__ void __launch_bounds__(kBlock, 2) rqs_resnet_kernel(const ResnetArgs a) {
    // dynamic LDS: two weight stages, then per wave a [32][dt|1] tile of transformed outputs
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ LayerTables T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, dt = a.dt;
    const int ystride = dt | 1;
    int my_status = build_layer_tables(T, a.perm, a.scatter, a.tidx, a.iidx, D, dt, a.di, tid, kBlock);

    WeightStream sm;
    sm.s_w = reinterpret_cast<vec4f*>(lds_dyn);
    sm.w = a.w;
    sm.it = 0;
    sm.next_stage = (a.num_stages > 1) ? 1 : 0;
    sm.num_stages = a.num_stages;
    sm.tid = tid;
    float* s_y = lds_dyn + 2 * kWTileVec4 * 4 + wave * 32 * ystride;
    const int half = lane >> 5, r = lane & 31;
    const int groups = dt >> 2;
    const int64_t num_quads = a.batch >> 7;

    {  // stage 0 -> LDS buffer 0
        vec4f w0[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w0[i] = a.w[tid + i * kBlock];
#pragma unroll
        for (int i = 0; i < 6; ++i) sm.s_w[tid + i * kBlock] = w0[i];
    }
    __syncthreads();

    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = (quad << 7) + (wave << 5);
        const float* xrow = a.x + (row0 + r) * D;
        const float* bias = a.bias + half * 16;  // + 32 per tile
        bf16x8 ph[8], pm[8], pl[8];  // the current activations (128 k per sample) as bf16 pieces

        // ---- identity features, gathered through the fused permutation: k = ks*16 + half*8 + j
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = ks * 16 + half * 8 + j;
                const float xv = xrow[T.isrc[i < a.di ? i : 0]];
                v[j] = i < a.di ? xv : 0.0f;
            }
            bf16x2 hh[4], mm[4], ll[4];
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) split3(vec2f{v[j2 * 2], v[j2 * 2 + 1]}, hh[j2], mm[j2], ll[j2]);
            ph[ks] = join4(hh[0], hh[1], hh[2], hh[3]);
            pm[ks] = join4(mm[0], mm[1], mm[2], mm[3]);
            pl[ks] = join4(ll[0], ll[1], ll[2], ll[3]);
        }

        // ---- initial layer: h = W_i x + b_i
        {
            f32x16 h[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) load_bias_tile(h[t], bias + t * 32);
            gemm_128_out<false, 2>(h, ph, pm, pl, sm, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                tile_to_pieces<false>(h[t], ph[2 * t], pm[2 * t], pl[2 * t], ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
        }
        bias += 128;

        // ---- residual blocks: h += W_1 relu(W_0 relu(h) + b_0) + b_1
        for (int blk = 0; blk < a.num_blocks; ++blk) {
            f32x16 u[4], v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) load_bias_tile(u[t], bias + t * 32);
            gemm_128_out<true, 8>(u, ph, pm, pl, sm, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                load_bias_tile(v[t], bias + 128 + t * 32);
                add_pieces(v[t], 0, ph[2 * t], pm[2 * t], pl[2 * t]);
                add_pieces(v[t], 8, ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
                tile_to_pieces<true>(u[t], ph[2 * t], pm[2 * t], pl[2 * t], ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
            }
            gemm_128_out<false, 8>(v, ph, pm, pl, sm, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                tile_to_pieces<false>(v[t], ph[2 * t], pm[2 * t], pl[2 * t], ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
            bias += 256;
        }

        // ---- final layer, three 32-row tiles (= 4 features) at a time, and the splines
        float lad_acc = 0.0f;
        for (int g = 0; g < groups; ++g) {
            const float xin0 = a.x[(row0 + r) * D + T.tsrc[g * 4 + half * 2]];
            const float xin1 = a.x[(row0 + r) * D + T.tsrc[g * 4 + half * 2 + 1]];
            f32x16 acc[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                load_bias_tile(acc[t], bias + (g * 3 + t) * 32);
                vec4f wnext[6];
                stage_fetch(sm, wnext);
                const vec4f* cur = sm.s_w + (sm.it & 1) * kWTileVec4 + lane;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(0 * 8 + ks) * 64]);
                    const bf16x8 am = __builtin_bit_cast(bf16x8, cur[(1 * 8 + ks) * 64]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, cur[(2 * 8 + ks) * 64]);
                    NFA_MFMA6(acc[t], ah, am, al, ph[ks], pm[ks], pl[ks]);
                }
                stage_commit(sm, wnext);
            }
            {
                NFA_K7_FEATURE_A(pa, acc[0], acc[1]);
                NFA_K7_FEATURE_B(pb, acc[1], acc[2]);
                float y0, l0, y1, l1;
                my_status |= rqs_eval_flat8<INVERSE>(xin0, pa, a.sp, y0, l0);
                my_status |= rqs_eval_flat8<INVERSE>(xin1, pb, a.sp, y1, l1);
                s_y[r * ystride + g * 4 + half * 2] = y0;
                s_y[r * ystride + g * 4 + half * 2 + 1] = y1;
                lad_acc += l0;
                lad_acc += l1;
            }
        }
        assemble_rows(T, s_y, ystride, a.x, a.out, row0, D, a.div_D, lane);
        lad_acc += __shfl_xor(lad_acc, 32, kWave);
        if (half == 0) {
            float* dst = a.lad + row0 + r;
            *dst = a.accumulate ? *dst + lad_acc : lad_acc;
        }
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_rqs_coupling_resnet_f32(const float* inputs, const void* weights_packed,
                                           const float* bias_packed, const int64_t* transform_idx,
                                           const int64_t* identity_idx, const int64_t* in_perm,
                                           const int64_t* out_scatter, float* outputs, float* logabsdet,
                                           int32_t* status, int64_t batch, int32_t features,
                                           int32_t num_transform, int32_t num_identity,
                                           int32_t hidden_features, int32_t num_blocks,
                                           const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET)) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1 || num_transform < 1 || num_identity < 1 ||
        num_transform + num_identity > features || num_blocks < 0)
        return NFA_ERR_INVALID_ARGUMENT;
    ResnetArgs a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (a.sp.K != 8 || !a.sp.linear || hidden_features != 128 || (num_transform & 3) != 0 ||
        num_transform > 64 || num_identity > 32 || features > 128 || (batch & 127) != 0 || num_blocks > 64)
        return NFA_ERR_UNSUPPORTED;
    if (batch == 0) return NFA_OK;
    if (!inputs || !weights_packed || !bias_packed || !transform_idx || !identity_idx || !outputs || !logabsdet)
        return NFA_ERR_INVALID_ARGUMENT;
    a.x = inputs;
    a.w = reinterpret_cast<const vec4f*>(weights_packed);
    a.bias = bias_packed;
    a.tidx = transform_idx;
    a.iidx = identity_idx;
    a.perm = in_perm;
    a.scatter = out_scatter;
    a.out = outputs;
    a.lad = logabsdet;
    a.status = status;
    a.batch = batch;
    a.D = features;
    a.dt = num_transform;
    a.di = num_identity;
    a.num_blocks = num_blocks;
    a.num_stages = 1 + 8 * num_blocks + num_transform * 24 / 32;
    a.div_D = make_fastdiv((uint32_t)features);
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    int64_t blocks = batch >> 7;
    const int64_t cap = (int64_t)device_cu_count() * 2;
    if (blocks > cap) blocks = cap;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    profile_next_launch(&e0, &e1);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(kBlock);
    const size_t lds = (size_t)(kBlock / kWave) * 32 * (num_transform | 1) * sizeof(float) + 2 * kWTileVec4 * 16;
    auto kern = (flags & NFA_FLAG_INVERSE) ? rqs_resnet_kernel<true> : rqs_resnet_kernel<false>;
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
