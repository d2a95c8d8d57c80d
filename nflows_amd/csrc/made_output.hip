// K13: the output layer of a MADE conditioner + the autoregressive spline layer in ONE kernel.
//
//   params  = hidden @ (W * mask)^T + b      MADE.final_layer, made.py:261-268, :282 (MaskedLinear :71-72)
//   outputs, logabsdet = RQ spline per feature, summed per sample
//                                            MaskedPiecewiseRationalQuadraticAutoregressiveTransform._elementwise,
//                                            autoregressive.py:453-489 (forward :38-41; the inverse's last pass)
//
// BASELINE configs[4] (D = 784 features, 23 logits each, H = 256): the [B, 18 032] parameter tensor is 295 MB at
// B = 4 096 -- written by a library fp32 GEMM (37.8 GFLOP, the whole cost of the layer's forward pass) and read back
// by the spline kernel.  Here it never exists: K7b's scheme (rqs_fused_linear.hip) for a 256-wide hidden vector and
// any number of features.  A wave owns 32 samples, hidden^T (split into three bf16 pieces once) is the MFMA B
// operand and stays in registers, the weights -- masked, re-tiled and split by the host (ops.pack_made_output) --
// are the A operand, shared by the four waves of a workgroup through a double-buffered LDS tile; six bf16 products
// per fp32 multiply-add (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid: fp32 accuracy, full fp32 range).  The rows
// of a group of four features are ordered so that the 48 accumulator values of a lane are the 24 + 24 logits of
// its two features (ops._k7_row_order): the spline is evaluated straight from the accumulators.
//
// Features are independent given the hidden vector, so the grid is (128-row blocks) x (chunks of up to 26 groups
// = 104 features): B = 4 096 x D = 784 gives 32 x 8 = 256 workgroups.  A chunk writes its own columns of the
// output rows and ITS part of every row's log-determinant into logabsdet_partial[chunk][row]; the caller adds the
// chunks up in a fixed order (no atomics: the same bits on every run).
//
// Serves the forward pass (all features, inputs = x) and the tail of the column-wise inverse (the features behind
// the last sequential one, autoregressive.py:43-52 restated in transforms/autoregressive.py: inputs = z, the hidden
// vector final).  Supported: 8 bins, linear tails, H <= 256 (H % 4 == 0), batch % 128 == 0.

#include "fused_common.hpp"

#include <hip/hip_ext.h>

namespace nfa {

constexpr int kMadeOutK = 256;                       // k columns of a weight tile (hidden width, zero-padded)
constexpr int kMadeOutTileVec4 = 3 * 16 * 64;        // one weight tile: [piece][k-step][lane] x 16 bytes = 48 KB
constexpr int kMadeOutGroups = NFA_MADE_OUTPUT_GROUPS_PER_CHUNK;
static_assert((kMadeOutGroups & 1) == 0, "the tile loop walks pairs of groups");

struct MadeOutArgs {
    const float* x;        // [B, ld] values the spline is applied to (columns c0 .. c0 + nf - 1)
    const float* hidden;   // [B, H]
    const vec4f* wpacked;  // [groups * 3 tiles][3 pieces][16 k-steps][64 lanes] x 16 bytes
    const float* bpad;     // [tiles][2 lane-halves][16]
    float* out;            // [B, ld]
    float* lad_part;       // [chunks][B]
    int32_t* status;
    int64_t batch, ld;
    int H, nf, groups, c0, row_blocks;
    RqsDev sp;
};

__device__ __forceinline__ void bias_into_tile(f32x16& acc, const vec4f* bias_tile_half) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const vec4f b = bias_tile_half[q4];
        acc[q4 * 4 + 0] = b.x;
        acc[q4 * 4 + 1] = b.y;
        acc[q4 * 4 + 2] = b.z;
        acc[q4 * 4 + 3] = b.w;
    }
}

template <bool INVERSE>
__global__ void __launch_bounds__(kBlock, 1) rqs_made_output_kernel(const MadeOutArgs a) {
    // dynamic LDS: two weight tiles, the chunk's biases, then per wave a [32][chunk features | 1] tile of results
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    vec4f* s_w = reinterpret_cast<vec4f*>(lds_dyn);
    float* s_bias = lds_dyn + 2 * kMadeOutTileVec4 * 4;                 // [tiles of the chunk][2 lane-halves][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, r = lane & 31;
    // XCD-aware numbering: workgroup ids go round-robin over the eight XCDs, each with its own 4 MB L2.  The row
    // blocks of one chunk read the same 3.7 MB of weights, so they are given ids of ONE XCD (id' = consecutive
    // within an XCD) and the chunk's weights stay in that L2 instead of every L2 streaming all chunks.
    int wid = blockIdx.x;
    {
        const int total = gridDim.x;
        if ((total & 7) == 0) wid = (wid & 7) * (total >> 3) + (wid >> 3);
    }
    const int chunk = wid / a.row_blocks, row_block = wid - chunk * a.row_blocks;
    const int g0 = chunk * kMadeOutGroups;
    const int ng = (a.groups - g0 < kMadeOutGroups) ? a.groups - g0 : kMadeOutGroups;   // even (the host pads)
    const int ntiles = ng * 3;
    constexpr int ystride = (kMadeOutGroups * 4) | 1;
    float* s_y = s_bias + kMadeOutGroups * 3 * 32 + wave * 32 * ystride;
    const int64_t row0 = ((int64_t)row_block << 7) + (wave << 5);
    const vec4f* wg = a.wpacked + (size_t)g0 * 3 * kMadeOutTileVec4;
    int my_status = 0;

    // weight tile 0 -> LDS buffer 0, tile 1 into registers: the NEXT tile is in flight (global -> registers) while
    // the current one is multiplied, and goes registers -> the other LDS buffer behind the MFMAs.  Nothing inside
    // the tile loop may wait for "all loads": the first version fetched every tile's biases in front of its MFMAs
    // (a vmcnt(0) that drained the prefetch: fetch and multiply ran one after the other, 3.7 us per tile) -- the
    // biases of the chunk now sit in LDS and the spline inputs are requested a pair of groups ahead.  (Two tiles in
    // flight in two register sets need more than the 512 registers of a lone wave: scratch traffic, which waits on
    // vmcnt as well.)
    // (round 4) the tiles come in by LDS-DMA: no staging registers, no ds_write pass, and no load the compiler knows
    // about -- so nothing makes it wait for "all loads" in the middle of a tile.  Tile nt + 1 is requested at the
    // beginning of tile nt, into the buffer whose readers all passed the barrier behind tile nt - 1, and awaited
    // (vmcnt(0): the spline inputs of the next pair ride along) in front of the barrier behind tile nt.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lane_off16 = (unsigned)lane * 16u;
    auto request_tile = [&](int tile_index, int buffer) {
        const char* src = reinterpret_cast<const char*>(wg + (size_t)tile_index * kMadeOutTileVec4) + wave_u * 1024;
        char* dst = reinterpret_cast<char*>(s_w + buffer * kMadeOutTileVec4) + wave_u * 1024;
#pragma unroll
        for (int i = 0; i < 12; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 4096 + lane_off16),
                                             (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, 0, 0);
    };
    {
        request_tile(0, 0);
        const float* bsrc = a.bpad + (size_t)g0 * 3 * 32;
        for (int i = tid; i < ntiles * 32; i += kBlock) s_bias[i] = bsrc[i];
    }

    // hidden^T, split into bf16 pieces: lane (sample r, half) covers k = half*128 + ks*8 + 0..7 (columns past the
    // hidden width: zero, like their zero-padded weights; the loads are unconditional, clamped)
    bf16x8 bh[16], bm[16], bl[16];
    {
        const float* hrow = a.hidden + (row0 + r) * a.H;
#pragma unroll
        for (int part = 0; part < 2; ++part) {   // eight k-steps' loads in flight at a time
            vec4f hv[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = half * 128 + (part * 8 + i) * 8;
                hv[2 * i] = *reinterpret_cast<const vec4f*>(hrow + (k < a.H ? k : 0));
                hv[2 * i + 1] = *reinterpret_cast<const vec4f*>(hrow + (k + 4 < a.H ? k + 4 : 0));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ks = part * 8 + i;
                const int k = half * 128 + ks * 8;
                const vec4f zero = {0.f, 0.f, 0.f, 0.f};
                const vec4f v0 = k < a.H ? hv[2 * i] : zero, v1 = k + 4 < a.H ? hv[2 * i + 1] : zero;
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
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile 0 (LDS-DMA) before the barrier
    __syncthreads();

    float lad_acc = 0.0f;
    const float* xrow = a.x + (row0 + r) * a.ld + a.c0;
    const int last = a.nf - 1;
    // the lane's features of the first pair of groups (padding features past nf: zero weights, results dropped;
    // their loads re-read the last feature)
    float xin[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = (g0 + (j >> 1)) * 4 + half * 2 + (j & 1);
        xin[j] = xrow[f < last ? f : last];
    }
    for (int g = 0; g < ng; g += 2) {
        // the next pair's spline inputs: requested now, in front of this pair's weight refills
        float xnext[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = (g0 + g + 2 + (j >> 1)) * 4 + half * 2 + (j & 1);
            xnext[j] = xrow[f < last ? f : last];
        }
        f32x16 acc[3];
#pragma unroll
        for (int u = 0; u < 6; ++u) {   // six tiles = two groups: the LDS buffers alternate statically
            const int t = u % 3;
            const int nt = g * 3 + u;
            request_tile(nt + 1 < ntiles ? nt + 1 : 0, (u + 1) & 1);   // (past the chunk's last: tile 0 again, never used)
            bias_into_tile(acc[t], reinterpret_cast<const vec4f*>(s_bias + nt * 32 + half * 16));
            const vec4f* cur = s_w + (u & 1) * kMadeOutTileVec4 + lane;
            bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(0 * 16) * 64]);
            bf16x8 am = __builtin_bit_cast(bf16x8, cur[(1 * 16) * 64]);
            bf16x8 al = __builtin_bit_cast(bf16x8, cur[(2 * 16) * 64]);
#ifdef NFA_K13_ABL_NO_MFMA
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#else
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
#endif
                // the next k-step's weight fragments are requested before this step's MFMAs
                const int kn = ks < 15 ? ks + 1 : 15;
                const bf16x8 nh = __builtin_bit_cast(bf16x8, cur[(0 * 16 + kn) * 64]);
                const bf16x8 nm = __builtin_bit_cast(bf16x8, cur[(1 * 16 + kn) * 64]);
                const bf16x8 nl = __builtin_bit_cast(bf16x8, cur[(2 * 16 + kn) * 64]);
                // smallest products first
                NFA_MFMA6(acc[t], ah, am, al, bh[ks], bm[ks], bl[ks]);   // (bf16x3_gemm.hpp: product order)
                ah = nh;
                am = nm;
                al = nl;
            }
            // tile nt + 1 has landed (this wave's requests), and every wave is done with tile nt
            asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (t == 2) {
                const int gg = g + u / 3;
                const int f0 = (g0 + gg) * 4 + half * 2;
                const bool has0 = f0 < a.nf, has1 = f0 + 1 < a.nf;
                NFA_K7_FEATURE_A(pa, acc[0], acc[1]);
                NFA_K7_FEATURE_B(pb, acc[1], acc[2]);
                float y0, l0, y1, l1;
#ifdef NFA_K13_ABL_NO_EVAL   // (measurement: the GEMM structure without the spline arithmetic)
                y0 = pa[0] + pa[23] + xin[2 * (u / 3)];
                l0 = pa[5];
                y1 = pb[0] + pb[23] + xin[2 * (u / 3) + 1];
                l1 = pb[7];
                const int st0 = 0, st1 = 0;
#else
                const int st0 = rqs_eval_flat8<INVERSE>(xin[2 * (u / 3)], pa, a.sp, y0, l0);
                const int st1 = rqs_eval_flat8<INVERSE>(xin[2 * (u / 3) + 1], pb, a.sp, y1, l1);
#endif
                float* y_slot = s_y + r * ystride + gg * 4 + half * 2;
                y_slot[0] = y0;
                y_slot[1] = y1;
                if (has0) {
                    lad_acc += l0;
                    my_status |= st0;
                }
                if (has1) {
                    lad_acc += l1;
                    my_status |= st1;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) xin[j] = xnext[j];
    }

    // the wave's 32 rows x the chunk's columns, row by row (consecutive lanes = consecutive columns)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const int first = g0 * 4;
        const int ncols = (a.nf - first < ng * 4) ? a.nf - first : ng * 4;
        float* obase = a.out + row0 * a.ld + a.c0 + first;
        for (int rr = 0; rr < 32; ++rr)
            for (int j = lane; j < ncols; j += kWave) obase[rr * a.ld + j] = s_y[rr * ystride + j];
    }
    lad_acc += __shfl_xor(lad_acc, 32, kWave);
    if (half == 0) a.lad_part[(int64_t)chunk * a.batch + row0 + r] = lad_acc;
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_rqs_made_output_f32(const float* inputs, int64_t row_stride, int32_t first_column,
                                       const float* hidden, int32_t hidden_features, const void* weight_packed,
                                       const float* bias_padded, float* outputs, float* logabsdet_partial,
                                       int32_t* status, int64_t batch, int32_t num_features, const nfa_rqs_spec* spec,
                                       int32_t flags, void* stream) {
    if (flags & ~NFA_FLAG_INVERSE) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || num_features < 1 || first_column < 0 || row_stride < (int64_t)first_column + num_features ||
        hidden_features < 1)
        return NFA_ERR_INVALID_ARGUMENT;
    MadeOutArgs a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (a.sp.K != 8 || !a.sp.linear || hidden_features > kMadeOutK || (hidden_features & 3) != 0 || (batch & 127) != 0 ||
        (batch >> 7) > 0x7fffffffLL / 64)
        return NFA_ERR_UNSUPPORTED;
    if (batch == 0) return NFA_OK;
    if (!inputs || !hidden || !weight_packed || !bias_padded || !outputs || !logabsdet_partial)
        return NFA_ERR_INVALID_ARGUMENT;
    a.x = inputs;
    a.hidden = hidden;
    a.wpacked = reinterpret_cast<const vec4f*>(weight_packed);
    a.bpad = bias_padded;
    a.out = outputs;
    a.lad_part = logabsdet_partial;
    a.status = status;
    a.batch = batch;
    a.ld = row_stride;
    a.H = hidden_features;
    a.nf = num_features;
    a.groups = ((num_features + 7) >> 3) << 1;   // groups of four features, an even number of them (zero rows)
    a.c0 = first_column;
    const int chunks = (a.groups + kMadeOutGroups - 1) / kMadeOutGroups;
    const size_t lds = (size_t)2 * kMadeOutTileVec4 * 16 + (size_t)kMadeOutGroups * 3 * 32 * sizeof(float) +
                       (size_t)(kBlock / kWave) * 32 * ((kMadeOutGroups * 4) | 1) * sizeof(float);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    profile_next_launch(&e0, &e1);
    hipStream_t st = (hipStream_t)stream;
    a.row_blocks = (int)(batch >> 7);
    const dim3 grid((unsigned)((batch >> 7) * chunks)), block(kBlock);
    const bool inverse = (flags & NFA_FLAG_INVERSE) != 0;
    void (*kern)(const MadeOutArgs) = inverse ? rqs_made_output_kernel<true> : rqs_made_output_kernel<false>;
    note_layer_kernel("rqs_made_output_kernel<inverse=%d>", inverse ? 1 : 0);
    static unsigned long long raised[2] = {};   // device masks (raise_dynamic_lds)
    const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[inverse ? 1 : 0], (int)lds);
    if (rc_lds != NFA_OK) return rc_lds;
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
