// Affine / additive coupling (K2), autoregressive affine (K2b), column permutation (K4),
// per-sample row sum (K3), standard-normal log-prob epilogue, and the library's host utilities.
// gfx950 (MI355X) only.  Reference lines are cited in include/nflows_amd.h.

#include "common.hpp"

#include <math.h>

namespace nfa {

static thread_local int g_last_hip_error = 0;

int set_hip_error(hipError_t e) {
    g_last_hip_error = (int)e;
    return NFA_ERR_HIP;
}

int device_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// ------------------------------------------------------------------------------------------
// K2: fused affine / additive coupling.  Same tile scheme as the spline layer: R whole samples
// per workgroup, conditioner output and inputs staged through LDS with 16-byte accesses.
struct AffineArgs {
    const float* x;
    const float* params;
    const float* scale;  // NFA_SCALE_GIVEN only
    const int64_t* tidx;
    const int64_t* perm;
    const int64_t* scatter;
    float* out;
    float* lad;
    int32_t* status;
    int64_t batch;
    int D, dt, R, pcols, activation, inverse, accumulate;
    FastDiv div_dt, div_D;
    int off_sc, off_x, off_out, off_lad, off_idx;
};

__global__ void __launch_bounds__(kBlock) affine_coupling_kernel(const AffineArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_p = lds;
    float* s_sc = lds + a.off_sc;
    float* s_x = lds + a.off_x;
    float* s_out = lds + a.off_out;
    float* s_lad = lds + a.off_lad;
    int* s_tidx = reinterpret_cast<int*>(lds + a.off_idx);
    int* s_src = s_tidx + a.dt;
    int* s_dst = s_src + a.D;
    unsigned char* s_ist = reinterpret_cast<unsigned char*>(s_dst + a.D);

    const int tid = threadIdx.x;
    const int D = a.D, dt = a.dt, pc = a.pcols;
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

    const int64_t num_tiles = (a.batch + a.R - 1) / a.R;
    for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * a.R;
        const int rows = (int)((a.batch - row0) < a.R ? (a.batch - row0) : a.R);
        const int nitems = rows * dt;
        const int mp = tile_load(a.params + row0 * pc, rows * pc, s_p, tid);
        int ms = 0;
        if (a.activation == NFA_SCALE_GIVEN) ms = tile_load(a.scale + row0 * dt, nitems, s_sc, tid);
        const int mx = tile_load(a.x + row0 * D, rows * D, s_x, tid);
        float* s_o = s_out + tile_store_offset(a.out + row0 * D);
        __syncthreads();

        for (int e = tid; e < rows * D; e += kBlock) {
            const int r = (int)fastdiv((uint32_t)e, a.div_D);
            const int c = e - r * D;
            if (!s_ist[c]) s_o[e - c + s_dst[c]] = s_x[mx + e - c + s_src[c]];
        }
        for (int i = tid; i < nitems; i += kBlock) {
            const int r = (int)fastdiv((uint32_t)i, a.div_dt);
            const int j = i - r * dt;
            const int col = s_tidx[j];
            const float xin = s_x[mx + r * D + s_src[col]];
            const float shift = s_p[mp + r * pc + j];
            float y, l;
            if (a.activation == NFA_SCALE_ADDITIVE) {
                // scale == 1: x*1 + shift, (x - shift)/1 and log(1) == 0 are exact
                y = a.inverse ? xin - shift : xin + shift;
                l = 0.0f;
            } else {
                const float sc = a.activation == NFA_SCALE_GIVEN
                                     ? s_sc[ms + i]
                                     : scale_of(s_p[mp + r * pc + dt + j], a.activation);
                if (a.inverse) affine_element<true>(xin, shift, sc, y, l);
                else affine_element<false>(xin, shift, sc, y, l);
            }
            s_o[r * D + s_dst[col]] = y;
            s_lad[i] = l;
        }
        __syncthreads();
        tile_store(a.out + row0 * D, rows * D, s_out, tid);
        const int wave = tid >> 6, lane = tid & 63;
        for (int r = wave; r < rows; r += kBlock / kWave) {
            float v = 0.0f;
            for (int m = lane; m < dt; m += kWave) v += s_lad[r * dt + m];
            v = wave_sum(v);
            if (lane == 0) a.lad[row0 + r] = a.accumulate ? a.lad[row0 + r] + v : v;
        }
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

// ------------------------------------------------------------------------------------------
// K2b: autoregressive affine, params [B, D, 2] interleaved (scale logit, shift).
// One wave per sample row segment; lanes stride over features, fixed-order reduction.
__global__ void __launch_bounds__(kBlock) affine_ar_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ params,
                                                           float* __restrict__ out,
                                                           float* __restrict__ lad, int64_t batch, int D,
                                                           int inverse) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / kWave);
    for (int64_t b = wave0; b < batch; b += nwaves) {
        float acc = 0.0f;
        for (int c = lane; c < D; c += kWave) {
            const float2 p = *reinterpret_cast<const float2*>(params + (b * D + c) * 2);
            const float sc = scale_of(p.x, NFA_SCALE_SOFTPLUS);
            const float xv = x[b * D + c];
            float y, l;
            if (inverse) affine_element<true>(xv, p.y, sc, y, l);
            else affine_element<false>(xv, p.y, sc, y, l);
            out[b * D + c] = y;
            acc += l;
        }
        acc = wave_sum(acc);
        if (lane == 0) lad[b] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// K4: out[b, c] = in[b, perm[c]] on 4-byte words.  R rows per workgroup through LDS so that both
// the global read and the global write are fully coalesced 16-byte accesses.
struct PermArgs {
    const float* x;
    const int64_t* perm;
    float* out;
    int32_t* status;
    int64_t batch;
    int D, R;
    FastDiv div_D;
    int off_out, off_idx;
};

__global__ void __launch_bounds__(kBlock) permute_cols_kernel(const PermArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_x = lds;
    float* s_out = lds + a.off_out;
    int* s_src = reinterpret_cast<int*>(lds + a.off_idx);
    const int tid = threadIdx.x, D = a.D;
    int bad = 0;
    for (int c = tid; c < D; c += kBlock) {
        const int64_t p = a.perm[c];
        if (p < 0 || p >= D) bad = NFA_STATUS_BAD_INDEX;
        s_src[c] = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
    }
    const int64_t num_tiles = (a.batch + a.R - 1) / a.R;
    for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * a.R;
        const int rows = (int)((a.batch - row0) < a.R ? (a.batch - row0) : a.R);
        const int mx = tile_load(a.x + row0 * D, rows * D, s_x, tid);
        float* s_o = s_out + tile_store_offset(a.out + row0 * D);
        __syncthreads();
        for (int e = tid; e < rows * D; e += kBlock) {
            const int r = (int)fastdiv((uint32_t)e, a.div_D);
            const int c = e - r * D;
            s_o[e] = s_x[mx + e - c + s_src[c]];
        }
        __syncthreads();
        tile_store(a.out + row0 * D, rows * D, s_out, tid);
    }
    if (bad && a.status) atomicOr(a.status, bad);
}

// ------------------------------------------------------------------------------------------
// K3 / normal epilogue: one wave per row, float4 loads when the row is 16-byte aligned.
template <bool NORMAL>
__global__ void __launch_bounds__(kBlock) rowsum_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ add,
                                                        float* __restrict__ out, int64_t rows,
                                                        int64_t cols, float log_z) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / kWave);
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float* row = x + r * cols;
        float acc = 0.0f;
        if ((cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(row) & 15) == 0)) {
            for (int64_t c = lane * 4; c < cols; c += kWave * 4) {
                const float4 q = *reinterpret_cast<const float4*>(row + c);
                if (NORMAL)
                    acc += ((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
                else
                    acc += ((q.x + q.y) + (q.z + q.w));
            }
        } else {
            for (int64_t c = lane; c < cols; c += kWave) {
                const float v = row[c];
                acc += NORMAL ? v * v : v;
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            if (NORMAL) {
                float lp = -0.5f * acc - log_z;  // distributions/normal.py:31-33
                if (add) lp = lp + add[r];       // flows/base.py:49
                out[r] = lp;
            } else {
                out[r] = acc;
            }
        }
    }
}

static dim3 wave_per_row_grid(int64_t rows) {
    int64_t blocks = (rows + (kBlock / kWave) - 1) / (kBlock / kWave);
    const int64_t cap = (int64_t)device_cu_count() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned)blocks);
}

constexpr int kMaxDynLdsMisc = 64 * 1024;

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_abi_version(void) { return NFA_ABI_VERSION; }
extern "C" const char* nfa_build_arch(void) { return "gfx950"; }
extern "C" int nfa_last_hip_error(void) { return g_last_hip_error; }
extern "C" const char* nfa_strerror(int code) {
    switch (code) {
        case NFA_OK: return "ok";
        case NFA_ERR_INVALID_ARGUMENT: return "invalid argument";
        case NFA_ERR_UNSUPPORTED: return "unsupported configuration for the fused kernel";
        case NFA_ERR_MIN_BIN_WIDTH: return "Minimal bin width too large for the number of bins";
        case NFA_ERR_MIN_BIN_HEIGHT: return "Minimal bin height too large for the number of bins";
        case NFA_ERR_HIP: return "HIP runtime error";
        default: return "unknown error";
    }
}

extern "C" int nfa_affine_coupling_f32(const float* inputs, const float* params, const float* scale,
                                       const int64_t* transform_idx, const int64_t* in_perm,
                                       const int64_t* out_scatter, float* outputs, float* logabsdet,
                                       int32_t* status, int64_t batch,
                                       int32_t features, int32_t num_transform,
                                       int32_t scale_activation, int32_t flags, void* stream) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET)) return NFA_ERR_INVALID_ARGUMENT;
    const int inverse = flags & NFA_FLAG_INVERSE;
    if (batch < 0 || features < 1 || num_transform < 0 || num_transform > features)
        return NFA_ERR_INVALID_ARGUMENT;
    if (scale_activation < NFA_SCALE_DEFAULT || scale_activation > NFA_SCALE_SOFTPLUS)
        return NFA_ERR_INVALID_ARGUMENT;
    if (batch == 0) return NFA_OK;
    if (!inputs || !outputs || !logabsdet || (num_transform > 0 && (!params || !transform_idx)))
        return NFA_ERR_INVALID_ARGUMENT;
    if (scale_activation == NFA_SCALE_GIVEN && num_transform > 0 && !scale) return NFA_ERR_INVALID_ARGUMENT;
    if (features > 65535) return NFA_ERR_UNSUPPORTED;
    AffineArgs a;
    const int D = features, dt = num_transform;
    a.pcols = scale_activation == NFA_SCALE_ADDITIVE ? dt : 2 * dt;
    int R = dt > 0 ? kBlock / dt : kBlock / (D < kBlock ? D : kBlock);
    if (R < 1) R = 1;
    if ((int64_t)R > batch) R = (int)batch;
    auto lds_floats = [&](int r) {
        int o = round_up4(r * a.pcols) + 4;
        a.off_sc = o;
        o += (scale_activation == NFA_SCALE_GIVEN ? round_up4(r * dt) + 4 : 0);
        a.off_x = o;
        o += round_up4(r * D) + 4;
        a.off_out = o;
        o += round_up4(r * D) + 4;
        a.off_lad = o;
        o += round_up4(r * dt);
        a.off_idx = o;
        o += dt + 2 * D + (D + 3) / 4;
        return o;
    };
    while (R > 1 && (size_t)lds_floats(R) * 4 > (size_t)kMaxDynLdsMisc) R >>= 1;
    const size_t lds = (size_t)lds_floats(R) * 4;
    if (lds > (size_t)kMaxDynLdsMisc || (int64_t)R * D >= 65536) return NFA_ERR_UNSUPPORTED;
    a.x = inputs;
    a.params = params;
    a.scale = scale;
    a.tidx = transform_idx;
    a.perm = in_perm;
    a.scatter = out_scatter;
    a.out = outputs;
    a.lad = logabsdet;
    a.status = status;
    a.batch = batch;
    a.D = D;
    a.dt = dt;
    a.R = R;
    a.activation = scale_activation;
    a.inverse = inverse;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    a.div_dt = make_fastdiv((uint32_t)(dt > 0 ? dt : 1));
    a.div_D = make_fastdiv((uint32_t)D);
    const int64_t tiles = (batch + R - 1) / R;
    int per_cu = (int)((size_t)(160 * 1024) / (lds + 256));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int64_t g = (int64_t)device_cu_count() * per_cu;
    if (g > tiles) g = tiles;
    hipLaunchKernelGGL(affine_coupling_kernel, dim3((unsigned)g), dim3(kBlock), lds,
                       (hipStream_t)stream, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_affine_autoregressive_f32(const float* inputs, const float* params, float* outputs,
                                             float* logabsdet, int64_t batch, int32_t features,
                                             int32_t inverse, void* stream) {
    if (batch < 0 || features < 1) return NFA_ERR_INVALID_ARGUMENT;
    if (batch == 0) return NFA_OK;
    if (!inputs || !params || !outputs || !logabsdet) return NFA_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(affine_ar_kernel, wave_per_row_grid(batch), dim3(kBlock), 0,
                       (hipStream_t)stream, inputs, params, outputs, logabsdet, batch, features, inverse);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_permute_cols_b32(const void* inputs, const int64_t* perm, void* outputs,
                                    int32_t* status, int64_t batch, int32_t features, void* stream) {
    if (batch < 0 || features < 1) return NFA_ERR_INVALID_ARGUMENT;
    if (batch == 0) return NFA_OK;
    if (!inputs || !perm || !outputs) return NFA_ERR_INVALID_ARGUMENT;
    if (features > 6000) return NFA_ERR_UNSUPPORTED;  // one row (x2) + index must fit in 64 KiB LDS
    PermArgs a;
    const int D = features;
    int R = (4 * kBlock) / D;  // ~4 words per lane
    if (R < 1) R = 1;
    if ((int64_t)R > batch) R = (int)batch;
    a.x = static_cast<const float*>(inputs);
    a.perm = perm;
    a.out = static_cast<float*>(outputs);
    a.status = status;
    a.batch = batch;
    a.D = D;
    a.R = R;
    a.div_D = make_fastdiv((uint32_t)D);
    a.off_out = round_up4(R * D) + 4;
    a.off_idx = a.off_out + round_up4(R * D) + 4;
    const size_t lds = (size_t)(a.off_idx + D) * 4;
    if (lds > (size_t)kMaxDynLdsMisc) return NFA_ERR_UNSUPPORTED;
    const int64_t tiles = (batch + R - 1) / R;
    int64_t g = (int64_t)device_cu_count() * 8;
    if (g > tiles) g = tiles;
    hipLaunchKernelGGL(permute_cols_kernel, dim3((unsigned)g), dim3(kBlock), lds, (hipStream_t)stream, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

// ------------------------------------------------------------------------------------------
// Sum of a vector in float64 + its length: the two numbers a rank contributes to the data-parallel
// log-likelihood.  Up to 64 workgroups write one partial sum each; the workgroup that draws the
// last ticket adds the partials in index order (the result does not depend on which one that is)
// and hands the ticket counter back at zero.
constexpr int kSumBlocks = 64;

__global__ void __launch_bounds__(kBlock) sum_count_kernel(const float* __restrict__ v, int64_t n,
                                                           double* __restrict__ out,
                                                           double* __restrict__ partial, unsigned* ticket) {
    __shared__ double part[kBlock];
    __shared__ unsigned drawn;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float v0 = v[i], v1 = v[i + stride], v2 = v[i + 2 * stride], v3 = v[i + 3 * stride];
        a0 += (double)v0;
        a1 += (double)v1;
        a2 += (double)v2;
        a3 += (double)v3;
    }
    for (; i < n; i += stride) a0 += (double)v[i];
    part[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    for (int step = kBlock / 2; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step) part[threadIdx.x] += part[threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partial[blockIdx.x], part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        drawn = atomicAdd(ticket, 1u);
    }
    __syncthreads();
    if (drawn != gridDim.x - 1) return;
    __threadfence();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (unsigned b = 0; b < gridDim.x; ++b)
            total += __hip_atomic_load(&partial[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out[0] = total;
        out[1] = (double)n;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

extern "C" size_t nfa_sum_count_workspace_bytes(void) { return (kSumBlocks + 1) * sizeof(double); }

extern "C" int nfa_sum_count_f64(const float* values, int64_t n, double* out, void* workspace, void* stream) {
    if (n < 0 || !out || !workspace || (n > 0 && !values)) return NFA_ERR_INVALID_ARGUMENT;
    int64_t blocks = (n + 4 * kBlock - 1) / (4 * kBlock);
    if (blocks > kSumBlocks) blocks = kSumBlocks;
    if (blocks < 1) blocks = 1;
    double* partial = reinterpret_cast<double*>(workspace);
    unsigned* ticket = reinterpret_cast<unsigned*>(partial + kSumBlocks);
    hipLaunchKernelGGL(sum_count_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, values, n,
                       out, partial, ticket);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_rowsum_f32(const float* x, float* out, int64_t rows, int64_t cols, void* stream) {
    if (rows < 0 || cols < 0) return NFA_ERR_INVALID_ARGUMENT;
    if (rows == 0) return NFA_OK;
    if (!out || (cols > 0 && !x)) return NFA_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL((rowsum_kernel<false>), wave_per_row_grid(rows), dim3(kBlock), 0,
                       (hipStream_t)stream, x, (const float*)nullptr, out, rows, cols, 0.0f);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_standard_normal_log_prob_f32(const float* z, const float* logabsdet, float* out,
                                                int64_t rows, int64_t cols, void* stream) {
    if (rows < 0 || cols < 1) return NFA_ERR_INVALID_ARGUMENT;
    if (rows == 0) return NFA_OK;
    if (!z || !out) return NFA_ERR_INVALID_ARGUMENT;
    const float log_z = (float)(0.5 * (double)cols * log(2.0 * 3.14159265358979323846));
    hipLaunchKernelGGL((rowsum_kernel<true>), wave_per_row_grid(rows), dim3(kBlock), 0,
                       (hipStream_t)stream, z, logabsdet, out, rows, cols, log_z);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

// ------------------------------------------------------------------------------------------
// torchutils.searchsorted (utils/torchutils.py:134-136): count of knots <= input, minus one, with eps
// added to the last knot.  HBM-bound: 4 (num_knots + 1) bytes read and 8 written per element.  Dense rows
// (row_stride == num_knots) come in as the contiguous image of a 256-row tile through LDS (coalesced
// 16-byte loads; a lane then reads its own row: stride num_knots words, conflict-free for odd counts --
// K + 1 knots of an even K); a shared row (row_stride == 0) is staged once per workgroup; other strides
// read their rows from global memory directly.  The ballot / shuffle form of a monotone search has no
// place here: the reference's count is defined for ANY knot row (monotone or not) and a count is what
// this kernel returns.
namespace nfa {
struct SearchArgs {
    const float* knots;
    int64_t stride;
    int nk;
    const float* x;
    int64_t* out;
    int64_t n;
    float eps;
    int T;       // rows per tile
    int dense;   // 1: rows are contiguous and a tile of them is staged through LDS
};

__global__ void __launch_bounds__(kBlock) searchsorted_kernel(const SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nk = a.nk;
    const int64_t num_tiles = (a.n + a.T - 1) / a.T;
    const bool dense = a.dense != 0, shared = a.stride == 0;
    if (shared) {
        for (int j = tid; j < nk; j += kBlock) lds[j] = a.knots[j];
        __syncthreads();
    }
    for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int64_t i0 = tile * a.T;
        const int cnt = (int)((a.n - i0) < a.T ? (a.n - i0) : a.T);
        int off = 0;
        if (dense) {
            off = tile_load(a.knots + i0 * nk, cnt * nk, lds, tid);
            __syncthreads();
        }
        for (int e = tid; e < cnt; e += kBlock) {
            const float x = a.x[i0 + e];
            int count = 0;
            if (dense || shared) {
                const float* row = shared ? lds : lds + off + e * nk;
                for (int j = 0; j < nk - 1; ++j) count += x >= row[j] ? 1 : 0;
                count += x >= row[nk - 1] + a.eps ? 1 : 0;
            } else {
                const float* row = a.knots + (i0 + e) * a.stride;
                for (int j = 0; j < nk - 1; ++j) count += x >= row[j] ? 1 : 0;
                count += x >= row[nk - 1] + a.eps ? 1 : 0;
            }
            a.out[i0 + e] = (int64_t)count - 1;
        }
        if (dense) __syncthreads();   // before the next tile overwrites the image
    }
}
}  // namespace nfa

extern "C" int nfa_searchsorted_f32(const float* bin_locations, int64_t row_stride, int32_t num_knots,
                                    const float* inputs, int64_t* bin_idx, int64_t n, double eps, void* stream) {
    if (n < 0 || num_knots < 1 || row_stride < 0) return NFA_ERR_INVALID_ARGUMENT;
    if (n == 0) return NFA_OK;
    if (!bin_locations || !inputs || !bin_idx) return NFA_ERR_INVALID_ARGUMENT;
    SearchArgs a;
    a.knots = bin_locations;
    a.stride = row_stride;
    a.nk = num_knots;
    a.x = inputs;
    a.out = bin_idx;
    a.n = n;
    a.eps = (float)eps;
    int T = kBlock;
    size_t lds = 16;
    a.dense = 0;
    if (row_stride == num_knots && num_knots <= 4096) {
        auto bytes = [&](int t) { return (size_t)(round_up4(t * num_knots) + 8) * 4; };
        while (T > 1 && bytes(T) > (size_t)kMaxDynLdsMisc) T >>= 1;
        if (bytes(T) <= (size_t)kMaxDynLdsMisc) {   // (a longer row is read in place)
            lds = bytes(T);
            a.dense = 1;
        } else {
            T = kBlock;
        }
    } else if (row_stride == 0) {
        lds = (size_t)round_up4(num_knots) * 4;
        if (lds > (size_t)kMaxDynLdsMisc) return NFA_ERR_UNSUPPORTED;
    }
    a.T = T;
    const int64_t tiles = (n + T - 1) / T;
    int64_t g = (int64_t)device_cu_count() * 8;
    if (g > tiles) g = tiles;
    hipLaunchKernelGGL(searchsorted_kernel, dim3((unsigned)g), dim3(kBlock), lds, (hipStream_t)stream, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
