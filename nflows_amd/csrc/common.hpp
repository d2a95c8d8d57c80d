// Shared host/device helpers for libnflows_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nflows_amd.h"

namespace nfa {

constexpr int kBlock = 256;  // 4 wave64 per workgroup
constexpr int kWave = 64;

int set_hip_error(hipError_t e);  // records e (thread-local) and returns NFA_ERR_HIP
int device_cu_count();            // multiProcessorCount of the current device (cached)
// measurement aid (nfa_profile_*): event pair for the next layer-kernel launch, or nulls
void profile_next_launch(hipEvent_t* start, hipEvent_t* stop);
// nfa_last_layer_kernel: the launchers of the layer kernels leave the instance they chose (printf-style)
void note_layer_kernel(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define NFA_HIP_CHECK(expr)                                   \
    do {                                                      \
        hipError_t nfa_e_ = (expr);                           \
        if (nfa_e_ != hipSuccess) return nfa::set_hip_error(nfa_e_); \
    } while (0)

// hipFuncSetAttribute applies to the CURRENT device only: the opt-in to more than 64 KB of dynamic LDS is
// remembered per (kernel, device) -- `seen` is one device bit mask per kernel instance.
inline int raise_dynamic_lds(const void* kern, unsigned long long* seen, int bytes) {
    int dev = 0;
    NFA_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (*seen & bit)) return 0;
    NFA_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev < 64) *seen |= bit;
    return 0;
}

// ------------------------------------------------------------------------------------------
// Exact division of a small unsigned number by a runtime constant: q = n / d for n < 2^16,
// 1 <= d < 2^16, with magic = ceil(2^32 / d) computed on the host.
struct FastDiv {
    uint32_t d;
    uint32_t magic;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.magic = d <= 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) + d - 1) / d);
    return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, FastDiv f) {
    return f.d <= 1 ? n : __umulhi(n, f.magic);
}

// ------------------------------------------------------------------------------------------
// Coalesced global -> LDS copy of the float range src[0 .. count).  The LDS image is the
// 16-byte aligned window that contains the range, so every lane moves one aligned float4
// (global_load_dwordx4 + ds_write_b128); elements of the first/last float4 that fall outside
// the range are never read.  Returns the offset (0..3) of src[0] inside dst.
// dst must be 16-byte aligned and hold ceil((count+3)/4)*4 + 4 floats.
template <int BLOCK = kBlock, int UNROLL = 6>
__device__ __forceinline__ int tile_load(const float* __restrict__ src, int count, float* dst,
                                         int tid) {
    const int mis = (int)((reinterpret_cast<uintptr_t>(src) >> 2) & 3);
    const float* win = src - mis;  // 16-byte aligned
    const int nvec = (mis + count + 3) >> 2;
    if (mis == 0 && (count & 3) == 0) {
        // aligned fast path (workgroup-uniform branch): issue UNROLL independent 16-byte loads per
        // lane before the first LDS write, so a lane has UNROLL KiB-rows in flight
        const float4* g = reinterpret_cast<const float4*>(src);
        float4* l = reinterpret_cast<float4*>(dst);
        for (int base = 0; base < nvec; base += BLOCK * UNROLL) {
            float4 r[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                // clamped index: the load is unconditional (stays in registers, no divergence);
                // lanes past the end re-read the last vector and simply do not store it
                const int v = base + u * BLOCK + tid;
                r[u] = g[v < nvec ? v : nvec - 1];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int v = base + u * BLOCK + tid;
                if (v < nvec) l[v] = r[u];
            }
        }
        return 0;
    }
    for (int v = tid; v < nvec; v += BLOCK) {
        const int e0 = v * 4 - mis;  // index into src of this vector's first element
        if (e0 >= 0 && e0 + 4 <= count) {
            const float4 q = *reinterpret_cast<const float4*>(win + v * 4);
            *reinterpret_cast<float4*>(dst + v * 4) = q;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = e0 + c;
                if (e >= 0 && e < count) dst[v * 4 + c] = src[e];
            }
        }
    }
    return mis;
}

// LDS -> global copy of count floats; src[off + i] -> dst[i], where off = (address of dst / 4) & 3
// so that full float4 stores are 16-byte aligned on both sides.
__device__ __forceinline__ int tile_store_offset(const float* dst) {
    return (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);
}
template <int BLOCK = kBlock>
__device__ __forceinline__ void tile_store(float* __restrict__ dst, int count, const float* src,
                                           int tid) {
    const int mis = tile_store_offset(dst);
    float* win = dst - mis;
    const int nvec = (mis + count + 3) >> 2;
    for (int v = tid; v < nvec; v += BLOCK) {
        const int e0 = v * 4 - mis;
        if (e0 >= 0 && e0 + 4 <= count) {
            *reinterpret_cast<float4*>(win + v * 4) = *reinterpret_cast<const float4*>(src + v * 4);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = e0 + c;
                if (e >= 0 && e < count) dst[e] = src[v * 4 + c];
            }
        }
    }
}

// Sum over the 64 lanes of a wave, fixed order (deterministic); result valid in lane 0.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

inline int round_up4(int n) { return (n + 3) & ~3; }

}  // namespace nfa
#include "affine_math.hpp"   // scale activations and the per-element map of the affine layers (K2, K2b, K11)
namespace nfa {

}  // namespace nfa
