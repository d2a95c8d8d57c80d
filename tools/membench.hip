// Ceiling probes for the K1 memory pipeline on MI355X (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/bin/membench && tools/bin/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// (a) pure read: every lane streams float4, one atomic-free sink
template <int UNROLL>
__global__ void __launch_bounds__(256) read_only(const float4* __restrict__ src, size_t nvec, float* sink) {
    float acc = 0.f;
    size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; base < nvec; base += stride) {
        float4 r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { size_t v = base + (size_t)u * 256; r[u] = src[v < nvec ? v : nvec - 1]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += r[u].x + r[u].y + r[u].z + r[u].w;
    }
    if (acc == 12345.678f) *sink = acc;
}

// (b) tile pipeline like K1: block owns contiguous 23552-B tiles; load -> LDS -> barrier -> (touch) -> barrier
template <int BLOCK, int TILE_VEC>
__global__ void __launch_bounds__(BLOCK) tile_pipe(const float4* __restrict__ src, size_t ntiles, float* __restrict__ out) {
    __shared__ float4 lds[TILE_VEC + 4];
    constexpr int UNROLL = (TILE_VEC + BLOCK - 1) / BLOCK;
    float acc = 0.f;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const float4* g = src + t * TILE_VEC;
        float4 r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { int v = u * BLOCK + threadIdx.x; r[u] = g[v < TILE_VEC ? v : TILE_VEC - 1]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { int v = u * BLOCK + threadIdx.x; if (v < TILE_VEC) lds[v] = r[u]; }
        __syncthreads();
        const float* f = reinterpret_cast<const float*>(lds);
        acc += f[threadIdx.x * 23 % (TILE_VEC * 4)];
        __syncthreads();
    }
    out[(size_t)blockIdx.x * BLOCK + threadIdx.x] = acc;
}

// (c) same but prefetching the next tile into registers before touching the current one
template <int BLOCK, int TILE_VEC>
__global__ void __launch_bounds__(BLOCK) tile_pipe_prefetch(const float4* __restrict__ src, size_t ntiles, float* __restrict__ out) {
    __shared__ float4 lds[TILE_VEC + 4];
    constexpr int UNROLL = (TILE_VEC + BLOCK - 1) / BLOCK;
    float acc = 0.f;
    float4 r[UNROLL];
    size_t t = blockIdx.x;
    if (t < ntiles) {
        const float4* g = src + t * TILE_VEC;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { int v = u * BLOCK + threadIdx.x; r[u] = g[v < TILE_VEC ? v : TILE_VEC - 1]; }
    }
    for (; t < ntiles; t += gridDim.x) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { int v = u * BLOCK + threadIdx.x; if (v < TILE_VEC) lds[v] = r[u]; }
        __syncthreads();
        size_t tn = t + gridDim.x;
        if (tn < ntiles) {
            const float4* g = src + tn * TILE_VEC;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { int v = u * BLOCK + threadIdx.x; r[u] = g[v < TILE_VEC ? v : TILE_VEC - 1]; }
        }
        const float* f = reinterpret_cast<const float*>(lds);
        acc += f[threadIdx.x * 23 % (TILE_VEC * 4)];
        __syncthreads();
    }
    out[(size_t)blockIdx.x * BLOCK + threadIdx.x] = acc;
}

template <typename F>
float time_it(F launch, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch(i);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch(i);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main() {
    const size_t rows = 65536, tile_vec = 1472, ntiles = rows / 8;
    const size_t nvec = ntiles * tile_vec;           // 193 MB
    const int NBUF = 4;                               // rotate buffers: defeat the 256 MiB Infinity Cache
    float4* buf[NBUF]; float* out; float* sink;
    for (int i = 0; i < NBUF; ++i) { CK(hipMalloc(&buf[i], nvec * 16)); CK(hipMemset(buf[i], 1, nvec * 16)); }
    CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&sink, 4));
    const double gb = nvec * 16 / 1e9;
    for (int g : {1280, 2048, 4096, 8192}) {
        float us = time_it([&](int i) { read_only<4><<<g, 256>>>(buf[i % NBUF], nvec, sink); }, 20);
        printf("read_only<4>        grid %5d : %7.1f us  %6.0f GB/s\n", g, us, gb / us * 1e6);
        us = time_it([&](int i) { read_only<8><<<g, 256>>>(buf[i % NBUF], nvec, sink); }, 20);
        printf("read_only<8>        grid %5d : %7.1f us  %6.0f GB/s\n", g, us, gb / us * 1e6);
    }
    for (int g : {1280, 1536, 2048}) {
        float us = time_it([&](int i) { tile_pipe<256, 1472><<<g, 256>>>(buf[i % NBUF], ntiles, out); }, 20);
        printf("tile_pipe<256>      grid %5d : %7.1f us  %6.0f GB/s\n", g, us, gb / us * 1e6);
        us = time_it([&](int i) { tile_pipe_prefetch<256, 1472><<<g, 256>>>(buf[i % NBUF], ntiles, out); }, 20);
        printf("tile_pipe_prefetch  grid %5d : %7.1f us  %6.0f GB/s\n", g, us, gb / us * 1e6);
    }
    for (int g : {4096, 5120}) {
        float us = time_it([&](int i) { tile_pipe<64, 368><<<g, 64>>>(buf[i % NBUF], ntiles * 4, out); }, 20);
        printf("tile_pipe<64>       grid %5d : %7.1f us  %6.0f GB/s\n", g, us, gb / us * 1e6);
        us = time_it([&](int i) { tile_pipe_prefetch<64, 368><<<g, 64>>>(buf[i % NBUF], ntiles * 4, out); }, 20);
        printf("tile_pipe_pf<64>    grid %5d : %7.1f us  %6.0f GB/s\n", g, us, gb / us * 1e6);
    }
    return 0;
}
