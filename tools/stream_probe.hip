// Probe (round 5): how fast can ONE CU pull a shared weight stream through LDS-DMA, as the whole-layer kernels do?
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/bin/stream_probe
//   tools/bin/stream_probe
// The small-batch floor of K8s (0.64 ms for any batch <= 16 384 rows) is one workgroup per CU streaming the flow's
// 21.5 MB of packed weights (1 344 stages of 16 KB) once: 0.47 us per stage = 35 GB/s per CU.  Is that the CU's limit
// for this access pattern, or the consumer's (MFMA chain of one wave per SIMD)?  This kernel keeps the stream and drops
// the consumer: NW waves per workgroup, ring of R slots of 16 KB, every stage requested by all waves
// (global_load_lds_dwordx4, 16 B per lane: the product's stream_request), per stage one counted s_waitcnt vmcnt + one
// workgroup barrier (the product's stream_advance), optionally `reads` ds_read_b128 per lane and stage (the fragment
// reads of a consumer: 16 per lane and stage in K8s).  Every workgroup reads the SAME bytes (weights), grid = G workgroups.
// Output: us per stage and GB/s per CU for NW in {4, 8}, R in {4, 7}, G in {1, 64, 128, 256}, reads in {0, 16}.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float vec4f __attribute__((ext_vector_type(4)));
constexpr int kStageBytes = 16384;

// a bandwidth hog for the second stream of the `verify` mode: read-modify-write sweeps over a large buffer
__global__ void hog_kernel(float* p, size_t n, int sweeps) {
    for (int s = 0; s < sweeps; ++s)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.0f;
}

template <int NW, int RING>
__global__ void __launch_bounds__(NW * 64) stream_kernel(const char* w, int num_stages, int passes, int reads, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    constexpr int kThreads = NW * 64;
    constexpr int PER = 16 / NW;   // requests per lane and stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    vec4f acc = {0, 0, 0, 0};
    int fetch = 0, slot = 0;
    auto request = [&](int dst_slot) {
        const char* stage = w + (size_t)fetch * kStageBytes;
        char* dst = ring + dst_slot * kStageBytes + wave * 1024;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stage + i * kThreads * 16 + tid * 16),
                                             (__attribute__((address_space(3))) void*)(dst + i * kThreads * 16), 16, 0, 0);
        fetch = fetch + 1 == num_stages ? 0 : fetch + 1;
    };
    for (int j = 0; j < RING - 1; ++j) request(j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long total = (long)num_stages * passes;
    for (long s = 0; s < total; ++s) {
        // the stage in `slot` is complete; request the stage RING - 1 ahead into the slot freed one stage ago
        request(slot == 0 ? RING - 1 : slot - 1);
        const unsigned base = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)ring + slot * kStageBytes + lane * 16;
        for (int r = 0; r < reads; ++r) {
            vec4f v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(base + (unsigned)(r & 15) * 1024u));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
            acc += v;
        }
        if (reads < 0) {
            // verify mode: every float of stage t holds (float)t; the stage in `slot` is complete by the protocol.  All 16
            // KiB-rows of the slot are checked (one 16-byte vector per lane and row); sink[1 + block] counts stale vectors.
            const float want = (float)(s % num_stages);
            int stale = 0;
            for (int r = 0; r < 16; ++r) {
                vec4f v;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + (unsigned)r * 1024u));
                stale += (v.x != want) + (v.y != want) + (v.z != want) + (v.w != want);
            }
            if (stale) atomicAdd(reinterpret_cast<int*>(sink) + 1 + blockIdx.x, stale);
        }
        // this wave's share of the NEXT stage has landed: all but the (RING - 2) youngest stages' requests
        if constexpr (PER * (RING - 2) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if constexpr (PER * (RING - 2) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (PER * (RING - 2) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (PER * (RING - 2) == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if constexpr (PER * (RING - 2) == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        slot = slot + 1 == RING ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[blockIdx.x] = acc.x;
}

template <int NW, int RING>
static void run(const char* w, int num_stages, int grid, int reads, float* sink) {
    const size_t lds = (size_t)RING * kStageBytes;
    hipFuncSetAttribute((const void*)stream_kernel<NW, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int passes = 4;
    // LDS request of 120 KB: one workgroup per CU whatever the ring depth
    const size_t lds_launch = lds < 120 * 1024 ? 120 * 1024 : lds;
    hipLaunchKernelGGL((stream_kernel<NW, RING>), dim3(grid), dim3(NW * 64), lds_launch, 0, w, num_stages, 1, reads, sink);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<NW, RING>), dim3(grid), dim3(NW * 64), lds_launch, 0, w, num_stages, passes, reads, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double stages = (double)num_stages * passes;
    const double us_per_stage = best * 1e3 / stages;
    printf("stream_probe waves=%d ring=%d grid=%3d reads=%2d: %.3f us per 16 KB stage, %.1f GB/s per CU, %.2f TB/s aggregate, one pass of %d stages %.3f ms\n",
           NW, RING, grid, reads, us_per_stage, kStageBytes / us_per_stage / 1e3, grid * kStageBytes / us_per_stage / 1e6, num_stages,
           best / passes);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

// verify mode: the stream's contents identify their stage; a second stream keeps HBM busy meanwhile
template <int NW, int RING>
static void verify(char* w, int num_stages, int grid, float* sink, bool hog, float* hogbuf, size_t hogn, hipStream_t side) {
    hipFuncSetAttribute((const void*)stream_kernel<NW, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    const size_t lds_launch = 120 * 1024;
    hipMemset(sink, 0, 4096 * sizeof(float));
    hipDeviceSynchronize();
    const int passes = 40;
    if (hog) hipLaunchKernelGGL(hog_kernel, dim3(2048), dim3(256), 0, side, hogbuf, hogn, 40);
    hipLaunchKernelGGL((stream_kernel<NW, RING>), dim3(grid), dim3(NW * 64), lds_launch, 0, w, num_stages, passes, -1, sink);
    hipDeviceSynchronize();
    std::vector<int> h(4096);
    hipMemcpy(h.data(), sink, 4096 * sizeof(int), hipMemcpyDeviceToHost);
    long stale = 0;
    int blocks = 0;
    for (int b = 0; b < grid; ++b) {
        stale += h[1 + b];
        blocks += h[1 + b] != 0;
    }
    printf("stream_probe verify waves=%d ring=%d grid=%3d %s: %ld stale 16-byte vectors in %d workgroups (of %.0f vectors checked)\n", NW, RING, grid,
           hog ? "under a bandwidth hog" : "quiet device         ", stale / 4, blocks, (double)grid * num_stages * passes * 64.0 * 16 * NW / NW);
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'v') {
        const int num_stages = 1344;
        char* w;
        float *sink, *hogbuf;
        const size_t hogn = (size_t)1 << 29;
        hipMalloc(&w, (size_t)num_stages * kStageBytes);
        hipMalloc(&sink, 4096 * sizeof(float));
        hipMalloc(&hogbuf, hogn * sizeof(float));
        hipMemset(hogbuf, 0, hogn * sizeof(float));
        std::vector<float> host((size_t)num_stages * kStageBytes / 4);
        for (int t = 0; t < num_stages; ++t)
            for (int i = 0; i < kStageBytes / 4; ++i) host[(size_t)t * (kStageBytes / 4) + i] = (float)t;
        hipMemcpy(w, host.data(), host.size() * 4, hipMemcpyHostToDevice);
        hipStream_t side;
        hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
        for (int rep = 0; rep < 3; ++rep)
            for (int hog = 0; hog < 2; ++hog) {
                verify<4, 4>(w, num_stages, 256, sink, hog, hogbuf, hogn, side);
                verify<8, 4>(w, num_stages, 256, sink, hog, hogbuf, hogn, side);
                verify<8, 7>(w, num_stages, 256, sink, hog, hogbuf, hogn, side);
                verify<4, 4>(w, num_stages, 128, sink, hog, hogbuf, hogn, side);
            }
        return 0;
    }
    const int num_stages = 1344;   // 32 layers x 42 stages: the bench flow's stream
    char* w;
    float* sink;
    hipMalloc(&w, (size_t)num_stages * kStageBytes);
    hipMalloc(&sink, 4096 * sizeof(float));
    hipMemset(w, 1, (size_t)num_stages * kStageBytes);
    for (int grid : {1, 64, 128, 256}) {
        for (int reads : {0, 16}) {
            run<4, 4>(w, num_stages, grid, reads, sink);
            run<4, 7>(w, num_stages, grid, reads, sink);
            run<8, 4>(w, num_stages, grid, reads, sink);
            run<8, 7>(w, num_stages, grid, reads, sink);
        }
    }
    hipFree(w);
    hipFree(sink);
    return 0;
}
