// Probe: can a wave in an MFMA-only burst give its idle issue slots to the VALU work of the other
// wave on its SIMD by padding the burst with s_nop?
//   hipcc --offload-arch=gfx950 -O3 tools/yield_probe.hip -o tools/bin/yield_probe
// Wave A (waves 0-3): MFMA, then PAD (nothing / s_nop 7 x k / s_sleep 1), repeated.
// Wave B (waves 4-7): independent v_fma_f32 only, a fixed amount of work.
// Reported: A's cycles per MFMA, B's total cycles next to A and alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int PAD>
__global__ void __launch_bounds__(512) probe(int iters, int b_alone, float* out, unsigned long long* span) {
    extern __shared__ float big[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
    float v[16], w[16];
    for (int j = 0; j < 16; ++j) { v[j] = 0.01f * (lane + j) + 0.5f; w[j] = 1.0f + 0.001f * j; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (!b_alone)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (PAD == 1) asm volatile("s_nop 7");
                    if (PAD == 2) asm volatile("s_nop 7\n\ts_nop 7");
                    if (PAD == 3) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
                    if (PAD == 4) asm volatile("s_sleep 1");
                    if (PAD == 5) asm volatile("s_nop 15\n\ts_nop 7");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < 48; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 15]) : "v"(w[q & 15]), "v"(w[(q + 1) & 15]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j] + acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j];
    if (s == 1.2345f) out[0] = s;
    if (lane == 0) span[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PAD>
static void run(const char* name) {
    const int grid = 256, iters = 2000;
    float* out; unsigned long long* sp;
    hipMalloc(&out, 64); hipMalloc(&sp, grid * 8 * 8);
    auto k = probe<PAD>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    std::vector<unsigned long long> s(8), s2(8);
    for (int i = 0; i < 2; ++i) k<<<grid, 512, 100 * 1024>>>(iters, 0, out, sp);
    hipDeviceSynchronize();
    hipMemcpy(s.data(), sp, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 2; ++i) k<<<grid, 512, 100 * 1024>>>(iters, 1, out, sp);
    hipDeviceSynchronize();
    hipMemcpy(s2.data(), sp, 64, hipMemcpyDeviceToHost);
    printf("%-28s A: %6.1f cycles per MFMA | B (48 v_fma per iteration = 6 per MFMA of A): %7.1f cycles per iteration beside A, %7.1f alone\n",
           name, (double)s[0] / iters / 8, (double)s[4] / iters, (double)s2[4] / iters);
    hipFree(out); hipFree(sp);
}

int main() {
    run<0>("A: back-to-back MFMAs");
    run<1>("A: MFMA + s_nop 7");
    run<2>("A: MFMA + 2 x s_nop 7");
    run<3>("A: MFMA + 3 x s_nop 7");
    run<5>("A: MFMA + s_nop 15, s_nop 7");
    run<4>("A: MFMA + s_sleep 1");
    return 0;
}
