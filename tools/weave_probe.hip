// Probe: which VALU instructions hide under a v_mfma_f32_32x32x16_f16 of the same wave on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/weave_probe.hip -o tools/bin/weave_probe
// One wave per SIMD (or two); stream = { MFMA ; FILL x filler } repeated; fillers of one kind:
//   0 v_fma_f32 v, v, s, s (independent)      1 v_fma_f32 v, v, v, v (3 VGPR sources, independent)
//   2 dependent chain v_fma_f32                3 v_exp_f32 (independent)
//   4 v_cmp_ge_f32 + v_cndmask_b32 pairs       5 v_cvt_pk_f16_f32
//   6 ds_read_b32 (one lgkmcnt wait per group) 7 v_add_f32 v, v, v (2 VGPR sources)
//   8 v_max3_f32                               9 v_rcp_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int FILL>
__global__ void __launch_bounds__(512) weave(int iters, float* out, unsigned long long* span) {
    extern __shared__ float big[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
    float v[16], w[16];
    for (int j = 0; j < 16; ++j) { v[j] = 0.01f * (lane + j) + 0.5f; w[j] = 1.0f + 0.001f * j; }
    big[threadIdx.x] = lane;
    const float c = 0.999f, d = 0.001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < FILL; ++j) {
                const int q = (m * FILL + j) % 16;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "s"(c), "v"(w[q]));
                else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(w[q]), "v"(w[(q + 1) & 15]));
                else if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(w[q]), "v"(w[(q + 1) & 15]));
                else if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[q]));
                else if (KIND == 4) {
                    if (j & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[q]) : "v"(w[q]) : );
                    else asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(v[q]), "v"(w[q]) : "vcc");
                } else if (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[q]) : "v"(w[q]));
                else if (KIND == 6) asm volatile("ds_read_b32 %0, %1" : "=v"(v[q]) : "v"(lane * 4 + q * 256));
                else if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[q]) : "v"(w[q]));
                else if (KIND == 8) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(w[q]), "v"(w[(q + 1) & 15]));
                else if (KIND == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[q]));
            }
            if (KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j] + acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j];
    if (s == 1.2345f) out[0] = s;
    if (lane == 0) span[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int FILL>
static double run(int waves) {
    const int grid = 256, iters = 2000;
    float* out; unsigned long long* sp;
    hipMalloc(&out, 64); hipMalloc(&sp, grid * 8 * 8);
    auto k = weave<KIND, FILL>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int i = 0; i < 2; ++i) k<<<grid, waves * 64, 100 * 1024>>>(iters, out, sp);
    hipDeviceSynchronize();
    std::vector<unsigned long long> s(waves);
    hipMemcpy(s.data(), sp, waves * 8, hipMemcpyDeviceToHost);
    hipFree(out); hipFree(sp);
    return (double)s[waves - 1] / iters / 8;
}

template <int KIND>
static void kind(const char* name) {
    printf("%-44s 1 wave/SIMD: F=2 %5.1f  F=4 %5.1f  F=6 %5.1f  F=8 %5.1f | 2 waves/SIMD (cycles per MFMA of the last wave / 2): F=4 %5.1f  F=6 %5.1f  F=8 %5.1f\n",
           name, run<KIND, 2>(4), run<KIND, 4>(4), run<KIND, 6>(4), run<KIND, 8>(4), run<KIND, 4>(8) / 2, run<KIND, 6>(8) / 2, run<KIND, 8>(8) / 2);
}

int main() {
    printf("cycles per MFMA (32x32x16 f16, 4 accumulators in rotation) with F fillers behind every MFMA\n");
    kind<0>("v_fma_f32 v, v, s, s");
    kind<1>("v_fma_f32 v, v, v, v");
    kind<2>("v_fma_f32 dependent chain");
    kind<3>("v_exp_f32");
    kind<4>("v_cmp_ge_f32 / v_cndmask_b32 pairs");
    kind<5>("v_cvt_pk_f16_f32");
    kind<6>("ds_read_b32 + one lgkmcnt(0)");
    kind<7>("v_add_f32 v, v, v");
    kind<8>("v_max3_f32");
    kind<9>("v_rcp_f32");
    return 0;
}
