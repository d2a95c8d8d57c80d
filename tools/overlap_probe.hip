// Probe: can the bf16 / f32 MFMA pipe and the VALU of one SIMD work at the same time on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o tools/bin/overlap_probe
// One workgroup per CU (LDS-limited), WAVES waves per workgroup.  Each wave runs a chain of MFMAs,
// a chain of VALU FMAs, or both interleaved in one instruction stream, as selected per wave by a
// role table; HW_ID is recorded so that the SIMD every wave ran on is known.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { R_IDLE = 0, R_MFMA = 1, R_VALU = 2, R_BOTH = 3, R_MFMA32 = 4, R_BOTH32 = 5, R_VALU_PRIO = 6, R_MFMA_PRIO = 7 };

struct Roles { int r[16]; };

__global__ void __launch_bounds__(1024) probe(Roles roles, int iters, float* out, unsigned* hwid,
                                              unsigned long long* span) {
    extern __shared__ float big[];  // forces one workgroup per CU
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = roles.r[wave];
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    float fa = 0.001f * lane, fb = 0.5f + 0.001f * lane;
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 0.01f * (lane + j);
    const float c = 0.999f, d = 0.001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (role == R_MFMA) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
    } else if (role == R_MFMA32) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc1, 0, 0, 0);
        }
    } else if (role == R_VALU) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], c, d);
        }
    } else if (role == R_VALU_PRIO) {
        __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], c, d);
        }
        __builtin_amdgcn_s_setprio(0);
    } else if (role == R_MFMA_PRIO) {
        __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    } else if (role == R_BOTH) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], c, d);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
#pragma unroll
            for (int j = 8; j < 16; ++j) v[j] = __builtin_fmaf(v[j], c, d);
        }
    } else if (role == R_BOTH32) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], c, d);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc1, 0, 0, 0);
#pragma unroll
            for (int j = 8; j < 16; ++j) v[j] = __builtin_fmaf(v[j], c, d);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j] + acc0[j] + acc1[j];
    if (s == 1.2345f) out[0] = s;
    if (lane == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + wave;
        hwid[w] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID
        span[w] = t1 - t0;
    }
}

static void run(const char* name, int waves, const std::vector<int>& roles, int iters) {
    Roles r; memset(&r, 0, sizeof r);
    for (int i = 0; i < waves; ++i) r.r[i] = roles[i];
    const int grid = 256;
    float* out; unsigned* hw; unsigned long long* sp;
    hipMalloc(&out, 64); hipMalloc(&hw, grid * 16 * 4); hipMalloc(&sp, grid * 16 * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) probe<<<grid, waves * 64, 100 * 1024>>>(r, iters, out, hw, sp);
    hipEventRecord(e0);
    probe<<<grid, waves * 64, 100 * 1024>>>(r, iters, out, hw, sp);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(waves); std::vector<unsigned long long> s(waves);
    hipMemcpy(h.data(), hw, waves * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s.data(), sp, waves * 8, hipMemcpyDeviceToHost);
    printf("%-34s %8.1f us | block 0:", name, ms * 1e3);
    for (int i = 0; i < waves; ++i)
        printf(" [w%d simd%u role%d %llu cyc/iter]", i, (h[i] >> 4) & 3, roles[i], s[i] / iters);
    printf("\n");
    hipFree(out); hipFree(hw); hipFree(sp);
}

int main() {
    const int it = 20000;
    // per iteration: MFMA roles issue 2 MFMAs (bf16: 2 x 8 passes = 64 cycles; f32: 2 x 16 = 128), VALU
    // roles 16 FMAs (64 cycles)
    run("1 wave/SIMD: bf16 MFMA", 4, {1, 1, 1, 1}, it);
    run("1 wave/SIMD: f32 MFMA", 4, {4, 4, 4, 4}, it);
    run("1 wave/SIMD: VALU", 4, {2, 2, 2, 2}, it);
    run("1 wave/SIMD: bf16 MFMA+VALU same wave", 4, {3, 3, 3, 3}, it);
    run("1 wave/SIMD: f32 MFMA+VALU same wave", 4, {5, 5, 5, 5}, it);
    run("2 waves/SIMD: bf16 MFMA | VALU", 8, {1, 1, 1, 1, 2, 2, 2, 2}, it);
    run("2 waves/SIMD: f32 MFMA | VALU", 8, {4, 4, 4, 4, 2, 2, 2, 2}, it);
    run("2/SIMD: bf16 MFMA | VALU prio3", 8, {1, 1, 1, 1, 6, 6, 6, 6}, it);
    run("2/SIMD: VALU prio3 | bf16 MFMA", 8, {6, 6, 6, 6, 1, 1, 1, 1}, it);
    run("2/SIMD: VALU | bf16 MFMA", 8, {2, 2, 2, 2, 1, 1, 1, 1}, it);
    run("2/SIMD: VALU | bf16 MFMA prio3", 8, {2, 2, 2, 2, 7, 7, 7, 7}, it);
    run("2 waves/SIMD: bf16 MFMA | bf16 MFMA", 8, {1, 1, 1, 1, 1, 1, 1, 1}, it);
    run("2 waves/SIMD: VALU | VALU", 8, {2, 2, 2, 2, 2, 2, 2, 2}, it);
    run("2 waves/SIMD: both | both (bf16)", 8, {3, 3, 3, 3, 3, 3, 3, 3}, it);
    return 0;
}
