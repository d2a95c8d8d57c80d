// Probe for the dual-set plan: ONE wave per SIMD, per "region" (the span
// between two stage barriers) 24 dependent bf16 MFMAs whose A fragments come from LDS, next to an
// independent chunk of VALU work of the size of a spline chunk.  How long does a region take with
// only the MFMAs, only the VALU chunk, and both (hipcc free to interleave them)?
//   hipcc --offload-arch=gfx950 -O3 tools/region_probe.hip -o tools/bin/region_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float vec4f __attribute__((ext_vector_type(4)));

template <int MODE, int PREFETCH>  // MODE bit 0: MFMAs, bit 1: VALU chunk
__global__ void __launch_bounds__(256, 1) probe(const vec4f* w, float* out, int regions, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    vec4f* ring = reinterpret_cast<vec4f*>(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 768 * 2; i += 256) ring[i] = w[i];
    __syncthreads();
    f32x16 acc = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    bf16x8 b;
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)(0.01f * (lane + j));
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 0.01f * (lane + j) + 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < regions; ++r) {
        const vec4f* cur = ring + (r & 1) * 768 + lane;
        if (MODE < 4 && (MODE & 1)) {
            if (PREFETCH) {  // all 12 fragments of the region requested up front
                bf16x8 a[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) a[i] = __builtin_bit_cast(bf16x8, cur[i * 64]);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(0 * 4 + k4) * 64]);
                    const bf16x8 am = __builtin_bit_cast(bf16x8, cur[(1 * 4 + k4) * 64]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, cur[(2 * 4 + k4) * 64]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b, acc, 0, 0, 0);
                }
            }
        }
        if (MODE < 4 && (MODE & 2)) {  // ~130 VALU ops incl. 16 exps: the size of one spline chunk
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float t = v[j] * 0.7f - 0.3f;
                t = __builtin_amdgcn_exp2f(t);
                t = __builtin_fmaf(t, 0.5f, v[(j + 1) & 15]);
                t = __builtin_fmaf(t, t, 0.25f);
                t = t * 0.9f + 0.01f;
                t = __builtin_fmaf(t, 0.3f, -0.1f);
                t = t - v[(j + 5) & 15] * 0.001f;
                v[j] = t * 0.5f + 0.1f;
            }
        }
        if (MODE == 8 || MODE == 9) {
            // 24 MFMAs spread over four independent accumulators: round robin (8) or in four chains of
            // six dependent ones (9, the order of K8's k-major GEMM)
            bf16x8 a[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) a[i] = __builtin_bit_cast(bf16x8, cur[i * 64]);
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int t = MODE == 8 ? (i & 3) : (i / 6);
                if (t == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % 12], b, acc, 0, 0, 0);
                if (t == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % 12], b, acc1, 0, 0, 0);
                if (t == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % 12], b, acc2, 0, 0, 0);
                if (t == 3) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % 12], b, acc3, 0, 0, 0);
            }
        }
        if (MODE == 5 || MODE == 6 || MODE == 7) {
            // 16 dependent MFMAs alone (5) / each followed by one 8-op slice of the VALU chunk (6)
            bf16x8 a[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) a[i] = __builtin_bit_cast(bf16x8, cur[i * 64]);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (MODE == 7 && (j & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j % 12], b, acc1, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j % 12], b, acc, 0, 0, 0);
                if (MODE >= 6) {
                    __builtin_amdgcn_sched_barrier(0);
                    float t = v[j] * 0.7f - 0.3f;
                    t = __builtin_amdgcn_exp2f(t);
                    t = __builtin_fmaf(t, 0.5f, v[(j + 1) & 15]);
                    t = __builtin_fmaf(t, t, 0.25f);
                    t = t * 0.9f + 0.01f;
                    t = __builtin_fmaf(t, 0.3f, -0.1f);
                    t = t - v[(j + 5) & 15] * 0.001f;
                    t = t * 0.5f + 0.1f;
                    asm volatile("" : "+v"(t));  // (keeps the slices from being packed pairwise into v_pk_*)
                    v[j] = t;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (MODE == 4) {
            // both, hand-placed: fragments up front, then 1-2 MFMAs, a fence, one 8-op slice of the VALU
            // chunk, a fence ... (sched_barrier(0): nothing may be scheduled across)
            bf16x8 a[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) a[i] = __builtin_bit_cast(bf16x8, cur[i * 64]);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(3 * j / 2) % 12], b, acc, 0, 0, 0);
                if (j & 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(3 * j / 2 + 1) % 12], b, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                float t = v[j] * 0.7f - 0.3f;
                t = __builtin_amdgcn_exp2f(t);
                t = __builtin_fmaf(t, 0.5f, v[(j + 1) & 15]);
                t = __builtin_fmaf(t, t, 0.25f);
                t = t * 0.9f + 0.01f;
                t = __builtin_fmaf(t, 0.3f, -0.1f);
                t = t - v[(j + 5) & 15] * 0.001f;
                v[j] = t * 0.5f + 0.1f;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j] + acc[j] + acc1[j] + acc2[j] + acc3[j];
    if (s == 1.2345f) out[0] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = (t1 - t0) / regions;
}

template <int MODE, int PREFETCH>
static void run(const char* name, vec4f* w, float* out, unsigned long long* cyc) {
    const int regions = 4000;
    hipFuncSetAttribute((const void*)probe<MODE, PREFETCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, PREFETCH><<<256, 256, 100 * 1024>>>(w, out, regions, cyc);
    hipEventRecord(e0);
    probe<MODE, PREFETCH><<<256, 256, 100 * 1024>>>(w, out, regions, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s %7.1f us  %5llu cycles / region\n", name, ms * 1e3, c);
}

int main() {
    vec4f* w; float* out; unsigned long long* cyc;
    hipMalloc(&w, 768 * 2 * 16); hipMalloc(&out, 64); hipMalloc(&cyc, 8);
    hipMemset(w, 0, 768 * 2 * 16);
    run<1, 0>("24 MFMAs, fragments read per k-step", w, out, cyc);
    run<1, 1>("24 MFMAs, fragments requested up front", w, out, cyc);
    run<2, 0>("VALU chunk (~130 ops, 16 exp)", w, out, cyc);
    run<3, 0>("both, fragments per k-step", w, out, cyc);
    run<3, 1>("both, fragments up front", w, out, cyc);
    run<4, 0>("both, hand-placed with sched_barrier fences", w, out, cyc);
    run<5, 0>("16 dependent MFMAs", w, out, cyc);
    run<6, 0>("16 MFMAs, each followed by one VALU slice (fenced)", w, out, cyc);
    run<7, 0>("16 MFMAs on two alternating accumulators, each followed by one VALU slice", w, out, cyc);
    run<8, 0>("24 MFMAs, 4 accumulators round robin", w, out, cyc);
    run<9, 0>("24 MFMAs, 4 chains of 6 dependent", w, out, cyc);
    return 0;
}
