// Probe: what does one k-step of K8h's hidden GEMMs cost, piece by piece?
//   hipcc --offload-arch=gfx950 -O3 tools/kstep_probe.hip -o tools/bin/kstep_probe
// A k-step = 12 x v_mfma_f32_32x32x16_f16 (4 output tiles x 3 products) whose A operands come from
// an 8 KB LDS stage (two ds_read_b128 per tile, requested one tile ahead, counted lgkmcnt), one
// LDS-DMA request per wave for a later stage, one workgroup barrier.  Variants switch the parts on
// one after the other; WAVES = 4 (one per SIMD) or 8 (two per SIMD), one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float vec4f __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kStage = 8192, kRing = 6;

// MODE bits: 1 = A operands from LDS (counted waits), 2 = barrier per k-step, 4 = LDS-DMA request per k-step,
//            8 = some VALU work per k-step (24 v_fma after the MFMAs), 16 = independent accumulators per product
//            32 = 64 v_fma per k-step instead of 24; 64 = staggered: the first wave of a SIMD does its VALU work
//            BEFORE its MFMAs, the second one AFTER them (so that one wave's VALU runs beside the other's MFMAs)
template <int MODE, int NW>
__global__ void __launch_bounds__(NW * 64, 2) probe(const vec4f* w, int ksteps, float* out, unsigned long long* span, unsigned* hwid) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    vec4f* ring = reinterpret_cast<vec4f*>(lds);
    for (int i = tid; i < kRing * kStage / 16; i += NW * 64) ring[i] = w[i];
    __syncthreads();
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    f32x16 acc2[4] = {{0}, {0}, {0}, {0}};
    f16x8 bh, bl;
    for (int j = 0; j < 8; ++j) { bh[j] = (_Float16)(0.01f * (lane + j)); bl[j] = (_Float16)(0.0001f * (lane - j)); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = lane + j;
    const unsigned base = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds + lane * 16;
    int slot = 0, fetch = 0;
    // which of the two waves of its SIMD this wave is: MODE & 128 -> by wave parity, else by wave < NW / 2
    const bool first_of_simd = (MODE & 128) ? (wave & 1) == 0 : wave < NW / 2;
    if (lane == 0 && blockIdx.x == 0) hwid[wave] = __builtin_amdgcn_s_getreg((15 << 11) | 4);   // HW_ID bits [15:0]
    vec4f fh = ring[lane], fl = ring[64 + lane];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int ks = 0; ks < ksteps; ++ks) {
        if (MODE & 4) {
            const int dst = slot >= 1 ? slot - 1 : kRing - 1;
            const char* src = reinterpret_cast<const char*>(w) + (size_t)fetch * kStage + tid * 16;
            char* d = reinterpret_cast<char*>(ring) + dst * kStage + __builtin_amdgcn_readfirstlane(wave) * 1024;
            for (int i = 0; i < 8 / NW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * NW * 1024),
                                                 (__attribute__((address_space(3))) void*)(d + i * NW * 1024), 16, 0, 0);
            fetch = fetch + 1 == 64 ? 0 : fetch + 1;
        }
        const unsigned cur = base + slot * kStage, nxt = base + (slot + 1 == kRing ? 0 : slot + 1) * kStage;
        constexpr int NV = (MODE & 32) ? 64 : 24;
        if ((MODE & 8) && (MODE & 64) && first_of_simd) {
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q & 7]) : "v"(v[(q + 1) & 7]));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            vec4f nh = fh, nl = fl;
            if (MODE & 1) {
                if (t < 3)
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=v"(nh), "=v"(nl) : "v"(cur + (t + 1) * 2048));
                else
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=v"(nh), "=v"(nl) : "v"(nxt));
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh), "+v"(fl));
            }
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 ah = __builtin_bit_cast(f16x8, fh), al = __builtin_bit_cast(f16x8, fl);
            if (MODE & 16) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            } else {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            fh = nh;
            fl = nl;
        }
        if ((MODE & 8) && !((MODE & 64) && first_of_simd)) {
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q & 7]) : "v"(v[(q + 1) & 7]));
        }
        if (MODE & 2) {
            if (MODE & 4) {
                if (NW == 8) asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        slot = slot + 1 == kRing ? 0 : slot + 1;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j] + acc2[0][j] + acc2[1][j] + acc2[2][j] + acc2[3][j];
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 1.2345f) out[0] = s;
    if (lane == 0) span[blockIdx.x * NW + wave] = t1 - t0;
}

template <int MODE, int NW>
static void run(const char* name, const vec4f* w) {
    const int grid = 256, ksteps = 2000;
    float* out; unsigned long long* sp; unsigned* hw;
    hipMalloc(&out, 64); hipMalloc(&sp, grid * 8 * 8); hipMalloc(&hw, 64);
    auto k = probe<MODE, NW>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<grid, NW * 64, 100 * 1024>>>(w, ksteps, out, sp, hw);
    hipEventRecord(e0);
    k<<<grid, NW * 64, 100 * 1024>>>(w, ksteps, out, sp, hw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> s(NW);
    hipMemcpy(s.data(), sp, NW * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (auto x : s) mx = x > mx ? x : mx;
    printf("%-62s %d waves/SIMD: %7.1f cycles per k-step (12 MFMAs; 384 = pipe time per wave)  [%.3f us per k-step]\n", name, NW / 4,
           (double)mx / ksteps, ms * 1e3 / ksteps);
    if (MODE & 64) {
        unsigned h[8];
        hipMemcpy(h, hw, NW * 4, hipMemcpyDeviceToHost);
        printf("    SIMD of waves 0..%d (HW_ID[5:4]):", NW - 1);
        for (int i = 0; i < NW; ++i) printf(" %u", (h[i] >> 4) & 3);
        printf("\n");
    }
    hipFree(out); hipFree(sp); hipFree(hw);
}

int main() {
    vec4f* w; hipMalloc(&w, 64 * kStage); hipMemset(w, 0x11, 64 * kStage);
    run<0, 4>("MFMAs only (A in registers)", w);
    run<16, 4>("MFMAs only, products on two accumulators", w);
    run<1, 4>("+ A from LDS, counted waits", w);
    run<3, 4>("+ barrier per k-step", w);
    run<7, 4>("+ LDS-DMA request per k-step", w);
    run<15, 4>("+ 24 VALU per k-step", w);
    run<0, 8>("MFMAs only (A in registers)", w);
    run<1, 8>("+ A from LDS, counted waits", w);
    run<3, 8>("+ barrier per k-step", w);
    run<7, 8>("+ LDS-DMA request per k-step", w);
    run<15, 8>("+ 24 VALU per k-step", w);
    run<31, 8>("same, products on two accumulators", w);
    run<15 + 64, 8>("24 VALU per k-step, staggered (wave 0-3 before, 4-7 after)", w);
    run<15 + 32, 8>("64 VALU per k-step, all after the MFMAs", w);
    run<15 + 32 + 64, 8>("64 VALU per k-step, staggered", w);
    run<15 + 32, 4>("64 VALU per k-step, all after the MFMAs", w);
    run<15 + 64 + 128, 8>("24 VALU per k-step, staggered by wave parity", w);
    run<15 + 32 + 64 + 128, 8>("64 VALU per k-step, staggered by wave parity", w);
    return 0;
}
