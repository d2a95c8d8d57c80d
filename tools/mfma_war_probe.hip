// Probe: is a VALU write to a register that a just-issued MFMA reads as its A or B operand safe on
// gfx950 when a second wave on the same SIMD keeps the matrix pipe busy?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_war_probe.hip -o tools/bin/mfma_war_probe
// A "victim" wave issues  v_mfma_f32_32x32x16_f16 acc, a, b, 0  and then, GAP independent VALU
// instructions later, overwrites the registers of b (or a) with zeros -- all inside one asm statement,
// so the instruction stream is exactly as written.  The accumulator is compared with the product of
// the ORIGINAL operands.  Run alone (one wave per SIMD) and beside an "aggressor" wave that issues
// MFMAs back to back on the same SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uvec4 __attribute__((ext_vector_type(4)));

#define GAP_NOPS_0 ""
#define GAP_NOPS_1 "v_mov_b32 %[d0], %[d0]\n\t"
#define GAP_NOPS_2 GAP_NOPS_1 "v_mov_b32 %[d1], %[d1]\n\t"
#define GAP_NOPS_4 GAP_NOPS_2 GAP_NOPS_2
#define GAP_NOPS_8 GAP_NOPS_4 GAP_NOPS_4
#define GAP_NOPS_16 GAP_NOPS_8 GAP_NOPS_8
#define GAP_NOPS_32 GAP_NOPS_16 GAP_NOPS_16

// WHICH: 0 = overwrite B, 1 = overwrite A
#define VICTIM_BODY(GAPSTR, WHICH)                                                            \
    asm volatile("v_mfma_f32_32x32x16_f16 %[acc], %[a], %[b], 0\n\t" GAPSTR                  \
                 "v_mov_b32 %[k0], 0\n\tv_mov_b32 %[k1], 0\n\tv_mov_b32 %[k2], 0\n\tv_mov_b32 %[k3], 0\n\t" \
                 "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"                               \
                 : [acc] "=&v"(acc), [k0] "+v"(kw[0]), [k1] "+v"(kw[1]), [k2] "+v"(kw[2]), [k3] "+v"(kw[3]), \
                   [d0] "+v"(d0), [d1] "+v"(d1)                                                \
                 : [a] "v"(WHICH ? __builtin_bit_cast(f16x8, kw) : a), [b] "v"(WHICH ? b : __builtin_bit_cast(f16x8, kw)))

template <int GAP, int WHICH>
__global__ void __launch_bounds__(512) probe(int iters, int aggressor, unsigned* bad, unsigned* badlanes) {
    extern __shared__ float big[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(0.25f * ((lane * 7 + j * 3) % 13) - 1.0f);
        b[j] = (_Float16)(0.125f * ((lane * 5 + j) % 11) - 0.5f);
    }
    __syncthreads();
    if (wave >= 4) {
        if (!aggressor) return;
        f32x16 acc0 = {0}, acc1 = {0};
        for (int i = 0; i < iters * 6; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        }
        if (acc0[0] + acc1[0] == 1.2345f) bad[0] = 1;
        return;
    }
    // reference product with untouched operands (no contention issue: nothing overwrites)
    f32x16 ref = {0};
    ref = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, ref, 0, 0, 0);
    unsigned nbad = 0, lanes_lo = 0;
    float d0 = 1.0f, d1 = 2.0f;
    for (int i = 0; i < iters; ++i) {
        uvec4 kw = __builtin_bit_cast(uvec4, WHICH ? a : b);
        asm volatile("" : "+v"(kw));
        f32x16 acc;
        if (GAP == 0) { VICTIM_BODY(GAP_NOPS_0, WHICH); }
        else if (GAP == 1) { VICTIM_BODY(GAP_NOPS_1, WHICH); }
        else if (GAP == 2) { VICTIM_BODY(GAP_NOPS_2, WHICH); }
        else if (GAP == 4) { VICTIM_BODY(GAP_NOPS_4, WHICH); }
        else if (GAP == 8) { VICTIM_BODY(GAP_NOPS_8, WHICH); }
        else if (GAP == 16) { VICTIM_BODY(GAP_NOPS_16, WHICH); }
        else { VICTIM_BODY(GAP_NOPS_32, WHICH); }
        bool wrong = false;
        for (int q = 0; q < 16; ++q) wrong |= (acc[q] != ref[q]);
        if (wrong) {
            ++nbad;
            lanes_lo |= 1u;
        }
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(nbad != 0);
    if (nbad) atomicAdd(bad + 1, nbad);
    if (lane == 0 && blockIdx.x == 0) {
        badlanes[wave * 2] = (unsigned)m;
        badlanes[wave * 2 + 1] = (unsigned)(m >> 32);
    }
    if (d0 + d1 == 1.2345f) bad[0] = 2;
}

template <int GAP, int WHICH>
static void run(int aggressor) {
    unsigned *bad, *bl;
    hipMalloc(&bad, 16); hipMalloc(&bl, 64);
    hipMemset(bad, 0, 16); hipMemset(bl, 0, 64);
    auto k = probe<GAP, WHICH>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    k<<<256, 512, 100 * 1024>>>(2000, aggressor, bad, bl);
    hipDeviceSynchronize();
    unsigned h[4], hl[8];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hl, bl, 32, hipMemcpyDeviceToHost);
    printf("overwrite %s  gap %2d VALU  %s: wrong lane-results %9u   bad-lane mask (block 0, wave 0) %08x %08x\n",
           WHICH ? "A" : "B", GAP, aggressor ? "beside an MFMA wave" : "alone              ", h[1], hl[1], hl[0]);
    hipFree(bad); hipFree(bl);
}

int main() {
    run<0, 0>(0); run<0, 0>(1); run<1, 0>(1); run<2, 0>(1); run<4, 0>(1); run<8, 0>(1); run<16, 0>(1); run<32, 0>(1);
    run<0, 1>(0); run<0, 1>(1); run<1, 1>(1); run<2, 1>(1); run<4, 1>(1); run<8, 1>(1); run<16, 1>(1); run<32, 1>(1);
    return 0;
}
