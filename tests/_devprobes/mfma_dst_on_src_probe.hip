// Test infrastructure (round 4): reproducer for the hazard of DESIGN.md section 4 ("K8s ... a compiler / hardware hazard"):
// hipcc (ROCm 7.2) may allocate the RESULT of v_mfma_f32_16x16x32_f16 on the registers of the instruction's own A operand
// when that operand dies there -- `v_mfma_f32_16x16x32_f16 v[6:9], v[6:9], v[124:127], v[42:45]` -- and in K8s one wave
// in a few thousand then came out 1e-4 off.  Here every wave runs chains of MFMAs in both forms, written in asm so that
// the register assignment is exactly as stated:
//     overlap: D = the A fragment's registers          safe: D = registers of their own
// with the A fragment rebuilt by VALU instructions right in front of each MFMA (as the piece conversion does) and two
// waves per SIMD (512 threads per workgroup, one workgroup per CU, every CU busy).  Counts the lane results that differ
// from the same products accumulated in a third, independent chain with s_nop padding.  Prints one line per form:
//     mfma_dst_on_src <form>: <wrong> wrong lane-results of <total>
// The product never emits the overlap form (tests/test_host_logic.py::test_no_mfma_result_lands_on_its_own_operands);
// tests/test_gpu_probes.py requires the SAFE form to be clean and records what the overlap form does on this
// driver / compiler / chip, so that a change of behaviour is seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int FORM>   // 0 = overlap (D on A), 1 = safe (D elsewhere)
__global__ void __launch_bounds__(512) probe(int iters, const unsigned* seed, unsigned* wrong) {
    const int lane = threadIdx.x & 63;
    const unsigned s = seed[(blockIdx.x * 512 + threadIdx.x) & 4095];
    u32x4 b = {0x3c003800u ^ (s & 0x03ff03ffu), 0x38003c00u ^ ((s >> 3) & 0x03ff03ffu), 0x3a003a00u, 0x3c003c00u};   // f16 pairs ~ 0.5 .. 2
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, ref = {0.f, 0.f, 0.f, 0.f};
        u32x4 a0 = {0x3c003c00u + (unsigned)((it * 37 + lane) & 0xff), 0x38003a00u, 0x3c003800u + (unsigned)(it & 0x7f), 0x3a003c00u};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // the A fragment of this step, made by VALU work right in front of the MFMA that reads it
            u32x4 a, a2;
            asm volatile("v_add_u32 %0, %4, %8\n\tv_xor_b32 %1, %5, %9\n\tv_add_u32 %2, %6, %8\n\tv_xor_b32 %3, %7, %9"
                         : "=&v"(a.x), "=&v"(a.y), "=&v"(a.z), "=&v"(a.w)
                         : "v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w), "v"((unsigned)k), "v"((unsigned)(k << 16)));
            a2 = a;
            // reference chain: padded, result on registers of its own
            asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(ref) : "v"(a2), "v"(b));
            if (FORM == 0) {
                // result written over the A fragment's registers: %0 is both the destination and srcA
                f32x4 d = __builtin_bit_cast(f32x4, a);
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(acc));
                acc = d;
            } else {
                f32x4 d;
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(acc));
                acc = d;
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int j = 0; j < 4; ++j) bad += (__builtin_bit_cast(unsigned, acc[j]) != __builtin_bit_cast(unsigned, ref[j])) ? 1u : 0u;
    }
    if (bad) atomicAdd(wrong, bad);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount;
    std::vector<unsigned> h(4096);
    srand(7);
    for (auto& v : h) v = (unsigned)rand();
    unsigned *seed, *wrong;
    hipMalloc(&seed, 4096 * 4);
    hipMalloc(&wrong, 4);
    hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    int rc = 0;
    for (int form = 0; form < 2; ++form) {
        hipMemset(wrong, 0, 4);
        if (form == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), 0, 0, iters, seed, wrong);
        else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(512), 0, 0, iters, seed, wrong);
        unsigned w = 0;
        hipMemcpy(&w, wrong, 4, hipMemcpyDeviceToHost);
        printf("mfma_dst_on_src %s: %u wrong lane-results of %llu\n", form == 0 ? "overlap" : "safe", w,
               (unsigned long long)blocks * 512ull * iters * 4ull);
        if (form == 1 && w) rc = 1;
    }
    return rc;
}
