// What the f16 matrix pipe sustains under the chip's power cap, by operand data.
// Every wave issues v_mfma_f32_32x32x16_f16 back to back from registers (four independent accumulators, two waves
// per SIMD, all 256 CUs) for a few seconds; operands are zeros, or Gaussian values (the weights scaled into
// [2^13, 2^14) like K8h's high pieces, activations ~ N(0, 1)) with fresh A operands for every instruction.
// Prints TFLOP/s per data kind; tools/mfma_power.py samples clock and power beside it.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o tools/bin/mfma_power_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512, 2) mfma_loop(const f16x8* a_in, const f16x8* b_in, float* out, int iters, int na) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 16 A fragments (rotated: every MFMA sees another one) and 4 B fragments per wave, all in registers
    f16x8 a[16], b[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = a_in[((size_t)((blockIdx.x * 8 + wave) * 16 + i) % na) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = b_in[((size_t)(blockIdx.x * 8 + wave) * 4 + i) * 64 + lane];
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i >> 2) & 3], acc[i & 3], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) s += acc[t][g];
    if (s == 1.2345f) out[0] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same rate of multiply-adds on v_mfma_f32_16x16x32_f16 (sixteen independent accumulators of four registers)
__global__ void __launch_bounds__(512, 2) mfma_loop_16(const f16x8* a_in, const f16x8* b_in, float* out, int iters, int na) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a[16], b[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = a_in[((size_t)((blockIdx.x * 8 + wave) * 16 + i) % na) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = b_in[((size_t)(blockIdx.x * 8 + wave) * 4 + i) * 64 + lane];
    f32x4 acc[16] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[(i + r) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) s += acc[t][g];
    if (s == 1.2345f) out[0] = s;
}

static float gauss() {
    const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = rand() / (double)RAND_MAX;
    return (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int blocks = 256, na = 4096;
    std::vector<_Float16> ha((size_t)na * 64 * 8), hb((size_t)blocks * 8 * 4 * 64 * 8);
    f16x8 *da, *db;
    float* dout;
    hipMalloc(&da, ha.size() * 2);
    hipMalloc(&db, hb.size() * 2);
    hipMalloc(&dout, 4);
    const char* kinds[] = {"zeros", "gaussian"};
    for (int kind = 0; kind < 2; ++kind) {
        srand(1);
        for (auto& v : ha) v = kind ? (_Float16)(gauss() * 3000.0f) : (_Float16)0.0f;   // weights x T
        for (auto& v : hb) v = kind ? (_Float16)gauss() : (_Float16)0.0f;               // activations
        hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        const int iters = 20000;
        for (int shape = 0; shape < 2; ++shape) {
            auto launch = [&](int n) {
                if (shape == 0) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, da, db, dout, n, na);
                else hipLaunchKernelGGL(mfma_loop_16, dim3(blocks), dim3(512), 0, 0, da, db, dout, n, na);
            };
            launch(100);
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            int launches = 0;
            double dt = 0;
            do {
                launch(iters);
                hipDeviceSynchronize();
                ++launches;
                dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            } while (dt < seconds);
            // (both loops issue the same number of multiply-adds per iteration: 16 x 32x32x16 = 32 x 16x16x32)
            const double flops = (double)launches * blocks * 8 * (double)iters * 16 * 2.0 * 32 * 32 * 16;
            printf("mfma_power_probe %s %s: %.1f TFLOP/s over %.1f s (%d launches)\n", kinds[kind],
                   shape == 0 ? "32x32x16" : "16x16x32", flops / dt / 1e12, dt, launches);
            fflush(stdout);
        }
    }
    return 0;
}
