// Feasibility probe for fusing the conditioner's final Linear(128 -> 736) into the spline kernel:
// fp32 MFMA (v_mfma_f32_32x32x2_f32), A (activations) resident in VGPRs, B (weights) streamed
// from a pre-packed, fully coalesced layout, K = 128, N = 768 (32 features x 24, padded).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float vec4 __attribute__((ext_vector_type(4)));
constexpr int H = 128, NT = 24 /* N tiles of 32 */, N = NT * 32;

// Wp layout: [ntile][j4 (16)][lane (64)][4]: lane l, element (j4*4+q) = W[ntile*32 + (l&31)][(l>>5)*64 + j4*4 + q]
__global__ void __launch_bounds__(256) gemm_probe(const float* __restrict__ h, const float* __restrict__ Wp,
                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                 int rows, int write_out) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const int half = lane >> 5, r = lane & 31;
    for (int tile = wave_global; tile * 32 < rows; tile += nwaves) {
        const int row0 = tile * 32;
        vec4 a[16];
        const vec4* hp = reinterpret_cast<const vec4*>(h + (size_t)(row0 + r) * H + half * 64);
#pragma unroll
        for (int j4 = 0; j4 < 16; ++j4) a[j4] = hp[j4];
        float sink = 0.f;
        vec4 b[16];
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc = {0};
            const vec4* wp = reinterpret_cast<const vec4*>(Wp) + ((size_t)nt * 16) * 64 + lane;
            static_assert(true, "");
            if (write_out != 2 || nt == 0) {
#pragma unroll
            for (int j4 = 0; j4 < 16; ++j4) b[j4] = wp[j4 * 64];
            }
#pragma unroll
            for (int j4 = 0; j4 < 16; ++j4) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j4].x, b[j4].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j4].y, b[j4].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j4].z, b[j4].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j4].w, b[j4].w, acc, 0, 0, 0);
            }
            const float bv = bias[nt * 32 + r];
            if (write_out == 1) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = (g & 3) + 8 * (g >> 2) + 4 * half;
                    out[(size_t)(row0 + row) * N + nt * 32 + r] = acc[g] + bv;
                }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) sink += acc[g] + bv;
            }
        }
        if (!write_out && sink == 1.2345f) out[0] = sink;
    }
}

int main() {
    const int B = 65536;
    std::vector<float> hh((size_t)B * H), W((size_t)N * H), bias(N), Wp((size_t)N * H);
    srand(1);
    for (auto& v : hh) v = (rand() / (float)RAND_MAX - 0.5f);
    for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (auto& v : bias) v = (rand() / (float)RAND_MAX - 0.5f);
    for (int nt = 0; nt < NT; ++nt)
        for (int j4 = 0; j4 < 16; ++j4)
            for (int l = 0; l < 64; ++l)
                for (int q = 0; q < 4; ++q)
                    Wp[(((size_t)nt * 16 + j4) * 64 + l) * 4 + q] = W[(size_t)(nt * 32 + (l & 31)) * H + (l >> 5) * 64 + j4 * 4 + q];
    float *dh, *dW, *db, *dout;
    hipMalloc(&dh, hh.size() * 4); hipMalloc(&dW, Wp.size() * 4); hipMalloc(&db, N * 4); hipMalloc(&dout, (size_t)B * N * 4);
    hipMemcpy(dh, hh.data(), hh.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, Wp.data(), Wp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bias.data(), N * 4, hipMemcpyHostToDevice);
    // correctness on a few rows
    gemm_probe<<<512, 256>>>(dh, dW, db, dout, B, 1);
    hipDeviceSynchronize();
    std::vector<float> o((size_t)64 * N);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int r = 0; r < 64; ++r)
        for (int c = 0; c < N; ++c) {
            double s = bias[c];
            for (int k = 0; k < H; ++k) s += (double)hh[(size_t)r * H + k] * W[(size_t)c * H + k];
            maxerr = fmax(maxerr, fabs(s - o[(size_t)r * N + c]));
        }
    printf("max |err| vs double reference on 64 rows: %.3e\n", maxerr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wo : {0, 2})
        for (int grid : {256, 512, 1024}) {
            for (int i = 0; i < 3; ++i) gemm_probe<<<grid, 256>>>(dh, dW, db, dout, B, wo);
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) gemm_probe<<<grid, 256>>>(dh, dW, db, dout, B, wo);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double us = ms / 10 * 1e3;
            printf("write_out=%d grid %4d: %7.1f us  %6.1f TFLOP/s (useful 736/768: %.1f)\n", wo, grid, us,
                   2.0 * B * H * N / us / 1e6, 2.0 * B * H * 736 / us / 1e6);
        }
    return 0;
}
