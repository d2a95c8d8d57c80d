// Probe: fp32-accurate GEMM (K=128, N=768) from split-bf16 MFMA: every fp32 operand is written
// as the exact sum of three bf16 numbers (hi + mid + lo), the 9 (or 6 largest) cross products
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Measures accuracy vs a double reference
// and time, next to the fp32 MFMA (v_mfma_f32_32x32x2_f32) version.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int H = 128, NT = 24, N = NT * 32;

__host__ __device__ inline unsigned short f2bf(float x) {  // round to nearest even
    unsigned u; memcpy(&u, &x, 4);
    unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(r >> 16);
}
__host__ __device__ inline float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

// Bp layout: [piece(3)][ntile][kb(8)][lane(64)][8 bf16]: lane l, element j = piece of W[ntile*32 + (l&31)][kb*16 + (l>>5)*8 + j]
template <int TERMS>
__global__ void __launch_bounds__(256) gemm_bf16x(const float* __restrict__ h, const bf16x8* __restrict__ Bp,
                                                 float* __restrict__ out, int rows, int write_out) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const int half = lane >> 5, r = lane & 31;
    for (int tile = wave_global; tile * 32 < rows; tile += nwaves) {
        const int row0 = tile * 32;
        bf16x8 a[3][8];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const float* hp = h + (size_t)(row0 + r) * H + kb * 16 + half * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = hp[j];
                const unsigned short xh = f2bf(x);
                const float r1 = x - bf2f(xh);
                const unsigned short xm = f2bf(r1);
                const float r2 = r1 - bf2f(xm);
                a[0][kb][j] = (short)xh; a[1][kb][j] = (short)xm; a[2][kb][j] = (short)f2bf(r2);
            }
        }
        float sink = 0.f;
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc = {0};
            const bf16x8* bp = Bp + ((size_t)nt * 8) * 64 + lane;
            const size_t piece = (size_t)NT * 8 * 64;
            bf16x8 b[3][8];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) b[p][kb] = bp[p * piece + kb * 64];
            // smallest terms first
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (TERMS == 9) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][kb], b[2][kb], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][kb], b[1][kb], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][kb], b[2][kb], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][kb], b[0][kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][kb], b[2][kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][kb], b[1][kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][kb], b[0][kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][kb], b[1][kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][kb], b[0][kb], acc, 0, 0, 0);
            }
            if (write_out) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = (g & 3) + 8 * (g >> 2) + 4 * half;
                    out[(size_t)(row0 + row) * N + nt * 32 + r] = acc[g];
                }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) sink += acc[g];
            }
        }
        if (!write_out && sink == 1.2345f) out[0] = sink;
    }
}

int main() {
    const int B = 65536;
    std::vector<float> hh((size_t)B * H), W((size_t)N * H);
    srand(1);
    for (auto& v : hh) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    std::vector<unsigned short> Bp((size_t)3 * NT * 8 * 64 * 8);
    for (int nt = 0; nt < NT; ++nt) for (int kb = 0; kb < 8; ++kb) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const float x = W[(size_t)(nt * 32 + (l & 31)) * H + kb * 16 + (l >> 5) * 8 + j];
        const unsigned short xh = f2bf(x); const float r1 = x - bf2f(xh);
        const unsigned short xm = f2bf(r1); const float r2 = r1 - bf2f(xm);
        const size_t idx = (((size_t)nt * 8 + kb) * 64 + l) * 8 + j, piece = (size_t)NT * 8 * 64 * 8;
        Bp[idx] = xh; Bp[piece + idx] = xm; Bp[2 * piece + idx] = f2bf(r2);
    }
    float *dh, *dout; bf16x8* dB;
    hipMalloc(&dh, hh.size() * 4); hipMalloc(&dB, Bp.size() * 2); hipMalloc(&dout, (size_t)B * N * 4);
    hipMemcpy(dh, hh.data(), hh.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> o((size_t)64 * N);
    for (int terms : {6, 9}) {
        if (terms == 6) gemm_bf16x<6><<<512, 256>>>(dh, dB, dout, B, 1); else gemm_bf16x<9><<<512, 256>>>(dh, dB, dout, B, 1);
        hipDeviceSynchronize();
        hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxerr32 = 0;
        for (int r = 0; r < 64; ++r) for (int c = 0; c < N; ++c) {
            double s = 0; float s32 = 0.f;
            for (int k = 0; k < H; ++k) { s += (double)hh[(size_t)r * H + k] * W[(size_t)c * H + k]; s32 = fmaf(hh[(size_t)r * H + k], W[(size_t)c * H + k], s32); }
            maxerr = fmax(maxerr, fabs(s - o[(size_t)r * N + c])); maxerr32 = fmax(maxerr32, fabs(s - (double)s32));
        }
        printf("bf16x%d: max |err| vs double %.3e   (sequential fp32 fma chain: %.3e)\n", terms, maxerr, maxerr32);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int terms : {6, 9}) for (int grid : {256, 512, 1024}) {
        auto launch = [&]() { if (terms == 6) gemm_bf16x<6><<<grid, 256>>>(dh, dB, dout, B, 0); else gemm_bf16x<9><<<grid, 256>>>(dh, dB, dout, B, 0); };
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("bf16x%d grid %4d: %7.1f us  (fp32-equivalent %.1f TFLOP/s)\n", terms, grid, ms / 10 * 1e3, 2.0 * B * H * N / (ms / 10 * 1e3) / 1e6);
    }
    return 0;
}
