// Round 6: accuracy of an fp32 GEMM (K = 128, the conditioner's hidden Linears) against float64 for every split-operand
// scheme the whole-layer kernels use or could use, next to sequential and multi-accumulator fp32 fma chains:
//   bf16 x 3: 6 products (K8, K11 until round 5), 9 products, 6 products with the leading product in an accumulator of its own
//   f16 x 2:  3 products (K8h)
//   f16 x 3:  5 products (K8x), 5 products with the leading product in an accumulator of its own, 6 products (+ lo lo)
// activations relu-like (half zeros) or Gaussian, at several magnitudes; f16 pieces of activations at scale S, weights
// at scale T = 2^12.
//   hipcc --offload-arch=gfx950 -O3 tools/gemm_numerics_probe.hip -o tools/bin/gemm_numerics_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int H = 128, NT = 4, N = NT * 32;

__host__ __device__ inline unsigned short f2bf(float x) {
    unsigned u; memcpy(&u, &x, 4);
    unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(r >> 16);
}
__host__ __device__ inline float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

enum { BF6 = 0, BF9 = 1, BF6_SPLIT = 2, F2_3 = 3, F3_5 = 4, F3_5_SPLIT = 5, F3_6 = 6 };

// W pieces: [piece][ntile][kb(8)][lane][8]; lane l, element j = W[ntile*32 + (l&31)][kb*16 + (l>>5)*8 + j]
template <int MODE>
__global__ void __launch_bounds__(256) gemm_split(const float* __restrict__ h, const void* __restrict__ Wp,
                                                 float* __restrict__ out, int rows, float act_scale, float out_scale) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const int half = lane >> 5, r = lane & 31;
    constexpr bool BF = MODE <= BF6_SPLIT;
    for (int tile = wave_global; tile * 32 < rows; tile += nwaves) {
        const int row0 = tile * 32;
        bf16x8 a3[3][8];
        f16x8 a2[3][8];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const float* hp = h + (size_t)(row0 + r) * H + kb * 16 + half * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = hp[j] * act_scale;
                if (BF) {
                    const unsigned short xh = f2bf(x);
                    const float r1 = x - bf2f(xh);
                    const unsigned short xm = f2bf(r1);
                    const float r2 = r1 - bf2f(xm);
                    a3[0][kb][j] = (short)xh; a3[1][kb][j] = (short)xm; a3[2][kb][j] = (short)f2bf(r2);
                } else {
                    const _Float16 xh = (_Float16)x;
                    const float t = x - (float)xh;
                    const _Float16 xl = (_Float16)t;
                    a2[0][kb][j] = xh;
                    a2[1][kb][j] = xl;
                    a2[2][kb][j] = (_Float16)(t - (float)xl);
                }
            }
        }
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc = {0}, small = {0};
            const size_t piece = (size_t)NT * 8 * 64;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (BF) {
                    const bf16x8* bp = reinterpret_cast<const bf16x8*>(Wp) + ((size_t)nt * 8 + kb) * 64 + lane;
                    const bf16x8 b0 = bp[0], b1 = bp[piece], b2 = bp[2 * piece];
#define M(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
                    if (MODE == BF9) { M(a3[2][kb], b2, acc); M(a3[1][kb], b2, acc); M(a3[2][kb], b1, acc); }
                    if (MODE == BF6_SPLIT) {
                        M(a3[2][kb], b0, small); M(a3[0][kb], b2, small); M(a3[1][kb], b1, small); M(a3[1][kb], b0, small); M(a3[0][kb], b1, small);
                        M(a3[0][kb], b0, acc);
                    } else {
                        M(a3[2][kb], b0, acc); M(a3[0][kb], b2, acc); M(a3[1][kb], b1, acc); M(a3[1][kb], b0, acc); M(a3[0][kb], b1, acc);
                        M(a3[0][kb], b0, acc);
                    }
#undef M
                } else {
                    const f16x8* bp = reinterpret_cast<const f16x8*>(Wp) + ((size_t)nt * 8 + kb) * 64 + lane;
                    const f16x8 b0 = bp[0], b1 = bp[piece], b2 = bp[2 * piece];
#define M(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
                    if (MODE == F2_3) { M(a2[1][kb], b0, acc); M(a2[0][kb], b1, acc); M(a2[0][kb], b0, acc); }
                    if (MODE == F3_5 || MODE == F3_6) {
                        if (MODE == F3_6) M(a2[1][kb], b1, acc);
                        M(a2[2][kb], b0, acc); M(a2[0][kb], b2, acc); M(a2[1][kb], b0, acc); M(a2[0][kb], b1, acc); M(a2[0][kb], b0, acc);
                    }
                    if (MODE == F3_5_SPLIT) {
                        M(a2[2][kb], b0, small); M(a2[0][kb], b2, small); M(a2[1][kb], b0, small); M(a2[0][kb], b1, small);
                        M(a2[0][kb], b0, acc);
                    }
#undef M
                }
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = (g & 3) + 8 * (g >> 2) + 4 * half;
                out[(size_t)(row0 + row) * N + nt * 32 + r] = (acc[g] + small[g]) * out_scale;
            }
        }
    }
}

template <int MODE>
static void launch(const float* dh, const void* dW, float* dout, int B, float S, float out_scale) {
    gemm_split<MODE><<<64, 256>>>(dh, dW, dout, B, S, out_scale);
}

int main() {
    const int B = 4096;
    std::vector<float> W((size_t)N * H);
    srand(1);
    for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.176f;   // U(-1/sqrt(128), 1/sqrt(128))
    float *dh, *dout; void* dW;
    hipMalloc(&dh, (size_t)B * H * 4); hipMalloc(&dW, (size_t)3 * N * H * 2); hipMalloc(&dout, (size_t)B * N * 4);
    std::vector<float> o((size_t)B * N);
    // weights: bf16 triples and f16 triples (x T = 2^12)
    const size_t piece = (size_t)NT * 8 * 64 * 8;
    std::vector<unsigned short> Pb(3 * piece);
    std::vector<_Float16> Pf(3 * piece);
    const float T = ldexpf(1.0f, 12);
    for (int nt = 0; nt < NT; ++nt) for (int kb = 0; kb < 8; ++kb) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const float x = W[(size_t)(nt * 32 + (l & 31)) * H + kb * 16 + (l >> 5) * 8 + j];
        const size_t idx = (((size_t)nt * 8 + kb) * 64 + l) * 8 + j;
        const unsigned short xh = f2bf(x); const float r1 = x - bf2f(xh);
        const unsigned short xm = f2bf(r1); const float r2 = r1 - bf2f(xm);
        Pb[idx] = xh; Pb[piece + idx] = xm; Pb[2 * piece + idx] = f2bf(r2);
        const float y = x * T;
        const _Float16 yh = (_Float16)y; const float t = y - (float)yh; const _Float16 yl = (_Float16)t;
        Pf[idx] = yh; Pf[piece + idx] = yl; Pf[2 * piece + idx] = (_Float16)(t - (float)yl);
    }
    for (int gaussian = 0; gaussian < 2; ++gaussian)
    for (float mag : {1.0f, 1e-2f, 30.0f}) {
        std::vector<float> hh((size_t)B * H);
        srand(7);
        for (auto& v : hh) {
            const float u = rand() / (float)RAND_MAX, w = rand() / (float)RAND_MAX, w2 = rand() / (float)RAND_MAX;
            if (gaussian) v = mag * sqrtf(-2.f * logf(w + 1e-9f)) * cosf(6.2831853f * w2);
            else v = (u < 0.5f) ? 0.f : mag * 2.f * w * w;   // relu-like: half zeros
        }
        hipMemcpy(dh, hh.data(), hh.size() * 4, hipMemcpyHostToDevice);
        std::vector<double> ref((size_t)B * N);
        std::vector<float> c1((size_t)B * N), c2((size_t)B * N), c4((size_t)B * N), c16((size_t)B * N);
        for (int r = 0; r < B; ++r) for (int c = 0; c < N; ++c) {
            double s = 0; float s1 = 0.f, s2[2] = {0, 0}, s4[4] = {0, 0, 0, 0}, s16[16] = {0};
            for (int k = 0; k < H; ++k) {
                const float a = hh[(size_t)r * H + k], w = W[(size_t)c * H + k];
                s += (double)a * w; s1 = fmaf(a, w, s1); s2[k & 1] = fmaf(a, w, s2[k & 1]); s4[k & 3] = fmaf(a, w, s4[k & 3]);
                s16[k & 15] = fmaf(a, w, s16[k & 15]);
            }
            float t16 = 0; for (int i = 8; i >= 1; i >>= 1) for (int j = 0; j < i; ++j) s16[j] += s16[j + i]; t16 = s16[0];
            ref[(size_t)r * N + c] = s; c1[(size_t)r * N + c] = s1; c2[(size_t)r * N + c] = s2[0] + s2[1];
            c4[(size_t)r * N + c] = (s4[0] + s4[1]) + (s4[2] + s4[3]); c16[(size_t)r * N + c] = t16;
        }
        auto report = [&](const char* name, const float* got) {
            double mx = 0, sq = 0, mean = 0, mref = 0;
            for (size_t i = 0; i < ref.size(); ++i) { const double e = fabs(ref[i] - got[i]); mx = fmax(mx, e); sq += e * e; mean += e; mref = fmax(mref, fabs(ref[i])); }
            printf("  %s x%-5g %-44s max %.3e  rms %.3e  mean %.3e   (max|ref| %.3g)\n", gaussian ? "gauss" : "relu ", mag, name, mx, sqrt(sq / ref.size()),
                   mean / ref.size(), mref);
        };
        report("fp32 fma chain (1 accumulator)", c1.data());
        report("fp32 fma, 2 accumulators over k", c2.data());
        report("fp32 fma, 4 accumulators over k", c4.data());
        report("fp32 fma, 16 accumulators over k (tree)", c16.data());
        auto run = [&](const char* name, void (*fn)(const float*, const void*, float*, int, float, float), bool bf, float S) {
            if (bf) hipMemcpy(dW, Pb.data(), Pb.size() * 2, hipMemcpyHostToDevice);
            else hipMemcpy(dW, Pf.data(), Pf.size() * 2, hipMemcpyHostToDevice);
            fn(dh, dW, dout, B, S, bf ? 1.0f : 1.0f / (T * S));
            hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
            report(name, o.data());
        };
        run("bf16 x3, 6 products (K8 / K11 r5)", launch<BF6>, true, 1.0f);
        run("bf16 x3, 9 products", launch<BF9>, true, 1.0f);
        run("bf16 x3, 6 products, hi hi in its own accumulator", launch<BF6_SPLIT>, true, 1.0f);
        run("f16 x2, 3 products, S = 1 (K8h)", launch<F2_3>, false, 1.0f);
        run("f16 x3, 5 products, S = 1", launch<F3_5>, false, 1.0f);
        run("f16 x3, 5 products, S = 16 (K8x)", launch<F3_5>, false, 16.0f);
        run("f16 x3, 5 products, S = 16, hi hi own accumulator", launch<F3_5_SPLIT>, false, 16.0f);
        run("f16 x3, 6 products (+ lo lo), S = 16", launch<F3_6>, false, 16.0f);
    }
    return 0;
}
