// Probe for the round-2 rebuild of K8:
//  (1) accuracy of an fp32 GEMM (K = 128) computed on the f16 matrix pipe from TWO f16 pieces per
//      operand (x = hi + lo, three cross products hi*hi, hi*lo, lo*hi) next to the three-piece bf16
//      scheme (six products) and a sequential fp32 fma chain; weights optionally pre-scaled by a
//      power of two so that their low pieces stay in the normal f16 range;
//  (2) whether v_mfma_f32_32x32x16_f16 honours f16 subnormal inputs;
//  (3) the price of VALU filler instructions between MFMAs that rotate over independent
//      accumulators (the structure the guide recommends), one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/f16x2_probe.hip -o tools/bin/f16x2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
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

// MODE 0: bf16 x 3 pieces, 6 products; MODE 1: f16 x 2 pieces, 3 products
// W pieces: [piece][ntile][kb(8)][lane][8]; lane l, element j = W[ntile*32 + (l&31)][kb*16 + (l>>5)*8 + j]
template <int MODE>
__global__ void __launch_bounds__(256) gemm_split(const float* __restrict__ h, const void* __restrict__ Wp,
                                                 float* __restrict__ out, int rows, float out_scale) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const int half = lane >> 5, r = lane & 31;
    for (int tile = wave_global; tile * 32 < rows; tile += nwaves) {
        const int row0 = tile * 32;
        bf16x8 a3[3][8];
        f16x8 a2[2][8];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const float* hp = h + (size_t)(row0 + r) * H + kb * 16 + half * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = hp[j];
                if (MODE == 0) {
                    const unsigned short xh = f2bf(x);
                    const float r1 = x - bf2f(xh);
                    const unsigned short xm = f2bf(r1);
                    const float r2 = r1 - bf2f(xm);
                    a3[0][kb][j] = (short)xh; a3[1][kb][j] = (short)xm; a3[2][kb][j] = (short)f2bf(r2);
                } else {
                    const _Float16 xh = (_Float16)x;
                    a2[0][kb][j] = xh;
                    a2[1][kb][j] = (_Float16)(x - (float)xh);
                }
            }
        }
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc = {0};
            const size_t piece = (size_t)NT * 8 * 64;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (MODE == 0) {
                    const bf16x8* bp = reinterpret_cast<const bf16x8*>(Wp) + ((size_t)nt * 8 + kb) * 64 + lane;
                    const bf16x8 b0 = bp[0], b1 = bp[piece], b2 = bp[2 * piece];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[2][kb], b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0][kb], b2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1][kb], b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1][kb], b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0][kb], b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0][kb], b0, acc, 0, 0, 0);
                } else {
                    const f16x8* bp = reinterpret_cast<const f16x8*>(Wp) + ((size_t)nt * 8 + kb) * 64 + lane;
                    const f16x8 b0 = bp[0], b1 = bp[piece];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[1][kb], b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[0][kb], b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[0][kb], b0, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = (g & 3) + 8 * (g >> 2) + 4 * half;
                out[(size_t)(row0 + row) * N + nt * 32 + r] = acc[g] * out_scale;
            }
        }
    }
}

// (2) one MFMA with hand-made inputs: a = subnormal f16 everywhere, b = 1.0
__global__ void subnormal_probe(float* out) {
    f16x8 a, b;
    const unsigned short sub = 0x0001;  // 2^-24
    _Float16 s; memcpy(&s, &sub, 2);
    for (int j = 0; j < 8; ++j) { a[j] = s; b[j] = (_Float16)1.0f; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];   // 16 * 2^-24 = 9.5e-7 if subnormals are honoured, 0 if flushed
    // and a conversion producing a subnormal
    const float tiny = 3.0e-6f;
    const _Float16 t = (_Float16)tiny;
    if (threadIdx.x == 0) out[1] = (float)t;
}

// (3) MFMA + filler stream.  ACCS accumulators in rotation, FILL independent v_fma between MFMAs.
template <int ACCS, int FILL, bool PK>
__global__ void __launch_bounds__(512) weave_probe(int iters, float* out, unsigned long long* span) {
    extern __shared__ float big[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 0.01f * (lane + j);
    const float c = 0.999f, d = 0.001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m % ACCS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % ACCS], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < FILL; ++j) {
                const int q = (m * FILL + j) % 16;
                v[q] = __builtin_fmaf(v[q], c, d);
                asm volatile("" : "+v"(v[q]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j] + acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j];
    if (s == 1.2345f) out[0] = s;
    if (lane == 0) span[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int ACCS, int FILL>
static void run_weave(int waves) {
    const int grid = 256, iters = 4000;
    float* out; unsigned long long* sp;
    hipMalloc(&out, 64); hipMalloc(&sp, grid * 8 * 8);
    auto k = weave_probe<ACCS, FILL, false>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int i = 0; i < 2; ++i) k<<<grid, waves * 64, 100 * 1024>>>(iters, out, sp);
    hipDeviceSynchronize();
    std::vector<unsigned long long> s(waves);
    hipMemcpy(s.data(), sp, waves * 8, hipMemcpyDeviceToHost);
    printf("weave accs=%d fill=%2d waves/SIMD=%d: %6.1f cyc per MFMA (wave 0), %6.1f (last wave)\n", ACCS, FILL, waves / 4,
           (double)s[0] / iters / 8, (double)s[waves - 1] / iters / 8);
    hipFree(out); hipFree(sp);
}

int main() {
    const int B = 2048;
    std::vector<float> W((size_t)N * H);
    srand(1);
    for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.176f;   // U(-1/sqrt(128), 1/sqrt(128))
    float *dh, *dout; void* dW;
    hipMalloc(&dh, (size_t)B * H * 4); hipMalloc(&dW, (size_t)3 * N * H * 2); hipMalloc(&dout, (size_t)B * N * 4);
    std::vector<float> o((size_t)B * N);

    for (float act_scale : {1.0f, 1e-2f, 1e-4f, 100.0f}) {
        std::vector<float> hh((size_t)B * H);
        srand(7);
        for (auto& v : hh) {   // relu-like: half zeros, half |N(0,1)|-ish
            const float u = rand() / (float)RAND_MAX, w = rand() / (float)RAND_MAX;
            v = (u < 0.5f) ? 0.f : act_scale * 2.f * w * w;
        }
        hipMemcpy(dh, hh.data(), hh.size() * 4, hipMemcpyHostToDevice);
        // references
        std::vector<double> ref((size_t)B * N); std::vector<float> ref32((size_t)B * N);
        for (int r = 0; r < B; ++r) for (int c = 0; c < N; ++c) {
            double s = 0; float s32 = 0.f;
            for (int k = 0; k < H; ++k) { s += (double)hh[(size_t)r * H + k] * W[(size_t)c * H + k]; s32 = fmaf(hh[(size_t)r * H + k], W[(size_t)c * H + k], s32); }
            ref[(size_t)r * N + c] = s; ref32[(size_t)r * N + c] = s32;
        }
        auto report = [&](const char* name, const float* got) {
            double mx = 0, sq = 0, mref = 0;
            for (size_t i = 0; i < ref.size(); ++i) { const double e = fabs(ref[i] - got[i]); mx = fmax(mx, e); sq += e * e; mref = fmax(mref, fabs(ref[i])); }
            printf("  act x%-7g %-28s max|err| %.3e  rms %.3e   (max|ref| %.3g)\n", act_scale, name, mx, sqrt(sq / ref.size()), mref);
        };
        report("fp32 fma chain", ref32.data());
        // bf16 x 3
        {
            std::vector<unsigned short> P((size_t)3 * NT * 8 * 64 * 8);
            const size_t piece = (size_t)NT * 8 * 64 * 8;
            for (int nt = 0; nt < NT; ++nt) for (int kb = 0; kb < 8; ++kb) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
                const float x = W[(size_t)(nt * 32 + (l & 31)) * H + kb * 16 + (l >> 5) * 8 + j];
                const unsigned short xh = f2bf(x); const float r1 = x - bf2f(xh);
                const unsigned short xm = f2bf(r1); const float r2 = r1 - bf2f(xm);
                const size_t idx = (((size_t)nt * 8 + kb) * 64 + l) * 8 + j;
                P[idx] = xh; P[piece + idx] = xm; P[2 * piece + idx] = f2bf(r2);
            }
            hipMemcpy(dW, P.data(), P.size() * 2, hipMemcpyHostToDevice);
            gemm_split<0><<<64, 256>>>(dh, dW, dout, B, 1.0f);
            hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
            report("bf16 x3, 6 products", o.data());
        }
        for (int tscale : {0, 8, 12}) {
            std::vector<_Float16> P((size_t)2 * NT * 8 * 64 * 8);
            const size_t piece = (size_t)NT * 8 * 64 * 8;
            const float T = ldexpf(1.0f, tscale);
            for (int nt = 0; nt < NT; ++nt) for (int kb = 0; kb < 8; ++kb) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
                const float x = W[(size_t)(nt * 32 + (l & 31)) * H + kb * 16 + (l >> 5) * 8 + j] * T;
                const _Float16 xh = (_Float16)x;
                const size_t idx = (((size_t)nt * 8 + kb) * 64 + l) * 8 + j;
                P[idx] = xh; P[piece + idx] = (_Float16)(x - (float)xh);
            }
            hipMemcpy(dW, P.data(), P.size() * 2, hipMemcpyHostToDevice);
            gemm_split<1><<<64, 256>>>(dh, dW, dout, B, 1.0f / T);
            hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
            char name[64]; snprintf(name, sizeof name, "f16 x2, 3 products, W*2^%d", tscale);
            report(name, o.data());
        }
    }
    {
        float* d; hipMalloc(&d, 64);
        subnormal_probe<<<1, 64>>>(d);
        float r[2]; hipMemcpy(r, d, 8, hipMemcpyDeviceToHost);
        printf("f16 MFMA with subnormal inputs: acc = %.3e (9.54e-07 = honoured, 0 = flushed); cvt(3e-6) -> %.3e\n", r[0], r[1]);
    }
    run_weave<1, 0>(4); run_weave<4, 0>(4);
    run_weave<1, 2>(4); run_weave<4, 2>(4);
    run_weave<1, 4>(4); run_weave<4, 4>(4);
    run_weave<4, 5>(4); run_weave<4, 6>(4); run_weave<4, 7>(4); run_weave<4, 8>(4); run_weave<4, 10>(4); run_weave<4, 12>(4);
    run_weave<4, 0>(8); run_weave<4, 4>(8); run_weave<4, 6>(8); run_weave<4, 8>(8); run_weave<4, 10>(8); run_weave<4, 12>(8);
    run_weave<1, 4>(8); run_weave<1, 8>(8);
    return 0;
}
