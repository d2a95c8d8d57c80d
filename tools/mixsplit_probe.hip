// Checks the piece conversion of csrc/rqs_resnet_f16.hip (round 3) on the device: for random fp32 values v and a
// power-of-two scale, hi = RN16(v * scale) and lo = RN16(v * scale - hi) made by v_fma_mixlo_f16 / v_fma_mixhi_f16
// must equal the convert / convert back / subtract / convert sequence bit for bit (both halves of the packed pair).
//   hipcc --offload-arch=gfx950 -O2 tools/mixsplit_probe.hip -o tools/bin/mixsplit_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float vec2f __attribute__((ext_vector_type(2)));

__global__ void probe(const float* v, float scale, int relu, unsigned* out_asm, unsigned* out_ref, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0 = v[2 * i], s1 = v[2 * i + 1];
    unsigned h, l;
    float peak = 0.0f;
    if (relu) {
        float m0, m1;
        asm("v_max_f32 %2, %5, 0\n\t"
            "v_max_f32 %3, %6, 0\n\t"
            "v_fma_mixlo_f16 %0, %2, %7, 0 op_sel_hi:[0,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %3, %7, 0 op_sel_hi:[0,0,0]\n\t"
            "v_max3_f32 %4, %4, %2, %3\n\t"
            "v_fma_mixlo_f16 %1, %2, %7, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %3, %7, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(h), "=&v"(l), "=&v"(m0), "=&v"(m1), "+v"(peak)
            : "v"(s0), "v"(s1), "v"(scale));
    } else {
        asm("v_fma_mixlo_f16 %0, %3, %5, 0 op_sel_hi:[0,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]\n\t"
            "v_max3_f32 %2, %2, |%3|, |%4|\n\t"
            "v_fma_mixlo_f16 %1, %3, %5, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %4, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(h), "=&v"(l), "+v"(peak)
            : "v"(s0), "v"(s1), "v"(scale));
    }
    out_asm[3 * i] = h;
    out_asm[3 * i + 1] = l;
    out_asm[3 * i + 2] = __builtin_bit_cast(unsigned, peak);
    // the round-2 sequence
    float v0 = s0 * scale, v1 = s1 * scale;
    if (relu) {
        v0 = (v0 < 0.0f) ? 0.0f : v0;
        v1 = (v1 < 0.0f) ? 0.0f : v1;
    }
    const f16x2 hh = __builtin_convertvector(vec2f{v0, v1}, f16x2);
    float r0 = v0 - (float)hh[0], r1 = v1 - (float)hh[1];
    const f16x2 ll = __builtin_convertvector(vec2f{r0, r1}, f16x2);
    out_ref[3 * i] = __builtin_bit_cast(unsigned, hh);
    out_ref[3 * i + 1] = __builtin_bit_cast(unsigned, ll);
    const float a0 = relu ? fmaxf(s0, 0.0f) : fabsf(s0), a1 = relu ? fmaxf(s1, 0.0f) : fabsf(s1);
    out_ref[3 * i + 2] = __builtin_bit_cast(unsigned, fmaxf(a0, a1));
}

int main() {
    const int n = 1 << 20;
    std::vector<float> h(2 * n);
    srand(7);
    for (int i = 0; i < 2 * n; ++i) {
        const double u = rand() / (double)RAND_MAX, e = (rand() % 40) - 26;   // magnitudes 2^-26 .. 2^13
        h[i] = (float)((2 * u - 1) * std::ldexp(1.0, (int)e));
    }
    h[0] = 0.0f; h[1] = -0.0f; h[2] = 65504.0f; h[3] = 1e-8f; h[4] = 70000.0f; h[5] = -70000.0f;
    float* dv; unsigned *da, *dr;
    hipMalloc(&dv, 2 * n * 4); hipMalloc(&da, 3 * n * 4); hipMalloc(&dr, 3 * n * 4);
    hipMemcpy(dv, h.data(), 2 * n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> a(3 * n), r(3 * n);
    int bad_total = 0;
    const float scales[] = {1.0f, 0.25f, 4.0f, 1.0f / 8192.0f};
    for (int relu = 0; relu < 2; ++relu)
        for (float sc : scales) {
            hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dv, sc, relu, da, dr, n);
            hipMemcpy(a.data(), da, 3 * n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(r.data(), dr, 3 * n * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < 3 * n; ++i)
                if (a[i] != r[i]) {
                    // (+0 / -0 pieces are the same number)
                    auto same_half = [](unsigned x, unsigned y) { return x == y || (((x | y) & 0x7fffu) == 0); };
                    const bool zero_pair = i % 3 != 2 && same_half(a[i] & 0xffffu, r[i] & 0xffffu) && same_half(a[i] >> 16, r[i] >> 16);
                    if (zero_pair) continue;
                    if (bad < 5) printf("  relu %d scale %g: item %d word %d: asm %08x ref %08x (v = %g, %g)\n", relu, sc, i / 3, i % 3, a[i], r[i], h[2 * (i / 3)], h[2 * (i / 3) + 1]);
                    ++bad;
                }
            printf("mixsplit relu=%d scale=%g: %d of %d words differ\n", relu, sc, bad, 3 * n);
            bad_total += bad;
        }
    printf("mixsplit_probe: %s\n", bad_total == 0 ? "OK" : "MISMATCH");
    return bad_total != 0;
}
