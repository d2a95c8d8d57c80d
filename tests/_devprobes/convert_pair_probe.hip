// Test infrastructure (round 4): the piece conversion of the f16 whole-layer kernels -- nfa::k8h::convert_pair from
// nflows_amd/csrc/k8h_common.hpp, THE source K8h / K8s compile, not a copy of its instructions -- against the plain
// sequence it replaces: hi = RN16(v * scale), lo = RN16(v * scale - hi), ReLU applied first when asked, `peak` the
// running maximum of |v| (after the ReLU).  25 M random words per (relu, guard, scale); prints the number of differing
// words and "convert_pair_probe: OK" / "MISMATCH".  Built by __graft_entry__.build() into tests/_devprobes/bin/ and run
// by tests/test_gpu_probes.py.
#include "k8h_common.hpp"

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <bool RELU, bool GUARD>
__global__ void probe(const float* v, float scale, unsigned* out_fn, unsigned* out_ref, float* peaks, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0 = v[2 * i], s1 = v[2 * i + 1];
    unsigned h, l;
    float peak = 0.25f;
    nfa::k8h::convert_pair<RELU, GUARD>(s0, s1, scale, peak, h, l);
    out_fn[2 * i] = h;
    out_fn[2 * i + 1] = l;
    peaks[2 * i] = peak;
    // the sequence the instructions replace
    float a0 = s0, a1 = s1;
    if (RELU) {
        a0 = a0 > 0.0f ? a0 : (a0 != a0 ? 0.0f : 0.0f);   // v_max_f32(x, 0): NaN -> 0
        a1 = a1 > 0.0f ? a1 : 0.0f;
    }
    const float p0 = a0 * scale, p1 = a1 * scale;          // (power-of-two scale: exact)
    const _Float16 h0 = (_Float16)p0, h1 = (_Float16)p1;
    const _Float16 l0 = (_Float16)(p0 - (float)h0), l1 = (_Float16)(p1 - (float)h1);
    half2_t hv = {h0, h1}, lv = {l0, l1};
    out_ref[2 * i] = __builtin_bit_cast(unsigned, hv);
    out_ref[2 * i + 1] = __builtin_bit_cast(unsigned, lv);
    float pk = 0.25f;
    pk = fmaxf(pk, fmaxf(RELU ? a0 : fabsf(a0), RELU ? a1 : fabsf(a1)));
    peaks[2 * i + 1] = pk;
}

template <bool RELU, bool GUARD>
static int run(const float* dv, const std::vector<float>& hv, float scale, unsigned* da, unsigned* dr, float* dp, int n) {
    hipLaunchKernelGGL((probe<RELU, GUARD>), dim3((n + 255) / 256), dim3(256), 0, 0, dv, scale, da, dr, dp, n);
    std::vector<unsigned> a(2 * n), r(2 * n);
    std::vector<float> p(2 * n);
    hipMemcpy(a.data(), da, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), dr, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
    hipMemcpy(p.data(), dp, sizeof(float) * 2 * n, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2 * n; ++i)
        if (a[i] != r[i]) {
            // (the low piece of an overflowed high piece is NaN / -inf either way: any NaN payload counts as equal)
            const unsigned short a0 = a[i] & 0xffff, a1 = a[i] >> 16, r0 = r[i] & 0xffff, r1 = r[i] >> 16;
            auto nan16 = [](unsigned short x) { return (x & 0x7c00) == 0x7c00 && (x & 0x3ff); };
            // (and the sign of a zero: the fused form computes v * scale + 0, which turns -0 into +0; a zero piece
            //  contributes nothing to a product either way)
            auto zero16 = [](unsigned short x) { return (x & 0x7fff) == 0; };
            const bool same0 = a0 == r0 || (nan16(a0) && nan16(r0)) || (zero16(a0) && zero16(r0));
            const bool same1 = a1 == r1 || (nan16(a1) && nan16(r1)) || (zero16(a1) && zero16(r1));
            if (same0 && same1) continue;
            if (bad < 5) printf("  relu %d guard %d scale %g: pair %d word %d: function %08x plain %08x (v = %g, %g)\n", RELU, GUARD, scale,
                                i / 2, i % 2, a[i], r[i], hv[2 * (i / 2)], hv[2 * (i / 2) + 1]);
            ++bad;
        }
    int bad_peak = 0;
    for (int i = 0; i < n; ++i)
        if (!(p[2 * i] == p[2 * i + 1]) && !(p[2 * i] != p[2 * i] && p[2 * i + 1] != p[2 * i + 1])) {
            if (bad_peak < 3) printf("  relu %d: pair %d: peak %g, expected %g\n", RELU, i, p[2 * i], p[2 * i + 1]);
            ++bad_peak;
        }
    printf("convert_pair<relu=%d, guard=%d> scale=%g: %d of %d words differ, %d of %d peaks differ\n", RELU, GUARD, scale, bad, 2 * n, bad_peak, n);
    return bad + bad_peak;
}

int main() {
    const int n = 1 << 22;   // pairs per configuration
    std::vector<float> h(2 * n);
    srand(12345);
    for (int i = 0; i < 2 * n; ++i) {
        // magnitudes from 1e-9 to 1e6 (beyond the f16 range on both sides), both signs, a few specials
        const float mag = expf(((float)rand() / RAND_MAX) * 34.5f - 20.7f);
        h[i] = ((rand() & 1) ? mag : -mag) * (0.5f + (float)rand() / RAND_MAX);
    }
    h[0] = 0.0f; h[1] = -0.0f; h[2] = 65504.0f; h[3] = 65519.9f; h[4] = 65520.0f; h[5] = 1e-8f; h[6] = 6.1e-5f; h[7] = -3.0f;
    float *dv, *dp;
    unsigned *da, *dr;
    hipMalloc(&dv, sizeof(float) * 2 * n);
    hipMalloc(&dp, sizeof(float) * 2 * n);
    hipMalloc(&da, sizeof(unsigned) * 2 * n);
    hipMalloc(&dr, sizeof(unsigned) * 2 * n);
    hipMemcpy(dv, h.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice);
    int bad = 0;
    for (float scale : {1.0f, 0.25f, 16.0f}) {
        bad += run<false, false>(dv, h, scale, da, dr, dp, n);
        bad += run<true, false>(dv, h, scale, da, dr, dp, n);
        bad += run<false, true>(dv, h, scale, da, dr, dp, n);
        bad += run<true, true>(dv, h, scale, da, dr, dp, n);
    }
    printf("convert_pair_probe: %s\n", bad == 0 ? "OK" : "MISMATCH");
    return bad == 0 ? 0 : 1;
}
