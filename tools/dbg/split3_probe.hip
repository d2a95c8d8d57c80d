// debug probe: split3_scaled / pieces_fma2 / relu_pieces of f16x3_gemm.hpp against host arithmetic
#include "f16x3_gemm.hpp"
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstring>
using namespace nfa;
__global__ void k(const float* v, float scale, unsigned* out, float* rec, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned h, l, r;
    k8x::split3_scaled(v[2 * i], v[2 * i + 1], scale, h, l, r);
    out[3 * i] = h; out[3 * i + 1] = l; out[3 * i + 2] = r;
    float a0 = 0.f, a1 = 0.f;
    k8x::pieces_fma2(h, l, r, 1.0f, a0, a1);
    rec[2 * i] = a0; rec[2 * i + 1] = a1;
}
static float h2f(unsigned short b) { _Float16 x; memcpy(&x, &b, 2); return (float)x; }
int main() {
    const int n = 1 << 16;
    std::vector<float> v(n);
    srand(3);
    for (auto& x : v) { float u = rand() / (float)RAND_MAX; float w = rand() / (float)RAND_MAX; x = (u - 0.5f) * 8.f * powf(10.f, (w - 0.5f) * 4.f); }
    float* dv; unsigned* dout; float* drec;
    hipMalloc(&dv, n * 4); hipMalloc(&dout, n / 2 * 12); hipMalloc(&drec, n * 4);
    hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
    for (float S : {1.0f, 16.0f}) {
        k<<<n / 2 / 256, 256>>>(dv, S, dout, drec, n);
        std::vector<unsigned> o(n / 2 * 3); std::vector<float> rec(n);
        hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(rec.data(), drec, n * 4, hipMemcpyDeviceToHost);
        int bad_h = 0, bad_l = 0, bad_r = 0, bad_rec = 0; double worst = 0;
        for (int i = 0; i < n; ++i) {
            const float x = v[i] * S;
            const _Float16 eh = (_Float16)x; const float t = x - (float)eh; const _Float16 el = (_Float16)t; const _Float16 er = (_Float16)(t - (float)el);
            const unsigned hw = o[3 * (i / 2)], lw = o[3 * (i / 2) + 1], rw = o[3 * (i / 2) + 2];
            const unsigned short gh = (i & 1) ? hw >> 16 : hw & 0xffff, gl = (i & 1) ? lw >> 16 : lw & 0xffff, gr = (i & 1) ? rw >> 16 : rw & 0xffff;
            if (h2f(gh) != (float)eh) ++bad_h;
            if (h2f(gl) != (float)el) ++bad_l;
            if (h2f(gr) != (float)er) ++bad_r;
            const double sum = (double)h2f(gh) + h2f(gl) + h2f(gr);
            if (rec[i] != (float)sum) ++bad_rec;
            worst = fmax(worst, fabs(sum - x) / fmax(fabs(x), 1e-30));
            if (i < 4) printf("  x %.9g -> %.9g %.9g %.9g (want %.9g %.9g %.9g) rec %.9g\n", x, h2f(gh), h2f(gl), h2f(gr), (float)eh, (float)el, (float)er, rec[i]);
        }
        printf("S=%g: bad hi %d lo %d r %d rec %d of %d; worst rel |sum - x| %.3e\n", S, bad_h, bad_l, bad_r, bad_rec, n, worst);
    }
    return 0;
}
