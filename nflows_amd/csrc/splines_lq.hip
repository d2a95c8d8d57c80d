// K9: the rational-quadratic spline's siblings as elementwise functionals (SURVEY.md section 8f,
// row f4): piecewise-linear (splines/linear.py:9-105), piecewise-quadratic
// (splines/quadratic.py:11-159) and piecewise-cubic (splines/cubic.py:15-267) splines, constrained
// and with linear tails, forward and inverse.
//
// Same skeleton as K5: a tile of T elements per workgroup pass, every lane stages its own logits
// in an LDS slot (coalesced when the logit arrays are one packed [n, P] view, a per-lane strided
// gather otherwise), runs softmax -> prefix sums -> bin search -> per-bin polynomial on them in
// place, and writes its output and log-derivative.  Arithmetic follows the reference step by step
// (fp contraction off, IEEE division, prefix sums accumulated in double like aten's CPU cumsum);
// the reference's quirks that are visible in the results are reproduced and marked below.

#include "rqs_math.hpp"

namespace nfa {

enum { kLinear = 0, kQuadratic = 1, kCubic = 2 };

struct LqArgs {
    const float* x;
    const float* a0;  // linear: unnormalized pdf [.., K]; quadratic: unnormalized widths [.., K]
    const float* a1;  // quadratic: unnormalized heights [.., nh]; cubic: [.., K]
    const float* a2;  // cubic: left / right boundary-derivative logits [.., 1]
    const float* a3;
    int64_t s0, s1, s2, s3;  // element strides of the rows of a0 .. a3
    float* y;
    float* lad;
    int32_t* status;
    int64_t n;
    int K, nh, slot, T, packed, unconstrained;
    float left, right, bottom, top;     // box (unconstrained: +-tail_bound)
    float span_in, span_out;            // (float)(right - left), (float)(top - bottom)
    float min_w, min_h, om_w, om_h;     // minimums, (float)(1 - min_w*K), quadratic: (float)(1 - min_h)
    float om_hk;                        // cubic: (float)(1 - min_h*K)
    float divisor, rdivisor;            // quadratic: logits / divisor first (0 = no scaling)
    float log_bin_width;                // linear: (float)log(1/K)
};

__device__ __forceinline__ float clamp01(float v) {  // torch.clamp(v, 0, 1): NaN stays NaN
    return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}

// softmax of K logits in place (optionally divided by `divisor` first); the fp32 denominator is
// accumulated in double (a long sequential fp32 sum would drift from aten's blocked sum).
// KT > 0: K known at compile time, loops unroll and `p` may be a register array.
template <int KT>
__device__ __forceinline__ void softmax_in_place(float* p, int Krt, float divisor, float rdivisor) {
#pragma clang fp contract(off)
    const int K = KT > 0 ? KT : Krt;
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        float u = p[i];
        if (divisor != 0.0f) u = div_with_rcp(u, divisor, rdivisor);
        p[i] = u;
        m = fmaxf(m, u);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const float e = exp_noclamp(p[i] - m);
        p[i] = e;
        s += (double)e;
    }
    const float sum = (float)s;
    const float rsum = rcp_refined(sum);  // a / b from RN(1/b): the correctly rounded quotient
#pragma unroll
    for (int i = 0; i < K; ++i) p[i] = div_with_rcp(p[i], sum, rsum);
}

// splines/linear.py:40-105.  p: the lane's K logits (overwritten).  x is inside the box.
// No data-dependent indexing: the bin is picked while walking the prefix sums.
template <int KT, bool INVERSE>
__device__ __forceinline__ int linear_eval(float x, float* p, const LqArgs& a, float& y, float& lad) {
#pragma clang fp contract(off)
    const int K = KT > 0 ? KT : a.K;
    const float u = INVERSE ? (x - a.bottom) / a.span_out : (x - a.left) / a.span_in;
    softmax_in_place<KT>(p, K, 0.0f, 0.0f);
    float out;
    if (INVERSE) {
        // cdf = pad0(cumsum(pdf)) with cdf[K] = 1; searchsorted adds 1e-6 to the last knot IN PLACE
        // (torchutils.py:135), so the last bin's slope and offset see 1 + 1e-6 as well (quirk)
        double acc = 0.0;
        float prev = 0.0f, lo = 0.0f, hi = 0.0f;
        int k = -1;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            acc += (double)p[i];
            const float next = (i == K - 1) ? 1.0f + 1e-6f : (float)acc;
            if (u >= prev) {
                k = i;
                lo = prev;
                hi = next;
            }
            prev = next;
        }
        if (k < 0 || u >= prev) {
            y = x;
            lad = 0.0f;
            return NFA_STATUS_OUTSIDE_DOMAIN;
        }
        // torch.linspace(0, 1, K + 1) in float32: start + i*step below the middle, end - (K-i)*step above
        const float step = 1.0f / (float)K;
        const int half = (K + 1) / 2;
        const float b0 = (k < half) ? (float)k * step : 1.0f - (float)(K - k) * step;
        const float b1 = (k + 1 < half) ? (float)(k + 1) * step : 1.0f - (float)(K - k - 1) * step;
        const float slope = (hi - lo) / (b1 - b0);
        const float offset = hi - slope * b1;
        out = clamp01((u - offset) / slope);
        lad = -log_normal(slope);
    } else {
        const float pos = u * (float)K;
        int k = (int)floorf(pos);
        k = k >= K ? K - 1 : (k < 0 ? 0 : k);
        const float alpha = pos - (float)k;
        double acc = 0.0;
        float ck = 0.0f, pk = 0.0f;  // cdf[k], pdf[k]
#pragma unroll
        for (int i = 0; i < K; ++i) {
            if (i == k) {
                ck = (float)acc;
                pk = p[i];
            }
            acc += (double)p[i];
        }
        out = clamp01(ck + alpha * pk);
        lad = log_normal(pk) - a.log_bin_width;
    }
    y = INVERSE ? out * a.span_in + a.left : out * a.span_out + a.bottom;
    return 0;
}

// splines/quadratic.py:55-159.  w: K width logits (overwritten by the widths); h: K+1 slots, the
// nh height logits sit at h[1..nh] (nh = K-1) or h[0..K] (nh = K+1) and are overwritten by the heights.
template <int KT, bool INVERSE>
__device__ __forceinline__ int quadratic_eval(float x, float* w, float* h, const LqArgs& a, float& y, float& lad) {
#pragma clang fp contract(off)
    const int K = KT > 0 ? KT : a.K;
    const float u = INVERSE ? (x - a.bottom) / a.span_out : (x - a.left) / a.span_in;
    softmax_in_place<KT>(w, K, a.divisor, a.rdivisor);
#pragma unroll
    for (int i = 0; i < K; ++i) w[i] = a.min_w + a.om_w * w[i];
    const bool derived = a.nh == K - 1;
#pragma unroll
    for (int i = 0; i <= K; ++i) {
        if (derived && (i == 0 || i == K)) continue;
        float v = h[i];
        if (a.divisor != 0.0f) v = div_with_rcp(v, a.divisor, a.rdivisor);
        h[i] = softplus_beta(v, 1.0f) + 1e-3f;
    }
    if (derived) {  // boundary heights such that the normalised ones are exactly 1 (:93-107)
        const float fw = 0.5f * w[0], lw = 0.5f * w[K - 1];
        float s = 0.0f;
#pragma unroll
        for (int i = 1; i + 1 < K; ++i) s += ((h[i] + h[i + 1]) / 2.0f) * w[i];
        const float num = (0.5f * fw) * h[1] + (0.5f * lw) * h[K - 1] + s;
        const float c = num / ((1.0f - 0.5f * fw) - 0.5f * lw);
        h[0] = c;
        h[K] = c;
    }
    float area = 0.0f;
#pragma unroll
    for (int i = 0; i < K; ++i) area += ((h[i] + h[i + 1]) / 2.0f) * w[i];
    const float rarea = rcp_refined(area);
#pragma unroll
    for (int i = 0; i <= K; ++i) h[i] = a.min_h + a.om_h * div_with_rcp(h[i], area, rarea);

    // knots: bin_left_cdf (searched in the inverse) and bin_locations (searched forward), both
    // pad0(cumsum(.)) with the last entry forced to 1 (+1e-6 for the search); the bin's width and
    // end heights are picked during the walk
    double acc_c = 0.0, acc_l = 0.0;
    float pc = 0.0f, pl = 0.0f, c0 = 0.0f, l0 = 0.0f, bw = 0.0f, hl = 0.0f, hr = 0.0f;
    int k = -1;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        acc_c += (double)(((h[i] + h[i + 1]) / 2.0f) * w[i]);
        acc_l += (double)w[i];
        const bool last = i == K - 1;
        const float nc = last ? 1.0f : (float)acc_c, nl = last ? 1.0f : (float)acc_l;
        if (u >= (INVERSE ? pc : pl)) {
            k = i;
            c0 = pc;
            l0 = pl;
            bw = w[i];
            hl = h[i];
            hr = h[i + 1];
        }
        pc = nc;
        pl = nl;
    }
    if (k < 0 || u >= 1.0f + 1e-6f) {
        y = x;
        lad = 0.0f;
        return NFA_STATUS_OUTSIDE_DOMAIN;
    }
    const float qa = (0.5f * (hr - hl)) * bw, qb = hl * bw, qc = c0;
    float out;
    if (INVERSE) {
        const float c_ = qc - u;
        const float alpha = (-qb + sqrtf(qb * qb - (4.0f * qa) * c_)) / (2.0f * qa);
        out = clamp01(alpha * bw + l0);
        lad = -log_normal(alpha * (hr - hl) + hl);
    } else {
        const float alpha = (u - l0) / bw;
        out = clamp01((qa * (alpha * alpha) + qb * alpha) + qc);
        lad = log_normal(alpha * (hr - hl) + hl);
    }
    y = INVERSE ? out * a.span_in + a.left : out * a.span_out + a.bottom;
    return 0;
}

// torchutils.cbrt (torchutils.py:139-141): sign(x) * exp(log|x| / 3)
__device__ __forceinline__ float cbrt_like_reference(float x) {
#pragma clang fp contract(off)
    const float sg = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
    return sg * expf(logf(fabsf(x)) / 3.0f);
}

__device__ __forceinline__ float sign_of(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// Root of ca s^3 + cb s^2 + cc s + cd = u reported as lcw + s, by the reference's case analysis
// (cubic.py:151-226): one real root, three real roots (the one inside the bin, +- 1e-5), and the
// "almost quadratic" override for |ca| < 1e-3.
__device__ __forceinline__ float cubic_inverse_root(float ca, float cb, float cc, float cd, float u, float lcw,
                                                    float rcw, bool& almost_quadratic) {
#pragma clang fp contract(off)
    const float b_ = (cb / ca) / 3.0f, c_ = (cc / ca) / 3.0f, d_ = (cd - u) / ca;
    const float d1 = -(b_ * b_) + c_;
    const float d2 = (-c_) * b_ + d_;
    const float d3 = b_ * d_ - c_ * c_;
    const float disc = (4.0f * d1) * d3 - d2 * d2;
    const float dep1 = (-2.0f * b_) * d1 + d2;
    float out = 0.0f;
    if (disc < 0.0f) {  // one real root
        const float sq = sqrtf(-disc);
        const float p = cbrt_like_reference((-dep1 + sq) / 2.0f);
        const float q = cbrt_like_reference((-dep1 - sq) / 2.0f);
        out = ((p + q) - b_) + lcw;
    } else if (disc >= 0.0f) {  // three real roots: the one inside the bin (+- eps)
        const float theta = atan2f(sqrtf(disc), -dep1) / 3.0f;
        const float c1 = cosf(theta), s1 = sinf(theta);
        const float hs3 = 0.8660254037844386f;  // 0.5 * sqrt(3)
        const float scale = 2.0f * sqrtf(-d1);
        const float shift = -b_ + lcw;
        const float r1 = c1 * scale + shift;
        const float r2 = (-0.5f * c1 - hs3 * s1) * scale + shift;
        const float r3 = (-0.5f * c1 + hs3 * s1) * scale + shift;
        const float lo = lcw - 1e-5f, hi = rcw + 1e-5f;
        const bool m1 = lo < r1 && r1 < hi, m2 = lo < r2 && r2 < hi, m3 = lo < r3 && r3 < hi;
        out = m1 ? r1 : (m2 ? r2 : (m3 ? r3 : r1));
    }
    if (fabsf(ca) < 1e-3f) {  // almost quadratic (:219-226)
        const float alpha = (-cc + sqrtf(cc * cc - (4.0f * cb) * (cd - u))) / (2.0f * cb);
        out = alpha + lcw;
    }
    almost_quadratic = fabsf(ca) < 1e-3f;
    return out;
}

// splines/cubic.py:63-267.  w / h: K width / height logits (overwritten by the bin widths / heights),
// udl / udr: the two boundary-derivative logits.  The coefficients of the searched bin are formed
// from its neighbours during one walk over the bins (no data-dependent indexing).
template <int KT, bool INVERSE>
__device__ __forceinline__ int cubic_eval(float x, float* w, float* h, float udl, float udr, const LqArgs& a,
                                          float& y, float& lad) {
#pragma clang fp contract(off)
    const int K = KT > 0 ? KT : a.K;
    const float u = INVERSE ? (x - a.bottom) / a.span_out : (x - a.left) / a.span_in;
    softmax_in_place<KT>(w, K, a.divisor, a.rdivisor);
    softmax_in_place<KT>(h, K, a.divisor, a.rdivisor);
#pragma unroll
    for (int i = 0; i < K; ++i) {
        w[i] = a.min_w + a.om_w * w[i];
        h[i] = a.min_h + a.om_hk * h[i];
    }
    // walk: knots (double-accumulated prefix sums, last forced to 1), slopes h/w, interior
    // derivatives 2*min(|s_i|, |s_i+1|, weighted mean) (:116-135); the bin's left / right
    // derivative, slope, width and left knots are captured when the search passes it
    double acc_w = 0.0, acc_h = 0.0;
    float pw = 0.0f, ph = 0.0f;                 // knot_i of cumwidths / cumheights
    float d_prev = 0.0f;                        // derivative at knot_i
    float s_prev = 0.0f, w_prev = 0.0f;         // slope / width of bin i-1
    int k = -1;
    float lcw = 0.0f, rcw = 0.0f, lch = 0.0f, bw = 0.0f, bs = 0.0f, dl = 0.0f, dr = 0.0f;
    bool take_right = false;                    // the bin picked in the previous iteration still needs its right derivative
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const float wi = w[i], si = h[i] / wi;
        float d_i;                              // derivative at knot_i
        if (i == 0) {
            d_i = ((1.0f / (1.0f + expf(-udl))) * 3.0f) * si;
        } else {
            const float m1 = fminf(fabsf(s_prev), fabsf(si));
            const float m2 = (0.5f * (wi * s_prev + w_prev * si)) / (w_prev + wi);
            d_i = fminf(m1, m2) * (sign_of(s_prev) + sign_of(si));
        }
        if (take_right) {
            dr = d_i;
            take_right = false;
        }
        acc_w += (double)wi;
        acc_h += (double)h[i];
        const bool last = i == K - 1;
        const float nw = last ? 1.0f : (float)acc_w, nh = last ? 1.0f : (float)acc_h;
        if (u >= (INVERSE ? ph : pw)) {
            k = i;
            lcw = pw;
            rcw = nw;
            lch = ph;
            bw = wi;
            bs = si;
            dl = d_i;
            take_right = true;
        }
        pw = nw;
        ph = nh;
        d_prev = d_i;
        s_prev = si;
        w_prev = wi;
    }
    if (take_right) dr = ((1.0f / (1.0f + expf(-udr))) * 3.0f) * s_prev;  // the last bin: right boundary derivative
    (void)d_prev;
    if (k < 0 || u >= 1.0f + 1e-6f) {
        y = x;
        lad = 0.0f;
        return NFA_STATUS_OUTSIDE_DOMAIN;
    }
    const float ca = ((dl + dr) - 2.0f * bs) / (bw * bw);
    const float cb = ((3.0f * bs - 2.0f * dl) - dr) / bw;
    const float cc = dl, cd = lch;
    if (INVERSE) {
        bool almost_quadratic;
        const float out = cubic_inverse_root(ca, cb, cc, cd, u, lcw, rcw, almost_quadratic);
        (void)almost_quadratic;
        const float so = out - lcw;
        lad = -logf(((3.0f * ca) * (so * so) + (2.0f * cb) * so) + cc);
        y = out * a.span_in + a.left;
    } else {
        const float si = u - lcw;
        const float out = ((ca * ((si * si) * si) + cb * (si * si)) + cc * si) + cd;
        lad = logf(((3.0f * ca) * (si * si) + (2.0f * cb) * si) + cc);
        y = out * a.span_out + a.bottom;
    }
    return 0;
}

template <int KIND, int KT, bool INVERSE>
__global__ void __launch_bounds__(kBlock) spline_lq_kernel(const LqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int K = a.K;
    const int P = KIND == kLinear ? K : (KIND == kCubic ? 2 * K + 2 : K + a.nh);
    int my_status = 0;
    const int64_t num_tiles = (a.n + a.T - 1) / a.T;
    for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int64_t i0 = tile * a.T;
        const int cnt = (int)((a.n - i0) < a.T ? (a.n - i0) : a.T);
        // every lane owns `slot` floats: [K widths / pdf logits][K+1 heights]
        float* mine = lds + tid * a.slot;
        const int hshift = (KIND == kQuadratic && a.nh == K - 1) ? 1 : 0;
        if (a.packed && KT > 0) {
            // one contiguous [n, P] view, K known at compile time: the tile's image is staged as it
            // is (16-byte coalesced loads) and every lane copies its row to registers below
            const int mp = tile_load(a.a0 + i0 * P, cnt * P, lds, tid);
            __syncthreads();
            mine = lds + mp + tid * P;
        } else if (a.packed) {
            const float* src = a.a0 + i0 * P;
            const int total = cnt * P;
            for (int e = tid; e < total; e += blockDim.x) {
                const int who = e / P, q = e - who * P;
                lds[who * a.slot + (q < K ? q : q + hshift)] = src[e];
            }
            __syncthreads();
        } else if (tid < cnt) {
            const int64_t i = i0 + tid;
            for (int q = 0; q < K; ++q) mine[q] = a.a0[i * a.s0 + q];
            if (KIND == kQuadratic)
                for (int q = 0; q < a.nh; ++q) mine[K + hshift + q] = a.a1[i * a.s1 + q];
            if (KIND == kCubic) {
                for (int q = 0; q < K; ++q) mine[K + q] = a.a1[i * a.s1 + q];
                mine[2 * K] = a.a2[i * a.s2];
                mine[2 * K + 1] = a.a3[i * a.s3];
            }
        }
        if (tid < cnt) {
            const float x = a.x[i0 + tid];
            float y = x, l = 0.0f;
            // linear tails: elements outside [-B, B] (NaN included) pass through (linear.py:12-22)
            const bool inside = x >= a.left && x <= a.right;
            if (inside) {
                constexpr int kRegs = KIND == kLinear ? (KT > 0 ? KT : 1) : 2 * (KT > 0 ? KT : 1) + 2;
                if (KT > 0) {  // K known at compile time: the lane's logits move to registers
                    float reg[kRegs];
                    if (a.packed && KIND == kQuadratic) {  // heights shifted by one slot when two are derived
#pragma unroll
                        for (int q = 0; q < 2 * KT + 1; ++q) {
                            const int srcq = q < KT ? q : q - hshift;
                            reg[q] = (q >= KT && (srcq < KT || srcq >= P)) ? 0.0f : mine[srcq];
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < (KIND == kLinear ? KT : (KIND == kCubic ? 2 * KT + 2 : 2 * KT + 1)); ++q)
                            reg[q] = mine[q];
                    }
                    if (KIND == kLinear) my_status |= linear_eval<KT, INVERSE>(x, reg, a, y, l);
                    else if (KIND == kQuadratic) my_status |= quadratic_eval<KT, INVERSE>(x, reg, reg + KT, a, y, l);
                    else my_status |= cubic_eval<KT, INVERSE>(x, reg, reg + KT, reg[2 * KT], reg[2 * KT + 1], a, y, l);
                } else {
                    if (KIND == kLinear) my_status |= linear_eval<0, INVERSE>(x, mine, a, y, l);
                    else if (KIND == kQuadratic) my_status |= quadratic_eval<0, INVERSE>(x, mine, mine + K, a, y, l);
                    else my_status |= cubic_eval<0, INVERSE>(x, mine, mine + K, mine[2 * K], mine[2 * K + 1], a, y, l);
                }
            } else if (!a.unconstrained) {
                my_status |= NFA_STATUS_OUTSIDE_DOMAIN;  // linear.py:47-48 / quadratic.py:66-67
            }
            a.y[i0 + tid] = y;
            a.lad[i0 + tid] = l;
        }
        __syncthreads();  // before the next tile overwrites the slots
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

static int launch_lq(LqArgs& a, int kind, int inverse, hipStream_t st) {
    const int K = a.K;
    a.slot = (kind == kLinear ? K : (kind == kCubic ? 2 * K + 2 : 2 * K + 1)) | 1;  // odd stride: conflict-free per-lane walks
    int T = kBlock;
    while (T > 32 && (size_t)T * a.slot * 4 > (size_t)64 * 1024) T >>= 1;
    if ((size_t)T * a.slot * 4 > (size_t)64 * 1024) return NFA_ERR_UNSUPPORTED;
    a.T = T;
    const size_t lds = (size_t)T * a.slot * 4 + 64;  // + slack of the staged tile image (tile_load)
    const int64_t tiles = (a.n + T - 1) / T;
    int per_cu = (int)((size_t)(160 * 1024) / (lds + 256));
    per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
    int64_t g = (int64_t)device_cu_count() * per_cu;
    if (g > tiles) g = tiles;
    const dim3 grid((unsigned)g), block((unsigned)T);
#define NFA_LQ_LAUNCH(KIND_, KT_)                                                                       \
    do {                                                                                              \
        if (inverse) hipLaunchKernelGGL((spline_lq_kernel<KIND_, KT_, true>), grid, block, lds, st, a); \
        else hipLaunchKernelGGL((spline_lq_kernel<KIND_, KT_, false>), grid, block, lds, st, a);       \
    } while (0)
    if (kind == kLinear) {
        if (K == 8) NFA_LQ_LAUNCH(kLinear, 8);
        else if (K == 10) NFA_LQ_LAUNCH(kLinear, 10);   // the reference's default num_bins
        else NFA_LQ_LAUNCH(kLinear, 0);
    } else if (kind == kQuadratic) {
        if (K == 8) NFA_LQ_LAUNCH(kQuadratic, 8);
        else if (K == 10) NFA_LQ_LAUNCH(kQuadratic, 10);
        else NFA_LQ_LAUNCH(kQuadratic, 0);
    } else {
        if (K == 8) NFA_LQ_LAUNCH(kCubic, 8);
        else if (K == 10) NFA_LQ_LAUNCH(kCubic, 10);
        else NFA_LQ_LAUNCH(kCubic, 0);
    }
#undef NFA_LQ_LAUNCH
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

static int fill_common(LqArgs& a, const nfa_rqs_spec* spec) {
    if (!spec) return NFA_ERR_INVALID_ARGUMENT;
    if (spec->num_bins < 1 || spec->num_bins > 4096) return NFA_ERR_INVALID_ARGUMENT;
    if (spec->tails != NFA_TAILS_NONE && spec->tails != NFA_TAILS_LINEAR) return NFA_ERR_INVALID_ARGUMENT;
    a.K = spec->num_bins;
    a.unconstrained = spec->tails == NFA_TAILS_LINEAR;
    a.left = (float)spec->left;
    a.right = (float)spec->right;
    a.bottom = (float)spec->bottom;
    a.top = (float)spec->top;
    a.span_in = (float)(spec->right - spec->left);
    a.span_out = (float)(spec->top - spec->bottom);
    a.log_bin_width = (float)log(1.0 / (double)spec->num_bins);
    a.min_w = (float)spec->min_bin_width;
    a.min_h = (float)spec->min_bin_height;
    a.om_w = (float)(1.0 - spec->min_bin_width * spec->num_bins);
    a.om_h = (float)(1.0 - spec->min_bin_height);
    a.om_hk = (float)(1.0 - spec->min_bin_height * spec->num_bins);
    a.divisor = (float)spec->wh_divisor;
    a.rdivisor = a.divisor != 0.0f ? 1.0f / a.divisor : 0.0f;
    return NFA_OK;
}


// ------------------------------------------------------------------------------------------
// Backward of the linear and quadratic spline functionals (the reference differentiates them by
// autograd through the eager ops of splines/linear.py:40-105 and splines/quadratic.py:55-159).
// One lane per element: it rebuilds the spline from the logits with the forward kernels' own
// arithmetic (same knots, same bin), then applies the closed-form adjoints of the map inside the
// bin, of the prefix sums, of the height normalisation and of the softmax / softplus.
// Logit rows are dense ([n, K], [n, nh]); the lane's working set lives in its LDS slot.
struct LqBwdArgs {
    const float* x;
    const float* a0;  // [n, K] pdf / width logits
    const float* a1;  // [n, nh] height logits (quadratic), [n, K] (cubic)
    const float* a2;  // [n] left / right boundary-derivative logits (cubic)
    const float* a3;
    const float* gy;  // [n] upstream gradient of outputs
    const float* gl;  // [n] upstream gradient of logabsdet (may be null)
    float* gx;        // [n]
    float* g0;        // [n, K]
    float* g1;        // [n, nh]
    float* g2;        // [n] (cubic)
    float* g3;
    int64_t n;
    LqArgs f;         // the forward description (box, minimums, divisor)
};

__device__ __forceinline__ float sigmoid_of(float v) { return v > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-v)); }

template <int KT, bool INVERSE>
__global__ void __launch_bounds__(kBlock) linear_spline_backward_kernel(const LqBwdArgs b) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LqArgs& a = b.f;
    const int K = KT > 0 ? KT : a.K;            // KT > 0: the bin loops unroll, slot offsets are immediates
    const int slot = KT > 0 ? (KT | 1) : a.slot;
    float* pdf = lds + threadIdx.x * slot;  // K probabilities, then their adjoints in place
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < b.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = b.x[i];
        const float gy = b.gy[i], gl = b.gl ? b.gl[i] : 0.0f;
        float* g0 = b.g0 + i * K;
        if (!(x >= a.left && x <= a.right)) {  // tails (identity) or outside the box: no logit gradient
            b.gx[i] = gy;
            for (int q = 0; q < K; ++q) g0[q] = 0.0f;
            continue;
        }
        for (int q = 0; q < K; ++q) pdf[q] = b.a0[i * K + q];
        softmax_in_place<KT>(pdf, K, 0.0f, 0.0f);
        const float u = INVERSE ? (x - a.bottom) / a.span_out : (x - a.left) / a.span_in;
        int k = -1;
        float g_u = 0.0f, g_lo = 0.0f, g_hi = 0.0f, g_pk = 0.0f;  // adjoints of u, cdf[k], cdf[k+1], pdf[k]
        bool hi_is_sum = false;
        if (!INVERSE) {
            const float pos = u * (float)K;
            k = (int)floorf(pos);
            k = k >= K ? K - 1 : (k < 0 ? 0 : k);
            const float alpha = pos - (float)k;
            double acc = 0.0;
            for (int q = 0; q < k; ++q) acc += (double)pdf[q];
            const float pk = pdf[k];
            const float out = (float)acc + alpha * pk;
            const float g_out = (out < 0.0f || out > 1.0f) ? 0.0f : gy * a.span_out;  // torch.clamp
            g_lo = g_out;
            g_pk = g_out * alpha + gl / pk;
            g_u = g_out * pk * (float)K;
        } else {
            double acc = 0.0;
            float prev = 0.0f, lo = 0.0f, hi = 0.0f;
            for (int q = 0; q < K; ++q) {
                acc += (double)pdf[q];
                const float next = (q == K - 1) ? 1.0f + 1e-6f : (float)acc;
                if (u >= prev) {
                    k = q;
                    lo = prev;
                    hi = next;
                }
                prev = next;
            }
            if (k < 0 || u >= prev) {  // the forward pass flagged it
                b.gx[i] = gy;
                for (int q = 0; q < K; ++q) g0[q] = 0.0f;
                continue;
            }
            const float step = 1.0f / (float)K;
            const int half = (K + 1) / 2;
            const float b0 = (k < half) ? (float)k * step : 1.0f - (float)(K - k) * step;
            const float b1 = (k + 1 < half) ? (float)(k + 1) * step : 1.0f - (float)(K - k - 1) * step;
            const float db = b1 - b0;
            const float slope = (hi - lo) / db;
            const float offset = hi - slope * b1;
            const float q_ = (u - offset) / slope;
            const float g_out = (q_ < 0.0f || q_ > 1.0f) ? 0.0f : gy * a.span_in;
            g_u = g_out / slope;
            const float g_off = -g_out / slope;
            float g_slope = -g_out * q_ / slope - gl / slope;  // lad = -log(slope)
            g_hi = g_off;
            g_slope -= g_off * b1;
            g_hi += g_slope / db;
            g_lo = -g_slope / db;
            hi_is_sum = k < K - 1;  // the last knot is the constant 1 (+1e-6)
        }
        // cdf[k] = sum_{q<k} pdf[q], cdf[k+1] = sum_{q<=k} pdf[q]; then the softmax
        float dot = 0.0f;
        for (int q = 0; q < K; ++q) {
            float g = 0.0f;
            if (q < k) g += g_lo;
            if (q <= k && hi_is_sum) g += g_hi;
            if (q == k) g += g_pk;
            dot += g * pdf[q];
        }
        for (int q = 0; q < K; ++q) {
            float g = 0.0f;
            if (q < k) g += g_lo;
            if (q <= k && hi_is_sum) g += g_hi;
            if (q == k) g += g_pk;
            g0[q] = pdf[q] * (g - dot);
        }
        b.gx[i] = INVERSE ? g_u / a.span_out : g_u / a.span_in;
    }
}

template <int KT, bool DERIVED, bool INVERSE>
__global__ void __launch_bounds__(kBlock) quadratic_spline_backward_kernel(const LqBwdArgs b) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LqArgs& a = b.f;
    // KT > 0: K and the number of height logits (DERIVED: K - 1, else K + 1) are compile-time constants --
    // the bin loops unroll and every slot offset is an immediate; KT == 0 reads both from the arguments
    const int K = KT > 0 ? KT : a.K;
    const int nh = KT > 0 ? (DERIVED ? KT - 1 : KT + 1) : a.nh;
    const bool derived = KT > 0 ? DERIVED : nh == K - 1;
    const int hs = derived ? 1 : 0;  // height logit q sits at slot q + hs
    const int slot = KT > 0 ? ((5 * KT + 3) | 1) : a.slot;
    // per lane: W[K] | H[K+1] (unnormalised) | Hn[K+1] | gW[K] | gHn[K+1] (later gH)
    float* W = lds + threadIdx.x * slot;
    float* H = W + K;
    float* Hn = H + K + 1;
    float* gW = Hn + K + 1;
    float* gH = gW + K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < b.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = b.x[i];
        const float gy = b.gy[i], gl = b.gl ? b.gl[i] : 0.0f;
        float* g0 = b.g0 + i * K;
        float* g1 = b.g1 + i * nh;
        bool live = x >= a.left && x <= a.right;
        int k = -1;
        float u = 0.0f, c0 = 0.0f, l0 = 0.0f, area = 1.0f, cst = 0.0f, cden = 1.0f;
        if (live) {
            u = INVERSE ? (x - a.bottom) / a.span_out : (x - a.left) / a.span_in;
            for (int q = 0; q < K; ++q) W[q] = b.a0[i * K + q];
            softmax_in_place<KT>(W, K, a.divisor, a.rdivisor);
            for (int q = 0; q < K; ++q) W[q] = a.min_w + a.om_w * W[q];
            for (int q = 0; q < nh; ++q) {
                float v = b.a1[i * nh + q];
                if (a.divisor != 0.0f) v = div_with_rcp(v, a.divisor, a.rdivisor);
                H[q + hs] = softplus_beta(v, 1.0f) + 1e-3f;
            }
            if (derived) {
                const float fw = 0.5f * W[0], lw = 0.5f * W[K - 1];
                float s = 0.0f;
                for (int q = 1; q + 1 < K; ++q) s += ((H[q] + H[q + 1]) / 2.0f) * W[q];
                const float num = (0.5f * fw) * H[1] + (0.5f * lw) * H[K - 1] + s;
                cden = (1.0f - 0.5f * fw) - 0.5f * lw;
                cst = num / cden;
                H[0] = cst;
                H[K] = cst;
            }
            area = 0.0f;
            for (int q = 0; q < K; ++q) area += ((H[q] + H[q + 1]) / 2.0f) * W[q];
            const float rarea = rcp_refined(area);
            for (int q = 0; q <= K; ++q) Hn[q] = a.min_h + a.om_h * div_with_rcp(H[q], area, rarea);
            double acc_c = 0.0, acc_l = 0.0;
            float pc = 0.0f, pl = 0.0f;
            for (int q = 0; q < K; ++q) {
                acc_c += (double)(((Hn[q] + Hn[q + 1]) / 2.0f) * W[q]);
                acc_l += (double)W[q];
                const bool last = q == K - 1;
                const float nc = last ? 1.0f : (float)acc_c, nl = last ? 1.0f : (float)acc_l;
                if (u >= (INVERSE ? pc : pl)) {
                    k = q;
                    c0 = pc;
                    l0 = pl;
                }
                pc = nc;
                pl = nl;
            }
            if (k < 0 || u >= 1.0f + 1e-6f) live = false;
        }
        if (!live) {
            b.gx[i] = gy;
            for (int q = 0; q < K; ++q) g0[q] = 0.0f;
            for (int q = 0; q < nh; ++q) g1[q] = 0.0f;
            continue;
        }
        for (int q = 0; q < K; ++q) gW[q] = 0.0f;
        for (int q = 0; q <= K; ++q) gH[q] = 0.0f;  // adjoints of the normalised heights first
        const float bw = W[k], hl = Hn[k], hr = Hn[k + 1], dH = hr - hl;
        const float qa = (0.5f * dH) * bw, qb = hl * bw;
        float g_u, g_c0, g_l0, g_a, g_b, g_dH, g_hl, g_bw;
        if (!INVERSE) {
            const float alpha = (u - l0) / bw;
            const float out = (qa * (alpha * alpha) + qb * alpha) + c0;
            const float g_out = (out < 0.0f || out > 1.0f) ? 0.0f : gy * a.span_out;
            const float D = alpha * dH + hl;
            const float g_D = gl / D;
            const float g_alpha = g_out * (2.0f * qa * alpha + qb) + g_D * dH;
            g_a = g_out * alpha * alpha;
            g_b = g_out * alpha;
            g_c0 = g_out;
            g_dH = g_D * alpha;
            g_hl = g_D;
            g_bw = -g_alpha * alpha / bw;
            g_l0 = -g_alpha / bw;
            g_u = g_alpha / bw;
        } else {
            const float c_ = c0 - u;
            const float disc = qb * qb - (4.0f * qa) * c_;
            const float r = sqrtf(disc);
            const float alpha = (-qb + r) / (2.0f * qa);
            const float out = alpha * bw + l0;
            const float g_out = (out < 0.0f || out > 1.0f) ? 0.0f : gy * a.span_in;
            const float D = alpha * dH + hl;
            const float g_D = -gl / D;
            const float g_alpha = g_out * bw + g_D * dH;
            g_bw = g_out * alpha;
            g_l0 = g_out;
            g_dH = g_D * alpha;
            g_hl = g_D;
            // alpha is the root of qa s^2 + qb s + c_ = 0 whose slope there is 2 qa alpha + qb = r: differentiated
            // implicitly, d alpha = -(alpha^2 d qa + alpha d qb + d c_) / r.  Equal to the chain through the closed
            // form (-qb + r) / (2 qa), without its cancellation (-1 + qb / r) for flat bins (qa -> 0): measured on
            // the fixtures against float64, up to 100 x closer than the reference's own fp32 autograd there.
            const float t = g_alpha / r;
            g_a = -t * (alpha * alpha);
            g_b = -t * alpha;
            const float g_c_ = -t;
            g_c0 = g_c_;
            g_u = -g_c_;
        }
        // a = 0.5 dH bw, b = hl bw
        g_dH += 0.5f * bw * g_a;
        g_bw += 0.5f * dH * g_a + hl * g_b;
        g_hl += bw * g_b;
        gH[k + 1] += g_dH;
        gH[k] += g_hl - g_dH;
        gW[k] += g_bw;
        // c0 = sum_{q<k} 0.5 (Hn[q] + Hn[q+1]) W[q],  l0 = sum_{q<k} W[q]
        for (int q = 0; q < k; ++q) {
            gH[q] += 0.5f * W[q] * g_c0;
            gH[q + 1] += 0.5f * W[q] * g_c0;
            gW[q] += 0.5f * (Hn[q] + Hn[q + 1]) * g_c0 + g_l0;
        }
        // Hn = min_h + om_h H / area,  area = sum 0.5 (H[q] + H[q+1]) W[q]
        float g_area = 0.0f;
        const float h_scale = a.om_h / area;   // (one division; gradients are not held to the reference's rounding)
        for (int q = 0; q <= K; ++q) {
            g_area -= gH[q] * H[q];
            gH[q] = gH[q] * h_scale;  // now the adjoint of H[q] (first part)
        }
        g_area = a.om_h * g_area / (area * area);
        for (int q = 0; q < K; ++q) {
            gH[q] += 0.5f * W[q] * g_area;
            gH[q + 1] += 0.5f * W[q] * g_area;
            gW[q] += 0.5f * (H[q] + H[q + 1]) * g_area;
        }
        if (derived) {  // H[0] = H[K] = num / cden
            const float g_c = gH[0] + gH[K];
            const float g_num = g_c / cden, g_den = -g_c * cst / cden;
            gW[0] += 0.25f * H[1] * g_num - 0.25f * g_den;
            gW[K - 1] += 0.25f * H[K - 1] * g_num - 0.25f * g_den;
            gH[1] += 0.25f * W[0] * g_num;
            gH[K - 1] += 0.25f * W[K - 1] * g_num;
            for (int q = 1; q + 1 < K; ++q) {
                gH[q] += 0.5f * W[q] * g_num;
                gH[q + 1] += 0.5f * W[q] * g_num;
                gW[q] += 0.5f * (H[q] + H[q + 1]) * g_num;
            }
        }
        const float sc = a.divisor != 0.0f ? a.rdivisor : 1.0f;
        for (int q = 0; q < nh; ++q) {
            float v = b.a1[i * nh + q];
            if (a.divisor != 0.0f) v = div_with_rcp(v, a.divisor, a.rdivisor);
            g1[q] = gH[q + hs] * sigmoid_of(v) * sc;
        }
        // W = min_w + om_w softmax(logits / divisor)
        float dot = 0.0f;
        const float r_om_w = 1.0f / a.om_w;
        for (int q = 0; q < K; ++q) {
            const float sw = (W[q] - a.min_w) * r_om_w;   // the softmax value back out of the width
            W[q] = sw;
            dot += a.om_w * gW[q] * sw;
        }
        for (int q = 0; q < K; ++q) g0[q] = W[q] * (a.om_w * gW[q] - dot) * sc;
        b.gx[i] = INVERSE ? g_u / a.span_out : g_u / a.span_in;
    }
}

// Cubic spline (splines/cubic.py:63-267).  Only the searched bin's cubic matters: its coefficients
// depend on the prefix sums below it, on its own width / height and on the knot derivatives at its
// two ends, each a function of the two bins that meet there (or of a boundary logit).  The inverse
// direction differentiates the root implicitly (ds/dp = -(dF/dp) / F'(s)), which equals the derivative
// of whichever closed-form root the forward pass took; the "almost quadratic" override (:219-226)
// has its own explicit formula and is differentiated as such.
template <int KT, bool INVERSE>
__global__ void __launch_bounds__(kBlock) cubic_spline_backward_kernel(const LqBwdArgs b) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LqArgs& a = b.f;
    const int K = KT > 0 ? KT : a.K;            // KT > 0: the bin loops unroll, slot offsets are immediates
    const int slot = KT > 0 ? ((4 * KT) | 1) : a.slot;
    float* w = lds + threadIdx.x * slot;  // widths | heights | their adjoints
    float* h = w + K;
    float* gw = h + K;
    float* gh = gw + K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < b.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = b.x[i];
        const float gy = b.gy[i], gl = b.gl ? b.gl[i] : 0.0f;
        float* g0 = b.g0 + i * K;
        float* g1 = b.g1 + i * K;
        bool live = x >= a.left && x <= a.right;
        int k = -1;
        float u = 0.0f, lcw = 0.0f, rcw = 0.0f, lch = 0.0f;
        if (live) {
            u = INVERSE ? (x - a.bottom) / a.span_out : (x - a.left) / a.span_in;
            for (int q = 0; q < K; ++q) {
                w[q] = b.a0[i * K + q];
                h[q] = b.a1[i * K + q];
            }
            softmax_in_place<KT>(w, K, a.divisor, a.rdivisor);
            softmax_in_place<KT>(h, K, a.divisor, a.rdivisor);
            for (int q = 0; q < K; ++q) {
                w[q] = a.min_w + a.om_w * w[q];
                h[q] = a.min_h + a.om_hk * h[q];
            }
            double acc_w = 0.0, acc_h = 0.0;
            float pw = 0.0f, ph = 0.0f;
            for (int q = 0; q < K; ++q) {
                acc_w += (double)w[q];
                acc_h += (double)h[q];
                const bool last = q == K - 1;
                const float nw = last ? 1.0f : (float)acc_w, nh = last ? 1.0f : (float)acc_h;
                if (u >= (INVERSE ? ph : pw)) {
                    k = q;
                    lcw = pw;
                    rcw = nw;
                    lch = ph;
                }
                pw = nw;
                ph = nh;
            }
            if (k < 0 || u >= 1.0f + 1e-6f) live = false;
        }
        if (!live) {
            b.gx[i] = gy;
            for (int q = 0; q < K; ++q) {
                g0[q] = 0.0f;
                g1[q] = 0.0f;
            }
            b.g2[i] = 0.0f;
            b.g3[i] = 0.0f;
            continue;
        }
        const float udl = b.a2[i], udr = b.a3[i];
        const float sgl = 1.0f / (1.0f + expf(-udl)), sgr = 1.0f / (1.0f + expf(-udr));
        // knot derivative at the left end of bin j (j = 0: boundary logit) -- value only
        auto knot_derivative = [&](int j) -> float {
            if (j == 0) return (sgl * 3.0f) * (h[0] / w[0]);
            if (j == K) return (sgr * 3.0f) * (h[K - 1] / w[K - 1]);
            const float sp = h[j - 1] / w[j - 1], sn = h[j] / w[j];
            const float m1 = fminf(fabsf(sp), fabsf(sn));
            const float m2 = (0.5f * (w[j] * sp + w[j - 1] * sn)) / (w[j - 1] + w[j]);
            return fminf(m1, m2) * (sign_of(sp) + sign_of(sn));
        };
        const float bw = w[k], bs = h[k] / w[k];
        const float dl = knot_derivative(k), dr = knot_derivative(k + 1);
        const float ca = ((dl + dr) - 2.0f * bs) / (bw * bw);
        const float cb = ((3.0f * bs - 2.0f * dl) - dr) / bw;
        const float cc = dl, cd = lch;
        float g_ca, g_cb, g_cc, g_cd, g_u, g_lcw;
        if (!INVERSE) {
            const float s = u - lcw;
            const float P = ((3.0f * ca) * (s * s) + (2.0f * cb) * s) + cc;  // f'(s)
            const float g_out = gy * a.span_out;
            const float g_P = gl / P;
            const float g_s = g_out * P + g_P * (6.0f * ca * s + 2.0f * cb);
            g_ca = g_out * s * s * s + g_P * 3.0f * s * s;
            g_cb = g_out * s * s + g_P * 2.0f * s;
            g_cc = g_out * s + g_P;
            g_cd = g_out;
            g_u = g_s;
            g_lcw = -g_s;
        } else {
            bool almost_quadratic;
            const float out = cubic_inverse_root(ca, cb, cc, cd, u, lcw, rcw, almost_quadratic);
            const float s = out - lcw;
            const float P = ((3.0f * ca) * (s * s) + (2.0f * cb) * s) + cc;
            const float g_out = gy * a.span_in;
            const float g_P = -gl / P;  // lad = -log P
            const float g_s = g_out + g_P * (6.0f * ca * s + 2.0f * cb);
            g_ca = g_P * 3.0f * s * s;
            g_cb = g_P * 2.0f * s;
            g_cc = g_P;
            g_cd = 0.0f;
            g_lcw = g_out;  // out = lcw + s
            if (almost_quadratic) {  // s = (-cc + sqrt(cc^2 - 4 cb (cd - u))) / (2 cb)
                const float r = sqrtf(cc * cc - (4.0f * cb) * (cd - u));
                g_cc += g_s * (-1.0f + cc / r) / (2.0f * cb);
                g_cb += g_s * (-(cd - u) / (r * cb) - s / cb);
                g_cd += -g_s / r;
                g_u = g_s / r;
            } else {  // F(s) = ca s^3 + cb s^2 + cc s + cd - u = 0
                const float t = g_s / P;
                g_ca -= t * s * s * s;
                g_cb -= t * s * s;
                g_cc -= t * s;
                g_cd -= t;
                g_u = t;
            }
        }
        // ca = (dl + dr - 2 bs) / bw^2,  cb = (3 bs - 2 dl - dr) / bw,  cc = dl,  cd = lch
        const float rb = 1.0f / bw, rb2 = rb * rb;
        const float g_dl = g_ca * rb2 - 2.0f * g_cb * rb + g_cc;
        const float g_dr = g_ca * rb2 - g_cb * rb;
        const float g_bs = -2.0f * g_ca * rb2 + 3.0f * g_cb * rb;
        const float g_bw = -2.0f * g_ca * ca * rb - g_cb * cb * rb;
        for (int q = 0; q < K; ++q) {
            gw[q] = q < k ? g_lcw : 0.0f;   // lcw = sum_{q<k} w, lch = sum_{q<k} h
            gh[q] = q < k ? g_cd : 0.0f;
        }
        gw[k] += g_bw - g_bs * bs * rb;     // bs = h[k] / w[k]
        gh[k] += g_bs * rb;
        float g_udl = 0.0f, g_udr = 0.0f;
        // adjoint of the knot derivative at the left end of bin j
        auto knot_derivative_adjoint = [&](int j, float g) {
            if (j == 0) {
                const float s0 = h[0] / w[0];
                g_udl += g * 3.0f * s0 * sgl * (1.0f - sgl);
                gh[0] += g * 3.0f * sgl / w[0];
                gw[0] -= g * 3.0f * sgl * s0 / w[0];
                return;
            }
            if (j == K) {
                const float s0 = h[K - 1] / w[K - 1];
                g_udr += g * 3.0f * s0 * sgr * (1.0f - sgr);
                gh[K - 1] += g * 3.0f * sgr / w[K - 1];
                gw[K - 1] -= g * 3.0f * sgr * s0 / w[K - 1];
                return;
            }
            const float wp = w[j - 1], wn = w[j];
            const float sp = h[j - 1] / wp, sn = h[j] / wn;
            const float sg = sign_of(sp) + sign_of(sn);
            const float m1 = fminf(fabsf(sp), fabsf(sn));
            const float den = wp + wn;
            const float m2 = (0.5f * (wn * sp + wp * sn)) / den;
            float g_sp = 0.0f, g_sn = 0.0f;
            const float gm = g * sg;
            if (m1 < m2 || (m1 == m2)) {
                const float share = m1 == m2 ? 0.5f : 1.0f;  // torch.min splits ties
                if (fabsf(sp) < fabsf(sn)) g_sp += share * gm * sign_of(sp);
                else if (fabsf(sn) < fabsf(sp)) g_sn += share * gm * sign_of(sn);
                else {
                    g_sp += 0.5f * share * gm * sign_of(sp);
                    g_sn += 0.5f * share * gm * sign_of(sn);
                }
            }
            if (m2 < m1 || (m1 == m2)) {
                const float share = m1 == m2 ? 0.5f : 1.0f;
                const float gm2 = share * gm;
                g_sp += gm2 * 0.5f * wn / den;
                g_sn += gm2 * 0.5f * wp / den;
                gw[j] += gm2 * (0.5f * sp / den - m2 / den);
                gw[j - 1] += gm2 * (0.5f * sn / den - m2 / den);
            }
            gh[j - 1] += g_sp / wp;
            gw[j - 1] -= g_sp * sp / wp;
            gh[j] += g_sn / wn;
            gw[j] -= g_sn * sn / wn;
        };
        knot_derivative_adjoint(k, g_dl);
        knot_derivative_adjoint(k + 1, g_dr);
        // w = min_w + om_w softmax(logits / divisor), h = min_h + om_hk softmax(logits / divisor)
        const float sc = a.divisor != 0.0f ? a.rdivisor : 1.0f;
        float dot_w = 0.0f, dot_h = 0.0f;
        const float r_om_w = 1.0f / a.om_w, r_om_hk = 1.0f / a.om_hk;
        for (int q = 0; q < K; ++q) {   // the softmax values back out of the widths / heights, kept in place
            w[q] = (w[q] - a.min_w) * r_om_w;
            h[q] = (h[q] - a.min_h) * r_om_hk;
            dot_w += a.om_w * gw[q] * w[q];
            dot_h += a.om_hk * gh[q] * h[q];
        }
        for (int q = 0; q < K; ++q) {
            g0[q] = w[q] * (a.om_w * gw[q] - dot_w) * sc;
            g1[q] = h[q] * (a.om_hk * gh[q] - dot_h) * sc;
        }
        b.g2[i] = g_udl;
        b.g3[i] = g_udr;
        b.gx[i] = INVERSE ? g_u / a.span_out : g_u / a.span_in;
    }
}

static int launch_lq_backward(LqBwdArgs& b, int kind, int inverse, hipStream_t st) {
    const int K = b.f.K;
    b.f.slot = (kind == kLinear ? K : (kind == kCubic ? 4 * K : 5 * K + 3)) | 1;  // odd stride: conflict-free per-lane walks
    int T = kBlock;
    while (T > 64 && (size_t)T * b.f.slot * 4 > (size_t)64 * 1024) T >>= 1;
    if ((size_t)T * b.f.slot * 4 > (size_t)64 * 1024) return NFA_ERR_UNSUPPORTED;
    const size_t lds = (size_t)T * b.f.slot * 4;
    int64_t g = (b.n + T - 1) / T;
    const int64_t cap = (int64_t)device_cu_count() * 8;
    if (g > cap) g = cap;
    const dim3 grid((unsigned)g), block((unsigned)T);
    // K = 8 (the benchmark configurations) and K = 10 (the reference's default num_bins) are compiled with
    // the bin count as a constant; every other K runs the generic instance (same arithmetic, same results)
#define NFA_LQB(KERNEL_, ...)                                                                     \
    do {                                                                                          \
        if (inverse) hipLaunchKernelGGL((KERNEL_<__VA_ARGS__, true>), grid, block, lds, st, b);   \
        else hipLaunchKernelGGL((KERNEL_<__VA_ARGS__, false>), grid, block, lds, st, b);          \
    } while (0)
    if (kind == kLinear) {
        if (K == 8) NFA_LQB(linear_spline_backward_kernel, 8);
        else if (K == 10) NFA_LQB(linear_spline_backward_kernel, 10);
        else NFA_LQB(linear_spline_backward_kernel, 0);
    } else if (kind == kQuadratic) {
        const bool derived = b.f.nh == K - 1;
        if (K == 8 && derived) NFA_LQB(quadratic_spline_backward_kernel, 8, true);
        else if (K == 8) NFA_LQB(quadratic_spline_backward_kernel, 8, false);
        else if (K == 10 && derived) NFA_LQB(quadratic_spline_backward_kernel, 10, true);
        else if (K == 10) NFA_LQB(quadratic_spline_backward_kernel, 10, false);
        else NFA_LQB(quadratic_spline_backward_kernel, 0, false);
    } else {
        if (K == 8) NFA_LQB(cubic_spline_backward_kernel, 8);
        else if (K == 10) NFA_LQB(cubic_spline_backward_kernel, 10);
        else NFA_LQB(cubic_spline_backward_kernel, 0);
    }
#undef NFA_LQB
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_linear_spline_f32(const float* inputs, const float* unnormalized_pdf, int64_t stride,
                                     float* outputs, float* logabsdet, int32_t* status, int64_t n,
                                     const nfa_rqs_spec* spec, int32_t inverse, void* stream) {
    if (n < 0) return NFA_ERR_INVALID_ARGUMENT;
    LqArgs a;
    int rc = fill_common(a, spec);
    if (rc != NFA_OK) return rc;
    if (n == 0) return NFA_OK;
    if (!inputs || !unnormalized_pdf || !outputs || !logabsdet) return NFA_ERR_INVALID_ARGUMENT;
    a.x = inputs;
    a.a0 = unnormalized_pdf;
    a.a1 = a.a2 = a.a3 = nullptr;
    a.s0 = stride;
    a.s1 = a.s2 = a.s3 = 0;
    a.nh = 0;
    a.y = outputs;
    a.lad = logabsdet;
    a.status = status;
    a.n = n;
    a.packed = stride == a.K;
    return launch_lq(a, kLinear, inverse, (hipStream_t)stream);
}

extern "C" int nfa_quadratic_spline_f32(const float* inputs, const float* unnormalized_widths,
                                        int64_t stride_w, const float* unnormalized_heights,
                                        int64_t stride_h, int32_t num_heights, float* outputs,
                                        float* logabsdet, int32_t* status, int64_t n,
                                        const nfa_rqs_spec* spec, int32_t inverse, void* stream) {
    if (n < 0) return NFA_ERR_INVALID_ARGUMENT;
    LqArgs a;
    int rc = fill_common(a, spec);
    if (rc != NFA_OK) return rc;
    if (spec->min_bin_width * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_WIDTH;
    if (spec->min_bin_height * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_HEIGHT;
    if (num_heights != a.K - 1 && num_heights != a.K + 1) return NFA_ERR_INVALID_ARGUMENT;
    if (a.K < 2 && num_heights == a.K - 1) return NFA_ERR_INVALID_ARGUMENT;
    if (n == 0) return NFA_OK;
    if (!inputs || !unnormalized_widths || !unnormalized_heights || !outputs || !logabsdet)
        return NFA_ERR_INVALID_ARGUMENT;
    a.x = inputs;
    a.a0 = unnormalized_widths;
    a.a1 = unnormalized_heights;
    a.a2 = a.a3 = nullptr;
    a.s0 = stride_w;
    a.s1 = stride_h;
    a.s2 = a.s3 = 0;
    a.nh = num_heights;
    a.y = outputs;
    a.lad = logabsdet;
    a.status = status;
    a.n = n;
    const int P = a.K + num_heights;
    a.packed = (unnormalized_heights == unnormalized_widths + a.K) && stride_w == P && stride_h == P;
    return launch_lq(a, kQuadratic, inverse, (hipStream_t)stream);
}

extern "C" int nfa_cubic_spline_f32(const float* inputs, const float* unnormalized_widths, int64_t stride_w,
                                    const float* unnormalized_heights, int64_t stride_h,
                                    const float* unnorm_derivatives_left, int64_t stride_l,
                                    const float* unnorm_derivatives_right, int64_t stride_r,
                                    float* outputs, float* logabsdet, int32_t* status, int64_t n,
                                    const nfa_rqs_spec* spec, int32_t inverse, void* stream) {
    if (n < 0) return NFA_ERR_INVALID_ARGUMENT;
    LqArgs a;
    int rc = fill_common(a, spec);
    if (rc != NFA_OK) return rc;
    if (spec->min_bin_width * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_WIDTH;
    if (spec->min_bin_height * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_HEIGHT;
    if (n == 0) return NFA_OK;
    if (!inputs || !unnormalized_widths || !unnormalized_heights || !unnorm_derivatives_left ||
        !unnorm_derivatives_right || !outputs || !logabsdet)
        return NFA_ERR_INVALID_ARGUMENT;
    a.x = inputs;
    a.a0 = unnormalized_widths;
    a.a1 = unnormalized_heights;
    a.a2 = unnorm_derivatives_left;
    a.a3 = unnorm_derivatives_right;
    a.s0 = stride_w;
    a.s1 = stride_h;
    a.s2 = stride_l;
    a.s3 = stride_r;
    a.nh = a.K;
    a.y = outputs;
    a.lad = logabsdet;
    a.status = status;
    a.n = n;
    const int P = 2 * a.K + 2;
    a.packed = unnormalized_heights == unnormalized_widths + a.K &&
               unnorm_derivatives_left == unnormalized_widths + 2 * a.K &&
               unnorm_derivatives_right == unnormalized_widths + 2 * a.K + 1 && stride_w == P &&
               stride_h == P && stride_l == P && stride_r == P;
    return launch_lq(a, kCubic, inverse, (hipStream_t)stream);
}

extern "C" int nfa_linear_spline_backward_f32(const float* inputs, const float* unnormalized_pdf,
                                              const float* grad_outputs, const float* grad_logabsdet,
                                              float* grad_inputs, float* grad_unnormalized_pdf, int64_t n,
                                              const nfa_rqs_spec* spec, int32_t inverse, void* stream) {
    if (n < 0) return NFA_ERR_INVALID_ARGUMENT;
    LqBwdArgs b;
    int rc = fill_common(b.f, spec);
    if (rc != NFA_OK) return rc;
    if (n == 0) return NFA_OK;
    if (!inputs || !unnormalized_pdf || !grad_outputs || !grad_inputs || !grad_unnormalized_pdf)
        return NFA_ERR_INVALID_ARGUMENT;
    b.x = inputs;
    b.a0 = unnormalized_pdf;
    b.a1 = b.a2 = b.a3 = nullptr;
    b.g2 = b.g3 = nullptr;
    b.gy = grad_outputs;
    b.gl = grad_logabsdet;
    b.gx = grad_inputs;
    b.g0 = grad_unnormalized_pdf;
    b.g1 = nullptr;
    b.n = n;
    b.f.nh = 0;
    return launch_lq_backward(b, kLinear, inverse, (hipStream_t)stream);
}

extern "C" int nfa_quadratic_spline_backward_f32(const float* inputs, const float* unnormalized_widths,
                                                 const float* unnormalized_heights, int32_t num_heights,
                                                 const float* grad_outputs, const float* grad_logabsdet,
                                                 float* grad_inputs, float* grad_unnormalized_widths,
                                                 float* grad_unnormalized_heights, int64_t n,
                                                 const nfa_rqs_spec* spec, int32_t inverse, void* stream) {
    if (n < 0) return NFA_ERR_INVALID_ARGUMENT;
    LqBwdArgs b;
    int rc = fill_common(b.f, spec);
    if (rc != NFA_OK) return rc;
    if (spec->min_bin_width * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_WIDTH;
    if (spec->min_bin_height * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_HEIGHT;
    if (num_heights != b.f.K - 1 && num_heights != b.f.K + 1) return NFA_ERR_INVALID_ARGUMENT;
    if (b.f.K < 2 && num_heights == b.f.K - 1) return NFA_ERR_INVALID_ARGUMENT;
    if (n == 0) return NFA_OK;
    if (!inputs || !unnormalized_widths || !unnormalized_heights || !grad_outputs || !grad_inputs ||
        !grad_unnormalized_widths || !grad_unnormalized_heights)
        return NFA_ERR_INVALID_ARGUMENT;
    b.x = inputs;
    b.a0 = unnormalized_widths;
    b.a1 = unnormalized_heights;
    b.a2 = b.a3 = nullptr;
    b.g2 = b.g3 = nullptr;
    b.gy = grad_outputs;
    b.gl = grad_logabsdet;
    b.gx = grad_inputs;
    b.g0 = grad_unnormalized_widths;
    b.g1 = grad_unnormalized_heights;
    b.n = n;
    b.f.nh = num_heights;
    return launch_lq_backward(b, kQuadratic, inverse, (hipStream_t)stream);
}

extern "C" int nfa_cubic_spline_backward_f32(const float* inputs, const float* unnormalized_widths,
                                             const float* unnormalized_heights,
                                             const float* unnorm_derivatives_left,
                                             const float* unnorm_derivatives_right, const float* grad_outputs,
                                             const float* grad_logabsdet, float* grad_inputs,
                                             float* grad_unnormalized_widths, float* grad_unnormalized_heights,
                                             float* grad_unnorm_derivatives_left,
                                             float* grad_unnorm_derivatives_right, int64_t n,
                                             const nfa_rqs_spec* spec, int32_t inverse, void* stream) {
    if (n < 0) return NFA_ERR_INVALID_ARGUMENT;
    LqBwdArgs b;
    int rc = fill_common(b.f, spec);
    if (rc != NFA_OK) return rc;
    if (spec->min_bin_width * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_WIDTH;
    if (spec->min_bin_height * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_HEIGHT;
    if (n == 0) return NFA_OK;
    if (!inputs || !unnormalized_widths || !unnormalized_heights || !unnorm_derivatives_left ||
        !unnorm_derivatives_right || !grad_outputs || !grad_inputs || !grad_unnormalized_widths ||
        !grad_unnormalized_heights || !grad_unnorm_derivatives_left || !grad_unnorm_derivatives_right)
        return NFA_ERR_INVALID_ARGUMENT;
    b.x = inputs;
    b.a0 = unnormalized_widths;
    b.a1 = unnormalized_heights;
    b.a2 = unnorm_derivatives_left;
    b.a3 = unnorm_derivatives_right;
    b.gy = grad_outputs;
    b.gl = grad_logabsdet;
    b.gx = grad_inputs;
    b.g0 = grad_unnormalized_widths;
    b.g1 = grad_unnormalized_heights;
    b.g2 = grad_unnorm_derivatives_left;
    b.g3 = grad_unnorm_derivatives_right;
    b.n = n;
    b.f.nh = b.f.K;
    return launch_lq_backward(b, kCubic, inverse, (hipStream_t)stream);
}
