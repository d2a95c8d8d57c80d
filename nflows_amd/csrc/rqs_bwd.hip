// Backward of the fused rational-quadratic coupling layer.
//
// The reference trains through autograd over ~520 eager ops per layer
// (examples/moons.ipynb cell 3: loss = -flow.log_prob(x).mean(); loss.backward()).  Here the
// gradient of one layer is ONE kernel that recomputes the spline from (inputs, params) -- nothing
// but the layer's own inputs is kept from the forward pass -- and writes
//     grad_inputs [B, D]        and        grad_params [B, d_t * P]
// given grad_outputs [B, D] and grad_logabsdet [B].
//
// Per spline (bin k, knots a=cw_k, b=cw_{k+1}, c=ch_k, e=ch_{k+1}, derivatives d0, d1):
//   forward map   y   = c + h (delta th^2 + d0 t) / (delta + s t)            rational_quadratic.py:162-181
//                 lad = log(delta^2 (d1 th^2 + 2 delta t + d0 (1-th)^2)) - 2 log(delta + s t)
//   with w=b-a, h=e-c, delta=h/w, th=(x-a)/w, t=th(1-th), s=d0+d1-2 delta.
// Reverse mode through that expression gives adjoints of (x, a, b, c, e, d0, d1); knots are affine
// in the prefix sums of min + (1-min K) softmax(logits / sqrt(H)) (:91-98), derivatives are
// min_d + softplus(logit) (:104), so the adjoints of the 3K-1 (3K+1) logits follow in closed form.
// The inverse direction (x = f^-1(y), lad = -log f'(x)) uses the same partials at the solution x
// through the implicit-function theorem.
//
// Same tile scheme as the generic forward kernel: R whole samples per workgroup, conditioner
// output staged through LDS with 16-byte accesses; each lane overwrites its own P logits in LDS
// with their gradients, which then leave as one coalesced copy.

#include "rqs_math.hpp"

#include <stdlib.h>

namespace nfa {

struct BwdArgs {
    const float* x;
    const float* params;
    const int64_t* tidx;
    const int64_t* perm;
    const int64_t* scatter;
    const float* gout;  // grad wrt outputs [B, D]
    const float* glad;  // grad wrt logabsdet [B], may be null (= 0)
    float* gin;         // grad wrt inputs [B, D]
    float* gparams;     // grad wrt params [B, dt*P]
    int32_t* status;
    int64_t batch;
    int D, dt, R, C;
    FastDiv div_dt, div_D;
    RqsDev sp;
    int off_x, off_gy, off_gx, off_idx;
};

// adjoints of (x, a, b, c, e, d0, d1) for upstream (gy, gl) on (y, lad) of the FORWARD map at x
struct BinAdj {
    float x, a, b, c, e, d0, d1;
};

__device__ __forceinline__ BinAdj forward_map_adjoint(float x, float a, float b, float c, float e, float d0,
                                                      float d1, float gy, float gl) {
    const float w = b - a, h = e - c;
    const float rw = 1.0f / w;
    const float delta = h * rw;
    const float th = (x - a) * rw;
    const float omt = 1.0f - th;
    const float t = th * omt;
    const float s = d0 + d1 - 2.0f * delta;
    const float Bn = delta * th * th + d0 * t;
    const float num = h * Bn;
    const float den = delta + s * t;
    const float rden = 1.0f / den;
    const float A = d1 * th * th + 2.0f * delta * t + d0 * omt * omt;
    const float dn = delta * delta * A;

    BinAdj g;
    g.c = gy;
    const float num_b = gy * rden;
    float den_b = -gy * num * rden * rden - 2.0f * gl * rden;
    const float dn_b = gl / dn;
    float delta_b = dn_b * 2.0f * delta * A;
    const float A_b = dn_b * delta * delta;
    g.d1 = A_b * th * th;
    float th_b = A_b * (2.0f * d1 * th - 2.0f * d0 * omt);
    delta_b += A_b * 2.0f * t;
    float t_b = A_b * 2.0f * delta;
    g.d0 = A_b * omt * omt;
    delta_b += den_b;
    const float s_b = den_b * t;
    t_b += den_b * s;
    float h_b = num_b * Bn;
    const float Bn_b = num_b * h;
    delta_b += Bn_b * th * th;
    th_b += Bn_b * 2.0f * delta * th;
    g.d0 += Bn_b * t;
    t_b += Bn_b * d0;
    g.d0 += s_b;
    g.d1 += s_b;
    delta_b -= 2.0f * s_b;
    th_b += t_b * (1.0f - 2.0f * th);
    g.x = th_b * rw;
    g.a = -th_b * rw;
    float w_b = -th_b * th * rw;
    h_b += delta_b * rw;
    w_b -= delta_b * delta * rw;
    g.b = w_b;
    g.a -= w_b;
    g.e = h_b;
    g.c -= h_b;
    return g;
}

// d lad / d x of the forward map (needed by the inverse direction before the full adjoint)
__device__ __forceinline__ float forward_lad_dx(float x, float a, float b, float c, float e, float d0, float d1) {
    const float w = b - a, h = e - c;
    const float rw = 1.0f / w;
    const float delta = h * rw;
    const float th = (x - a) * rw;
    const float omt = 1.0f - th;
    const float t = th * omt;
    const float s = d0 + d1 - 2.0f * delta;
    const float den = delta + s * t;
    const float A = d1 * th * th + 2.0f * delta * t + d0 * omt * omt;
    const float dA = 2.0f * d1 * th - 2.0f * d0 * omt + 2.0f * delta * (1.0f - 2.0f * th);
    const float dden = s * (1.0f - 2.0f * th);
    return (dA / A - 2.0f * dden / den) * rw;
}

__device__ __forceinline__ float sigmoid_beta(float u, float beta) {
    const float z = u * beta;
    return z > 20.0f ? 1.0f : 1.0f / (1.0f + exp_noclamp(-z));  // d/du [softplus(u; beta)]
}

// One spline: overwrites the lane's P logits in `sl` with their gradients; returns d loss / d input.
template <int KT, bool INVERSE, bool LINEAR>
__device__ __forceinline__ float rqs_backward(float x, float* sl, const RqsDev& sp, float g_out, float gl,
                                              int& status) {
    const int K = KT > 0 ? KT : sp.K;
    const int P = sp.P;
    const float left = LINEAR ? -sp.right : sp.left;
    const float right = sp.right;
    const float bottom = LINEAR ? -sp.right : sp.bottom;
    const float top = LINEAR ? sp.right : sp.top;
    const float span_w = sp.span_w;
    const float span_h = LINEAR ? sp.span_w : sp.span_h;
    bool inside = x >= left && x <= right;
    if (!LINEAR && !inside) status |= NFA_STATUS_OUTSIDE_DOMAIN;

    int k = -1;
    float cw0 = 0.f, cw1 = 1.f, ch0 = 0.f, ch1 = 1.f, den_w = 1.f, den_h = 1.f;
    Slots<KT> ew, eh;
    ew.bind(sl);
    eh.bind(sl + K);
    if (inside) {
        den_w = softmax_numerators<KT>(ew, sl, K, sp.divisor, sp.rdivisor);
        den_h = softmax_numerators<KT>(eh, sl + K, K, sp.divisor, sp.rdivisor);
        if (INVERSE) {
            walk_bins<KT, true>(eh, K, den_h, sp.min_h, sp.om_h, span_h, bottom, top, x, k, ch0, ch1);
            if (k >= 0) walk_bins<KT, false>(ew, K, den_w, sp.min_w, sp.om_w, span_w, left, right, x, k, cw0, cw1);
        } else {
            walk_bins<KT, true>(ew, K, den_w, sp.min_w, sp.om_w, span_w, left, right, x, k, cw0, cw1);
            if (k >= 0) walk_bins<KT, false>(eh, K, den_h, sp.min_h, sp.om_h, span_h, bottom, top, x, k, ch0, ch1);
        }
        if (k < 0 || x >= (INVERSE ? (LINEAR ? sp.right_eps : sp.top_eps) : sp.right_eps)) {
            status |= NFA_STATUS_OUTSIDE_DOMAIN;
            inside = false;
        }
    }
    if (!inside) {  // identity (tails) or out of domain: no parameter gradient
        for (int q = 0; q < P; ++q) sl[q] = 0.0f;
        return g_out;
    }

    float* sd = sl + 2 * K;
    int i0, i1;  // positions of the bin's two derivative logits (-1: the constant tail logit)
    if (LINEAR) {
        i0 = k >= 1 ? k - 1 : -1;
        i1 = k < sp.nd ? k : -1;
    } else {
        i0 = k;
        i1 = k + 1;
    }
    const float u0 = i0 >= 0 ? sd[i0] : sp.tail_logit;
    const float u1 = i1 >= 0 ? sd[i1] : sp.tail_logit;
    const float d0 = sp.min_d + softplus_beta(u0, sp.beta);
    const float d1 = sp.min_d + softplus_beta(u1, sp.beta);

    BinAdj g;
    float g_in;
    if (!INVERSE) {
        g = forward_map_adjoint(x, cw0, cw1, ch0, ch1, d0, d1, g_out, gl);
        g_in = g.x;
    } else {
        // x is the layer INPUT y_in here; solve for xs = f^-1(y_in) exactly like the forward pass
        const float in_w = cw1 - cw0, in_h = ch1 - ch0;
        const float delta = in_h / in_w;
        const float s = (d0 + d1) - 2.0f * delta;
        const float yc = x - ch0;
        const float qa = yc * s + in_h * (delta - d0);
        const float qb = in_h * d0 - yc * s;
        const float qc = (-delta) * yc;
        const float disc = qb * qb - (4.0f * qa) * qc;
        if (!(disc >= 0.0f)) status |= NFA_STATUS_NEG_DISCRIMINANT;
        const float root = (2.0f * qc) / ((-qb) - sqrtf(disc));
        const float xs = root * in_w + cw0;
        // f'(xs) = exp(lad_fwd(xs))
        const float t1 = root * (1.0f - root), omr = 1.0f - root;
        const float den = delta + s * t1;
        const float fprime = (delta * delta) * ((d1 * root * root + 2.0f * delta * t1) + d0 * omr * omr) / (den * den);
        // outputs: xs (adjoint g_out) and lad_inv = -log f'(xs) (adjoint gl)
        const float Lx = forward_lad_dx(xs, cw0, cw1, ch0, ch1, d0, d1);
        const float xbar = g_out - gl * Lx;
        g_in = xbar / fprime;
        g = forward_map_adjoint(xs, cw0, cw1, ch0, ch1, d0, d1, -g_in, -gl);
    }

    // ---- knots -> logits.  p_j = softmax numerators / denominator (ew, eh hold the numerators)
    const bool has_b = k < K - 1;  // knot k+1 is the fixed end knot when k == K-1
    const float rdw = 1.0f / den_w, rdh = 1.0f / den_h;
    const float ka_w = sp.om_w * span_w, ka_h = sp.om_h * span_h;
    float dot_w = 0.0f, dot_h = 0.0f;
#pragma unroll
    for (int j = 0; j < (KT > 0 ? KT : K); ++j) {
        const float pw = ew.get(j) * rdw, ph = eh.get(j) * rdh;
        const float pbw = ka_w * ((j < k ? g.a : 0.0f) + ((j <= k && has_b) ? g.b : 0.0f));
        const float pbh = ka_h * ((j < k ? g.c : 0.0f) + ((j <= k && has_b) ? g.e : 0.0f));
        dot_w += pw * pbw;
        dot_h += ph * pbh;
    }
    const float sc = sp.divisor != 0.0f ? sp.rdivisor : 1.0f;
    // derivative-logit gradients first (they read sd[] which the loop below does not touch)
    const float gd0 = i0 >= 0 ? g.d0 * sigmoid_beta(u0, sp.beta) : 0.0f;
    const float gd1 = i1 >= 0 ? g.d1 * sigmoid_beta(u1, sp.beta) : 0.0f;
#pragma unroll
    for (int j = 0; j < (KT > 0 ? KT : K); ++j) {
        const float pw = ew.get(j) * rdw, ph = eh.get(j) * rdh;
        const float pbw = ka_w * ((j < k ? g.a : 0.0f) + ((j <= k && has_b) ? g.b : 0.0f));
        const float pbh = ka_h * ((j < k ? g.c : 0.0f) + ((j <= k && has_b) ? g.e : 0.0f));
        sl[j] = pw * (pbw - dot_w) * sc;
        sl[K + j] = ph * (pbh - dot_h) * sc;
    }
    for (int q = 0; q < sp.nd; ++q) sd[q] = 0.0f;
    if (i0 >= 0) sd[i0] = gd0;
    if (i1 >= 0) sd[i1] = gd1;
    return g_in;
}

template <int KT, bool INVERSE>
__global__ void __launch_bounds__(kBlock) rqs_coupling_backward_kernel(const BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_p = lds;
    float* s_x = lds + a.off_x;
    float* s_gy = lds + a.off_gy;
    float* s_gx = lds + a.off_gx;
    int* s_tidx = reinterpret_cast<int*>(lds + a.off_idx);
    int* s_src = s_tidx + a.dt;
    int* s_dst = s_src + a.D;
    unsigned char* s_ist = reinterpret_cast<unsigned char*>(s_dst + a.D);

    const int tid = threadIdx.x;
    const int D = a.D, dt = a.dt, P = a.sp.P;
    int my_status = 0;
    for (int c = tid; c < D; c += kBlock) {
        int src = c, dst = c;
        if (a.perm) {
            const int64_t p = a.perm[c];
            if (p < 0 || p >= D) my_status |= NFA_STATUS_BAD_INDEX;
            src = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
        }
        if (a.scatter) {
            const int64_t p = a.scatter[c];
            if (p < 0 || p >= D) my_status |= NFA_STATUS_BAD_INDEX;
            dst = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
        }
        s_src[c] = src;
        s_dst[c] = dst;
        s_ist[c] = 0;
    }
    __syncthreads();
    for (int j = tid; j < dt; j += kBlock) {
        const int64_t t = a.tidx[j];
        if (t < 0 || t >= D) my_status |= NFA_STATUS_BAD_INDEX;
        const int col = (int)(t < 0 ? 0 : (t >= D ? D - 1 : t));
        s_tidx[j] = col;
        s_ist[col] = 1;
    }

    const int64_t num_tiles = (a.batch + a.R - 1) / a.R;
    for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * a.R;
        const int rows = (int)((a.batch - row0) < a.R ? (a.batch - row0) : a.R);
        const int nitems = rows * dt;
        const int mx = tile_load(a.x + row0 * D, rows * D, s_x, tid);
        const int mg = tile_load(a.gout + row0 * D, rows * D, s_gy, tid);
        float* s_gi = s_gx + tile_store_offset(a.gin + row0 * D);
        for (int c0 = 0; c0 < nitems; c0 += a.C) {
            const int cn = (nitems - c0) < a.C ? (nitems - c0) : a.C;
            float* gdst = a.gparams + (row0 * dt + c0) * (int64_t)P;
            const int mp = tile_load(a.params + (row0 * dt + c0) * (int64_t)P, cn * P, s_p, tid);
            __syncthreads();
            if (c0 == 0) {
                // pass-through columns: out[:, dst[c]] = in[:, src[c]]  =>  gin[:, src[c]] = gout[:, dst[c]]
                for (int e = tid; e < rows * D; e += kBlock) {
                    const int r = (int)fastdiv((uint32_t)e, a.div_D);
                    const int c = e - r * D;
                    if (!s_ist[c]) s_gi[e - c + s_src[c]] = s_gy[mg + e - c + s_dst[c]];
                }
            }
            for (int ii = tid; ii < cn; ii += kBlock) {
                const int i = c0 + ii;
                const int r = (int)fastdiv((uint32_t)i, a.div_dt);
                const int j = i - r * dt;
                const int col = s_tidx[j];
                const float xin = s_x[mx + r * D + s_src[col]];
                const float gy = s_gy[mg + r * D + s_dst[col]];
                const float gl = a.glad ? a.glad[row0 + r] : 0.0f;
                float* sl = s_p + mp + ii * P;
                const float gx = a.sp.linear ? rqs_backward<KT, INVERSE, true>(xin, sl, a.sp, gy, gl, my_status)
                                             : rqs_backward<KT, INVERSE, false>(xin, sl, a.sp, gy, gl, my_status);
                s_gi[r * D + s_src[col]] = gx;
            }
            __syncthreads();
            // the LDS image now holds the logit gradients in the layout of the global chunk
            if (mp == tile_store_offset(gdst)) {
                tile_store(gdst, cn * P, s_p, tid);
            } else {
                for (int e = tid; e < cn * P; e += kBlock) gdst[e] = s_p[mp + e];
            }
            __syncthreads();
        }
        tile_store(a.gin + row0 * D, rows * D, s_gx, tid);
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

// ------------------------------------------------------------------------------------------
// Software-pipelined form for 16-byte aligned layouts with full tiles (the scheme of the forward
// kernel `rqs_coupling_pipelined`, rqs.hip): every lane carries the NEXT tile -- NV float4 of
// conditioner output, one float4 each of inputs and upstream gradients, its row's grad_logabsdet
// -- in registers while the current tile is differentiated and stored, so HBM reads stay in flight
// during the whole evaluation.  Workgroup barriers are LDS-only (`s_waitcnt lgkmcnt(0); s_barrier`).
// A lane differentiates one spline in place in the LDS image of the conditioner output; the image
// then leaves as NV coalesced float4 stores per lane.
__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef float vec4 __attribute__((ext_vector_type(4)));

template <int KT, bool INVERSE, bool LINEAR>
__global__ void __launch_bounds__(kBlock, 3) rqs_coupling_backward_pipelined(const BwdArgs a) {
    constexpr int NV = (3 * KT + (LINEAR ? -1 : 1) + 3) / 4;
    static_assert(NV <= 8, "add prefetch registers");
    // Preconditions (checked by the host): a.batch % a.R == 0, R*dt <= 256, R*D <= 512,
    // dt*P % 4 == 0, D % 4 == 0, all five arrays 16-byte aligned.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_p = lds;
    float* s_x = lds + a.off_x;
    float* s_gy = lds + a.off_gy;
    float* s_gi = lds + a.off_gx;
    int* s_tidx = reinterpret_cast<int*>(lds + a.off_idx);
    int* s_src = s_tidx + a.dt;
    int* s_dst = s_src + a.D;
    unsigned char* s_ist = reinterpret_cast<unsigned char*>(s_dst + a.D);

    const int tid = threadIdx.x;
    const int D = a.D, dt = a.dt, P = a.sp.P, R = a.R;
    int my_status = 0;
    for (int c = tid; c < D; c += kBlock) {
        int src = c, dst = c;
        if (a.perm) {
            const int64_t p = a.perm[c];
            if (p < 0 || p >= D) my_status |= NFA_STATUS_BAD_INDEX;
            src = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
        }
        if (a.scatter) {
            const int64_t p = a.scatter[c];
            if (p < 0 || p >= D) my_status |= NFA_STATUS_BAD_INDEX;
            dst = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
        }
        s_src[c] = src;
        s_dst[c] = dst;
        s_ist[c] = 0;
    }
    __syncthreads();
    for (int j = tid; j < dt; j += kBlock) {
        const int64_t t = a.tidx[j];
        if (t < 0 || t >= D) my_status |= NFA_STATUS_BAD_INDEX;
        const int col = (int)(t < 0 ? 0 : (t >= D ? D - 1 : t));
        s_tidx[j] = col;
        s_ist[col] = 1;
    }
    __syncthreads();

    const int nitems = R * dt;          // <= 256
    const int nvp = (nitems * P) >> 2;  // float4 of conditioner output (and of its gradient) per tile
    const int nvx = (R * D) >> 2;       // float4 of inputs / gradients per tile (<= 128)
    const bool has_item = tid < nitems;
    int it_in = 0, it_out = 0, it_row = 0;
    {
        const int i = has_item ? tid : 0;
        it_row = (int)fastdiv((uint32_t)i, a.div_dt);
        const int col = s_tidx[i - it_row * dt];
        it_in = it_row * D + s_src[col];   // the spline's input, and where its input gradient goes
        it_out = it_row * D + s_dst[col];  // where its upstream gradient sits
    }
    float* it_p = s_p + (has_item ? tid : 0) * P;
    // pass-through columns: out[:, dst[c]] = in[:, src[c]]  =>  gin[:, src[c]] = gout[:, dst[c]]
    int cp_from0 = -1, cp_to0 = 0, cp_from1 = -1, cp_to1 = 0;
    {
        const int e0 = tid, e1 = tid + kBlock;
        if (e0 < R * D) {
            const int r = (int)fastdiv((uint32_t)e0, a.div_D), c = e0 - r * D;
            if (!s_ist[c]) { cp_from0 = e0 - c + s_dst[c]; cp_to0 = e0 - c + s_src[c]; }
        }
        if (e1 < R * D) {
            const int r = (int)fastdiv((uint32_t)e1, a.div_D), c = e1 - r * D;
            if (!s_ist[c]) { cp_from1 = e1 - c + s_dst[c]; cp_to1 = e1 - c + s_src[c]; }
        }
    }
    const int64_t tile_stride_p = (int64_t)nitems * P;
    const int tile_stride_x = R * D;

    vec4 pr0, pr1, pr2, pr3, pr4, pr5, pr6, pr7;
    vec4 xr, gr;
    float glr = 0.0f;
#define NFA_LD(k)                                                  \
    if (NV > k) {                                                  \
        const int v_ = k * kBlock + tid;                           \
        pr##k = gp_[v_ < nvp ? v_ : nvp - 1];                      \
    }
#define NFA_ST(k)                                                  \
    if (NV > k) {                                                  \
        const int v_ = k * kBlock + tid;                           \
        if (v_ < nvp) reinterpret_cast<vec4*>(s_p)[v_] = pr##k;    \
    }
#define NFA_GST(k)                                                 \
    if (NV > k) {                                                  \
        const int v_ = k * kBlock + tid;                           \
        if (v_ < nvp) gq_[v_] = reinterpret_cast<const vec4*>(s_p)[v_]; \
    }
    // index-clamped, unconditional loads; past the last tile the lanes re-read tile 0 (unused)
#define NFA_ISSUE_TILE(TILE)                                                                 \
    {                                                                                        \
        const int64_t t_ = (TILE) < num_tiles ? (TILE) : 0;                                  \
        const vec4* gp_ = reinterpret_cast<const vec4*>(a.params + t_ * tile_stride_p);      \
        const vec4* gx_ = reinterpret_cast<const vec4*>(a.x + t_ * tile_stride_x);          \
        const vec4* gg_ = reinterpret_cast<const vec4*>(a.gout + t_ * tile_stride_x);       \
        NFA_LD(0) NFA_LD(1) NFA_LD(2) NFA_LD(3) NFA_LD(4) NFA_LD(5) NFA_LD(6) NFA_LD(7)      \
        xr = gx_[tid < nvx ? tid : nvx - 1];                                                 \
        gr = gg_[tid < nvx ? tid : nvx - 1];                                                 \
        if (a.glad) glr = a.glad[t_ * R + it_row];                                           \
    }

    const int64_t num_tiles = a.batch / R;
    int64_t tile = blockIdx.x;
    NFA_ISSUE_TILE(tile)
    for (; tile < num_tiles; tile += gridDim.x) {
        NFA_ST(0) NFA_ST(1) NFA_ST(2) NFA_ST(3) NFA_ST(4) NFA_ST(5) NFA_ST(6) NFA_ST(7)
        if (tid < nvx) {
            reinterpret_cast<vec4*>(s_x)[tid] = xr;
            reinterpret_cast<vec4*>(s_gy)[tid] = gr;
        }
        const float gl = glr;
        const int64_t next = tile + gridDim.x;
        NFA_ISSUE_TILE(next)  // in flight until the next iteration's LDS writes
        lds_only_barrier();

        if (cp_from0 >= 0) s_gi[cp_to0] = s_gy[cp_from0];
        if (cp_from1 >= 0) s_gi[cp_to1] = s_gy[cp_from1];
        if (has_item)
            s_gi[it_in] = rqs_backward<KT, INVERSE, LINEAR>(s_x[it_in], it_p, a.sp, s_gy[it_out], gl, my_status);
        lds_only_barrier();

        const int64_t row0 = tile * R;
        vec4* gq_ = reinterpret_cast<vec4*>(a.gparams + tile * tile_stride_p);
        NFA_GST(0) NFA_GST(1) NFA_GST(2) NFA_GST(3) NFA_GST(4) NFA_GST(5) NFA_GST(6) NFA_GST(7)
        if (tid < nvx)
            reinterpret_cast<vec4*>(a.gin + row0 * D)[tid] = reinterpret_cast<const vec4*>(s_gi)[tid];
        lds_only_barrier();  // the image has been read out before the next tile overwrites it
    }
#undef NFA_ISSUE_TILE
#undef NFA_LD
#undef NFA_ST
#undef NFA_GST
    if (my_status && a.status) atomicOr(a.status, my_status);
}

constexpr int kMaxDynLdsBwd = 64 * 1024;

template <int KT>
static int launch_backward_pipelined(const BwdArgs& a, int inverse, dim3 grid, size_t lds, hipStream_t st) {
    if (a.sp.linear) {
        if (inverse)
            hipLaunchKernelGGL((rqs_coupling_backward_pipelined<KT, true, true>), grid, dim3(kBlock), lds, st, a);
        else
            hipLaunchKernelGGL((rqs_coupling_backward_pipelined<KT, false, true>), grid, dim3(kBlock), lds, st, a);
    } else {
        if (inverse)
            hipLaunchKernelGGL((rqs_coupling_backward_pipelined<KT, true, false>), grid, dim3(kBlock), lds, st, a);
        else
            hipLaunchKernelGGL((rqs_coupling_backward_pipelined<KT, false, false>), grid, dim3(kBlock), lds, st, a);
    }
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}


template <int KT>
static int launch_backward(const BwdArgs& a, int inverse, dim3 grid, size_t lds, hipStream_t st) {
    if (inverse)
        hipLaunchKernelGGL((rqs_coupling_backward_kernel<KT, true>), grid, dim3(kBlock), lds, st, a);
    else
        hipLaunchKernelGGL((rqs_coupling_backward_kernel<KT, false>), grid, dim3(kBlock), lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_rqs_coupling_backward_f32(const float* inputs, const float* params,
                                             const int64_t* transform_idx, const int64_t* in_perm,
                                             const int64_t* out_scatter, const float* grad_outputs,
                                             const float* grad_logabsdet, float* grad_inputs,
                                             float* grad_params, int32_t* status, int64_t batch,
                                             int32_t features, int32_t num_transform,
                                             const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    if (flags & ~NFA_FLAG_INVERSE) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1 || num_transform < 0 || num_transform > features)
        return NFA_ERR_INVALID_ARGUMENT;
    BwdArgs a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (batch == 0) return NFA_OK;
    if (!inputs || !grad_outputs || !grad_inputs || (num_transform > 0 && (!params || !transform_idx || !grad_params)))
        return NFA_ERR_INVALID_ARGUMENT;
    if (features > 65535) return NFA_ERR_UNSUPPORTED;
    const int P = a.sp.P, D = features, dt = num_transform;
    int R = dt > 0 ? kBlock / dt : kBlock / (D < kBlock ? D : kBlock);
    if (R < 1) R = 1;
    if ((int64_t)R > batch) R = (int)batch;
    int C = 0;
    auto lds_floats = [&](int r) {
        const int chunk_items = C > 0 ? C : r * dt;
        int o = round_up4(chunk_items * P) + 8;
        a.off_x = o;
        o += round_up4(r * D) + 4;
        a.off_gy = o;
        o += round_up4(r * D) + 4;
        a.off_gx = o;
        o += round_up4(r * D) + 4;
        a.off_idx = o;
        o += dt + 2 * D + (D + 3) / 4;
        return o;
    };
    while (R > 1 && (size_t)lds_floats(R) * 4 > (size_t)kMaxDynLdsBwd) R >>= 1;
    if (R == 1 && (size_t)lds_floats(1) * 4 > (size_t)kMaxDynLdsBwd) {
        const size_t fixed = (size_t)(3 * (round_up4(D) + 4) + dt + 2 * D + (D + 3) / 4 + 16) * 4;
        if (fixed + (size_t)kBlock * P * 4 > (size_t)kMaxDynLdsBwd) return NFA_ERR_UNSUPPORTED;
        C = (int)(((size_t)kMaxDynLdsBwd - fixed) / ((size_t)P * 4));
        C = (C / kBlock) * kBlock;
        if (C >= dt) C = 0;
    }
    const size_t lds = (size_t)lds_floats(R) * 4;
    if (lds > (size_t)kMaxDynLdsBwd || (int64_t)R * dt >= 65536 || (int64_t)R * D >= 65536)
        return NFA_ERR_UNSUPPORTED;
    a.C = C > 0 ? C : (R * dt > 0 ? R * dt : 1);
    a.x = inputs;
    a.params = params;
    a.tidx = transform_idx;
    a.perm = in_perm;
    a.scatter = out_scatter;
    a.gout = grad_outputs;
    a.glad = grad_logabsdet;
    a.gin = grad_inputs;
    a.gparams = grad_params;
    a.status = status;
    a.batch = batch;
    a.D = D;
    a.dt = dt;
    a.R = R;
    a.div_dt = make_fastdiv((uint32_t)(dt > 0 ? dt : 1));
    a.div_D = make_fastdiv((uint32_t)D);
    const int64_t tiles = (batch + R - 1) / R;
    int per_cu = (int)((size_t)(160 * 1024) / (lds + 256));
    if (per_cu > 6) per_cu = 6;
    if (per_cu < 1) per_cu = 1;
    int64_t g = (int64_t)device_cu_count() * per_cu;
    if (g > tiles) g = tiles;
    const int inverse = flags & NFA_FLAG_INVERSE;
    // aligned layouts with K = 8 take the software-pipelined kernel; leftover rows (< R) follow in
    // one workgroup of the generic kernel
    static const int use_pipe = [] {
        const char* e = getenv("NFA_K1_BWD_PIPELINE");
        return e ? atoi(e) : 1;
    }();
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (use_pipe && a.sp.K == 8 && C == 0 && dt > 0 && (dt * P) % 4 == 0 && D % 4 == 0 && R * dt <= kBlock &&
        R * D <= 2 * kBlock && (int64_t)R <= batch && aligned16(params) && aligned16(inputs) &&
        aligned16(grad_outputs) && aligned16(grad_inputs) && aligned16(grad_params)) {
        const int64_t full_rows = (batch / R) * R;
        BwdArgs f = a;
        f.batch = full_rows;
        int64_t gp = (int64_t)device_cu_count() * (per_cu > 3 ? 3 : per_cu);
        if (gp > full_rows / R) gp = full_rows / R;
        rc = launch_backward_pipelined<8>(f, inverse, dim3((unsigned)gp), lds, (hipStream_t)stream);
        if (rc != NFA_OK || full_rows == batch) return rc;
        a.x = inputs + full_rows * D;
        a.params = params + full_rows * (int64_t)dt * P;
        a.gout = grad_outputs + full_rows * D;
        a.glad = grad_logabsdet ? grad_logabsdet + full_rows : nullptr;
        a.gin = grad_inputs + full_rows * D;
        a.gparams = grad_params + full_rows * (int64_t)dt * P;
        a.batch = batch - full_rows;
        return launch_backward<8>(a, inverse, dim3(1), lds, (hipStream_t)stream);
    }
    switch (a.sp.K) {
        case 8: return launch_backward<8>(a, inverse, dim3((unsigned)g), lds, (hipStream_t)stream);
        default: return launch_backward<0>(a, inverse, dim3((unsigned)g), lds, (hipStream_t)stream);
    }
}
