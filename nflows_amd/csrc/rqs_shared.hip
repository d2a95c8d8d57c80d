// K6: rational-quadratic CDF transform whose parameters are shared by the whole batch
// (SURVEY.md section 8f, row f4):  PiecewiseRationalQuadraticCDF._spline,
// nflows/transforms/nonlinearities.py:431-467 -- the reference broadcasts the [F, K] logits to
// [B, F, K] (`_share_across_batch`, :226-227) and runs the generic functional over B*F splines.
//
// Here the knots are per FEATURE, not per sample: each workgroup builds the F tables
// (K+1 width knots, K+1 height knots, K+1 knot derivatives) ONCE in LDS with exactly the
// arithmetic of the per-sample kernels (softmax -> min + (1-min K) p -> double prefix sums ->
// affine -> forced end knots; min_d + softplus), then streams its rows through them: a lane
// handles one (sample, feature) element -- bin search by comparing against the feature's knots
// in LDS (lanes of one feature read the same words: broadcast), the in-bin map, and a
// fixed-order per-sample reduction of the log-derivative.  HBM traffic is the ideal
// 8 bytes per element (+4 per sample).

#include "rqs_math.hpp"

namespace nfa {

struct SharedArgs {
    const float* x;
    const float* uw;  // [F, K]
    const float* uh;  // [F, K]
    const float* ud;  // [F, nd]
    float* y;
    float* lad;       // [B]
    int32_t* status;
    int64_t batch;
    int F;
    int R;            // rows per tile
    FastDiv div_F;
    RqsDev sp;
    int off_x, off_lad, off_raw;
};

template <bool INVERSE>
__global__ void __launch_bounds__(kBlock) rqs_shared_kernel(const SharedArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = a.sp.K, F = a.F, T = 3 * (K + 1);
    float* s_tab = lds;              // [F][cw(K+1) | ch(K+1) | d(K+1)]
    float* s_x = lds + a.off_x;      // tile of inputs, later outputs (in place)
    float* s_l = lds + a.off_lad;    // per-element log-derivatives of the tile
    const int tid = threadIdx.x;
    const bool linear = a.sp.linear;
    const float left = linear ? -a.sp.right : a.sp.left, right = a.sp.right;
    const float bottom = linear ? -a.sp.right : a.sp.bottom, top = linear ? a.sp.right : a.sp.top;
    const float span_w = a.sp.span_w, span_h = linear ? a.sp.span_w : a.sp.span_h;
    int my_status = 0;

    // ---- per-feature tables.  All logits are first staged into LDS by the whole workgroup (one
    //      round of global latency instead of 3K dependent ones per lane), then one lane per
    //      (feature, role) builds the width knots, the height knots or the knot derivatives; the
    //      softmax numerators overwrite the staged logits in place.
    float* s_raw = lds + a.off_raw;  // [F][K] width logits | [F][K] height logits | [F][nd] derivative logits
    const int nd = a.sp.nd;
    for (int i = tid; i < F * K; i += kBlock) {
        s_raw[i] = a.uw[i];
        s_raw[F * K + i] = a.uh[i];
    }
    for (int i = tid; i < F * nd; i += kBlock) s_raw[2 * F * K + i] = a.ud[i];
    __syncthreads();
    for (int task = tid; task < 3 * F; task += kBlock) {
        const int role = (int)fastdiv((uint32_t)task, a.div_F);
        const int f = task - role * F;
        float* tab = s_tab + f * T;
        if (role < 2) {
            const int side = role;
            float* tmp = s_raw + side * F * K + f * K;
            float* knots = tab + side * (K + 1);
            const float lo = side ? bottom : left, hi = side ? top : right;
            const float span = side ? span_h : span_w;
            const float minbin = side ? a.sp.min_h : a.sp.min_w, om = side ? a.sp.om_h : a.sp.om_w;
            float m = -INFINITY;
            for (int i = 0; i < K; ++i) {
                float v = tmp[i];
                if (a.sp.divisor != 0.0f) v = div_with_rcp(v, a.sp.divisor, a.sp.rdivisor);
                tmp[i] = v;
                m = fmaxf(m, v);
            }
            double ssum = 0.0;
            for (int i = 0; i < K; ++i) {
                const float e = exp_noclamp(tmp[i] - m);
                tmp[i] = e;
                ssum += (double)e;
            }
            const float den = (float)ssum, rden = rcp_refined(den);
            double acc = 0.0;
            knots[0] = lo;
            for (int i = 0; i < K; ++i) {
                const float p = div_with_rcp(tmp[i], den, rden);
                const float w = minbin + om * p;
                acc += (double)w;
                knots[i + 1] = (i == K - 1) ? hi : span * (float)acc + lo;
            }
        } else {
            float* dv = tab + 2 * (K + 1);
            const float* ud = s_raw + 2 * F * K + f * nd;
            for (int i = 0; i <= K; ++i) {
                float logit;
                if (linear)
                    logit = (i == 0 || i - 1 >= nd) ? a.sp.tail_logit : ud[i - 1];
                else
                    logit = ud[i];
                dv[i] = a.sp.min_d + softplus_beta(logit, a.sp.beta);
            }
        }
    }
    __syncthreads();

    const int64_t num_tiles = (a.batch + a.R - 1) / a.R;
    for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * a.R;
        const int rows = (int)((a.batch - row0) < a.R ? (a.batch - row0) : a.R);
        const int n = rows * F;
        const int mx = tile_load(a.x + row0 * F, n, s_x, tid);
        __syncthreads();
        for (int e = tid; e < n; e += kBlock) {
            const int r = (int)fastdiv((uint32_t)e, a.div_F);
            const int f = e - r * F;
            const float x = s_x[mx + e];
            const float* tab = s_tab + f * T;
            float y = x, l = 0.0f;
            const bool inside = linear ? (x >= left && x <= right) : !(x < left || x > right);
            if (!inside) {
                if (!linear) my_status |= NFA_STATUS_OUTSIDE_DOMAIN;
            } else {
                const float* loc = tab + (INVERSE ? (K + 1) : 0);
                int k = -1;
                for (int i = 0; i < K; ++i) k = (x >= loc[i]) ? i : k;
                const float last_eps = loc[K] + 1e-6f;
                if (k < 0 || x >= last_eps) {
                    my_status |= NFA_STATUS_OUTSIDE_DOMAIN;
                } else {
                    my_status |= rqs_bin_eval<INVERSE>(x, tab[k], tab[k + 1], tab[K + 1 + k], tab[K + 2 + k],
                                                       tab[2 * (K + 1) + k], tab[2 * (K + 1) + k + 1], y, l);
                }
            }
            s_x[mx + e] = y;  // in place: the tile is stored from the same LDS image
            s_l[e] = l;
        }
        __syncthreads();
        if (mx == tile_store_offset(a.y + row0 * F)) {
            tile_store(a.y + row0 * F, n, s_x, tid);
        } else {
            for (int e = tid; e < n; e += kBlock) a.y[row0 * F + e] = s_x[mx + e];
        }
        const int wave = tid >> 6, lane = tid & 63;
        for (int r = wave; r < rows; r += kBlock / kWave) {
            float v = 0.0f;
            for (int m = lane; m < F; m += kWave) v += s_l[r * F + m];
            v = wave_sum(v);
            if (lane == 0) a.lad[row0 + r] = v;
        }
        __syncthreads();
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_rqs_shared_f32(const float* inputs, const float* unnormalized_widths,
                                  const float* unnormalized_heights,
                                  const float* unnormalized_derivatives, float* outputs,
                                  float* logabsdet, int32_t* status, int64_t batch, int32_t features,
                                  const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    if (flags & ~NFA_FLAG_INVERSE) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1) return NFA_ERR_INVALID_ARGUMENT;
    SharedArgs a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (batch == 0) return NFA_OK;
    if (!inputs || !outputs || !logabsdet || !unnormalized_widths || !unnormalized_heights ||
        (a.sp.nd > 0 && !unnormalized_derivatives))
        return NFA_ERR_INVALID_ARGUMENT;
    const int K = a.sp.K, F = features;
    const int64_t tab = (int64_t)F * 3 * (K + 1);
    int R = (4 * kBlock) / F;  // ~4 elements per lane per tile
    if (R < 1) R = 1;
    if ((int64_t)R > batch) R = (int)batch;
    auto lds_floats = [&](int r) {
        int64_t o = (tab + 3) & ~3;
        a.off_x = (int)o;
        o += round_up4(r * F) + 8;
        a.off_lad = (int)o;
        o += round_up4(r * F);
        a.off_raw = (int)o;  // staged logits: only live while the tables are built
        o += round_up4(F * (2 * K + a.sp.nd));
        return o;
    };
    while (R > 1 && lds_floats(R) * 4 > 64 * 1024) R >>= 1;
    const int64_t lds = lds_floats(R) * 4;
    if (lds > 64 * 1024 || (int64_t)R * F >= 65536) return NFA_ERR_UNSUPPORTED;
    a.x = inputs;
    a.uw = unnormalized_widths;
    a.uh = unnormalized_heights;
    a.ud = unnormalized_derivatives;
    a.y = outputs;
    a.lad = logabsdet;
    a.status = status;
    a.batch = batch;
    a.F = F;
    a.R = R;
    a.div_F = make_fastdiv((uint32_t)F);
    const int64_t tiles = (batch + R - 1) / R;
    int per_cu = (int)((160 * 1024) / (lds + 256));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int64_t g = (int64_t)device_cu_count() * per_cu;
    if (g > tiles) g = tiles;
    if (flags & NFA_FLAG_INVERSE)
        hipLaunchKernelGGL((rqs_shared_kernel<true>), dim3((unsigned)g), dim3(kBlock), (size_t)lds,
                           (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((rqs_shared_kernel<false>), dim3((unsigned)g), dim3(kBlock), (size_t)lds,
                           (hipStream_t)stream, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
