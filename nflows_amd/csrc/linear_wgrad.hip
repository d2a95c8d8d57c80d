// K10: weight and bias gradient of a conditioner layer  y = x W^T + b  (torch.nn.Linear):
//     grad_weight[O, I] = grad_outputs[B, O]^T . inputs[B, I]        grad_bias[O] = sum_b grad_outputs[b, :]
// The reference trains through autograd (examples/moons.ipynb cell 3); its conditioner layers are
// nn/nets/resnet.py:44,49,94,99 and nn/nets/mlp.py:47-68.  For these layers the reduction runs
// over the BATCH (65 536 rows) while the result is tiny (128 x 128): the library GEMM behind
// autograd tiles the result only (16 workgroups on a 256-CU chip, 223 us per layer measured,
// 56 % of the whole training step, profiles/r1_train_kernel_stats.csv).  Here the batch is split:
//
//   wgrad_partial_kernel   grid (result blocks, batch slices): a workgroup streams its rows of
//                          grad_outputs and inputs through a 3-slot LDS ring with LDS-DMA
//                          (global_load_lds, 16 bytes per lane, no VGPR round trip) and feeds them
//                          to v_mfma_f32_32x32x2_f32 -- rows of the two arrays ARE the k-index of
//                          this product, so both MFMA operands are plain row reads: lanes 0-31 take
//                          32 consecutive columns of row 2s, lanes 32-63 of row 2s+1.  True fp32
//                          products and sums.  Each wave owns TO x TI tiles of 32 x 32; the wave on
//                          the first input tile also sums the grad_outputs values it loads: the
//                          bias gradient.  Partials go to a workspace [slices][O*I + O].
//   wgrad_reduce_kernel    sums the slices in a fixed order (deterministic, no atomics) and adds
//                          the rows behind the last full 32-row stage.
//
// Bound: fp32 MFMA issue (64 cycles per 32x32x2) or HBM (each input element is read once per
// result block column/row it belongs to); for 128 x 128 at B = 65 536 both are ~14 us.
//
// Round 4: (1) up to eight same-shaped problems per launch pair (blockIdx.z: the four hidden Linears of a conditioner);
// (2) wgrad_partial_bf16_kernel, the default for the 128 x 128 result blocks: the same ring and layout with the operands
// split into three bf16 pieces by the lane that reads them and six products per multiply-add on
// v_mfma_f32_32x32x16_bf16 (fp32-accurate, 2.7 x less matrix-pipe time than the fp32 MFMA), 16-row stages with two
// workgroups per CU: 128 -> 736 at B = 65 536 130 -> 97 us, four 128 x 128 layers 90 -> 72 us.

#include "fused_common.hpp"

#include <stdlib.h>

namespace nfa {

constexpr int kWgRows = 32;  // batch rows per stage
constexpr int kWgRing = 3;

constexpr int kWgMaxProblems = 8;   // same-shaped problems per launch (nfa_linear_wgrad_batched_f32)

struct WgradArgs {
    const float* x[kWgMaxProblems];   // [B, I] per problem (blockIdx.z)
    const float* gy[kWgMaxProblems];  // [B, O]
    float* ws;        // [problems][ksplit][O*I + O]
    int I, O;
    int stages_total;  // full 32-row stages of the batch
    int ksplit;
    int blocks_i;
};

template <int TO, int TI, int WO, int WI>
__global__ void __launch_bounds__(kBlock) wgrad_partial_kernel(const WgradArgs a) {
    static_assert(WO * WI == 4, "four waves per workgroup");
    constexpr int BO = 32 * TO * WO, BI = 32 * TI * WI;
    constexpr int VA = BO / 4, VB = BI / 4;  // float4 per staged row
    constexpr int NA = (kWgRows * VA) / kBlock, NB = (kWgRows * VB + kBlock - 1) / kBlock;
    static_assert(NA * kBlock == kWgRows * VA && kWgRows * VB >= kBlock, "whole requests only");
    constexpr int NREQ = NA + NB;
    constexpr int kStage = kWgRows * (VA + VB);  // float4 per ring slot
    extern __shared__ __attribute__((aligned(16))) vec4f wg_ring[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, lh = lane >> 5;
    const int wo = wave / WI, wi = wave % WI;
    const int bo = blockIdx.x / a.blocks_i, bi = blockIdx.x - bo * a.blocks_i;
    const int ob = bo * BO, ib = bi * BI;
    const int kz = blockIdx.y;
    const int s_begin = (int)(((int64_t)a.stages_total * kz) / a.ksplit);
    const int s_end = (int)(((int64_t)a.stages_total * (kz + 1)) / a.ksplit);
    const int O = a.O, I = a.I;

    // this thread's share of a stage: NA float4 of grad_outputs, NB of inputs.  Columns past the
    // arrays' width are clamped onto valid ones: they only reach result elements that are never stored.
    int col_a = ob + (tid % VA) * 4, col_b = ib + (tid % VB) * 4;
    if (col_a > O - 4) col_a = O - 4;
    if (col_b > I - 4) col_b = I - 4;
    const int pz = blockIdx.z;   // the problem of a batched launch
    const float* src_a = a.gy[pz] + (int64_t)(tid / VA) * O + col_a;
    const float* src_b = a.x[pz] + (int64_t)(tid / VB) * I + col_b;
    constexpr int rows_per_req_a = kBlock / VA, rows_per_req_b = kBlock / VB;

    auto request = [&](int stage, int slot) {
        const float* ga = src_a + (int64_t)stage * kWgRows * O;
        const float* gb = src_b + (int64_t)stage * kWgRows * I;
        vec4f* dst = wg_ring + slot * kStage + tid;
#pragma unroll
        for (int q = 0; q < NA; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + (int64_t)q * rows_per_req_a * O),
                                             (__attribute__((address_space(3))) void*)(dst + q * kBlock), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < NB; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + (int64_t)q * rows_per_req_b * I),
                                             (__attribute__((address_space(3))) void*)(dst + kWgRows * VA + q * kBlock), 16, 0, 0);
    };
    // the next stage's requests of this wave have landed (those of the stage after it may still be
    // in flight) and every wave is done with the slot that is overwritten next
    auto advance = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(NREQ) : "memory");
    };

    f32x16 acc[TO][TI];
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[to][ti][j] = 0.0f;
    float colsum[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) colsum[to] = 0.0f;

    const int last = s_end - 1;
    request(s_begin, 0);
    request(s_begin + 1 <= last ? s_begin + 1 : last, 1);
    advance();
    int slot = 0;
    for (int s = s_begin; s < s_end; ++s) {
        const int ahead = s + 2 <= last ? s + 2 : last;  // (past the end: a harmless re-read into the free slot)
        request(ahead, slot >= 1 ? slot - 1 : kWgRing - 1);
        const float* sa = reinterpret_cast<const float*>(wg_ring + slot * kStage) + lh * BO + wo * 32 * TO + ln;
        const float* sb = reinterpret_cast<const float*>(wg_ring + slot * kStage + kWgRows * VA) + lh * BI + wi * 32 * TI + ln;
        // operands are read two k-steps ahead of the MFMAs that consume them (an LDS read issued
        // behind the four MFMAs of a k-step would otherwise land after the matrix pipe has drained)
        constexpr int NK = kWgRows / 2;
        float av[NK][TO], bv[NK][TI];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int to = 0; to < TO; ++to) av[k2][to] = sa[k2 * 2 * BO + to * 32];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) bv[k2][ti] = sb[k2 * 2 * BI + ti * 32];
        }
#pragma unroll
        for (int k2 = 0; k2 < NK; ++k2) {
            if (k2 + 2 < NK) {
#pragma unroll
                for (int to = 0; to < TO; ++to) av[k2 + 2][to] = sa[(k2 + 2) * 2 * BO + to * 32];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) bv[k2 + 2][ti] = sb[(k2 + 2) * 2 * BI + ti * 32];
            }
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the reads to their first use)
#pragma unroll
            for (int to = 0; to < TO; ++to) {
#pragma unroll
                for (int ti = 0; ti < TI; ++ti)
                    acc[to][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k2][to], bv[k2][ti], acc[to][ti], 0, 0, 0);
                if (wi == 0) colsum[to] += av[k2][to];
            }
        }
        advance();
        slot = slot + 1 == kWgRing ? 0 : slot + 1;
    }

    float* out = a.ws + ((int64_t)pz * a.ksplit + kz) * ((int64_t)O * I + O);
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        const int o0 = ob + wo * 32 * TO + to * 32;
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const int i = ib + wi * 32 * TI + ti * 32 + ln;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int o = o0 + 8 * (j >> 2) + 4 * lh + (j & 3);
                if (o < O && i < I) out[(int64_t)o * I + i] = acc[to][ti][j];
            }
        }
        if (wi == 0 && bi == 0) {  // (wave-uniform)
            const float v = colsum[to] + __shfl_xor(colsum[to], 32, kWave);
            if (lh == 0 && o0 + ln < O) out[(int64_t)O * I + o0 + ln] = v;
        }
    }
}

// ---- the same partial products on the bf16 matrix pipe (round 4).  v_mfma_f32_32x32x2_f32 runs at the vector rate
// (64 cycles for 2 k: under the power cap the chain sustains 121 TFLOP/s); three bf16 pieces per fp32 operand and the
// six largest cross products on v_mfma_f32_32x32x16_bf16 (fp32 accumulation, fp32-accurate: the scheme of K8 / K14,
// fused_common.hpp) take 6 x 32 cycles for 16 k -- 2.7 x less matrix-pipe time.  The batch rows are the k index: a lane
// holds, of its column (feature ln of the tile), the eight rows 16 ks + 8 lh + j -- eight ds_read_b32 a row pitch apart
// (the same number of LDS reads per 16 rows as the fp32 form) -- and splits them itself: ~56 VALU instructions per
// fragment that run beside the MFMAs of the previous k-step.  Same ring, same requests, same result layout, same
// reduction kernel; the bias gradient is summed from the raw fp32 values as before.
template <int TO, int TI, int WO, int WI, int ROWS>
__global__ void __launch_bounds__(kBlock) wgrad_partial_bf16_kernel(const WgradArgs a) {
    static_assert(ROWS == 16 || ROWS == 32, "stage rows");
    static_assert(WO * WI == 4, "four waves per workgroup");
    constexpr int BO = 32 * TO * WO, BI = 32 * TI * WI;
    constexpr int VA = BO / 4, VB = BI / 4;
    constexpr int NA = (ROWS * VA) / kBlock, NB = (ROWS * VB + kBlock - 1) / kBlock;
    static_assert(NA * kBlock == ROWS * VA && ROWS * VB >= kBlock, "whole requests only");
    constexpr int NREQ = NA + NB;
    constexpr int kStage = ROWS * (VA + VB);
    extern __shared__ __attribute__((aligned(16))) vec4f wg_ring[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, lh = lane >> 5;
    const int wo = wave / WI, wi = wave % WI;
    const int bo = blockIdx.x / a.blocks_i, bi = blockIdx.x - bo * a.blocks_i;
    const int ob = bo * BO, ib = bi * BI;
    const int kz = blockIdx.y;
    const int s_begin = (int)(((int64_t)a.stages_total * kz) / a.ksplit);
    const int s_end = (int)(((int64_t)a.stages_total * (kz + 1)) / a.ksplit);
    const int O = a.O, I = a.I;
    int col_a = ob + (tid % VA) * 4, col_b = ib + (tid % VB) * 4;
    if (col_a > O - 4) col_a = O - 4;
    if (col_b > I - 4) col_b = I - 4;
    const int pz = blockIdx.z;
    const float* src_a = a.gy[pz] + (int64_t)(tid / VA) * O + col_a;
    const float* src_b = a.x[pz] + (int64_t)(tid / VB) * I + col_b;
    constexpr int rows_per_req_a = kBlock / VA, rows_per_req_b = kBlock / VB;
    auto request = [&](int stage, int slot) {
        const float* ga = src_a + (int64_t)stage * ROWS * O;
        const float* gb = src_b + (int64_t)stage * ROWS * I;
        vec4f* dst = wg_ring + slot * kStage + tid;
#pragma unroll
        for (int q = 0; q < NA; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + (int64_t)q * rows_per_req_a * O),
                                             (__attribute__((address_space(3))) void*)(dst + q * kBlock), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < NB; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + (int64_t)q * rows_per_req_b * I),
                                             (__attribute__((address_space(3))) void*)(dst + ROWS * VA + q * kBlock), 16, 0, 0);
    };
    auto advance = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(NREQ) : "memory");
    };

    f32x16 acc[TO][TI];
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[to][ti][j] = 0.0f;
    float colsum[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) colsum[to] = 0.0f;

    // raw fp32 operands of one 16-row k-step: column (ln) of tile to / ti, rows 8 lh + j
    auto read_raw = [&](const float* sa, const float* sb, int ks, float (&ar)[TO][8], float (&br)[TI][8]) {
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int j = 0; j < 8; ++j) ar[to][j] = sa[(ks * 16 + j) * BO + to * 32];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int j = 0; j < 8; ++j) br[ti][j] = sb[(ks * 16 + j) * BI + ti * 32];
    };
    auto split8 = [](const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
        bf16x2 hh[4], mm[4], ll[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split3(vec2f{v[2 * q], v[2 * q + 1]}, hh[q], mm[q], ll[q]);
        h = join4(hh[0], hh[1], hh[2], hh[3]);
        m = join4(mm[0], mm[1], mm[2], mm[3]);
        l = join4(ll[0], ll[1], ll[2], ll[3]);
    };

    const int last = s_end - 1;
    request(s_begin, 0);
    request(s_begin + 1 <= last ? s_begin + 1 : last, 1);
    advance();
    int slot = 0;
    for (int s = s_begin; s < s_end; ++s) {
        const int ahead = s + 2 <= last ? s + 2 : last;
        request(ahead, slot >= 1 ? slot - 1 : kWgRing - 1);
        const float* sa = reinterpret_cast<const float*>(wg_ring + slot * kStage) + 8 * lh * BO + wo * 32 * TO + ln;
        const float* sb = reinterpret_cast<const float*>(wg_ring + slot * kStage + ROWS * VA) + 8 * lh * BI + wi * 32 * TI + ln;
        constexpr int NK = ROWS / 16;
        float ar[2][TO][8], br[2][TI][8];
        read_raw(sa, sb, 0, ar[0], br[0]);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            bf16x8 ah[TO], am[TO], al[TO], bh[TI], bm[TI], bl[TI];
#pragma unroll
            for (int to = 0; to < TO; ++to) {
                split8(ar[ks & 1][to], ah[to], am[to], al[to]);
                if (wi == 0) {
                    const float (&v)[8] = ar[ks & 1][to];
                    colsum[to] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                }
            }
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) split8(br[ks & 1][ti], bh[ti], bm[ti], bl[ti]);
            if (ks + 1 < NK) read_raw(sa, sb, ks + 1, ar[(ks + 1) & 1], br[(ks + 1) & 1]);   // in flight beside the MFMAs
#pragma unroll
            for (int to = 0; to < TO; ++to)
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) { NFA_MFMA6(acc[to][ti], ah[to], am[to], al[to], bh[ti], bm[ti], bl[ti]); }
        }
        advance();
        slot = slot + 1 == kWgRing ? 0 : slot + 1;
    }

    float* out = a.ws + ((int64_t)pz * a.ksplit + kz) * ((int64_t)O * I + O);
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        const int o0 = ob + wo * 32 * TO + to * 32;
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const int i = ib + wi * 32 * TI + ti * 32 + ln;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int o = o0 + 8 * (j >> 2) + 4 * lh + (j & 3);
                if (o < O && i < I) out[(int64_t)o * I + i] = acc[to][ti][j];
            }
        }
        if (wi == 0 && bi == 0) {  // (wave-uniform)
            const float v = colsum[to] + __shfl_xor(colsum[to], 32, kWave);
            if (lh == 0 && o0 + ln < O) out[(int64_t)O * I + o0 + ln] = v;
        }
    }
}

// grad_weight / grad_bias = sum over slices (fixed order) + the rows behind the last full stage
struct WgradReduceArgs {
    const float* x[kWgMaxProblems];
    const float* gy[kWgMaxProblems];
    float* gw[kWgMaxProblems];
    float* gb[kWgMaxProblems];   // null: no bias gradient for that problem
};

__global__ void __launch_bounds__(kBlock) wgrad_reduce_kernel(const float* __restrict__ ws_all, const WgradReduceArgs ra, int I,
                                                              int O, int ksplit, int64_t tail_begin, int64_t batch) {
    __shared__ float part[4][64];
    const int pz = blockIdx.y;
    const float* __restrict__ x = ra.x[pz];
    const float* __restrict__ gy = ra.gy[pz];
    float* __restrict__ gw = ra.gw[pz];
    float* __restrict__ gb = ra.gb[pz];
    const float* __restrict__ ws = ws_all + (int64_t)pz * ksplit * ((int64_t)O * I + O);
    const int tid = threadIdx.x, lane = tid & 63, p = tid >> 6;
    const int64_t n_w = (int64_t)O * I, stride = n_w + O;
    const int64_t n = gb ? stride : n_w;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    const int64_t ec = e < n ? e : n - 1;
    // slice quarter p: ksplit*p/4 .. ksplit*(p+1)/4
    const int k0 = (int)(((int64_t)ksplit * p) / 4), k1 = (int)(((int64_t)ksplit * (p + 1)) / 4);
    float v = 0.0f;
    const float* src = ws + ec;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)(k + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; k < k1; ++k) v += src[(int64_t)k * stride];
    part[p][lane] = v;
    __syncthreads();
    if (p != 0 || e >= n) return;
    v = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    if (e < n_w) {
        const int o = (int)(e / I), i = (int)(e - (int64_t)o * I);
        for (int64_t r = tail_begin; r < batch; ++r) v += gy[r * O + o] * x[r * I + i];
        gw[e] = v;
    } else {
        const int o = (int)(e - n_w);
        for (int64_t r = tail_begin; r < batch; ++r) v += gy[r * O + o];
        gb[o] = v;
    }
}

struct WgradPlan {
    int variant;  // 0: 128 x 128 result blocks, 1: 128 x 32
    int blocks_o, blocks_i, stages_total, ksplit;
    int rows;     // batch rows per stage
};

// `count` same-shaped problems share the chip: fewer batch slices each (fewer partial results to write and to sum,
// a longer stream per workgroup: the two-stage ramp of the ring is paid once per 32 stages instead of once per 8)
// rows per stage of the 128 x 128 form: 32, or -- NFA_K10_ROWS=16, the bf16 engine only -- 16 with two workgroups per CU
// (the three switches are read once per process: experiments set them in the environment of a fresh process)
static const char* wgrad_engine_env() {
    static const char* const eng = getenv("NFA_K10_ENGINE");
    return eng;
}
static int wgrad_rows(int variant, int problems_times_blocks) {
    static const char* const e = getenv("NFA_K10_ROWS");
    const char* eng = wgrad_engine_env();
    if (variant != 0 || (eng && eng[0] == 'f')) return kWgRows;
    if (e) return atoi(e) == 16 ? 16 : kWgRows;
    // 16-row stages (49 KB of LDS: two workgroups per CU, the split of one runs beside the MFMAs of the other) where the
    // launch has enough result blocks to feed them: 128 -> 736 110.9 -> 96.3 us, four 128 x 128 layers 75.2 -> 71.4 us;
    // a lone 128 x 128 problem is better off with 32 rows (30.7 vs 32.0 us)
    return problems_times_blocks >= 4 ? 16 : kWgRows;
}
static int wgrad_wgs_per_cu(int rows) {
    static const char* const e = getenv("NFA_K10_WGS");   // (experiment)
    return rows == 16 ? (e && atoi(e) > 0 ? atoi(e) : 2) : 1;
}

static WgradPlan plan_wgrad(int64_t batch, int I, int O, int count = 1) {
    WgradPlan p;
    p.variant = I <= 32 ? 1 : 0;
    const int BO = 128, BI = p.variant ? 32 : 128;
    p.blocks_o = (O + BO - 1) / BO;
    p.blocks_i = (I + BI - 1) / BI;
    const int rows = wgrad_rows(p.variant, p.blocks_o * p.blocks_i * (count > 0 ? count : 1));
    p.rows = rows;
    p.stages_total = (int)(batch / rows);
    int ks = device_cu_count() * wgrad_wgs_per_cu(rows) / (p.blocks_o * p.blocks_i * (count > 0 ? count : 1));
    if (ks < 1) ks = 1;
    if (ks > p.stages_total) ks = p.stages_total;
    p.ksplit = ks;
    return p;
}

}  // namespace nfa

using namespace nfa;

extern "C" size_t nfa_linear_wgrad_batched_workspace_bytes(int32_t count, int64_t batch, int32_t in_features,
                                                           int32_t out_features) {
    if (count < 1 || count > kWgMaxProblems || batch < 0 || in_features < 1 || out_features < 1) return 0;
    const WgradPlan p = plan_wgrad(batch, in_features, out_features, count);
    return (size_t)count * (p.ksplit > 0 ? p.ksplit : 1) * ((size_t)out_features * in_features + out_features) * sizeof(float);
}

extern "C" size_t nfa_linear_wgrad_workspace_bytes(int64_t batch, int32_t in_features, int32_t out_features) {
    return nfa_linear_wgrad_batched_workspace_bytes(1, batch, in_features, out_features);
}

extern "C" int nfa_linear_wgrad_batched_f32(int32_t count, const float* const* inputs, const float* const* grad_outputs,
                                            float* const* grad_weight, float* const* grad_bias, void* workspace,
                                            int64_t batch, int32_t in_features, int32_t out_features, int32_t flags,
                                            void* stream) {
    if (flags != 0 || batch < 0 || in_features < 1 || out_features < 1 || count < 1 || count > kWgMaxProblems)
        return NFA_ERR_INVALID_ARGUMENT;
    if (!inputs || !grad_outputs || !grad_weight) return NFA_ERR_INVALID_ARGUMENT;
    const int I = in_features, O = out_features;
    if ((I & 3) || (O & 3)) return NFA_ERR_UNSUPPORTED;  // (the LDS-DMA moves aligned 16-byte pieces of a row)
    for (int q = 0; q < count; ++q) {
        if (!grad_weight[q] || (batch > 0 && (!inputs[q] || !grad_outputs[q]))) return NFA_ERR_INVALID_ARGUMENT;
        if ((reinterpret_cast<uintptr_t>(inputs[q]) & 15) || (reinterpret_cast<uintptr_t>(grad_outputs[q]) & 15))
            return NFA_ERR_UNSUPPORTED;
    }
    if ((int64_t)O * I + O > (int64_t)1 << 30) return NFA_ERR_UNSUPPORTED;
    const WgradPlan p = plan_wgrad(batch, I, O, count);
    if (p.ksplit > 0 && !workspace) return NFA_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    WgradReduceArgs r;
    for (int q = 0; q < kWgMaxProblems; ++q) {
        const int s_ = q < count ? q : 0;
        r.x[q] = inputs[s_];
        r.gy[q] = grad_outputs[s_];
        r.gw[q] = grad_weight[s_];
        r.gb[q] = grad_bias ? grad_bias[s_] : nullptr;
    }
    if (p.ksplit > 0) {
        WgradArgs a;
        for (int q = 0; q < kWgMaxProblems; ++q) {
            a.x[q] = r.x[q];
            a.gy[q] = r.gy[q];
        }
        a.ws = static_cast<float*>(workspace);
        a.I = I;
        a.O = O;
        a.stages_total = p.stages_total;
        a.ksplit = p.ksplit;
        a.blocks_i = p.blocks_i;
        const dim3 grid((unsigned)(p.blocks_o * p.blocks_i), (unsigned)p.ksplit, (unsigned)count);
        // engine of the 128 x 128 result blocks: "bf16x3" (default since round 4) or "f32" (NFA_K10_ENGINE)
        const char* eng = wgrad_engine_env();
        const bool bf16 = !eng || eng[0] != 'f';
        if (p.variant == 0 && bf16 && p.rows == 16) {
            constexpr size_t lds = (size_t)kWgRing * 16 * (128 + 128) * 4;
            hipLaunchKernelGGL((wgrad_partial_bf16_kernel<2, 2, 2, 2, 16>), grid, dim3(kBlock), lds, st, a);
        } else if (p.variant == 0 && bf16) {
            constexpr size_t lds = (size_t)kWgRing * kWgRows * (128 + 128) * 4;
            static unsigned long long raised_b = 0;   // device mask (raise_dynamic_lds)
            const int rc_lds = raise_dynamic_lds((const void*)wgrad_partial_bf16_kernel<2, 2, 2, 2, kWgRows>, &raised_b, (int)lds);
            if (rc_lds != NFA_OK) return rc_lds;
            hipLaunchKernelGGL((wgrad_partial_bf16_kernel<2, 2, 2, 2, kWgRows>), grid, dim3(kBlock), lds, st, a);
        } else if (p.variant == 0) {
            constexpr size_t lds = (size_t)kWgRing * kWgRows * (128 + 128) * 4;
            static unsigned long long raised = 0;   // device mask (raise_dynamic_lds: the opt-in is per device)
            const int rc_lds = raise_dynamic_lds((const void*)wgrad_partial_kernel<2, 2, 2, 2>, &raised, (int)lds);
            if (rc_lds != NFA_OK) return rc_lds;
            hipLaunchKernelGGL((wgrad_partial_kernel<2, 2, 2, 2>), grid, dim3(kBlock), lds, st, a);
        } else {
            constexpr size_t lds = (size_t)kWgRing * kWgRows * (128 + 32) * 4;
            hipLaunchKernelGGL((wgrad_partial_kernel<1, 1, 4, 1>), grid, dim3(kBlock), lds, st, a);
        }
        NFA_HIP_CHECK(hipGetLastError());
    }
    const int64_t n = (int64_t)O * I + O;   // (problems without a bias gradient skip the last O elements themselves)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)count), dim3(kBlock), 0, st,
                       static_cast<const float*>(workspace), r, I, O, p.ksplit, (int64_t)p.stages_total * p.rows, batch);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_linear_wgrad_f32(const float* inputs, const float* grad_outputs, float* grad_weight,
                                    float* grad_bias, void* workspace, int64_t batch, int32_t in_features,
                                    int32_t out_features, int32_t flags, void* stream) {
    return nfa_linear_wgrad_batched_f32(1, &inputs, &grad_outputs, &grad_weight, &grad_bias, workspace, batch, in_features,
                                        out_features, flags, stream);
}
