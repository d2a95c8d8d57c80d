// K8c (round 6): the whole-layer kernel of rqs_resnet_f16s.hip (K8s: ResidualNet conditioner, nn/nets/resnet.py:55-100,
// + everything K1 replaces, coupling.py:73-130, :549-582, for a run of layers in one launch; GEMMs on two f16 pieces per
// fp32 operand, 16-sample tiles on v_mfma_f32_16x16x32_f16) with the GEMMs split by COLUMNS over the four waves of a
// 32-row (or 64-row) workgroup -- the small-batch form: `Flow.sample(n)` / `log_prob` of a few thousand rows (flows/base.py:51-75,
// distributions/base.py:69-84) and the per-GPU shard of a many-GPU job.
//
// Why.  K8s gives every wave 16 rows and ALL output columns: one wave per SIMD then walks dependent chains -- fragment
// read -> 24 MFMAs on one or two accumulator tiles per stage -> conversion -> next GEMM -- with nothing to overlap them
// (0.65 ms for 8 192 rows whatever the ring depth; the weight stream alone would allow 0.2).
// Here wave w owns output tiles 2 w, 2 w + 1 (32 of the 128 hidden columns) of every hidden GEMM for ALL RT 16-row
// tiles of the workgroup (RT = 2 or 4): 2 RT accumulator tiles, RT independent MFMA chains per weight fragment, a quarter
// of the fragment reads.  The next GEMM needs every column of every row tile as its B operand: each wave converts its 32 x 16 RT
// block to f16 pieces and leaves them in an LDS exchange buffer -- in B-fragment layout: k-step S of the next GEMM IS
// wave S's block (K8s's chaining rule, ops._k8s_column_order) -- one extra barrier per GEMM.  The final layer is split by
// FEATURES: wave w takes group G = 4 r + w of four features in round r (six 16-row tiles: the 24 logits of feature
// 4 G + g land in lane group g as in K8s) for all RT row tiles, its B operand -- all 128 k of all row tiles --
// resident in registers; a stage of the final layer carries two k-steps of ONE tile of
// every wave (pairs 2 w, 2 w + 1), so that every stage feeds all four waves.
//
// Stream: K8s's stages and parameter words for the initial layer and the blocks (16 KB stages of eight (hi, lo) fragment
// pairs, pair T = output tile T of one 32-wide k-step); final layer: per round of four groups 12 stages, stage 2 i + s =
// k-steps 2 s, 2 s + 1 of tile i of every wave's group (ops.pack_resnet_conditioner_f16(..., colsplit=True); groups
// beyond d_t / 4 are zero fragments).  Software pipeline as in K8x: the fragments a stage's first MFMAs need are read
// right behind the previous stage's barrier, the barrier stands between the two halves of a stage's MFMAs.
//
// Restrictions: K8s's (8 bins, linear tails, no context, hidden width 128, ReLU blocks, d_i <= 64, d_t % 4 == 0,
// d_t <= 64, D % 4 == 0, D <= 128, batch % 128 == 0).  Workgroups of four waves on RT 16-row tiles: RT = 2 (32 rows, two
// workgroups per CU: the default -- half of every serial term of the 64-row form) or RT = 4 (64 rows: NFA_K8C_ROWS=64);
// `redo_blocks`: bits 3 .. 6 of a 128-row block's word = its four 32-row quarters (RT = 2), bit 1 / bit 2 = its lower /
// upper 64 rows (RT = 4, as K8s's four-wave form).

#include "k8h_common.hpp"

namespace nfa {
namespace k8c {

using namespace k8h;

constexpr int kNW = 4, kThreads = kNW * kWave;
// RT = 16-row tiles per workgroup: 4 (64 rows) or 2 (32 rows: batches with no more 32-row blocks than CUs -- twice the
// workgroups, half of every serial term; `redo_blocks` then carries a bit per 32 rows)
constexpr int rows_of(int RT) { return 16 * RT; }
constexpr int row_pad_of(int RT) { return 16 * RT + 1; }              // [column][rows + 1]
constexpr int x_vec4_of(int RT) { return 4 * RT * 2 * 64; }           // exchange buffer: [k-step S][row tile][hi, lo][64 lanes] x 16 B
typedef vec4f f32x4;

#ifdef NFA_ABL_NO_MFMA   // (timing ablations: results are garbage)
#define NFA_K8C_MFMA(a, b, c) (c)
#else
#define NFA_K8C_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif
#define NFA_K8C_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef NFA_K8C_TRACE   // (debug build: cycle stamps of wave 0 of every workgroup, second layer of its first row block)
#define NFA_K8C_STAMP(i)                                                                              \
    if (a.trace != nullptr && layer == 1 && quad == blockIdx.x && tid == 0) a.trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter();
#else
#define NFA_K8C_STAMP(i)
#endif

// The wave's two fragment pairs (2 w, 2 w + 1) of a 16 KB stage, straight from global memory into registers: no wave
// reads another wave's pairs, so there is nothing to share through LDS -- and an LDS-DMA request costs its issuing wave
// 60-185 cycles (MI355X_MICROARCH.md), four of them per wave and stage against a stage's 24 MFMAs = 400 cycles: the ring
// of the first build cost more than the matrix work.  Plain loads, requested one GEMM (or three final-layer tiles) ahead;
// hipcc counts them (loads return in order).
struct Frag {
    vec4f h0, l0, h1, l1;
};
__device__ __forceinline__ Frag load_frag(const vec4f* wlane, int stage) {
    const vec4f* p = wlane + (size_t)stage * kStageVec4;   // pair g at g * 128 vec4: hi, + 64: lo
    return Frag{p[0], p[64], p[128], p[192]};
}

// the three products of one weight fragment pair with the pieces of the RT row tiles: RT independent chains
template <int RT>
__device__ __forceinline__ void products(f32x4* const (&a)[RT], vec4f ahw, vec4f alw, const uvec4 (&bh)[RT], const uvec4 (&bl)[RT]) {
    const f16x8 ah = __builtin_bit_cast(f16x8, ahw), al = __builtin_bit_cast(f16x8, alw);
    f16x8 h[RT], l[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        h[rt] = __builtin_bit_cast(f16x8, bh[rt]);
        l[rt] = __builtin_bit_cast(f16x8, bl[rt]);
    }
    // (smallest terms first, as K8s: the same products in the same order -- z is K8s's bit for bit)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *a[rt] = NFA_K8C_MFMA(al, h[rt], *a[rt]);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *a[rt] = NFA_K8C_MFMA(ah, l[rt], *a[rt]);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *a[rt] = NFA_K8C_MFMA(ah, h[rt], *a[rt]);
    // The operands stay live past the last product (rqs_resnet_f16s.hip: hipcc otherwise puts a renamed four-register
    // result on the registers of an operand that has just had its last use while the matrix pipe may still read it).  The
    // statement takes the accumulator as an in / out operand: a plain input-only statement has no dependence on the
    // MFMAs, and hipcc moved it in FRONT of them (tests/test_host_logic.py's assembly check found `v_mfma v[74:77],
    // v[74:77], ...` in the two-per-CU instances).  It is empty: nothing reads the result early.
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) asm volatile("" : "+v"(*a[rt]) : "v"(ah), "v"(al), "v"(h[rt]), "v"(l[rt]));
}

// pieces of k-step S for the row tiles from the exchange buffer
template <int RT>
__device__ __forceinline__ void read_pieces(const uvec4* X, int S, int lane, uvec4 (&bh)[RT], uvec4 (&bl)[RT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        bh[rt] = X[((S * RT + rt) * 2 + 0) * 64 + lane];
        bl[rt] = X[((S * RT + rt) * 2 + 1) * 64 + lane];
    }
}

// One k-major stage = one 32-wide k-step: the wave's two output tiles x four row tiles.  NEXT_S >= 0: the pieces of
// k-step NEXT_S are read between the two halves.
template <int NEXT_S, int RT>
__device__ __forceinline__ void stage_kmajor(f32x4 (&acc)[RT][2], const Frag& fr, uvec4 (&bh)[RT], uvec4 (&bl)[RT], const uvec4* X, int lane) {
    f32x4* c0[RT];
    f32x4* c1[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        c0[rt] = &acc[rt][0];
        c1[rt] = &acc[rt][1];
    }
    NFA_K8C_FENCE();
    products<RT>(c0, fr.h0, fr.l0, bh, bl);
    NFA_K8C_FENCE();
    uvec4 nh[RT], nl[RT];
    if constexpr (NEXT_S >= 0) read_pieces<RT>(X, NEXT_S, lane, nh, nl);
    NFA_K8C_FENCE();
    products<RT>(c1, fr.h1, fr.l1, bh, bl);
    NFA_K8C_FENCE();
    if constexpr (NEXT_S >= 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            bh[rt] = nh[rt];
            bl[rt] = nl[rt];
        }
    }
}

// Where the stages behind the initial layer live: a ring of four register buffers, stage k (counted from the first
// block's first stage through the final layer's last) in buffer k mod 4 -- a hidden GEMM is four stages, a final-layer tile
// two, a round twelve: every use has a compile-time buffer -- reloaded with stage k + 4 as soon as it is consumed; past the
// layer's last stage: the next layer's first ones.
struct Ahead {
    const vec4f* wlane;
    int here, next, total;   // stage numbers of this / the next layer's first hidden stage, stages per layer behind the initial one
    __device__ __forceinline__ Frag load(int k) const { return load_frag(wlane, k < total ? here + k : next + (k - total)); }
};

// 128 -> 128 GEMM on the pieces in the exchange buffer (four stages, k0 = the first one's number)
template <int RT>
__device__ __forceinline__ void gemm_hidden(f32x4 (&acc)[RT][2], Frag (&rb)[4], const Ahead& ah, int k0, const uvec4* X, int lane) {
    uvec4 bh[RT], bl[RT];
    read_pieces<RT>(X, 0, lane, bh, bl);
    stage_kmajor<1, RT>(acc, rb[0], bh, bl, X, lane);
    rb[0] = ah.load(k0 + 4);
    stage_kmajor<2, RT>(acc, rb[1], bh, bl, X, lane);
    rb[1] = ah.load(k0 + 5);
    stage_kmajor<3, RT>(acc, rb[2], bh, bl, X, lane);
    rb[2] = ah.load(k0 + 6);
    stage_kmajor<-1, RT>(acc, rb[3], bh, bl, X, lane);
    rb[3] = ah.load(k0 + 7);
    NFA_K8C_FENCE();
}

// The wave's 32 columns x 64 rows (x `scale`, ReLU'd when RELU) -> f16 pieces in the exchange buffer: k-step `wave` of
// the next GEMM (accumulator tiles 2 w, 2 w + 1 of a row tile are the eight k values lane (n, g) holds of it).  The buffer
// has two halves used in turn, so that one barrier per exchange is enough (the new contents are visible behind it).  (The guard orders the conversions -- asm blocks the hazard recogniser does not look
// into -- behind the matrix pipe's write-back of the GEMM's last products.)
template <bool RELU, int RT>
__device__ __forceinline__ void exchange(f32x4 (&acc)[RT][2], uvec4*& X, uvec4* X0, int wave, int lane, float scale, float& worst) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc[rt][0]), "+v"(acc[rt][1]));
    float peak = 0.0f;
    uvec4 h[RT], l[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        unsigned hi, lo;
        convert_pair<RELU ? kActRelu : kActNone, false>(acc[rt][0][0], acc[rt][0][1], scale, peak, hi, lo);
        h[rt][0] = hi;
        l[rt][0] = lo;
        convert_pair<RELU ? kActRelu : kActNone, false>(acc[rt][0][2], acc[rt][0][3], scale, peak, hi, lo);
        h[rt][1] = hi;
        l[rt][1] = lo;
        convert_pair<RELU ? kActRelu : kActNone, false>(acc[rt][1][0], acc[rt][1][1], scale, peak, hi, lo);
        h[rt][2] = hi;
        l[rt][2] = lo;
        convert_pair<RELU ? kActRelu : kActNone, false>(acc[rt][1][2], acc[rt][1][3], scale, peak, hi, lo);
        h[rt][3] = hi;
        l[rt][3] = lo;
    }
    worst = __builtin_fmaxf(worst, peak * scale);
    // the OTHER half of the buffer: a slower wave may still be reading this GEMM's pieces from the current one; the other
    // half was last read a GEMM ago, and every wave has passed a barrier since
    X = X == X0 ? X0 + x_vec4_of(RT) : X0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        X[((wave * RT + rt) * 2 + 0) * 64 + lane] = h[rt];
        X[((wave * RT + rt) * 2 + 1) * 64 + lane] = l[rt];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// one tile of the final layer (two stages: k-steps 0, 1 and 2, 3), RT row tiles
template <int RT>
__device__ __forceinline__ void tile_final(f32x4 (&t)[RT], Frag& f0, Frag& f1, const Ahead& ah, int k0, const uvec4 (&fh)[4][RT],
                                           const uvec4 (&fl)[4][RT]) {
    f32x4* c[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) c[rt] = &t[rt];
    NFA_K8C_FENCE();
    products<RT>(c, f0.h0, f0.l0, fh[0], fl[0]);
    NFA_K8C_FENCE();
    products<RT>(c, f0.h1, f0.l1, fh[1], fl[1]);
    NFA_K8C_FENCE();
    f0 = ah.load(k0 + 4);
    products<RT>(c, f1.h0, f1.l0, fh[2], fl[2]);
    NFA_K8C_FENCE();
    products<RT>(c, f1.h1, f1.l1, fh[3], fl[3]);
    NFA_K8C_FENCE();
    f1 = ah.load(k0 + 5);
    NFA_K8C_FENCE();
}

template <class Steps, int I, int END>
__device__ __forceinline__ void run_range(Steps& f, const RqsDev& sp) {
    if constexpr (I < END) {
        constexpr int N = Steps::kNumSlices;
        if constexpr (I < N) f.template num_w<I>();
        else if constexpr (I < 2 * N) f.template num_h<I - N>();
        else f.template finish<I - 2 * N>(sp);
        run_range<Steps, I + 1, END>(f, sp);
    }
}

__device__ __forceinline__ void load_bias4(f32x4& acc, const float* p) { acc = *reinterpret_cast<const vec4f*>(p); }

// a parameter stage's words (its first 8 KB) -> the parameter block, table entries checked and clamped
__device__ __forceinline__ void store_params(float* prm, vec4f v, int i, int nvec, const Args& a, int& my_status) {
    if (i < kTabWords / 4) {
        uvec4 u = __builtin_bit_cast(uvec4, v);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = i * 4 + c;
            const int e = (int)u[c];
            const bool used_entry = idx < kTabTr ? idx < a.di : idx - kTabTr < a.dt;
            my_status |= (used_entry && (e < 0 || e >= a.D)) ? NFA_STATUS_BAD_INDEX : 0;
            u[c] = (unsigned)(e < 0 ? 0 : (e >= a.D ? a.D - 1 : e));
        }
        v = __builtin_bit_cast(vec4f, u);
    }
    if (i < nvec) reinterpret_cast<vec4f*>(prm)[i] = v;
}

// OCC: workgroups per CU the instance is compiled for (2: 256 registers per lane, a few dozen bytes of scratch; measured no
// slower than the 512-register build of the same 32-row form even with one workgroup per CU)
template <bool INVERSE, int INIT_KS, int RT, int OCC = 1>
__global__ void __launch_bounds__(kThreads, OCC) rqs_resnet_f16c_kernel(const Args a) {
    constexpr int kRows = rows_of(RT), kRowPadC = row_pad_of(RT), kXVec4 = x_vec4_of(RT);
    constexpr int kParts = kThreads / kRows;   // threads per row in the epilogue's column sums
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ int s_final[128];
    __shared__ int s_bad[kNW];
    __shared__ float s_red[2][8][64];   // [log-determinant shares per wave | squared sums per column part][row]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = a.D, dt = a.dt;
    int my_status = 0;
    {   // (every thread: entry tid mod 128)
        const int te = tid & 127;
        const int v = a.final_tab[te];
        my_status |= (te < D && (v < 0 || v >= D)) ? NFA_STATUS_BAD_INDEX : 0;
        s_final[te] = v < 0 ? 0 : (v >= D ? D - 1 : v);
    }

    const int pblock = (a.param_words + 3) & ~3, pvec = pblock >> 2;   // (<= 512 vec4: one parameter stage, two per thread)
    uvec4* const X0 = reinterpret_cast<uvec4*>(lds_dyn);  // two halves of kXVec4 vectors (exchange)
    uvec4* X = X0;
    float* s_row = lds_dyn + 2 * kXVec4 * 4;              // [D][kRowPadC]
    float* s_param = s_row + D * kRowPadC;                // [2][pblock]
    const int groups = dt >> 2;
    const int rounds = (groups + 3) >> 2;
    const int nb = a.num_blocks;
    const int64_t num_quads = a.batch / kRows;
    const int g = lane >> 4, n = lane & 15;
    const vec4f* wlane = a.w + wave * 256 + lane;          // this lane's 16 bytes of pair 2 w of stage 0
    // stage numbers inside a layer: parameters, initial layer, blocks, final layer
    const int st_init = 1, st_hidden = 1 + INIT_KS;
    int pb = 0;
    // layer 0's parameters and initial-layer fragments
    {
        const vec4f* src = a.w;
        const int i0 = tid, i1 = tid + kThreads;
        const vec4f v0 = src[i0 < pvec ? i0 : 0], v1 = src[i1 < pvec ? i1 : 0];
        store_params(s_param, v0, i0, pvec, a, my_status);
        store_params(s_param, v1, i1, pvec, a, my_status);
    }
    Frag fi[INIT_KS];
#pragma unroll
    for (int S = 0; S < INIT_KS; ++S) fi[S] = load_frag(wlane, st_init + S);
    Frag rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[j] = load_frag(wlane, st_hidden + j);
    int next_layer_stage0 = a.num_layers > 1 ? a.num_stages : 0;   // stage 0 of the layer after the current one (cyclic)

    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = quad * kRows;
        // ---- the workgroup's 64 rows: one coalesced read; slot j of the tile = input column j
        {
            const vec4f* xv = reinterpret_cast<const vec4f*>(a.x + row0 * D);
            const int nvec = D * (kRows / 4);
            for (int e = tid; e < nvec; e += kThreads) {
                const vec4f v = xv[e];
                const int rr = (e * 4) / D, c0 = e * 4 - rr * D;
                s_row[(c0 + 0) * kRowPadC + rr] = v.x;
                s_row[(c0 + 1) * kRowPadC + rr] = v.y;
                s_row[(c0 + 2) * kRowPadC + rr] = v.z;
                s_row[(c0 + 3) * kRowPadC + rr] = v.w;
            }
        }
        __syncthreads();

        float lad_acc[RT];   // per row tile: rows 16 rt + n (this lane's features)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) lad_acc[rt] = 0.0f;
        float worst = 0.0f;
        int quad_status = 0;
        for (int layer = 0; layer < a.num_layers; ++layer) {
            const int stage0 = layer * a.num_stages;             // this layer's stage 0 in the stream
            NFA_K8C_STAMP(0)
            // ---- the next layer's parameter words: requested now, stored at the end of the layer
            const vec4f* pnext = a.w + (size_t)next_layer_stage0 * kStageVec4;
            const vec4f pn0 = pnext[tid < pvec ? tid : 0], pn1 = pnext[tid + kThreads < pvec ? tid + kThreads : 0];
            float* prm = s_param + pb * pblock;
            const int* tab = reinterpret_cast<const int*>(prm);
            const float* gemm = prm + kTabWords;   // header + biases of the next GEMM
            const Ahead ah{wlane, stage0 + st_hidden, next_layer_stage0 + st_hidden, 8 * nb + 12 * rounds};
            f32x4 hacc[RT][2];   // the residual stream h of the wave's 32 columns, fp32 (x the scale of the GEMM that wrote it)
            // ---- initial layer on the identity features (scale 1): k = 32 S + 8 g + j
            {
                const float* bias = gemm + kHdr + (2 * wave) * 16 + g * 4;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    load_bias4(hacc[rt][0], bias);
                    load_bias4(hacc[rt][1], bias + 16);
                }
#pragma unroll
                for (int S = 0; S < INIT_KS; ++S) {
                    uvec4 ih[RT], il[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                        for (int j2 = 0; j2 < 4; ++j2) {
                            const int i0 = S * 32 + g * 8 + j2 * 2;
                            float v0 = s_row[tab[kTabId + i0] * kRowPadC + rt * 16 + n];
                            float v1 = s_row[tab[kTabId + i0 + 1] * kRowPadC + rt * 16 + n];
                            v0 = i0 < a.di ? v0 : 0.0f;
                            v1 = i0 + 1 < a.di ? v1 : 0.0f;
                            unsigned hi, lo;
                            split2(v0, v1, hi, lo);
                            ih[rt][j2] = hi;
                            il[rt][j2] = lo;
                        }
                    }
                    stage_kmajor<-1, RT>(hacc, fi[S], ih, il, X, lane);
                }
            }
            float conv_scale = gemm[0];
            gemm += kHdr + 128;
            NFA_K8C_STAMP(2)
            if (nb > 0) exchange<true, RT>(hacc, X, X0, wave, lane, conv_scale, worst);
            else exchange<false, RT>(hacc, X, X0, wave, lane, conv_scale, worst);
            NFA_K8C_STAMP(3)

            // ---- residual blocks: h += W_1 relu(W_0 relu(h) + b_0) + b_1
            for (int blk = 0; blk < nb; ++blk) {
                f32x4 u[RT][2];
                {
                    const float* bias = gemm + kHdr + (2 * wave) * 16 + g * 4;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        load_bias4(u[rt][0], bias);
                        load_bias4(u[rt][1], bias + 16);
                    }
                    gemm_hidden<RT>(u, rb, ah, blk * 8, X, lane);
                    conv_scale = gemm[0];
                }
                gemm += kHdr + 128;
                NFA_K8C_STAMP(4 + blk * 4)
                exchange<true, RT>(u, X, X0, wave, lane, conv_scale, worst);
                NFA_K8C_STAMP(5 + blk * 4)
                {
                    // the second Linear accumulates into the residual stream itself: hacc = hacc * ratio + bias, then + W_1 relu(u)
                    const float* bias = gemm + kHdr + (2 * wave) * 16 + g * 4;
                    const float ratio = gemm[1];
                    const vec4f b0 = *reinterpret_cast<const vec4f*>(bias), b1 = *reinterpret_cast<const vec4f*>(bias + 16);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            hacc[rt][0][i] = __builtin_fmaf(hacc[rt][0][i], ratio, b0[i]);
                            hacc[rt][1][i] = __builtin_fmaf(hacc[rt][1][i], ratio, b1[i]);
                        }
                    }
                    gemm_hidden<RT>(hacc, rb, ah, blk * 8 + 4, X, lane);
                    conv_scale = gemm[0];
                }
                gemm += kHdr + 128;
                NFA_K8C_STAMP(6 + blk * 4)
                // pieces of relu(h) for the next block, of h itself for the final layer (no ReLU in front of it: resnet.py:99-100)
                if (blk + 1 < nb) exchange<true, RT>(hacc, X, X0, wave, lane, conv_scale, worst);
                else exchange<false, RT>(hacc, X, X0, wave, lane, conv_scale, worst);
                NFA_K8C_STAMP(7 + blk * 4)
            }

            // ---- final layer: wave w evaluates group G = 4 r + w in round r; the six tiles of the group hold the 24 logits
            //      of this lane's feature 4 G + g for the four row tiles
            {
                using Steps = FusedSteps<INVERSE, 8>;
                const float kappa = gemm[0];
                const float tail_s = a.sp.tail_logit * gemm[1];   // gemm[1] = 1 / kappa
                uvec4 fh[4][RT], fl[4][RT];   // [k-step][row tile]
#pragma unroll
                for (int S = 0; S < 4; ++S) read_pieces<RT>(X, S, lane, fh[S], fl[S]);
                NFA_K8C_STAMP(20)
                for (int r = 0; r < rounds; ++r) {
                    const int G = 4 * r + wave;
                    const int Gc = G < groups ? G : groups - 1;   // (a wave without a group: zero fragments, nothing evaluated)
                    const float* fbias = gemm + kHdr + (Gc * 6) * 16 + g * 4;
                    f32x4 t[6][RT];
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        f32x4 b;
                        load_bias4(b, fbias + i * 16);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) t[i][rt] = b;
                        tile_final<RT>(t[i], rb[(2 * i) % 4], rb[(2 * i + 1) % 4], ah, 8 * nb + r * 12 + 2 * i, fh, fl);
                        // (behind the layer's last tiles: the next layer's initial-layer fragments)
                        if (r == rounds - 1 && i == 5) {
#pragma unroll
                            for (int S = 0; S < INIT_KS; ++S) fi[S] = load_frag(wlane, next_layer_stage0 + st_init + S);
                        }
                        NFA_K8C_FENCE();
                    }
                    NFA_K8C_STAMP(21 + 2 * r)
                    if (G < groups) {
                        const int slot_col = tab[kTabTr + G * 4 + g] * kRowPadC;
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            Steps f;
                            f.kappa = kappa;
                            f.kl2e = 1.44269502162933349609375f * kappa;
                            f.tail_s = tail_s;
                            float* slot = s_row + slot_col + rt * 16 + n;
                            f.x = *slot;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                f.ew[j] = t[0][rt][j];
                                f.ew[4 + j] = t[1][rt][j];
                                f.eh[j] = t[2][rt][j];
                                f.eh[4 + j] = t[3][rt][j];
                                f.sd[j] = t[4][rt][j];
                                if (j < 3) f.sd[4 + j] = t[5][rt][j];
                            }
#ifndef NFA_ABL_NO_WEAVE
                            run_range<Steps, 0, 2 * Steps::kNumSlices + Steps::kFinishSlices>(f, a.sp);
#else
                            f.y = f.x + f.ew[0] + f.sd[6]; f.lad = f.eh[7]; f.status = 0;
#endif
                            *slot = f.y;
                            lad_acc[rt] += f.lad;
                            quad_status |= f.status;
                        }
                    }
                    NFA_K8C_STAMP(22 + 2 * r)
                }
            }
            // ---- the next layer's parameters (the other block), and the layer's end: every wave's spline results are in
            //      the row tile, the exchange buffer is free
            store_params(s_param + (pb ^ 1) * pblock, pn0, tid, pvec, a, my_status);
            store_params(s_param + (pb ^ 1) * pblock, pn1, tid + kThreads, pvec, a, my_status);
            pb ^= 1;
            next_layer_stage0 = next_layer_stage0 + a.num_stages;
            if (next_layer_stage0 >= a.num_stages * a.num_layers) next_layer_stage0 = 0;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }

        // ---- results: position p of a row comes from slot final[p]; a block with any non-finite value or an
        //      activation beyond the f16 range is not written: the exact kernel redoes it from the inputs
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            lad_acc[rt] += __shfl_xor(lad_acc[rt], 16, kWave);
            lad_acc[rt] += __shfl_xor(lad_acc[rt], 32, kWave);
        }
        // (lane group g leaves row tile g's sum -- RT = 2: groups 0, 1 --: one store per lane)
        {
            float mine = lad_acc[0];
#pragma unroll
            for (int rt = 1; rt < RT; ++rt) mine = g == rt ? lad_acc[rt] : mine;
            if (g < RT) s_red[0][wave][lane] = mine;
        }
        {
            const int row = tid % kRows, part = tid / kRows;
            float sumsq = 0.0f;
            for (int j = part; j < a.Ds; j += kParts) {
                const float v = s_row[j * kRowPadC + row];
                sumsq = __builtin_fmaf(v, v, sumsq);
            }
            s_red[1][part][row] = sumsq;
        }
        const bool wave_bad = __builtin_amdgcn_ballot_w64(!(worst < kF16Overflow)) != 0;
        if (lane == 0) s_bad[wave] = wave_bad ? 1 : 0;
        __syncthreads();
        // (every wave computes all rows' totals: row = lane mod rows)
        const int erow = lane % kRows;
        const float lad_row = (s_red[0][0][erow] + s_red[0][1][erow]) + (s_red[0][2][erow] + s_red[0][3][erow]);
        float sumsq_row = 0.0f;
#pragma unroll
        for (int part = 0; part < kParts; ++part) sumsq_row += s_red[1][part][erow];
        const bool bad = !(__builtin_fabsf(lad_row) < INFINITY) || !(__builtin_fabsf(sumsq_row) < INFINITY);
        const bool quad_bad = (__builtin_amdgcn_ballot_w64(bad) != 0) || ((s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) != 0);
        if (!quad_bad) {
            if (!a.skip_out) {
                const int nvec = D * (kRows / 4);
                vec4f* ov = reinterpret_cast<vec4f*>(a.out + row0 * D);
                for (int e = tid; e < nvec; e += kThreads) {
                    const int rr = (e * 4) / D, c0 = e * 4 - rr * D;
                    vec4f v;
                    v.x = s_row[s_final[c0 + 0] * kRowPadC + rr];
                    v.y = s_row[s_final[c0 + 1] * kRowPadC + rr];
                    v.z = s_row[s_final[c0 + 2] * kRowPadC + rr];
                    v.w = s_row[s_final[c0 + 3] * kRowPadC + rr];
                    ov[e] = v;
                }
            }
            if (wave == 0 && lane < kRows) {
                float* dst = a.lad + row0 + lane;
                float v = a.accumulate ? *dst + lad_row : lad_row;
                if (a.normal) v = (-0.5f * sumsq_row - a.log_z) + v;   // normal.py:31-33, flows/base.py:49
                *dst = v;
            }
            my_status |= quad_status;
        }
        // one flag word per 128 rows, zeroed by the launcher: 64-row workgroups set bit 1 / bit 2 (its lower / upper 64 rows are
        // open), 32-row workgroups bits 3 .. 6 (its four quarters): rqs_resnet_kernel.hpp gives the rows to the redo pass's waves
        if (tid == 0 && quad_bad) {
            if constexpr (RT == 4) atomicOr(a.redo + (quad >> 1), 2 << (quad & 1));
            else atomicOr(a.redo + (quad >> 2), 8 << (quad & 3));
        }
        __syncthreads();  // the row tile, s_bad and s_red are rewritten by the next row block
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

__global__ void zero_words_kernel(int32_t* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

}  // namespace k8c
}  // namespace nfa

using namespace nfa;

extern "C" int nfa_rqs_flow_resnet_f16x2_colsplit_f32(const float* inputs, const void* stream_packed, int32_t param_stages,
                                                      const int32_t* final_positions, int32_t num_layers, float* outputs,
                                                      float* logabsdet, int32_t* redo_blocks, int32_t* status, int64_t batch,
                                                      int32_t features, int32_t num_transform, int32_t num_identity,
                                                      int32_t hidden_features, int32_t num_blocks, const nfa_rqs_spec* spec,
                                                      int32_t flags, void* stream) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET | NFA_FLAG_STANDARD_NORMAL_LOG_PROB |
                  NFA_FLAG_SKIP_OUTPUTS | NFA_FLAG_PAD_COLUMNS_MASK))
        return NFA_ERR_INVALID_ARGUMENT;
    if (!density_flags_valid(flags)) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1 || num_transform < 1 || num_identity < 1 || num_transform > features ||
        num_identity > features || num_blocks < 0 || num_layers < 1 || param_stages < 1)
        return NFA_ERR_INVALID_ARGUMENT;
    k8h::Args a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (a.sp.beta != 1.0f) return NFA_ERR_UNSUPPORTED;
    if (a.sp.K != 8 || !a.sp.linear || hidden_features != 128 || (num_transform & 3) != 0 || num_transform > 64 ||
        num_identity > 64 || features > 128 || (features & 3) != 0 || (batch & 127) != 0 || num_blocks > 64 ||
        num_layers > 4096)
        return NFA_ERR_UNSUPPORTED;
    const int param_words = k8h::kTabWords + (k8h::kHdr + 128) * (1 + 2 * num_blocks) + k8h::kHdr + num_transform * 24;
    if (param_stages * 2048 < param_words || param_stages > 4) return NFA_ERR_INVALID_ARGUMENT;
    if (param_stages != 1) return NFA_ERR_UNSUPPORTED;   // (the next layer's words travel in two registers per thread)
    if (batch == 0) return NFA_OK;
    if (!inputs || !stream_packed || !final_positions || !logabsdet || !redo_blocks ||
        (!outputs && !(flags & NFA_FLAG_SKIP_OUTPUTS)))
        return NFA_ERR_INVALID_ARGUMENT;
    a.ctx = nullptr;
    a.ce = 0;
    a.dbg_bins = nullptr;
    a.dbg_logits = nullptr;
    a.normal = (flags & NFA_FLAG_STANDARD_NORMAL_LOG_PROB) ? 1 : 0;
    a.skip_out = (flags & NFA_FLAG_SKIP_OUTPUTS) ? 1 : 0;
    a.Ds = density_columns(flags, features);
    if (a.Ds < 1) return NFA_ERR_INVALID_ARGUMENT;
    a.log_z = standard_normal_log_z(a.Ds);
    a.x = inputs;
    a.w = reinterpret_cast<const vec4f*>(stream_packed);
    a.final_tab = final_positions;
    a.out = outputs;
    a.lad = logabsdet;
    a.redo = redo_blocks;
    a.status = status;
    a.batch = batch;
    a.D = features;
    a.dt = num_transform;
    a.di = num_identity;
    a.num_blocks = num_blocks;
    a.num_layers = num_layers;
    a.param_stages = param_stages;
    a.param_words = param_words;
    const int init_ks = num_identity > 32 ? 2 : 1;
    // stages per layer: parameters, one per k-step of the initial layer, four per hidden Linear, twelve per round of four
    // groups of four transformed features
    const int rounds = (num_transform / 4 + 3) / 4;
    a.num_stages = param_stages + init_ks + 8 * num_blocks + 12 * rounds;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    a.trace = nullptr;
#ifdef NFA_K8C_TRACE
    static unsigned long long* trace_dev = nullptr;
    if (!trace_dev) hipMalloc(&trace_dev, 256 * 64 * 8);
    hipMemset(trace_dev, 0, 256 * 64 * 8);
    a.trace = trace_dev;
#endif
    const int cus = device_cu_count();
    // batches with no more 32-row blocks than CUs: 32-row workgroups (twice the workgroups, half of every serial term)
    // 32-row workgroups, two per CU (twice the workgroups of the 64-row form, half of every serial term per workgroup: 8 192
    // rows 0.37 ms against 0.60; 16 384 rows 0.59 against 0.61); NFA_K8C_ROWS=64: the 64-row form, one per CU
    static const int rt_env = getenv("NFA_K8C_ROWS") ? atoi(getenv("NFA_K8C_ROWS")) : 0;
    const int RT = rt_env == 64 ? 4 : 2;
    const int occ = RT == 2 ? 2 : 1;
    const size_t lds = (size_t)2 * k8c::x_vec4_of(RT) * 16 + (size_t)features * k8c::row_pad_of(RT) * sizeof(float) +
                       (size_t)2 * ((param_words + 3) & ~3) * sizeof(float);
    const size_t lds_cap = 160 * 1024 - 8192;   // (beside 4.6 KB of static arrays)
    if (lds > lds_cap) return NFA_ERR_UNSUPPORTED;
    int64_t blocks = batch / k8c::rows_of(RT);
    if (blocks > (int64_t)cus * occ) blocks = (int64_t)cus * occ;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    profile_next_launch(&e0, &e1);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(k8c::kThreads);
    const bool inv = (flags & NFA_FLAG_INVERSE) != 0;
    void (*kern)(const k8h::Args) = nullptr;
    const int which = (inv ? 1 : 0) + (init_ks == 2 ? 2 : 0) + (RT == 2 ? 4 : 0);
    switch (which) {
        case 0: kern = k8c::rqs_resnet_f16c_kernel<false, 1, 4, 1>; break;
        case 1: kern = k8c::rqs_resnet_f16c_kernel<true, 1, 4, 1>; break;
        case 2: kern = k8c::rqs_resnet_f16c_kernel<false, 2, 4, 1>; break;
        case 3: kern = k8c::rqs_resnet_f16c_kernel<true, 2, 4, 1>; break;
        case 4: kern = k8c::rqs_resnet_f16c_kernel<false, 1, 2, 2>; break;
        case 5: kern = k8c::rqs_resnet_f16c_kernel<true, 1, 2, 2>; break;
        case 6: kern = k8c::rqs_resnet_f16c_kernel<false, 2, 2, 2>; break;
        default: kern = k8c::rqs_resnet_f16c_kernel<true, 2, 2, 2>; break;
    }
    note_layer_kernel("k8c::rqs_resnet_f16c_kernel<inverse=%d, init_ks=%d, waves=4, rows=%d, per_cu=%d, K=8>", inv ? 1 : 0, init_ks, 16 * RT, occ);
    hipLaunchKernelGGL(k8c::zero_words_kernel, dim3((unsigned)((batch / 128 + 255) / 256)), dim3(256), 0, st, redo_blocks,
                       (int)(batch / 128));
    if (lds > 64 * 1024) {
        static unsigned long long raised[8] = {};   // device masks (raise_dynamic_lds)
        const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[which], (int)lds_cap);
        if (rc_lds != NFA_OK) return rc_lds;
    }
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
#ifdef NFA_K8C_TRACE
    {
        static int calls = 0;
        if (++calls == 20) {
            hipDeviceSynchronize();
            static unsigned long long host[256 * 64];
            hipMemcpy(host, trace_dev, sizeof(host), hipMemcpyDeviceToHost);
            for (int b = 0; b < 2; ++b) {
                fprintf(stderr, "k8c trace block %d:", b);
                for (int i = 1; i < 30; ++i)
                    if (host[b * 64 + i]) fprintf(stderr, " [%d]%lld", i, (long long)(host[b * 64 + i] - host[b * 64]));
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    return NFA_OK;
}
