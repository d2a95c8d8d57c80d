// K12: the sequential part of the autoregressive spline layer's inverse (sampling direction) as ONE
// persistent kernel -- AutoregressiveTransform.inverse (autoregressive.py:43-52) for
// MaskedPiecewiseRationalQuadraticAutoregressiveTransform (autoregressive.py:404-495) over a MADE
// conditioner (made.py:233-311, masked residual / feed-forward blocks :67-231).
//
// The reference runs the whole MADE D times.  What actually changes from one iteration to the next is
// little: a hidden unit of degree d is connected to inputs of degree <= d only (made.py:60-63), i.e. it
// is a function of features 0 .. d-1 and final as soon as feature d-1 has been found.  So step t
//   1. evaluates the hidden units of degree t, layer by layer (for H < D - 1 sequential degrees that is
//      one unit per layer: five 256-term dot products instead of four 256 x 256 products per sample),
//   2. takes feature t's P rows of the output layer on the hidden vector as it stands (units of higher
//      degree are still zero and meet exactly-zero masked weights),
//   3. inverts the spline of feature t and records x_t.
// After step T - 1 (T = largest hidden degree) plus one more round of unit updates nothing changes
// any more; the caller finishes features T .. D-1 with one GEMM and one elementwise launch
// (transforms/autoregressive.py).  Same sums as the reference's masked GEMMs in another order.
//
// Samples are independent, so there is no grid-wide step: a workgroup of four waves owns 16 samples for
// all steps.  Round 3: every wave owns FOUR of them for the whole step -- unit chain, output rows, spline --
// with SIXTEEN lanes per sample (lane = (sample sw = lane / 16, part q = lane % 16)): a 256-term dot product
// is four 16-byte chunks per lane and a four-stage DPP butterfly (quad_perm x 2, row_half_mirror,
// row_mirror: no LDS round trip), all sixteen lanes end up with every sum, so the 3K - 1 logits of a
// feature never leave the registers.  (Round 2 gave the chain to wave 0 with four lanes per sample -- 16
// chunks per lane and two ds_bpermute exchanges per unit, 14.7 k of the step's 18.8 k cycles -- and the
// output rows to waves 1..3, joined by three workgroup barriers and a rank-1 correction.)  The waves only
// meet once per step, for the hand-over of the staged block.
// Per-sample state lives in LDS as [vector][16 samples][chunks, padded to a multiple of 16][4 floats]: the
// sixteen lanes of a sample read sixteen consecutive chunks (conflict-free; the samples of a wave are a
// multiple of 64 banks apart), a weight read is one address per part q, broadcast to the four samples.
//
// Everything step t needs from global memory is ONE contiguous block prepared by the host (weights
// pre-masked, rows sorted by degree, zero-padded): a header with the number of units per layer, the
// units' rows, the feature's output rows, the biases and unit indices.  Which block comes next does not
// depend on data, so block t + 1 is pulled into the other half of an LDS double buffer by LDS-DMA
// (global_load_lds, no registers, no waiting) while step t computes; the dot products read weights and
// state from LDS only (a weight read is a 4-address broadcast).  The first version read the rows from
// global memory inside the dot-product loops and spent its time waiting for L2: 35 us per step.
//
// Supported: 8 or 10 bins with linear tails (P = 23 / 29), ReLU, residual blocks (sequential masks) or
// feed-forward blocks (any masks), no context, no batch norm; state that fits the LDS.

#include "fused_common.hpp"

#include <hip/hip_ext.h>
#include <stdlib.h>

namespace nfa {

constexpr int kMadeMaxLinears = 12;
constexpr int kMadeSamples = 16;     // per workgroup
constexpr int kMadeWaves = 4;
constexpr int kMadeHeader = 16;      // ints in front of a step block: {units, offset of the unit rows, of the output rows, of the output biases}
constexpr int kMadeUnitWords = 8;    // per unit, right behind the header: {bias, index, columns, src vector, dst vector, add_stream, set_stream, 0}
#ifndef NFA_K12_ROWS_AHEAD
#define NFA_K12_ROWS_AHEAD 4   // groups of the output rows' dot products requested ahead (dot_rows_64); measured 2: 5.2 k, 3: 4.6 k, 4: 4.0 k cycles per step
#endif
constexpr int kMadeUnitsAhead = 4;   // unit entries every lane reads together with the header (more units: read one by one)
constexpr int kMadeGrain = 256;      // blocks are multiples of 256 floats (one LDS-DMA request of the wave)

struct MadeInvArgs {
    const float* z;          // [B, D] values to invert
    float* x;                // [B, D] features found (columns < T written)
    float* lad;              // [B]   sum of the log-derivatives of columns < T
    float* hidden;           // [B, H] the output layer's input once all hidden units are final
    const float* blocks;     // the step blocks, one after the other
    const int32_t* block_at; // [T + 2] start of block t in grains
    int32_t* status;
    int64_t batch;
    int D, H, Hp, Xp, T, num_linears, residual, final_src, stream_vec, num_vectors, max_block;  // max_block in floats
    int kp[kMadeMaxLinears], src[kMadeMaxLinears], dst[kMadeMaxLinears], add_stream[kMadeMaxLinears],
        set_stream[kMadeMaxLinears];
    RqsDev sp;
    unsigned long long* trace;   // debug (nfa_debug_k7_trace): cycle stamps of workgroup 0, wave 0 over the first steps
};

// sum over the sixteen lanes of a sample (a DPP row), every lane gets it
__device__ __forceinline__ float row_sum16(float v) {
    auto dpp = [](float x, int ctrl) {
        return __builtin_bit_cast(float, ctrl == 0xB1   ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false)
                                       : ctrl == 0x4E  ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false)
                                       : ctrl == 0x141 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false)
                                                       : __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
    };
    v += dpp(v, 0xB1);    // quad_perm [1, 0, 3, 2]
    v += dpp(v, 0x4E);    // quad_perm [2, 3, 0, 1]
    v += dpp(v, 0x141);   // row_half_mirror: the other quad of the half row
    v += dpp(v, 0x140);   // row_mirror: the other half row
    return v;
}

// sum over the LPS lanes of a sample (LPS = 16: a DPP row; LPS = 32 -- round 4, two samples per wave --: the two
// rows of a 32-lane half exchange their sums through ds_swizzle, lane i <-> i ^ 16)
template <int LPS>
__device__ __forceinline__ float sample_sum(float v) {
    v = row_sum16(v);
    if constexpr (LPS == 32)
        v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // and 0x1f, or 0, xor 0x10
    return v;
}

// `R` dot products of weight rows (`pitch` floats apart) with one sample's state vector `vec` (its `chunks`
// 16-byte chunks), both in LDS: this lane takes chunks q, q + 16, ...; every lane of the sample gets the sums
template <int R, int UNROLL, int LPS = 16>
__device__ __forceinline__ void dot_rows(float (&acc)[R], const float* rows, int pitch, const float* vec, int chunks,
                                         int q, int nrows) {
    // (no conditionals inside the loop: rows beyond `nrows` re-read row 0 and their sums are ignored --
    // a branch per row would keep the compiler from batching the LDS reads of several chunks)
    const float* rp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r] = 0.0f;
        rp[r] = rows + (r < nrows ? r : 0) * pitch;
    }
#pragma unroll UNROLL
    for (int c = q; c < chunks; c += LPS) {
        const vec4f v = *reinterpret_cast<const vec4f*>(vec + c * 4);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const vec4f w = *reinterpret_cast<const vec4f*>(rp[r] + c * 4);
            acc[r] = __builtin_fmaf(w.x, v.x, acc[r]);
            acc[r] = __builtin_fmaf(w.y, v.y, acc[r]);
            acc[r] = __builtin_fmaf(w.z, v.z, acc[r]);
            acc[r] = __builtin_fmaf(w.w, v.w, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = sample_sum<LPS>(acc[r]);
}

// the same for exactly 64 chunks (256 columns: BASELINE configs[4]'s hidden width), software-pipelined by hand:
// the reads of the next 16 chunks are in flight while the current ones are multiplied (hipcc waits for ALL
// outstanding LDS reads in front of the first FMA of a loop body otherwise: four exposed round trips per call)
template <int R, int AHEAD_, int LPS = 16>
__device__ __forceinline__ void dot_rows_64(float (&acc)[R], const float* rows, int pitch, const float* vec, int q, int nrows) {
    // G groups of LPS chunks (LPS = 16: four, LPS = 32: two); AHEAD = how many groups are requested before the first
    // one is multiplied (G = everything up front: right for a single row, whose four FMAs per group hide nothing)
    constexpr int G = 64 / LPS;
    constexpr int AHEAD = AHEAD_ < G ? AHEAD_ : G;
    static_assert(AHEAD >= 1, "");
    const float* rp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rp[r] = rows + (r < nrows ? r : 0) * pitch + q * 4;
    const float* vp = vec + q * 4;
    vec4f v[G], w[G][R];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) {
        v[i] = *reinterpret_cast<const vec4f*>(vp + i * LPS * 4);
#pragma unroll
        for (int r = 0; r < R; ++r) w[i][r] = *reinterpret_cast<const vec4f*>(rp[r] + i * LPS * 4);
    }
    float part[G][R];   // one partial sum per group: short dependency chains instead of one of sixteen
#pragma unroll
    for (int i = 0; i < G; ++i) {
        if (i + AHEAD < G) {
            v[i + AHEAD] = *reinterpret_cast<const vec4f*>(vp + (i + AHEAD) * LPS * 4);
#pragma unroll
            for (int r = 0; r < R; ++r) w[i + AHEAD][r] = *reinterpret_cast<const vec4f*>(rp[r] + (i + AHEAD) * LPS * 4);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = w[i][r].x * v[i].x;
            t = __builtin_fmaf(w[i][r].y, v[i].y, t);
            t = __builtin_fmaf(w[i][r].z, v[i].z, t);
            part[i][r] = __builtin_fmaf(w[i][r].w, v[i].w, t);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float t;
        if constexpr (G == 4) t = (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
        else t = part[0][r] + part[1][r];
        acc[r] = sample_sum<LPS>(t);
    }
}

// float offset of element k of a sample's vector whose samples are `pitch` chunks apart
__device__ __forceinline__ int state_index(int k, int s, int pitch) { return (s * pitch + (k >> 2)) * 4 + (k & 3); }

// block `t` -> LDS at `dst`: a wave requests one grain (64 lanes x 16 bytes) per instruction, grain g is
// wave g % 4's
template <int NW>
__device__ __forceinline__ void request_block(const MadeInvArgs& a, int t, float* dst, int lane, int wave) {
    const int g0 = a.block_at[t], g1 = a.block_at[t + 1];
    const char* src = reinterpret_cast<const char*>(a.blocks) + (size_t)g0 * (kMadeGrain * 4) + lane * 16;
    char* d = reinterpret_cast<char*>(dst);
    for (int g = wave; g < g1 - g0; g += NW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)g * (kMadeGrain * 4)),
                                         (__attribute__((address_space(3))) void*)(d + g * (kMadeGrain * 4)), 16, 0, 0);
}

// LPS = lanes per sample.  16: four waves x four samples (round 3).  32 (round 4): EIGHT waves x two samples -- the same
// sixteen samples and the same LDS per workgroup, but two waves per SIMD: a batch that gives every CU at most one
// workgroup (configs[4]: 4 096 samples = 256 workgroups) has nothing else to hide the chain's LDS round trips with.
template <int KT, int LPS>
__global__ void __launch_bounds__((LPS / 4) * kWave) made_rqs_inverse_kernel(const MadeInvArgs a) {
#pragma clang fp contract(off)
    constexpr int P = 3 * KT - 1;
    constexpr int RB = 8;                                           // output rows per pass of dot_rows
    constexpr int NW = LPS / 4;                                     // waves per workgroup (16 samples)
    constexpr int SPW = kWave / LPS;                                // samples per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & (LPS - 1), s = wave * SPW + lane / LPS;
    const int64_t row = (int64_t)blockIdx.x * kMadeSamples + s;
    const bool live = row < a.batch;
    const int64_t rrow = live ? row : a.batch - 1;
    const int px = ((a.Xp >> 2) + 15) & ~15, ph = ((a.Hp >> 2) + 15) & ~15;   // chunks between two samples
    const int x_floats = kMadeSamples * px * 4, vec_floats = kMadeSamples * ph * 4;
    const int state_floats = x_floats + a.num_vectors * vec_floats;
    float* xs = lds;                                                // [16][px][4]
    float* vecs = lds + x_floats;                                   // [num_vectors][16][ph][4]
    float* buf0 = lds + state_floats;                               // two step blocks
    float* buf1 = buf0 + a.max_block;
    request_block<NW>(a, 0, buf0, lane, wave);
    for (int i = threadIdx.x; i < state_floats; i += NW * kWave) lds[i] = 0.0f;

    float lad_acc = 0.0f;
    int my_status = 0;
    const float* zrow = a.z + rrow * a.D;
    float* stream = a.residual ? vecs + a.stream_vec * vec_floats : nullptr;
    float z_next = zrow[0];
    unsigned long long* tr = (a.trace && blockIdx.x == 0 && threadIdx.x == 0) ? a.trace : nullptr;
    int ti = 0;
#define NFA_K12_STAMP() if (tr && ti < 250) tr[ti++] = __builtin_readcyclecounter();
    for (int t = 0; t <= a.T; ++t) {
        float* blk = (t & 1) ? buf1 : buf0;
        NFA_K12_STAMP()
        // block t has landed (every wave's share, requested a whole step ago), every wave is done with the
        // other buffer half (and, at t = 0, the state is zeroed)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        NFA_K12_STAMP()
        const float z_t = z_next;
        if (t < a.T) {
            request_block<NW>(a, t + 1, (t & 1) ? buf0 : buf1, lane, wave);
            if (t + 1 < a.T) z_next = zrow[t + 1];
        }
        // ---- the step's control data in ONE round trip: header and the first unit entries (a dependent LDS
        //      read per field was most of the unit chain's time: five to six round trips per unit)
        typedef int vec4i __attribute__((ext_vector_type(4)));
        const vec4i* bv = reinterpret_cast<const vec4i*>(blk);
        const vec4i h0 = bv[0];
        vec4i ue[kMadeUnitsAhead][2];
#pragma unroll
        for (int u = 0; u < kMadeUnitsAhead; ++u) {
            ue[u][0] = bv[kMadeHeader / 4 + 2 * u];
            ue[u][1] = bv[kMadeHeader / 4 + 2 * u + 1];
        }
        const int units = __builtin_amdgcn_readfirstlane(h0.x);
        const float* rows = blk + __builtin_amdgcn_readfirstlane(h0.y);
        const float* wf = blk + __builtin_amdgcn_readfirstlane(h0.z);
        const float* fbias = blk + __builtin_amdgcn_readfirstlane(h0.w);
        // ---- 1. hidden units of degree t in layer order (typically one unit per layer), for this wave's samples
        auto unit = [&](vec4i e0, vec4i e1) {
            const float bias = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(e0.x));
            const int j = __builtin_amdgcn_readfirstlane(e0.y), kp = __builtin_amdgcn_readfirstlane(e0.z);
            const int src_v = __builtin_amdgcn_readfirstlane(e0.w), dst_v = __builtin_amdgcn_readfirstlane(e1.x);
            const bool add_stream = __builtin_amdgcn_readfirstlane(e1.y) != 0, set_stream = __builtin_amdgcn_readfirstlane(e1.z) != 0;
            const float* src = src_v < 0 ? xs + s * px * 4 : vecs + src_v * vec_floats + s * ph * 4;
            const int at = state_index(j, s, ph);
            const float carried = add_stream ? stream[at] : 0.0f;   // (in flight beside the dot product's reads)
            float acc[1];
            if (kp == 256) dot_rows_64<1, 4, LPS>(acc, rows, kp, src, q, 1);
            else dot_rows<1, 4, LPS>(acc, rows, kp, src, kp >> 2, q, 1);
            rows += kp;
            float v = acc[0] + bias;
            if (add_stream) v = carried + v;                    // residual connection (made.py:128)
            if (q == 0) {
                if (set_stream) stream[at] = v;
                if (dst_v >= 0) vecs[dst_v * vec_floats + at] = v < 0.0f ? 0.0f : v;   // ReLU'd for the next Linear (NaN stays)
            }
            // a unit just written is an input of the next Linear (same wave: LDS is in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
#pragma unroll
        for (int u = 0; u < kMadeUnitsAhead; ++u)
            if (u < units) unit(ue[u][0], ue[u][1]);
        for (int u = kMadeUnitsAhead; u < units; ++u) unit(bv[kMadeHeader / 4 + 2 * u], bv[kMadeHeader / 4 + 2 * u + 1]);
        NFA_K12_STAMP()
        if (t == a.T) break;
        // ---- 2. feature t's P output rows on the hidden vector as it stands: all sums in every lane of the sample
        const float* fin = vecs + a.final_src * vec_floats + s * ph * 4;
        float p[P];
#pragma unroll
        for (int g = 0; g < (P + RB - 1) / RB; ++g) {
            float acc[RB];
            const int left = P - g * RB;
            if (a.Hp == 256) dot_rows_64<RB, NFA_K12_ROWS_AHEAD, LPS>(acc, wf + g * RB * a.Hp, a.Hp, fin, q, left < RB ? left : RB);
            else dot_rows<RB, 4, LPS>(acc, wf + g * RB * a.Hp, a.Hp, fin, a.Hp >> 2, q, left < RB ? left : RB);
#pragma unroll
            for (int i = 0; i < RB; ++i)
                if (g * RB + i < P) p[g * RB + i] = acc[i] + fbias[g * RB + i];
        }
        NFA_K12_STAMP()
        // ---- 3. invert feature t (rational_quadratic.py:66-181 through the same evaluation as K5); every
        //      lane of a sample does it, one records the result
        float y, l;
        my_status |= rqs_eval<KT, true, true, true>(z_t, p, a.sp, y, l);
        lad_acc += l;
        // (kept in LDS only: a global store per step would sit in front of the next step's vmcnt(0))
        if (q == 0) xs[state_index(t, s, px)] = y;
        NFA_K12_STAMP()
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the features found -> the first T columns of the samples' rows; the final hidden vector of every sample:
    // input of the output layer for features >= T
    if (live) {
        for (int k = q; k < a.T; k += LPS) a.x[row * a.D + k] = xs[state_index(k, s, px)];
        if (q == 0) a.lad[row] = lad_acc;
        const float* fin = vecs + a.final_src * vec_floats;
        for (int k = q; k < a.H; k += LPS) a.hidden[row * a.H + k] = fin[state_index(k, s, ph)];
    } else {
        my_status = 0;
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_made_rqs_inverse_f32(const float* inputs, const float* step_blocks, const int32_t* block_starts,
                                        const int32_t* layout, int32_t layout_len, float* outputs,
                                        float* logabsdet, float* hidden_out, int32_t* status, int64_t batch,
                                        int32_t features, int32_t hidden_features, int32_t sequential_steps,
                                        const nfa_rqs_spec* spec, void* stream) {
    if (batch < 0 || features < 1 || hidden_features < 1 || sequential_steps < 0 || sequential_steps > features ||
        !layout || layout_len < 8)
        return NFA_ERR_INVALID_ARGUMENT;
    MadeInvArgs a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (a.sp.beta != 1.0f || !a.sp.linear || (a.sp.K != 8 && a.sp.K != 10)) return NFA_ERR_UNSUPPORTED;
    // layout: [num_linears, residual, final_src, stream_vec, num_vectors, Hp, Xp, max_block,
    //          then per Linear: padded_columns, src, dst, add_stream, set_stream]
    const int n = layout[0];
    if (n < 1 || n > kMadeMaxLinears || layout_len != 8 + 5 * n) return NFA_ERR_INVALID_ARGUMENT;
    a.num_linears = n;
    a.residual = layout[1];
    a.final_src = layout[2];
    a.stream_vec = layout[3];
    a.num_vectors = layout[4];
    a.Hp = layout[5];
    a.Xp = layout[6];
    a.max_block = layout[7];
    for (int l = 0; l < n; ++l) {
        const int32_t* e = layout + 8 + 5 * l;
        a.kp[l] = e[0];
        a.src[l] = e[1];
        a.dst[l] = e[2];
        a.add_stream[l] = e[3];
        a.set_stream[l] = e[4];
        if ((a.kp[l] & 15) != 0 || a.src[l] >= a.num_vectors || a.dst[l] >= a.num_vectors) return NFA_ERR_INVALID_ARGUMENT;
    }
    if ((a.Hp & 15) != 0 || (a.Xp & 15) != 0 || a.Hp < hidden_features || a.final_src < 0 || a.num_vectors < 1 ||
        a.final_src >= a.num_vectors || a.max_block < kMadeGrain || (a.max_block % kMadeGrain) != 0 ||
        (a.residual && (a.stream_vec < 0 || a.stream_vec >= a.num_vectors)))
        return NFA_ERR_INVALID_ARGUMENT;
    const size_t px = (((size_t)a.Xp >> 2) + 15) & ~(size_t)15, ph = (((size_t)a.Hp >> 2) + 15) & ~(size_t)15;   // chunks per sample
    const size_t lds = ((px + (size_t)a.num_vectors * ph) * 4 * kMadeSamples + 2 * (size_t)a.max_block) * sizeof(float);
    if (lds + 4096 > 160 * 1024) return NFA_ERR_UNSUPPORTED;   // (+ the 2 KB of logits and alignment)
    if (batch == 0) return NFA_OK;
    if (!inputs || !step_blocks || !block_starts || !outputs || !logabsdet || !hidden_out) return NFA_ERR_INVALID_ARGUMENT;
    a.z = inputs;
    a.x = outputs;
    a.lad = logabsdet;
    a.hidden = hidden_out;
    a.blocks = step_blocks;
    a.block_at = block_starts;
    a.status = status;
    a.batch = batch;
    a.D = features;
    a.H = hidden_features;
    a.T = sequential_steps;
    a.trace = g_k7_trace;
    // 32 lanes per sample (eight waves per workgroup: two per SIMD) was built in round 4 on the theory that the step
    // is a chain of LDS round trips which a second wave on the SIMD would hide; it is slower (every lane of a sample
    // repeats the spline inversion, the DPP sum grows a ds_swizzle stage): NFA_K12_LPS=32 selects it, 16 is the default
    const int64_t blocks = (batch + kMadeSamples - 1) / kMadeSamples;
    const char* lps_env = getenv("NFA_K12_LPS");
    const int lps = lps_env && atoi(lps_env) == 32 ? 32 : 16;   // (measured: 32 is SLOWER, 2.15 vs 1.97 ms at configs[4] -- profiles/r4/k12_lanes_per_sample.txt)
    void (*kern)(const MadeInvArgs) = a.sp.K == 8 ? (lps == 32 ? made_rqs_inverse_kernel<8, 32> : made_rqs_inverse_kernel<8, 16>)
                                                  : (lps == 32 ? made_rqs_inverse_kernel<10, 32> : made_rqs_inverse_kernel<10, 16>);
    if (lds > 64 * 1024) {
        static unsigned long long raised[4] = {};   // device masks (raise_dynamic_lds)
        const int which = (a.sp.K == 8 ? 0 : 1) + (lps == 32 ? 2 : 0);
        {
            const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[which], 160 * 1024 - 4096);
            if (rc_lds != NFA_OK) return rc_lds;
        }
    }
    note_layer_kernel("made_rqs_inverse_kernel<K=%d, lanes_per_sample=%d>", a.sp.K, lps);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3((lps / 4) * kWave), lds, (hipStream_t)stream, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
