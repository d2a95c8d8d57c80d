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
// all steps.  Lane = (sample s, quarter q): the four lanes of a sample split every dot product by
// 16-byte chunks (chunk c belongs to quarter c % 4) and add their parts with two cross-lane exchanges.
// The 3K - 1 output rows of a feature are dealt to the four waves (row p to wave p % 4), the few unit
// rows of a step are wave 0's; three workgroup barriers per step hand the results on through LDS.
// Per-sample state lives in LDS as [vector][chunk][16 samples][4 floats]: the features found so far
// and one vector per hidden Linear (its ReLU'd output) plus, for residual nets, the raw residual
// stream; a lane's read of chunk c is one ds_read_b128, conflict-free across the wave.
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
constexpr int kMadeHeader = 16;      // ints in front of a step block: units per layer [12], tail offset, output-row offset
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

// one lane's part of `R` dot products of weight rows (`pitch` floats apart) with a state vector, both
// in LDS: chunks q, q + 4, ... of `chunks`; then the sum over the four quarters (all four lanes get it)
template <int R, int UNROLL>
__device__ __forceinline__ void dot_rows(float (&acc)[R], const float* rows, int pitch, const float* vec, int chunks,
                                         int q, int s, int nrows, int skip_k = -1) {
    // (no conditionals inside the loop: rows beyond `nrows` re-read row 0 and their sums are ignored --
    // a branch per row would keep the compiler from batching the LDS reads of several chunks)
    const float* rp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r] = 0.0f;
        rp[r] = rows + (r < nrows ? r : 0) * pitch;
    }
    // (deep unrolling: the LDS reads of several chunks are in flight before the first FMA needs one;
    // a wave runs alone on its SIMD here, nothing else hides the latency)
#pragma unroll UNROLL
    for (int c = q; c < chunks; c += 4) {
        vec4f v = *reinterpret_cast<const vec4f*>(vec + (c * kMadeSamples + s) * 4);
        if (R > 1) {   // element `skip_k` of the vector is being written by another wave: leave it out
            const bool hit = c == (skip_k >> 2);
            v.x = (hit && (skip_k & 3) == 0) ? 0.0f : v.x;
            v.y = (hit && (skip_k & 3) == 1) ? 0.0f : v.y;
            v.z = (hit && (skip_k & 3) == 2) ? 0.0f : v.z;
            v.w = (hit && (skip_k & 3) == 3) ? 0.0f : v.w;
        }
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
    for (int r = 0; r < R; ++r) {
        acc[r] += __shfl_xor(acc[r], 16, kWave);
        acc[r] += __shfl_xor(acc[r], 32, kWave);
    }
}

__device__ __forceinline__ int state_index(int k, int s) { return ((k >> 2) * kMadeSamples + s) * 4 + (k & 3); }

// block `t` -> LDS at `dst`: a wave requests one grain (64 lanes x 16 bytes) per instruction, grain g is
// wave g % 4's
__device__ __forceinline__ void request_block(const MadeInvArgs& a, int t, float* dst, int lane, int wave) {
    const int g0 = a.block_at[t], g1 = a.block_at[t + 1];
    const char* src = reinterpret_cast<const char*>(a.blocks) + (size_t)g0 * (kMadeGrain * 4) + lane * 16;
    char* d = reinterpret_cast<char*>(dst);
    for (int g = wave; g < g1 - g0; g += kMadeWaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)g * (kMadeGrain * 4)),
                                         (__attribute__((address_space(3))) void*)(d + g * (kMadeGrain * 4)), 16, 0, 0);
}

template <int KT>
__global__ void __launch_bounds__(kMadeWaves * kWave) made_rqs_inverse_kernel(const MadeInvArgs a) {
#pragma clang fp contract(off)
    constexpr int P = 3 * KT - 1;
    constexpr int kRowWaves = kMadeWaves - 1;                    // waves 1..3 take the output rows while wave 0
    constexpr int RB = (P + kRowWaves - 1) / kRowWaves;          // walks the step's unit chain
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float s_params[32 * kMadeSamples];                   // a feature's logits of the 16 samples
    __shared__ int s_cfg[kMadeMaxLinears][5];                       // per Linear: columns, src, dst, add, set (a
                                                                    // dynamically indexed kernel argument is a
                                                                    // scalar memory load every time it is read)
    if (threadIdx.x < kMadeMaxLinears) {
        const int l = threadIdx.x;
        s_cfg[l][0] = a.kp[l];
        s_cfg[l][1] = a.src[l];
        s_cfg[l][2] = a.dst[l];
        s_cfg[l][3] = a.add_stream[l];
        s_cfg[l][4] = a.set_stream[l];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 15, q = lane >> 4;
    const int64_t row = (int64_t)blockIdx.x * kMadeSamples + s;
    const bool live = row < a.batch;
    const int64_t rrow = live ? row : a.batch - 1;
    const int vec_floats = (a.Hp >> 2) * kMadeSamples * 4;          // one hidden vector of all 16 samples
    const int state_floats = (a.Xp >> 2) * kMadeSamples * 4 + a.num_vectors * vec_floats;
    float* xs = lds;                                                // [Xp / 4][16][4]
    float* vecs = lds + (a.Xp >> 2) * kMadeSamples * 4;             // [num_vectors][Hp / 4][16][4]
    float* buf0 = lds + state_floats;                               // two step blocks
    float* buf1 = buf0 + a.max_block;
    request_block(a, 0, buf0, lane, wave);
    for (int i = threadIdx.x; i < state_floats; i += kMadeWaves * kWave) lds[i] = 0.0f;

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
        // block t has landed (every wave's share, requested a whole step ago); everything the previous
        // step wrote to LDS is visible, every read of the other buffer half is done
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        NFA_K12_STAMP()
        const float z_t = z_next;
        if (t < a.T) {
            request_block(a, t + 1, (t & 1) ? buf0 : buf1, lane, wave);
            if (t + 1 < a.T) z_next = zrow[t + 1];
        }
        const int* hdr = reinterpret_cast<const int*>(blk);
        const float* tail = blk + hdr[12];          // per unit (bias, index), then the feature's P biases
        // ---- 1. hidden units of degree t, layer by layer (wave 0; typically one unit per layer) and, beside
        //      it, 2. feature t's P output rows on the hidden vector (waves 1..3; row p is wave 1 + p % 3's).
        //      The only element of that vector the units of this step change is the new unit of the layer in
        //      front of the output layer: the row sums leave it out and receive it as a rank-1 term afterwards
        //      (with more than one such unit in a step the rows simply wait for the units).
        int units = 0;
        for (int l = 0; l < a.num_linears; ++l) units += hdr[l];
        const int nlast = hdr[a.num_linears - 1];
        const bool beside = nlast <= 1 && t < a.T;
        const int jnew = nlast == 1 ? __builtin_bit_cast(int, tail[2 * (units - 1) + 1]) : -1;
        const float* fin = vecs + a.final_src * vec_floats;
        const float* wf = blk + hdr[13];
        const float* fbias = tail + 2 * units;
        auto output_rows = [&](int skip) {
            float acc[RB];
            const int mine = (P - (wave - 1) + kRowWaves - 1) / kRowWaves;
            dot_rows<RB, 4>(acc, wf + (wave - 1) * a.Hp, kRowWaves * a.Hp, fin, a.Hp >> 2, q, s, mine, skip);
            if (q == 0) {
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int p_ = (wave - 1) + kRowWaves * i;
                    if (p_ < P) s_params[p_ * kMadeSamples + s] = acc[i] + fbias[p_];
                }
            }
        };
        if (wave == 0) {
            const float* rows = blk + kMadeHeader;
            const float* ut = tail;
            for (int l = 0; l < a.num_linears; ++l) {
                const int n = hdr[l];
                if (n == 0) continue;
                const int kp = s_cfg[l][0], chunks = kp >> 2, src_v = s_cfg[l][1], dst_v = s_cfg[l][2];
                const bool add_stream = s_cfg[l][3] != 0, set_stream = s_cfg[l][4] != 0;
                const float* src = src_v < 0 ? xs : vecs + src_v * vec_floats;
                float* dst = dst_v < 0 ? nullptr : vecs + dst_v * vec_floats;
                for (int u = 0; u < n; ++u) {
                    float acc[1];
                    dot_rows<1, 8>(acc, rows, kp, src, chunks, q, s, 1);
                    rows += kp;
                    const int j = __builtin_bit_cast(int, ut[1]);
                    float v = acc[0] + ut[0];
                    ut += 2;
                    const int at = state_index(j, s);
                    if (add_stream) v = stream[at] + v;               // residual connection (made.py:128)
                    if (q == 0) {
                        if (set_stream) stream[at] = v;
                        if (dst) dst[at] = v < 0.0f ? 0.0f : v;       // ReLU'd for the next Linear (NaN stays)
                    }
                }
                // the units just written are inputs of the next Linear
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else if (beside) {
            output_rows(jnew);
        }
        NFA_K12_STAMP()
        __syncthreads();
        if (t == a.T) break;
        if (!beside) {
            if (wave != 0) output_rows(-1);
            __syncthreads();
        }
        NFA_K12_STAMP()
        NFA_K12_STAMP()
        // ---- 3. invert feature t (rational_quadratic.py:66-181 through the same evaluation as K5); every
        //      thread of a sample does it, one records the result
        float p[P];
#pragma unroll
        for (int j = 0; j < P; ++j) p[j] = s_params[j * kMadeSamples + s];
        if (beside && jnew >= 0) {   // the rank-1 term of the unit the row sums left out
            const float vj = fin[state_index(jnew, s)];
#pragma unroll
            for (int j = 0; j < P; ++j) p[j] = __builtin_fmaf(wf[j * a.Hp + jnew], vj, p[j]);
        }
        float y, l;
        my_status |= rqs_eval<KT, true, true, true>(z_t, p, a.sp, y, l);
        lad_acc += l;
        // (kept in LDS only: a global store per step would sit in front of the next step's vmcnt(0))
        if (wave == 0 && q == 0) xs[state_index(t, s)] = y;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the features found, [16 samples][T] -> the first T columns of the samples' rows
    if (live)
        for (int k = q + 4 * wave; k < a.T; k += 4 * kMadeWaves) a.x[row * a.D + k] = xs[state_index(k, s)];
    if (wave == 0) {
        if (live && q == 0) a.lad[row] = lad_acc;
        // the final hidden vector of every sample: input of the output layer for features >= T
        const float* fin = vecs + a.final_src * vec_floats;
        if (live)
            for (int k = q; k < a.H; k += 4) a.hidden[row * a.H + k] = fin[state_index(k, s)];
        if (!live) my_status = 0;
        if (my_status && a.status) atomicOr(a.status, my_status);
    }
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
    const size_t lds = (((size_t)a.Xp + (size_t)a.num_vectors * a.Hp) * kMadeSamples + 2 * (size_t)a.max_block) * sizeof(float);
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
    void (*kern)(const MadeInvArgs) = a.sp.K == 8 ? made_rqs_inverse_kernel<8> : made_rqs_inverse_kernel<10>;
    if (lds > 64 * 1024) {
        static unsigned long long raised[2] = {};   // device masks (raise_dynamic_lds)
        const int which = a.sp.K == 8 ? 0 : 1;
        {
            const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[which], 160 * 1024 - 4096);
            if (rc_lds != NFA_OK) return rc_lds;
        }
    }
    const int64_t blocks = (batch + kMadeSamples - 1) / kMadeSamples;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kMadeWaves * kWave), lds, (hipStream_t)stream, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
