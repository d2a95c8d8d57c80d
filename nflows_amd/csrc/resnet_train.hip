// K14: the hidden part of the ResidualNet conditioner (nn/nets/resnet.py:92-100 without the final layer:
// initial Linear, then blocks h += W_1 relu(W_0 relu(h) + b_0) + b_1) for TRAINING -- one kernel for its forward
// pass that also leaves behind what the backward pass needs, one kernel for the chain of input gradients.
// The reference trains through autograd over eager ops (examples/moons.ipynb cell 3: `loss.backward()`); per layer
// that is five library GEMMs + six elementwise kernels forward and as many backward, each a round trip of a
// [B, 128] activation through HBM.  Here:
//
//   forward  (nfa_resnet_hidden_forward_f32):   x_id [B, d_i] -> h [B, 128], saved[2 k] = relu(h_k) (input of block
//            k's first Linear), saved[2 k + 1] = relu(a_k) (input of its second Linear): exactly the `inputs` the
//            weight-gradient kernel K10 wants, and the ReLU masks of the backward pass.  (With no block: nothing.)
//   backward (nfa_resnet_hidden_backward_f32):  g_h [B, 128] (+ saved) -> per block g_a_k = (g_c W_1) . [a_k > 0]
//            and g_h_k = g_h_{k+1} + (g_a_k W_0) . [h_k > 0], finally g_x = g_h_0 W_in.  Every g_* that is a
//            `grad_outputs` of a Linear is written once ([B, 128]) for K10; the weight gradients stay K10's.
//
// Same skeleton and GEMM machinery as K8 / K11 (bf16x3_gemm.hpp): a wave owns 32 rows, activations (and
// gradients) stay in registers as three bf16 pieces -- fp32-accurate products on the bf16 matrix pipe, full fp32
// range, so gradients need no scaling --, weights (W for the forward stream, W^T for the backward stream, packed
// by the host) arrive through the LDS-DMA ring.  Activations enter and leave in the MFMA accumulator layout: lane
// (half, r) holds, of row r and 32-feature tile t, features 8 q4 + 4 half + (0..3) for q4 = 0..3 -- four 16-byte
// accesses per tile which together cover whole 128-byte lines.
//
// Ordinary global LOADS are only issued with the weight ring drained (a `s_waitcnt vmcnt(n)` of the compiler counts
// on in-order return, which LDS-DMA requests sharing the counter do not give it): both kernels load at the top of
// a row block and wait for everything.  Stores are issued between the GEMMs: they only make the ring's counted
// wait more conservative (requests complete in order among themselves; `vmcnt(3)` with stores outstanding still
// implies that at most the three youngest requests are pending).
//
// Restrictions (the host keeps the eager path otherwise): hidden width <= 128 and a multiple of 4 (narrower nets are
// zero-padded into the 128-wide streams by the packer), ReLU, no context / batch norm / active dropout, at most three
// blocks, d_i <= 64 and d_i % 4 == 0, batch % 128 == 0.

#include "bf16x3_gemm.hpp"

#include <hip/hip_ext.h>

namespace nfa {

struct TrainArgs {
    const float* x;      // forward: [B, d_i];  backward: g_h [B, 128]
    const vec4f* w;      // the stream of 12 KB stages
    const float* bias;   // forward: accumulator-order biases (128 + 256 per block)
    float* saved;        // [2 nb][B][128] (forward: written, backward: read)
    float* out;          // forward: h [B, 128];  backward: g_x [B, d_i]
    float* grads;        // backward: [2 nb][B][128]: grads[2 k] = g_h_k (also the gradient w.r.t. block k's input
                         // = `grad_outputs` of the Linear in front of it), grads[2 k + 1] = g_a_k
    int64_t batch;       // multiple of 128
    int di, num_blocks, num_stages;
    // forward, optional: the net's final Linear (128 -> out_features) behind the blocks, its stages appended to the stream
    const float* fbias;  // accumulator-order bias, 32 per tile (zeros past out_features)
    float* params;       // [B, out_features]
    int out_features, final_tiles;
    // backward with the final Linear's input gradient (round 4): g_params [B, out_features] instead of g_h -- the kernel
    // starts with g_h = g_params W_f (k-major over out_features, W_f^T's stages at `final_first` of the backward
    // stream) and writes g_h to `grad_hidden` for K10 (grad_outputs of the last block's second Linear)
    const float* gparams;
    float* grad_hidden;
    int final_ksteps, final_first;
};

// accumulator tile <-> rows of a [B, 128] array: element (row, 32 t + 8 q4 + 4 half + i) = acc[4 q4 + i]
//
// Stores (round 4).  In the accumulator layout a lane holds 16-byte pieces of ONE row: a `global_store_dwordx4` of the
// wave then touches 64 separate 16-byte segments in 32 different 128-byte lines, and the address coalescer handles them
// one by one -- removing the stores of the saved activations took 26 of the forward kernel's 78 us although they are
// only 2.2 TB/s of traffic, and spreading them over the GEMMs changed nothing (profiles/r4/k14_ablations.txt).  With
// NFA_K14_LDS_STORES (default) a tile goes through a wave-private 32 x 36-float LDS image and leaves as four stores
// of EIGHT FULL 128-byte lines each: lane l writes bytes 16 (l % 8) .. of the tile's 128 bytes of row 8 j + l / 8.
#ifndef NFA_K14_LDS_STORES
#define NFA_K14_LDS_STORES 1
#endif
constexpr int kTrainTilePitch = 36;                          // floats per row of the staging image (32 + 4: b128 accesses)
constexpr int kTrainTileFloats = 32 * kTrainTilePitch;       // per wave

// the direct form (round 3): four 16-byte stores per lane, lane (half, r) -> row r
template <bool RELU>
__device__ __forceinline__ void store_tile_direct(float* base, int64_t row, int t, int half, const f32x16& a) {
    vec4f* p = reinterpret_cast<vec4f*>(base + row * 128 + 32 * t + 4 * half);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        vec4f v = {a[4 * q4], a[4 * q4 + 1], a[4 * q4 + 2], a[4 * q4 + 3]};
        if (RELU) {   // NaN stays NaN (torch.relu)
            v.x = v.x < 0.0f ? 0.0f : v.x;
            v.y = v.y < 0.0f ? 0.0f : v.y;
            v.z = v.z < 0.0f ? 0.0f : v.z;
            v.w = v.w < 0.0f ? 0.0f : v.w;
        }
        p[2 * q4] = v;   // 8 floats = two vec4 apart
    }
}

// the staged form: through the wave's LDS image, eight full 128-byte lines per store instruction; `ld` floats between
// rows, `cols` valid columns of this tile (a multiple of 4)
template <bool RELU>
__device__ __forceinline__ void store_tile_staged(float* base, int64_t row, int t, int half, const f32x16& a, float* stage,
                                                  int ld, int cols) {
    // `row` = the lane's own row (row0 + r); the wave's first row is row - r
    const int lane = __lane_id();
    const int r = lane & 31;
    float* mine = stage + r * kTrainTilePitch + 4 * half;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        vec4f v = {a[4 * q4], a[4 * q4 + 1], a[4 * q4 + 2], a[4 * q4 + 3]};
        if (RELU) {
            v.x = v.x < 0.0f ? 0.0f : v.x;
            v.y = v.y < 0.0f ? 0.0f : v.y;
            v.z = v.z < 0.0f ? 0.0f : v.z;
            v.w = v.w < 0.0f ? 0.0f : v.w;
        }
        *reinterpret_cast<vec4f*>(mine + 8 * q4) = v;
    }
    const int c = lane & 7, rr = lane >> 3;
    float* dst = base + (row - r) * ld + 32 * t + 4 * c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const vec4f w = *reinterpret_cast<const vec4f*>(stage + (8 * j + rr) * kTrainTilePitch + 4 * c);
        if (4 * c < cols) *reinterpret_cast<vec4f*>(dst + (int64_t)(8 * j + rr) * ld) = w;
    }
}

template <bool RELU>
__device__ __forceinline__ void store_tile(float* base, int64_t row, int t, int half, const f32x16& a, float* stage = nullptr) {
#ifdef NFA_K14_ABL_NO_SAVE   // (measurement: what the activation stores cost; results are garbage downstream)
    if (a[0] != 123.456f) return;
#endif
    if (NFA_K14_LDS_STORES && stage) store_tile_staged<RELU>(base, row, t, half, a, stage, 128, 32);
    else store_tile_direct<RELU>(base, row, t, half, a);
}

// The weight stream of these kernels: bf16x3_gemm.hpp's ring (12 KB stages, three LDS-DMA requests per stage and
// wave-quarter, counted waits) with the depth as a build parameter -- kTrainRing slots, kTrainAhead = kTrainRing - 1
// stages requested ahead of the one being consumed.  Stage c lives in slot c % kTrainRing; at k-step c stage
// c + kTrainAhead is requested into the slot stage c - 1 left (every wave passed the previous advance's barrier), and
// the advance waits until only the requests of stages c + 2 .. c + kTrainAhead may be pending:
// vmcnt(3 (kTrainAhead - 1)).  Measured: six slots (five stages in flight, 72 KB of LDS) against K8 / K11's three
// change nothing (forward 78.9 vs 77.4 us, with the final Linear 153.0 vs 147.9, backward 90.6 vs 90.2 at 65 536 rows;
// same results): like K8s' small batches the stage time (1.85 us for 0.32 us of MFMA per wave) is the SIMDs' own --
// fragment reads, conversions and the per-stage barrier between two co-resident waves --, not the fill rate.
#ifndef NFA_K14_RING
#define NFA_K14_RING 3
#endif
#ifndef NFA_K14_KSTEP
#define NFA_K14_KSTEP 0   // 1: the pairwise k-step (round-4 experiment)
#endif
constexpr int kTrainRing = NFA_K14_RING, kTrainAhead = kTrainRing - 1;
static_assert(kTrainRing >= 3 && 5 * (kTrainAhead - 1) <= 63, "vmcnt is a 6-bit count");

struct TrainStream {
    const vec4f* w;
    vec4f* ring;
    int slot;        // ring slot of the stage being consumed
    int fetch;       // stage index (in the stream) to request next
    int num_stages;
    int tid;
};

__device__ __forceinline__ void tstream_request_into(TrainStream& sm, int dst_slot) {
    const char* stage = reinterpret_cast<const char*>(sm.w) + (size_t)sm.fetch * (kStageVec4 * 16);
    const int wave = __builtin_amdgcn_readfirstlane(sm.tid >> 6);
    char* slot = reinterpret_cast<char*>(sm.ring) + dst_slot * (kStageVec4 * 16) + wave * (kWave * 16);
    const unsigned lane_off = (unsigned)sm.tid * 16u;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)((stage + i * kBlock * 16) + lane_off),
            (__attribute__((address_space(3))) void*)(slot + i * kBlock * 16), 16, 0, 0);
    sm.fetch = (sm.fetch + 1 == sm.num_stages) ? 0 : sm.fetch + 1;
}

__device__ __forceinline__ void tstream_request(TrainStream& sm) {
    const int dst = sm.slot + kTrainAhead;
    tstream_request_into(sm, dst >= kTrainRing ? dst - kTrainRing : dst);
}

__device__ __forceinline__ void tstream_advance(TrainStream& sm) {
#ifdef NFA_K14_ABL_NO_BARRIER
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(3 * (kTrainAhead - 1)) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(3 * (kTrainAhead - 1)) : "memory");
#endif
    sm.slot = (sm.slot + 1 == kTrainRing) ? 0 : sm.slot + 1;
}

__device__ __forceinline__ void start_stream(TrainStream& sm, const vec4f* w, float* lds, int num_stages, int tid,
                                             int first_stage = 0) {
    sm.w = w;
    sm.ring = reinterpret_cast<vec4f*>(lds);
    sm.fetch = first_stage;
    sm.num_stages = num_stages;
    sm.tid = tid;
#pragma unroll
    for (int s = 0; s < kTrainAhead; ++s) tstream_request_into(sm, s);   // stages 0 .. kTrainAhead - 1
    sm.slot = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// one k-step of a k-major GEMM (bf16x3_gemm.hpp: gemm_kmajor): out^T[128 x 32 samples] += W[128 x 16] x act^T,
// stage = [4 tiles][3 pieces][64 lanes] x 16 bytes
__device__ __forceinline__ void kstep(f32x16 (&acc)[4], const bf16x8& bh, const bf16x8& bm, const bf16x8& bl,
                                      TrainStream& sm, int lane) {
    tstream_request(sm);
    const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
#if NFA_K14_KSTEP == 0   // (round 3: tile by tile, six dependent products each, every read awaited where it is used)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(t * 3 + 0) * 64]);
        const bf16x8 am = __builtin_bit_cast(bf16x8, cur[(t * 3 + 1) * 64]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, cur[(t * 3 + 2) * 64]);
        NFA_MFMA6(acc[t], ah, am, al, bh, bm, bl);
    }
#else
    // (round 4) two tiles at a time: their six fragments are requested together, the next pair's while this pair's
    // twelve MFMAs run; consecutive MFMAs alternate between the two accumulators and are grouped by their srcB piece
    // (bl, bl, bm, bm, bm, bm, bh x 6: three changes of srcB per twelve instead of twelve)
    bf16x8 f[2][6];
#pragma unroll
    for (int q = 0; q < 6; ++q) f[0][q] = __builtin_bit_cast(bf16x8, cur[q * 64]);
#pragma unroll
    for (int pair = 0; pair < 2; ++pair) {
        if (pair == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) f[1][q] = __builtin_bit_cast(bf16x8, cur[(6 + q) * 64]);
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 (&g)[6] = f[pair];   // [tile][h, m, l]
        f32x16& a0 = acc[2 * pair];
        f32x16& a1 = acc[2 * pair + 1];
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[0], bl, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[3], bl, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[1], bm, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[4], bm, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[0], bm, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[3], bm, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[2], bh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[5], bh, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[1], bh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[4], bh, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[0], bh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[3], bh, a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    tstream_advance(sm);
}

// k-major GEMM over 128 inputs given as four fp32 accumulator tiles (ReLU'd first when RELU): tile t becomes the
// pieces of k-steps 2 t and 2 t + 1 right before they are consumed, so that no 96-register piece array is ever
// live next to the accumulators (K8 keeps the residual stream as pieces; here it stays in fp32 tiles)
// `save` (round 4): the source tiles are ALSO an array the other pass / K10 needs in HBM (forward: relu(h_k), relu(a_k);
// backward: g_a_k, g_h_k).  Tile t is stored right before the two k-steps that consume it -- four stores per lane in
// front of 48 MFMAs -- instead of all sixteen at once in front of the GEMM: every workgroup of the launch reaches that
// point at about the same time, the burst of 33 MB per array waited for itself in the next counted vmcnt (stores share
// the counter with the ring's requests) and nothing overlapped it: 26 of the forward kernel's 78 us at 65 536 rows
// (profiles/r4/k14_ablations.txt).
// 64 ReLU masks of a lane (4 tiles x 16 accumulator registers) as two words: bit 16 t + q of the pair
struct Mask64 {
    unsigned lo, hi;   // tiles 0-1, tiles 2-3
};

// Packed masks (round 4, last change): the forward kernel leaves, behind the 2 nb saved planes, one Mask64 per lane,
// wave tile and plane -- [2 nb][batch / 32][64 lanes] x 8 bytes, 4 MB at 65 536 rows and two blocks -- and the
// backward kernel reads those instead of rebuilding the bits from the 134 MB of saved activations (both kernels give
// lane (half, r) the same elements of a row: the accumulator layout).  mask bit = value > 0 (NaN: 0, as
// threshold_backward's `output > 0`).
__device__ __forceinline__ unsigned long long* packed_masks(float* saved, int64_t batch, int num_blocks) {
    return reinterpret_cast<unsigned long long*>(saved + (int64_t)2 * num_blocks * batch * 128);
}

// bits of tile t (value > 0) into the lane's mask
__device__ __forceinline__ void mask_tile(Mask64& m, const f32x16& v, int t) {
    unsigned bits = 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) bits |= (v[q] > 0.0f ? 1u : 0u) << q;
    if (t < 2) m.lo |= bits << (16 * (t & 1));
    else m.hi |= bits << (16 * (t & 1));
}

template <bool RELU, bool SAVE = false>
__device__ __forceinline__ void gemm_from_tiles(f32x16 (&acc)[4], const f32x16 (&src)[4], TrainStream& sm, int lane,
                                                float* save = nullptr, int64_t row = 0, int half = 0, float* stage = nullptr,
                                                Mask64* mask = nullptr) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (mask) mask_tile(*mask, src[t], t);   // (forward: the packed ReLU masks, tile by tile while the tile is at hand)
#ifndef NFA_K14_BURST_STORES
        // (SAVE as a template flag or `save` as a run-time pointer: the same code, another register allocation -- hipcc
        //  spills 12-20 bytes in the forward kernel with the flag and 780 in the backward kernel with the pointer)
        if constexpr (SAVE) store_tile<RELU>(save, row, t, half, src[t], stage);
        else if (save) store_tile<RELU>(save, row, t, half, src[t], stage);
#endif
        bf16x8 h0, m0, l0, h1, m1, l1;
        tile_to_pieces<RELU>(src[t], h0, m0, l0, h1, m1, l1);
        kstep(acc, h0, m0, l0, sm, lane);
        kstep(acc, h1, m1, l1, sm, lane);
    }
}

// one 32-row output tile of a 128-wide layer from fp32 input tiles (bf16x3_gemm.hpp: gemm_tile): two stages of
// [3 pieces][4 k-steps][64 lanes] x 16 bytes
__device__ __forceinline__ void gemm_tile_from_tiles(f32x16& acc, const f32x16 (&src)[4], TrainStream& sm, int lane) {
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        tstream_request(sm);
        const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            bf16x8 bh[2], bm[2], bl[2];
            tile_to_pieces<false>(src[2 * hs + tt], bh[0], bm[0], bl[0], bh[1], bm[1], bl[1]);
#pragma unroll
            for (int hk = 0; hk < 2; ++hk) {
                const int k4 = 2 * tt + hk;
                const bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(0 * 4 + k4) * 64]);
                const bf16x8 am = __builtin_bit_cast(bf16x8, cur[(1 * 4 + k4) * 64]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, cur[(2 * 4 + k4) * 64]);
                NFA_MFMA6(acc, ah, am, al, bh[hk], bm[hk], bl[hk]);
            }
        }
        tstream_advance(sm);
    }
}

// bf16x3_gemm.hpp's gemm_tile on this file's stream: one 32-row output tile from the 128 inputs given as pieces
__device__ __forceinline__ void gemm_tile_pieces(f32x16& acc, const bf16x8 (&ph)[8], const bf16x8 (&pm)[8],
                                                 const bf16x8 (&pl)[8], TrainStream& sm, int lane) {
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        tstream_request(sm);
        const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int ks = hs * 4 + k4;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(0 * 4 + k4) * 64]);
            const bf16x8 am = __builtin_bit_cast(bf16x8, cur[(1 * 4 + k4) * 64]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, cur[(2 * 4 + k4) * 64]);
            NFA_MFMA6(acc, ah, am, al, ph[ks], pm[ks], pl[ks]);
        }
        tstream_advance(sm);
    }
}

template <int INIT_KS>
__global__ void __launch_bounds__(kBlock, 2) resnet_hidden_forward_kernel(const TrainArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TrainStream sm;
    start_stream(sm, a.w, lds_dyn, a.num_stages, tid);
    float* stage = lds_dyn + kTrainRing * kStageVec4 * 4 + wave * kTrainTileFloats;   // this wave's store image
    const int64_t num_quads = a.batch >> 7;
    const int64_t plane = a.batch * 128;   // one saved activation
    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = (quad << 7) + (wave << 5);
        int lane_here = lane, di = a.di;
        asm volatile("" : "+v"(lane_here), "+s"(di));
        const int half = lane_here >> 5, r = lane_here & 31;
        const int64_t row = row0 + r;
        const float* bias = a.bias + half * 16;  // + 32 per tile
        f32x16 hs[4];   // the residual stream h of the wave's 32 rows, fp32, accumulator layout
        // ---- initial layer: h_0 = W_in x + b_in; the row's identity features are k = ks*16 + half*8 + j (zeros past
        //      d_i), loaded while nothing else is in flight
        {
            vec4f xv[INIT_KS][2];
            const float* xr = a.x + row * di;
#pragma unroll
            for (int ks = 0; ks < INIT_KS; ++ks)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int k0 = ks * 16 + half * 8 + g * 4;
                    xv[ks][g] = k0 < di ? *reinterpret_cast<const vec4f*>(xr + k0) : vec4f{0.0f, 0.0f, 0.0f, 0.0f};
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t) load_bias_tile(hs[t], bias + t * 32);
#pragma unroll
            for (int ks = 0; ks < INIT_KS; ++ks) {
                bf16x2 hh[4], mm[4], ll[4];
                split3(vec2f{xv[ks][0].x, xv[ks][0].y}, hh[0], mm[0], ll[0]);
                split3(vec2f{xv[ks][0].z, xv[ks][0].w}, hh[1], mm[1], ll[1]);
                split3(vec2f{xv[ks][1].x, xv[ks][1].y}, hh[2], mm[2], ll[2]);
                split3(vec2f{xv[ks][1].z, xv[ks][1].w}, hh[3], mm[3], ll[3]);
                kstep(hs, join4(hh[0], hh[1], hh[2], hh[3]), join4(mm[0], mm[1], mm[2], mm[3]),
                      join4(ll[0], ll[1], ll[2], ll[3]), sm, lane);
            }
        }
        bias += 128;
        // ---- residual blocks: a = W_0 relu(h) + b_0;  h += W_1 relu(a) + b_1
        for (int blk = 0; blk < a.num_blocks; ++blk) {
            float* in0 = a.saved + (2 * blk) * plane;
            float* in1 = a.saved + (2 * blk + 1) * plane;
            f32x16 u[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#ifdef NFA_K14_BURST_STORES   // (round 3: all sixteen stores in front of the GEMM)
                store_tile<true>(in0, row, t, half, hs[t]);
#endif
                load_bias_tile(u[t], bias + t * 32);
            }
            unsigned long long* pm = packed_masks(a.saved, a.batch, a.num_blocks) + ((row0 >> 5) << 6) + lane_here;
            Mask64 mh = {0u, 0u}, ma = {0u, 0u};   // [h_k > 0], [a_k > 0]
            gemm_from_tiles<true>(u, hs, sm, lane, in0, row, half, stage, &mh);    // (stores relu(h), the first Linear's input, on the way)
            pm[(int64_t)(2 * blk) * (a.batch << 1)] = ((unsigned long long)mh.hi << 32) | mh.lo;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#ifdef NFA_K14_BURST_STORES
                store_tile<true>(in1, row, t, half, u[t]);
#endif
                f32x16 b1;
                load_bias_tile(b1, bias + 128 + t * 32);
#pragma unroll
                for (int q = 0; q < 16; ++q) hs[t][q] += b1[q];   // skip connection: accumulate onto h
            }
            {
                // (the second GEMM accumulates INTO hs while it reads u: relu(a), the second Linear's input, is stored
                //  from u on the way)
                gemm_from_tiles<true>(hs, u, sm, lane, in1, row, half, stage, &ma);
                pm[(int64_t)(2 * blk + 1) * (a.batch << 1)] = ((unsigned long long)ma.hi << 32) | ma.lo;
            }
            bias += 256;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) store_tile<false>(a.out, row, t, half, hs[t], stage);
        // ---- optionally the final Linear: params = W_f h + b_f, one 32-column tile of the [B, out_features] result at a
        //      time (h as pieces from here on: converted once, the fp32 tiles are dead)
        if (a.final_tiles > 0) {
            bf16x8 ph[8], pm[8], pl[8];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                tile_to_pieces<false>(hs[t], ph[2 * t], pm[2 * t], pl[2 * t], ph[2 * t + 1], pm[2 * t + 1], pl[2 * t + 1]);
            const float* fb = a.fbias + half * 16;
            const int out_features = a.out_features;
            for (int t = 0; t < a.final_tiles; ++t) {
                f32x16 acc;
                load_bias_tile(acc, fb + t * 32);
                gemm_tile_pieces(acc, ph, pm, pl, sm, lane);
                store_tile_staged<false>(a.params, row, t, half, acc, stage, out_features, out_features - 32 * t < 32 ? out_features - 32 * t : 32);
            }
        }
        // drain (stores, and the two stages requested past this row block) before the next block's loads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the final Linear's input gradient inside the backward kernel (round 4): g_h^T [128 x 32 samples] = W_f^T
//      [128 x out] g_params^T, k-major over the out_features columns of g_params, 16 per k-step.  The wave's share
//      of g_params -- 32 rows x 16 columns = 2 KB per k-step -- arrives by LDS-DMA like the weights (ordinary loads
//      may not share the counter with the ring's requests: see the file header) in a wave-private ring of
//      kTrainRing 2-KB images behind the weight ring: two requests per k-step (16 rows x 64 bytes each), issued with
//      the weight stage of the same k-step, kTrainAhead k-steps ahead.  Image layout: row r at 64 r bytes, the
//      row's four 16-byte chunks XOR-swizzled by (r >> 1) & 3 -- lane (half, r) reads chunks 2 half, 2 half + 1 of
//      its row (its eight k values, as in the forward kernel's initial layer), eight consecutive lanes hit eight
//      different bank quads.  Columns past out_features (the last k-step of a width that is not a multiple of 16)
//      are fetched from the row's last chunk instead: their weights in the stream are zero.
constexpr int kGpImageBytes = 32 * 64;

__device__ __forceinline__ void gp_request(const float* gparams, int out_features, int64_t row0, int ks, char* image, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * i + (lane >> 2);
        const int chunk = (lane & 3) ^ ((r >> 1) & 3);
        int col = ks * 16 + chunk * 4;
        col = col < out_features ? col : out_features - 4;
        const float* src = gparams + (row0 + r) * (int64_t)out_features + col;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(image + i * 1024), 16, 0, 0);
    }
}

// one k-step of the final Linear's input gradient: requests (weights and g_params of k-step ks + kTrainAhead), the
// lane's eight values from the image of k-step ks, 24 MFMAs, the counted wait for k-step ks + 1
__device__ __forceinline__ void kstep_final(f32x16 (&acc)[4], TrainStream& sm, const float* gparams, int out_features,
                                            int64_t row0, int ks, int final_ksteps, char* images, int lane) {
    tstream_request(sm);
    const bool more = ks + kTrainAhead < final_ksteps;   // (wave-uniform)
    if (more) gp_request(gparams, out_features, row0, ks + kTrainAhead, images + ((ks + kTrainAhead) % kTrainRing) * kGpImageBytes, lane);
    const int half = lane >> 5, r = lane & 31;
    const char* img = images + (ks % kTrainRing) * kGpImageBytes + r * 64;
    const int sw = (r >> 1) & 3;
    const vec4f v0 = *reinterpret_cast<const vec4f*>(img + (((2 * half) ^ sw) << 4));
    const vec4f v1 = *reinterpret_cast<const vec4f*>(img + (((2 * half + 1) ^ sw) << 4));
    bf16x2 hh[4], mm[4], ll[4];
    split3(vec2f{v0.x, v0.y}, hh[0], mm[0], ll[0]);
    split3(vec2f{v0.z, v0.w}, hh[1], mm[1], ll[1]);
    split3(vec2f{v1.x, v1.y}, hh[2], mm[2], ll[2]);
    split3(vec2f{v1.z, v1.w}, hh[3], mm[3], ll[3]);
    const bf16x8 bh = join4(hh[0], hh[1], hh[2], hh[3]), bm = join4(mm[0], mm[1], mm[2], mm[3]),
                 bl = join4(ll[0], ll[1], ll[2], ll[3]);
    const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(t * 3 + 0) * 64]);
        const bf16x8 am = __builtin_bit_cast(bf16x8, cur[(t * 3 + 1) * 64]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, cur[(t * 3 + 2) * 64]);
        NFA_MFMA6(acc[t], ah, am, al, bh, bm, bl);
    }
    // stage ks + 1 (weights and image) has landed when only this k-step's own requests may be pending
    if (more) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(5 * (kTrainAhead - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(3 * (kTrainAhead - 1)) : "memory");
    sm.slot = (sm.slot + 1 == kTrainRing) ? 0 : sm.slot + 1;
}

// loads the lane's 64 values of a [B, 128] array (accumulator layout); the caller waits
__device__ __forceinline__ void load_tiles_raw(vec4f (&v)[16], const float* base, int64_t row, int half) {
    const vec4f* p = reinterpret_cast<const vec4f*>(base + row * 128 + 4 * half);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) v[t * 4 + q4] = p[8 * t + 2 * q4];
}

__device__ __forceinline__ Mask64 load_mask(const float* base, int64_t row, int half) {
    vec4f v[16];
    load_tiles_raw(v, base, row, half);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Mask64 m = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned bits = (v[i].x > 0.0f ? 1u : 0u) | (v[i].y > 0.0f ? 2u : 0u) | (v[i].z > 0.0f ? 4u : 0u) |
                              (v[i].w > 0.0f ? 8u : 0u);
        if (i < 8) m.lo |= bits << (4 * i);
        else m.hi |= bits << (4 * (i - 8));
    }
    return m;
}

// acc[q] = mask bit (16 t + q) ? acc[q] : 0   (threshold_backward: grad * (output > 0))
__device__ __forceinline__ void apply_mask(f32x16& acc, const Mask64& m, int t) {
    const unsigned w = (t < 2 ? m.lo : m.hi) >> (16 * (t & 1));
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = ((w >> q) & 1u) ? acc[q] : 0.0f;
}

__device__ __forceinline__ void zero_tile(f32x16& acc) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
}

template <int NB, bool FINAL = false>   // number of blocks (their masks live in registers): 0 .. 3 (four would spill: no
                                        // scratch traffic may share the counter of the LDS-DMA ring); FINAL: the incoming
                                        // gradient is g_params and the final Linear's input gradient is computed first
__global__ void __launch_bounds__(kBlock, 2) resnet_hidden_backward_kernel(const TrainArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TrainStream sm;
    start_stream(sm, a.w, lds_dyn, a.num_stages, tid, FINAL ? a.final_first : 0);
    // FINAL: this wave's ring of g_params images, behind the weight ring
    [[maybe_unused]] char* gp_images = reinterpret_cast<char*>(lds_dyn + kTrainRing * kStageVec4 * 4) +
                                       __builtin_amdgcn_readfirstlane(wave) * (kTrainRing * kGpImageBytes);
    // (the staged full-line stores measured no gain in this kernel -- 92.1 vs 91.5 us -- and cost 92 bytes of scratch:
    //  the backward pass keeps the direct stores)
    float* stage = nullptr;
    const int64_t num_quads = a.batch >> 7;
    const int64_t plane = a.batch * 128;
    const int tiles_x = (a.di + 31) >> 5;
    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = (quad << 7) + (wave << 5);
        f32x16 gs[4];   // g_h: the gradient of the residual stream, fp32, accumulator layout
        if constexpr (FINAL) {
            // ---- g_h = g_params W_f, BEFORE anything else of the row block is live (the masks of two blocks held across
            //      this loop pushed the allocation into 115 spilled registers): the first kTrainAhead images of the row
            //      block -- nothing else is in flight, the ring's requests were drained at the end of the previous block
            int lane_f = lane;
            asm volatile("" : "+v"(lane_f));
#pragma unroll
            for (int ks = 0; ks < kTrainAhead; ++ks)
                if (ks < a.final_ksteps) gp_request(a.gparams, a.out_features, row0, ks, gp_images + ks * kGpImageBytes, lane_f);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t) zero_tile(gs[t]);
            for (int ks = 0; ks < a.final_ksteps; ++ks)
                kstep_final(gs, sm, a.gparams, a.out_features, row0, ks, a.final_ksteps, gp_images, lane_f);
            // g_h leaves at once (K10's grad_outputs of the last block's second Linear)
#pragma unroll
            for (int t = 0; t < 4; ++t) store_tile<false>(a.grad_hidden, row0 + (lane_f & 31), t, lane_f >> 5, gs[t], nullptr);
            // the masks' ordinary loads follow: everything in flight -- these stores, the ring's two stages ahead -- is
            // awaited first and the loads are awaited explicitly (load_mask), so no compiler-counted wait meets an LDS-DMA
            // request
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        int lane_here = lane, di = a.di;
        asm volatile("" : "+v"(lane_here), "+s"(di));
        const int half = lane_here >> 5, r = lane_here & 31;
        const int64_t row = row0 + r;
        // ---- (without the final Linear: with the ring drained) the ReLU masks of every block, then the incoming gradient
        Mask64 masks[2 * NB > 0 ? 2 * NB : 1];
        {
            // (packed by the forward kernel: one 8-byte load per plane; all of them in flight together)
            const unsigned long long* pm = packed_masks(a.saved, a.batch, NB) + ((row0 >> 5) << 6) + lane_here;
            unsigned long long w[2 * NB > 0 ? 2 * NB : 1];
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) w[i] = pm[(int64_t)i * (a.batch << 1)];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) {
                masks[i].lo = (unsigned)w[i];
                masks[i].hi = (unsigned)(w[i] >> 32);
            }
        }
        if constexpr (!FINAL) {
            vec4f v[16];
            load_tiles_raw(v, a.x, row, half);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    gs[t][4 * q4] = v[4 * t + q4].x;
                    gs[t][4 * q4 + 1] = v[4 * t + q4].y;
                    gs[t][4 * q4 + 2] = v[4 * t + q4].z;
                    gs[t][4 * q4 + 3] = v[4 * t + q4].w;
                }
        }
        // ---- blocks, last to first: g_a = (g_h W_1) . [a > 0];  g_h += (g_a W_0) . [h > 0]
#pragma unroll
        for (int blk = NB - 1; blk >= 0; --blk) {
            f32x16 u[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) zero_tile(u[t]);
            // (round 4: an array is stored tile by tile by the GEMM that consumes it -- gemm_from_tiles' `save`: g_h of the
            //  block behind this one here, g_a of this block in the second GEMM, g_h_0 in the initial layer's GEMM below)
#ifdef NFA_K14_BWD_GH_IN_GEMM
            if (blk < NB - 1) gemm_from_tiles<false, true>(u, gs, sm, lane, a.grads + (2 * (blk + 1)) * plane, row, half, stage);
            else gemm_from_tiles<false>(u, gs, sm, lane);
#else
            gemm_from_tiles<false>(u, gs, sm, lane);
#endif
            float* ga = a.grads + (2 * blk + 1) * plane;
            // (three blocks: the interleaved form spills 12 bytes per lane -- scratch reloads share vmcnt with the ring --,
            //  so that instance keeps the stores in front of the GEMM)
            constexpr bool kSpread = NB <= 2;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                apply_mask(u[t], masks[2 * blk + 1], t);
#ifndef NFA_K14_BURST_STORES
                if constexpr (!kSpread)
#endif
                    store_tile<false>(ga, row, t, half, u[t]);
            }
            f32x16 v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) zero_tile(v[t]);
#ifndef NFA_K14_BURST_STORES
            if constexpr (kSpread) gemm_from_tiles<false, true>(v, u, sm, lane, ga, row, half, stage);
            else
#endif
                gemm_from_tiles<false>(v, u, sm, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                apply_mask(v[t], masks[2 * blk], t);
#pragma unroll
                for (int q = 0; q < 16; ++q) gs[t][q] += v[t][q];   // the skip connection's gradient
#if defined(NFA_K14_BURST_STORES) || !defined(NFA_K14_BWD_GH_IN_GEMM)
                store_tile<false>(a.grads + (2 * blk) * plane, row, t, half, gs[t], stage);
#endif
            }
        }
#if !defined(NFA_K14_BURST_STORES) && defined(NFA_K14_BWD_GH_IN_GEMM)
        if (NB > 0) {   // g_h_0 (the initial Linear's grad_outputs): nothing GEMM-shaped left to hide it behind but the
                        // one or two output tiles of g_x
#pragma unroll
            for (int t = 0; t < 4; ++t) store_tile<false>(a.grads, row, t, half, gs[t], stage);
        }
#endif
        // ---- initial layer: g_x = g_h_0 W_in, one 32-column tile at a time
        for (int t = 0; t < tiles_x; ++t) {
            f32x16 acc;
            zero_tile(acc);
            gemm_tile_from_tiles(acc, gs, sm, lane);
            float* gx = a.out + row * di + 32 * t + 4 * half;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                if (32 * t + 8 * q4 + 4 * half < di)
                    *reinterpret_cast<vec4f*>(gx + 8 * q4) = vec4f{acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the packer: fp32 parameters -> the two streams of split-bf16 stages (weights change every optimiser step, so
//      packing is part of the training step: one launch, one workgroup per 12 KB stage, one more for the biases).
//      Same bytes as ops.pack_resnet_hidden_train's tensor-operation reference (split3 rounds like torch's
//      .to(bfloat16)).
struct PackArgs {
    const float* w_in;
    const float* b_in;
    const float* blk[3][4];   // W_0, b_0, W_1, b_1 per block
    const float* w_f;         // optional final Linear [out_features, 128] and its bias
    const float* b_f;
    __bf16* fwd;
    float* fwd_bias;
    float* final_bias;        // [final_tiles * 32], accumulator order, zeros past out_features
    __bf16* bwd;
    int di, nb, init_ks, out_features, final_tiles;
    int H;   // the net's hidden width (<= 128): rows / columns past it are zero in the streams (units that stay 0)
};

// input feature consumed at (k-step ks, lane-half hf, element j) when the GEMM's input is the previous layer's
// accumulator tiles (ops._k8_column_order)
__device__ __forceinline__ int acc_col(int ks, int hf, int j) {
    return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * hf + (j & 3);
}

__global__ void __launch_bounds__(kBlock) pack_resnet_hidden_kernel(const PackArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;   // grp: tile t (k-major) or k4 (tile-major)
    const int hf = lane >> 5, i = lane & 31;
    const int n_hid = a.init_ks + 16 * a.nb, n_fwd = n_hid + 2 * a.final_tiles, n_bwd_k = 16 * a.nb,
              tiles_x = (a.di + 31) >> 5, n_final_t = (a.out_features + 15) >> 4;
    int s = blockIdx.x;
    if (s == n_fwd + n_bwd_k + 2 * tiles_x + n_final_t) {   // the biases, accumulator order: [tile][half][q] = b[32 tile + 8 (q / 4) + 4 half + q % 4]
        for (int e = tid; e < 128 * (1 + 2 * a.nb); e += kBlock) {
            const int v = e >> 7, r = e & 127;
            const float* b = v == 0 ? a.b_in : a.blk[(v - 1) >> 1][((v - 1) & 1) ? 3 : 1];
            const int tile = r >> 5, half = (r >> 4) & 1, q = r & 15;
            const int src = 32 * tile + 8 * (q >> 2) + 4 * half + (q & 3);
            a.fwd_bias[e] = src < a.H ? b[src] : 0.0f;
        }
        for (int e = tid; e < 32 * a.final_tiles; e += kBlock) {
            const int tile = e >> 5, half = (e >> 4) & 1, q = e & 15;
            const int src = 32 * tile + 8 * (q >> 2) + 4 * half + (q & 3);
            a.final_bias[e] = src < a.out_features ? a.b_f[src] : 0.0f;
        }
        return;
    }
    float v[8];
    __bf16* dst;
    bool tile_major = false;
    if (s >= n_hid && s < n_fwd) {   // the final Linear, tile-major: [piece][k4][lane], rows past out_features are zero
        dst = a.fwd + (size_t)s * 6144;
        const int s2 = s - n_hid, tile = s2 >> 1, hs = s2 & 1;
        const int row = 32 * tile + i, ks = 4 * hs + grp;
        tile_major = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = acc_col(ks, hf, j);
            v[j] = (row < a.out_features && col < a.H) ? a.w_f[row * a.H + col] : 0.0f;
        }
    } else if (s < n_hid) {
        dst = a.fwd + (size_t)s * 6144;
        const int row = 32 * grp + i;
        if (s < a.init_ks) {   // initial layer: k = ks*16 + hf*8 + j, zeros past d_i
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = s * 16 + hf * 8 + j;
                v[j] = (col < a.di && row < a.H) ? a.w_in[row * a.di + col] : 0.0f;
            }
        } else {
            const int lin = (s - a.init_ks) >> 3, ks = (s - a.init_ks) & 7;
            const float* w = a.blk[lin >> 1][(lin & 1) ? 2 : 0];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = acc_col(ks, hf, j);
                v[j] = (row < a.H && col < a.H) ? w[row * a.H + col] : 0.0f;
            }
        }
    } else if (s < n_fwd + n_bwd_k) {   // W_1^T, W_0^T per block, last block first
        s -= n_fwd;
        dst = a.bwd + (size_t)s * 6144;
        const int lin = s >> 3, ks = s & 7;
        const float* w = a.blk[a.nb - 1 - (lin >> 1)][(lin & 1) ? 0 : 2];
        const int row = 32 * grp + i;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = acc_col(ks, hf, j);
            v[j] = (row < a.H && col < a.H) ? w[col * a.H + row] : 0.0f;
        }
    } else if (s >= n_fwd + n_bwd_k + 2 * tiles_x) {   // W_f^T (round 4: the final Linear's input gradient inside the
        // backward kernel), k-major over the out_features columns of g_params in their natural order: A[h][c] = W_f[c][h]
        s -= n_fwd;
        dst = a.bwd + (size_t)s * 6144;
        const int ks = s - n_bwd_k - 2 * tiles_x;
        const int row = 32 * grp + i;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = ks * 16 + hf * 8 + j;
            v[j] = (col < a.out_features && row < a.H) ? a.w_f[col * a.H + row] : 0.0f;
        }
    } else {                            // W_in^T, tile-major: [piece][k4][lane], rows past d_i are zero
        s -= n_fwd;
        dst = a.bwd + (size_t)s * 6144;
        const int s2 = s - n_bwd_k, tile = s2 >> 1, hs = s2 & 1;
        const int row = 32 * tile + i, ks = 4 * hs + grp;
        tile_major = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = acc_col(ks, hf, j);
            v[j] = (row < a.di && col < a.H) ? a.w_in[col * a.di + row] : 0.0f;
        }
    }
    bf16x2 hh[4], mm[4], ll[4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) split3(vec2f{v[2 * j2], v[2 * j2 + 1]}, hh[j2], mm[j2], ll[j2]);
    const bf16x8 ph = join4(hh[0], hh[1], hh[2], hh[3]), pm = join4(mm[0], mm[1], mm[2], mm[3]),
                 pl = join4(ll[0], ll[1], ll[2], ll[3]);
    bf16x8* out = reinterpret_cast<bf16x8*>(dst);
    if (tile_major) {   // ((piece * 4 + k4) * 64 + lane)
        out[(0 * 4 + grp) * 64 + lane] = ph;
        out[(1 * 4 + grp) * 64 + lane] = pm;
        out[(2 * 4 + grp) * 64 + lane] = pl;
    } else {            // ((tile * 3 + piece) * 64 + lane)
        out[(grp * 3 + 0) * 64 + lane] = ph;
        out[(grp * 3 + 1) * 64 + lane] = pm;
        out[(grp * 3 + 2) * 64 + lane] = pl;
    }
}

// `hidden_features` is the width of the arrays the two kernels see (always 128); the packer takes narrower nets and
// pads them into the 128-wide streams (units past the net's width have zero weights and biases on both sides:
// relu(0) = 0 forward, zero masks backward)
static int check_train(int64_t batch, int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                       bool packer = false) {
    if (batch < 0 || num_identity < 1 || num_blocks < 0) return NFA_ERR_INVALID_ARGUMENT;
    if (packer ? (hidden_features < 4 || hidden_features > 128 || (hidden_features & 3) != 0) : hidden_features != 128)
        return NFA_ERR_UNSUPPORTED;
    if (num_identity > 64 || (num_identity & 3) != 0 || (batch & 127) != 0 || num_blocks > 3) return NFA_ERR_UNSUPPORTED;
    return NFA_OK;
}

static dim3 train_grid(int64_t batch) {
    int64_t blocks = batch >> 7;
    const int64_t cap = (int64_t)device_cu_count() * 2;
    return dim3((unsigned)(blocks > cap ? cap : blocks));
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_resnet_hidden_forward_f32(const float* identity_inputs, const void* weights_packed,
                                             const float* bias_packed, float* saved, float* hidden,
                                             const float* final_bias_packed, float* params, int32_t out_features,
                                             int64_t batch, int32_t num_identity, int32_t hidden_features,
                                             int32_t num_blocks, void* stream) {
    const int rc = check_train(batch, num_identity, hidden_features, num_blocks);
    if (rc != NFA_OK) return rc;
    if (out_features < 0) return NFA_ERR_INVALID_ARGUMENT;
    if ((out_features & 3) != 0 || out_features > 32 * 1024) return NFA_ERR_UNSUPPORTED;
    if (batch == 0) return NFA_OK;
    if (!identity_inputs || !weights_packed || !bias_packed || !hidden || (num_blocks > 0 && !saved) ||
        (out_features > 0 && (!final_bias_packed || !params)))
        return NFA_ERR_INVALID_ARGUMENT;
    TrainArgs a;
    a.x = identity_inputs;
    a.w = reinterpret_cast<const vec4f*>(weights_packed);
    a.bias = bias_packed;
    a.saved = saved;
    a.out = hidden;
    a.grads = nullptr;
    a.batch = batch;
    a.di = num_identity;
    a.num_blocks = num_blocks;
    const int init_ks = num_identity > 32 ? 4 : 2;
    a.fbias = final_bias_packed;
    a.params = params;
    a.out_features = out_features;
    a.final_tiles = (out_features + 31) / 32;
    a.gparams = nullptr;
    a.grad_hidden = nullptr;
    a.final_ksteps = a.final_first = 0;
    a.num_stages = init_ks + 16 * num_blocks + 2 * a.final_tiles;
    const size_t lds = (size_t)kTrainRing * kStageVec4 * 16 + (size_t)(kBlock / kWave) * kTrainTileFloats * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    void (*kern)(const TrainArgs) = init_ks == 2 ? resnet_hidden_forward_kernel<2> : resnet_hidden_forward_kernel<4>;
    if (lds > 64 * 1024) {
        static unsigned long long raised[2] = {};   // device masks (raise_dynamic_lds)
        const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[init_ks == 2 ? 0 : 1], (int)lds);
        if (rc_lds != NFA_OK) return rc_lds;
    }
    hipLaunchKernelGGL(kern, train_grid(batch), dim3(kBlock), lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

static int launch_train_backward(const float* grad_hidden, const float* grad_params, int32_t out_features,
                                 float* grad_hidden_out, const void* weights_packed, const float* saved, float* grads,
                                 float* grad_identity_inputs, int64_t batch, int32_t num_identity,
                                 int32_t hidden_features, int32_t num_blocks, void* stream) {
    const int rc = check_train(batch, num_identity, hidden_features, num_blocks);
    if (rc != NFA_OK) return rc;
    const bool with_final = grad_params != nullptr || out_features != 0;
    if (with_final && out_features < 4) return NFA_ERR_INVALID_ARGUMENT;
    if (with_final && ((out_features & 3) != 0 || out_features > 32 * 1024)) return NFA_ERR_UNSUPPORTED;
    if (batch == 0) return NFA_OK;
    if ((!with_final && !grad_hidden) || (with_final && (!grad_params || !grad_hidden_out)) || !weights_packed ||
        !grad_identity_inputs || (num_blocks > 0 && (!saved || !grads)))
        return NFA_ERR_INVALID_ARGUMENT;
    TrainArgs a;
    a.x = grad_hidden;
    a.w = reinterpret_cast<const vec4f*>(weights_packed);
    a.bias = nullptr;
    a.saved = const_cast<float*>(saved);
    a.out = grad_identity_inputs;
    a.grads = grads;
    a.batch = batch;
    a.di = num_identity;
    a.num_blocks = num_blocks;
    a.fbias = nullptr;
    a.params = nullptr;
    a.out_features = 0;
    a.final_tiles = 0;
    a.num_stages = 16 * num_blocks + 2 * ((num_identity + 31) / 32);
    a.gparams = grad_params;
    a.grad_hidden = grad_hidden_out;
    a.final_ksteps = 0;
    a.final_first = 0;
    if (with_final) {   // W_f^T's k-major stages sit behind W_in^T's in the backward stream and are consumed FIRST
        a.out_features = out_features;
        a.final_ksteps = (out_features + 15) / 16;
        a.final_first = a.num_stages;
        a.num_stages += a.final_ksteps;
    }
    // behind the weight ring: the store images (unused here: the backward pass stores directly) or, with the final
    // Linear, every wave's ring of g_params images
    const size_t lds = (size_t)kTrainRing * kStageVec4 * 16 +
                       (with_final ? (size_t)(kBlock / kWave) * kTrainRing * kGpImageBytes
                                   : (size_t)(kBlock / kWave) * kTrainTileFloats * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = train_grid(batch), block(kBlock);
    void (*kern)(const TrainArgs) = nullptr;
    if (with_final)
        kern = num_blocks == 0 ? resnet_hidden_backward_kernel<0, true> : num_blocks == 1 ? resnet_hidden_backward_kernel<1, true>
               : num_blocks == 2 ? resnet_hidden_backward_kernel<2, true> : resnet_hidden_backward_kernel<3, true>;
    else
        kern = num_blocks == 0 ? resnet_hidden_backward_kernel<0> : num_blocks == 1 ? resnet_hidden_backward_kernel<1>
               : num_blocks == 2 ? resnet_hidden_backward_kernel<2> : resnet_hidden_backward_kernel<3>;
    if (lds > 64 * 1024) {
        static unsigned long long raised[8] = {};   // device masks (raise_dynamic_lds)
        const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[num_blocks + (with_final ? 4 : 0)], (int)lds);
        if (rc_lds != NFA_OK) return rc_lds;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_resnet_hidden_backward_f32(const float* grad_hidden, const void* weights_packed, const float* saved,
                                              float* grads, float* grad_identity_inputs, int64_t batch,
                                              int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                                              void* stream) {
    if (!grad_hidden && batch > 0) return NFA_ERR_INVALID_ARGUMENT;
    return launch_train_backward(grad_hidden, nullptr, 0, nullptr, weights_packed, saved, grads, grad_identity_inputs, batch,
                                 num_identity, hidden_features, num_blocks, stream);
}

extern "C" int nfa_resnet_backward_f32(const float* grad_params, int32_t out_features, const void* weights_packed,
                                       const float* saved, float* grads, float* grad_hidden,
                                       float* grad_identity_inputs, int64_t batch, int32_t num_identity,
                                       int32_t hidden_features, int32_t num_blocks, void* stream) {
    if (out_features < 4) return NFA_ERR_INVALID_ARGUMENT;
    if (!grad_params && batch > 0) return NFA_ERR_INVALID_ARGUMENT;
    return launch_train_backward(nullptr, grad_params, out_features, grad_hidden, weights_packed, saved, grads,
                                 grad_identity_inputs, batch, num_identity, hidden_features, num_blocks, stream);
}

extern "C" int nfa_pack_resnet_hidden_train_f32(const float* initial_weight, const float* initial_bias,
                                                const float* const* block_params, const float* final_weight,
                                                const float* final_bias, int32_t out_features, int32_t num_identity,
                                                int32_t hidden_features, int32_t num_blocks, void* forward_stages,
                                                float* forward_bias, float* final_bias_packed, void* backward_stages,
                                                void* stream) {
    const int rc = check_train(0, num_identity, hidden_features, num_blocks, true);
    if (rc != NFA_OK) return rc;
    if (out_features < 0) return NFA_ERR_INVALID_ARGUMENT;
    if ((out_features & 3) != 0 || out_features > 32 * 1024) return NFA_ERR_UNSUPPORTED;
    if (!initial_weight || !initial_bias || !forward_stages || !forward_bias || !backward_stages ||
        (num_blocks > 0 && !block_params) || (out_features > 0 && (!final_weight || !final_bias || !final_bias_packed)))
        return NFA_ERR_INVALID_ARGUMENT;
    PackArgs a;
    a.w_in = initial_weight;
    a.b_in = initial_bias;
    for (int k = 0; k < 3; ++k)
        for (int q = 0; q < 4; ++q) {
            a.blk[k][q] = k < num_blocks ? block_params[4 * k + q] : nullptr;
            if (k < num_blocks && !a.blk[k][q]) return NFA_ERR_INVALID_ARGUMENT;
        }
    a.fwd = reinterpret_cast<__bf16*>(forward_stages);
    a.fwd_bias = forward_bias;
    a.bwd = reinterpret_cast<__bf16*>(backward_stages);
    a.di = num_identity;
    a.nb = num_blocks;
    a.init_ks = num_identity > 32 ? 4 : 2;
    a.H = hidden_features;
    a.w_f = final_weight;
    a.b_f = final_bias;
    a.final_bias = final_bias_packed;
    a.out_features = out_features;
    a.final_tiles = (out_features + 31) / 32;
    const int stages = a.init_ks + 32 * num_blocks + 2 * a.final_tiles + 2 * ((num_identity + 31) / 32) + (out_features + 15) / 16;
    hipLaunchKernelGGL(pack_resnet_hidden_kernel, dim3(stages + 1), dim3(kBlock), 0, (hipStream_t)stream, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
