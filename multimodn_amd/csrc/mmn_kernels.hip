// mmn_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the MultiModN sequential-fusion training
// step and the C ABI declared in include/mmn_hip.h.  Written for wave64 + fp32 MFMA
// (v_mfma_f32_16x16x4_f32: bit-exact fp32 FMA chains, so fp32 parity with the reference's ATen
// path holds up to summation order).  No CUDA names, no dual paths.
//
// Launch structure of one training step (reference: multimodn/multimodn.py:137-204):
//   k_prepare          any(isnan(x_k)) per data slot (:168) + repack of every weight matrix the
//                      chain kernels multiply by into MFMA-fragment order (forward W, backward W^T)
//   k_fb8<TILED>       FUSED forward + reverse chain of one 16-row tile, 8 waves (shapes like the
//                      MIMIC config, E <= 4): init broadcast, hidden MLPs, state updates, state-change
//                      partials, all D decoders on all E+1 states, CE-over-sigmoid, argmax, confusion
//                      counts, then grads wrt states / pre-activations; TILED = per-sample mode
//     k_fwd8 + k_bwd8  the same math as two 8-wave kernels (E <= 8; also the forward-only path)
//     k_chain_*_par    4-wave parallel-phase kernels (dims that fit LDS)
//     k_chain_fwd/bwd  sequential form, any shape, 16- or 32-row tiles
//     k_genf_fwd/bwd   generic tier, fast form: MIMIC_MLPEncoder (state enters the FIRST layer, activation on
//                      every layer, dropout multipliers from mmn_batch.drop_mask) + MLPDecoder heads; decoder
//                      operands / biases / item tables in LDS, descriptor as kernel argument
//     k_gen_fwd/bwd    generic tier, sequential form: any mix of encoder / decoder kinds and shapes
//   k_wgrad            grouped split-K "A^T B" GEMM: weight, bias and init-state grads as
//                      flat-gradient-shaped partial slabs
//   k_reduce           fixed-order slab reduction -> grads (+ Adam on the element just summed, :204);
//                      tile partials -> stats block (+ loss combination and epoch accumulators, :194-212)
//   k_adam / k_adam_accumulate   optimizer.step() over the flat buffers as its own launch
//                      (data parallel: after the all-reduce, together with the epoch accumulation)
//   k_ps_code / k_ps_layout / k_ps_gather   per-sample mode: regrouping of the rows into tiles of one
//                      executed sequence (mmn_regroup)
// File order: plan structs and device helpers; k_prepare; sequential chain kernels; generic tier
// (sequential form, fast form); 4-wave parallel kernels; 8-wave kernels (k_fwd8, k_bwd8, k_fb8); k_wgrad;
// k_ps_*; k_adam; k_reduce; host side (layout, plan, C ABI).
//
// Data layout in HBM (all fp32 row-major, B = batch rows):
//   states[e][B][S]     output state of encoder e          hid[e][l][B][H_l]  hidden activations
//   dz[r][B][2D]        d loss / d decoder logits, row r   dS[e][B][S]        d loss / d state_e (+dS0)
//   dpre[e][l][B][H_l]  d loss / d hidden pre-activation   slabs              split-K partial grads
//   sin[e][B][S]        per-sample mode: state that fed encoder e
//   gact / gdpre        generic tier: xin[e][B][F+S] = the (masked) cat[x, state] of a MIMIC encoder; per grid row
//                       the decoders' hidden activations / pre-activation gradients side by side, [B][dcols]
//   pack                weights in fragment order: [col tile][k-step][lane][4] (zero padded)
//
// Tiling.  A workgroup owns 16 batch rows (32 in the sequential tier's RT = 2 form); the state tiles
// stay in LDS for the whole chain.  Every product is "tile[rows x K] x W'[N x K]^T" with the
// ACTIVATION tile in LDS and the WEIGHT fragments loaded straight from L2 into registers (each
// weight element is used by exactly one wave of the workgroup, so an LDS round trip would be pure
// overhead).  The contraction is walked 16 at a time: one 16-byte fragment per operand feeds four
// MFMAs.
//
// What bounds the chain kernels is not the MFMAs (~12 us for both directions at B = 4096) but (a) the
// number of DEPENDENT global round trips (~0.7-2.5 us each on a busy chip), (b) the rate at which one
// CU can pull the weights through its vector-memory pipe: every workgroup needs ALL weights (~700 KB
// of fragments over both directions at the MIMIC shape), and (c) hipcc's wait counters: a branch
// around a load makes it drain the whole queue at the next use.  Hence: descriptors in kernel
// arguments; weights read from a per-step repack in which one wave-level load is 1 KB contiguous (a
// row-major fragment touches 16 half cache lines and measured ~12 GB/s per CU); every request
// unconditional and issued well ahead of its use.  DESIGN.md section 3 has the measurements.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <type_traits>
#include <algorithm>
#include <vector>

#include "mmn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT = 256;       // threads per workgroup
constexpr int XCH = 128;      // x columns staged per chunk
constexpr int LDX = 132;      // row stride of the x chunk image
constexpr int LDZ = 20;       // row stride of the dz tile (16 + 4)
constexpr int TILE_LD = 65;   // row stride of the wgrad reduction tile

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// smallest row stride = 4 (mod 64) floats that holds round_up(k, 16) floats: rows stay 16-byte
// aligned and the 16-byte fragment reads of 16 consecutive rows spread over all 64 banks
__host__ __device__ inline int pick_ld(int k) {
    const int k16 = round_up(k, 16);
    int ld = 68;
    while (ld < k16) ld += 64;
    return ld;
}

// ------------------------------------------------------------------------------------------------
// device-side plan (lives at the start of the workspace)
// ------------------------------------------------------------------------------------------------
enum { A_DPRE = 0, A_DS = 1, A_DZ = 2, A_GEN = 3 };
enum { IN_NONE = 0, IN_X = 1, IN_HID = 2, IN_STATE_ROW = 3, IN_PREV_STATE = 4 };   // + IN_INIT = 5, IN_GEN = 6 below

struct WTask {
    int32_t a_kind, a_enc, a_idx;      // A_DPRE: (enc, layer) ; A_DS: idx ; A_DZ: row
    int32_t M;
    int32_t in0_kind, in0_enc, in0_idx, k0;
    int32_t in1_kind, k1;
    int32_t bias, ntot;
    int32_t gate;                      // exec_flags index that must be set, else slab tile = 0
    int32_t w_ld;                      // row stride of the weight gradient in flat order (in_dim; decoders: S)
    // Partial slabs are FLAT-GRADIENT shaped: partial k of a task lives at part_base + k * part_stride +
    // (flat index of the gradient element), so k_reduce sums element i straight from its own index.
    int64_t part_base, part_stride;    // encoders / init state: 0, nA ; decoder row r: nA*ks + r*ks*nB, nB
    int64_t w_flat, b_flat;            // flat index (inside the region) of the weight / bias gradient tensor
    int32_t dec_stride, pad2;          // decoder tasks: flat distance between consecutive decoders, else 0
    // generic tier (MIMIC_MLPEncoder / MLPDecoder): operands in the gdpre / gact regions
    int64_t a_gen_off, in_gen_off;     // A_GEN: float offset into gdpre ; IN_GEN: float offset into gact
    int32_t gen_lda, gen_ldi;          // their row strides
    int32_t a_col, pad4;               // A_DZ: first column (decoder d's pair) when M < 2D
};

struct WItem {
    int32_t task, m0, n0, ks, nks;     // n0 relative to the source `src`; ks of nks row-range splits
    int16_t mt, nt;                    // interleave widths (1, 2 or 4): tile = 16*mt x 16*nt
    int16_t src, bias;                 // src 0: in0, 1: in1, 2: none (bias only)
};

// One k_wgrad work item, fully resolved at plan creation: a workgroup reads ONE record (a
// wave-uniform, scalar load) and knows its operands, instead of chasing plan -> item -> task ->
// flags through four dependent global loads.
enum { IN_INIT = 5, IN_GEN = 6 };
struct WRec {
    int64_t a_off;                      // float offset of A inside its region (dpre / dS / dz)
    int64_t in_off;                     // float offset of the In source inside its region (hid / states)
    int64_t w_off, b_off;               // float offsets of this item's partial of the weight / bias gradient (flat order)
    int32_t a_kind, lda, M, m0;
    int32_t in_kind, in_enc, ldi, ncols;
    int32_t n0, w_ld, col_off, gate;
    int32_t ks;
    int16_t mt, nt, has_in, bias;
    int32_t nks;                        // row-range splits of this item's task
    int32_t dec_stride;                 // decoder tasks: row m belongs to decoder m >> 1, class m & 1
    int32_t pad3;
};
struct WgArgs {
    const WRec* recs;
    const float* dpre; const float* dS; const float* dz; const float* states; const float* hid; const float* init;
    float* slabs;
    long long* stamps;
    const float* sin;                   // per-sample mode: In operand of the state-update weights
    int32_t maxB, S;
    const float* gact; const float* gdpre;   // generic tier: layer inputs / pre-activation gradients
};

struct Seg {                            // one gradient tensor
    float* dst;
    int64_t start;                      // first flat element index
    int64_t slab_base, pstride;
    int32_t count, n_partials, kdiv, ntot, coff, row_off;
    int32_t gate, pad;                  // state row that must exist this step for the tensor to have a gradient
};

// one matrix to repack: value(n, kk) = src[n*ld + col(kk)] (mode 0) or src[col(kk)*ld + n] (mode 1,
// i.e. the transposed matrix), col(kk) through the same two-segment map the kernels use; element
// (n, kk) lands at dst[((n/16 * T + (kk+kk_off)/16) * 64 + 16*((kk+kk_off)%16/4) + n%16) * 4 + (kk+kk_off)%4].
struct PackTask {
    const float* src; float* dst;
    int32_t ld_src, mode, N, T;
    int32_t len0, col0, len1, col1;
    int32_t kk_off, ntiles;
    int64_t start;                      // first flat element index of this task
};

struct DevPlan {
    mmn_model m;
    int32_t S, E, D, R, S16, ldS, ldH, maxB, max_tiles, KS, ldX, par_ok;
    int64_t hid_off[MMN_MAX_ENCODERS][MMN_MAX_LAYERS];   // float offsets into hid / dpre
    int64_t pkf_off[MMN_MAX_ENCODERS][MMN_MAX_LAYERS];   // float offsets into pack: forward operand of layer l
    int64_t pkb_off[MMN_MAX_ENCODERS][MMN_MAX_LAYERS];   // backward operand (W^T) of layer l, -1 if unused;
                                                         // for the state update: the carry part (columns of the state)
    int64_t pkh_off[MMN_MAX_ENCODERS];                   // state update, dh part (columns of h), -1 if no hidden layer
    int64_t pkd_off;                                     // backward decoder operand Wdec^T [S x 2D]
    float* states; float* hid; float* dpre; float* dz; float* dS;
    float* sin;               // per-sample mode: state that fed encoder e, [E][maxB][S]
    float* pack;              // fragment-ordered weights, rewritten by k_prepare every step
    float* lossp; float* scp; int32_t* cnt;   // cnt[tile][R*D][5]: per-tile integer counter partials
    int32_t* exec_flags;      // [R]   1 if state row r was produced this step
    int32_t* prev_row;        // [E]   state row that fed encoder e this step
    int32_t* nan_flags;       // [MMN_MAX_ENCODERS] NaN-found flag per data slot
    float* slabs; float* stats; double* epoch;
    long long* stamps;        // diagnostic phase timestamps (MMN_STAMPS=1), else nullptr
    WTask* tasks; WItem* items; Seg* segs; PackTask* ptasks; WRec* recs;
    int32_t n_tasks, n_items, n_segs, n_ptasks;
    int64_t n_grad_elems, n_pack_elems;
    // ---- generic tier (k_gen_fwd / k_gen_bwd: models with a MIMIC_MLPEncoder or an MLPDecoder)
    int32_t generic, pad5;
    float* gact;               // xin[e] [maxB x (F_e+S)] (MIMIC encoders: (masked) cat[x, state]) and the decoders'
                               // hidden activations dhid[r][d][l] [maxB x H]
    float* gdpre;              // d loss / d pre-activation of the decoders' hidden layers, same shape as dhid
    int64_t xin_off[MMN_MAX_ENCODERS];                           // float offsets into gact (-1: not a MIMIC encoder)
    int64_t dh_off[MMN_MAX_DECODERS][MMN_MAX_DEC_HIDDEN];        // first COLUMN of (d, l) inside a grid row's [maxB x dcols] block
    int64_t dh_row_stride;                                       // floats per grid row = maxB * dcols (gact and gdpre alike)
    int32_t dcols, dec_maxnh;                                    // hidden widths of all decoders side by side; deepest decoder
    int32_t dwf_off[MMN_MAX_DECODERS];                           // raw output Linear [2 x in] of decoder d inside biasbuf
    int64_t dh_base;                                             // start of the dhid blocks inside gact
    int64_t pkdf_off[MMN_MAX_DECODERS][MMN_MAX_DEC_HIDDEN + 1];  // forward operand of decoder layer l (last = output Linear)
    int64_t pkdb_off[MMN_MAX_DECODERS][MMN_MAX_DEC_HIDDEN + 1];  // backward operand (W^T)
    // the decoders' operands are reused on all E+1 state rows: when they fit they are copied into LDS once per
    // workgroup (forward operands in k_gen_fwd, backward ones in k_gen_bwd) and no decoder layer waits on global memory
    int64_t dec_f_off, dec_b_off;      // start of the contiguous forward / backward decoder operands inside pack
    int32_t dec_f_floats, dec_b_floats, dec_lds, gen_fast;
    // all biases of the model in one buffer (k_prepare gathers them every step), so that the fast kernels copy them
    // into LDS with one coalesced read instead of one dependent global load per layer epilogue
    float* biasbuf; const float* const* bias_src;
    int32_t n_bias, pad8;
    int32_t ebias_off[MMN_MAX_ENCODERS][MMN_MAX_LAYERS];
    int32_t dbias_off[MMN_MAX_DECODERS][MMN_MAX_DEC_HIDDEN + 1];
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Explicit address spaces.  Pointers reach the hot loops through structs, so the compiler cannot
// prove LDS vs global and would emit flat_* accesses whose waits (vmcnt(0) & lgkmcnt(0)) drain the
// weight prefetch at every k-step.  Every hot access goes through these helpers instead.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// wave index as a provably wave-uniform (SGPR) value: everything derived from it (tile ownership,
// "two column tiles?" flags) then compiles to scalar branches instead of exec masking
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
#define MMN_AS3 __attribute__((address_space(3)))
#define MMN_AS1 __attribute__((address_space(1)))
typedef MMN_AS3 float* lp;              // LDS pointers stay address-space typed end to end
typedef const MMN_AS3 float* clp;
__device__ __forceinline__ float lds_ld(clp p) { return *p; }
__device__ __forceinline__ f32x4 lds_ld4(clp p) { return *(const MMN_AS3 f32x4*)p; }
__device__ __forceinline__ void lds_st(lp p, float v) { *p = v; }
__device__ __forceinline__ void lds_st4(lp p, f32x4 v) { *(MMN_AS3 f32x4*)p = v; }
__device__ __forceinline__ float g_ld(const float* p) { return *(const MMN_AS1 float*)p; }
__device__ __forceinline__ int g_ldi(const int32_t* p) { return *(const MMN_AS1 int32_t*)p; }
__device__ __forceinline__ f32x2 g_ld2(const float* p) { return *(const MMN_AS1 f32x2*)p; }
__device__ __forceinline__ f32x4 g_ld4(const float* p) { return *(const MMN_AS1 f32x4*)p; }
__device__ __forceinline__ void g_st(float* p, float v) { *(MMN_AS1 float*)p = v; }
__device__ __forceinline__ void g_st2(float* p, f32x2 v) { *(MMN_AS1 f32x2*)p = v; }
__device__ __forceinline__ void g_st4(float* p, f32x4 v) { *(MMN_AS1 f32x4*)p = v; }
__device__ __forceinline__ void g_sti(int32_t* p, int v) { *(MMN_AS1 int32_t*)p = v; }

// Diagnostic only (MMN_STAMPS=1): 100 MHz timestamps of one workgroup's phases, written to a
// buffer nothing else reads.  With stamps == nullptr (always, outside diagnosis) this is one
// scalar compare per phase.
__device__ __forceinline__ void stamp(const long long* const* dummy, long long* stamps, int& k, int block) {
    (void)dummy;
    if (stamps && (int)blockIdx.x == block && threadIdx.x == 0 && k < 250) stamps[k] = (long long)wall_clock64();
    ++k;
}
#define STAMP() stamp(nullptr, p.stamps, stamp_k, stamp_block)

__device__ __forceinline__ float act_fwd(float v, int kind) {
    if (kind == MMN_ACT_RELU) return fmaxf(v, 0.0f);
    if (kind == MMN_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}
__device__ __forceinline__ float act_grad_from_out(float h, int kind) {
    if (kind == MMN_ACT_RELU) return h > 0.0f ? 1.0f : 0.0f;
    if (kind == MMN_ACT_SIGMOID) return h * (1.0f - h);
    return 1.0f;
}

__device__ __forceinline__ bool slot_present(const mmn_batch& b, int slot) {
    return b.nan_flags == nullptr || g_ldi(b.nan_flags + slot) == 0;
}

// The NaN flags of all data slots as one wave-uniform bit mask (bit = slot present), fetched ONCE
// per kernel: a per-use global load would sit behind every outstanding weight prefetch (vmcnt is
// in order), i.e. cost a full memory round trip at each phase boundary.
__device__ __forceinline__ unsigned present_mask(const mmn_batch& b) {
    if (b.nan_flags == nullptr) return 0xFFFFu;
    const int v = g_ldi(b.nan_flags + (threadIdx.x & 15));
    return (unsigned)(__ballot(v == 0) & 0xFFFFull);
}
__device__ __forceinline__ bool slot_present(unsigned pm, int slot) { return (pm >> slot) & 1u; }
__device__ __forceinline__ int next_exec(const mmn_batch& b, unsigned pm, int t) {
    while (t < b.n_seq && !slot_present(pm, b.seq_data[t])) ++t;
    return t;
}
__device__ __forceinline__ bool row_executed(const mmn_batch& b, unsigned pm, int r) {
    if (r == 0) return true;
    for (int t = 0; t < b.n_seq; ++t)
        if (b.seq_enc[t] == r - 1) return slot_present(pm, b.seq_data[t]);
    return false;
}
__device__ __forceinline__ int prev_row_of(const mmn_batch& b, unsigned pm, int e) {
    int prev_row = 0;
    for (int u = 0, pr = 0; u < b.n_seq; ++u) {
        if (!slot_present(pm, b.seq_data[u])) continue;
        if (b.seq_enc[u] == e) prev_row = pr;
        pr = b.seq_enc[u] + 1;
    }
    return prev_row;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- W' operand in fragment order: fragment (tile, t) = 64 lanes x float4 = 1 KB contiguous
struct PB {
    const float* pk;
    int T;        // k-steps (16 contraction elements each)
    int T0;       // k-steps whose activations come from A.a0 (the rest from A.a1)
    int N;        // valid output columns
};
__device__ __forceinline__ PB make_pb(const float* pk, int N, int len0, int len1) {
    PB b;
    b.pk = pk; b.N = N;
    b.T0 = (len0 + 15) >> 4;
    b.T = b.T0 + ((len1 + 15) >> 4);
    return b;
}
// activation operand: LDS images for the two contraction segments (zero/finite padded to 16);
// step t reads a0 + 16 t (t < T0) or a1 + 16 (t - T0)
struct ASrc { clp a0; int lda0; clp a1; int lda1; };

// The GEMM of one layer is split into ISSUE (request NS k-steps of weight fragments for this
// wave's two column tiles) and CONSUME (LDS activation fragments x those registers -> MFMA), so a
// caller can put unrelated work -- or a whole earlier layer -- between the two.  No branches and
// no bounds tests around the loads (the pack is zero padded): hipcc keeps counted vmcnt waits.
constexpr int TQ = 12;
template <int NS>
__device__ __forceinline__ void issue_b(f32x4 (&bq)[NS][2], const PB& B, const int (&n0)[2], int t_begin) {
    const int lane = threadIdx.x & 63;
    const int ntiles = (B.N + 15) >> 4;
    const int tl0 = min(n0[0] >> 4, ntiles - 1), tl1 = min(n0[1] >> 4, ntiles - 1);   // clamp: surplus re-reads a hot line
    const float* p0 = B.pk + ((int64_t)tl0 * B.T * 64 + lane) * 4;
    const float* p1 = B.pk + ((int64_t)tl1 * B.T * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int t = min(t_begin + j, B.T - 1);
        bq[j][0] = g_ld4(p0 + t * 256);
        bq[j][1] = g_ld4(p1 + t * 256);
    }
}

// NEXT != nullptr: after the MFMAs of slot j have been issued, slot j is immediately re-requested
// from *NEXT (the same layer of the next encoder), so the next fetch streams underneath this
// layer's MFMAs instead of stalling the wave for ~2 us of pure load issue afterwards.
template <int RT, int NS>
__device__ __forceinline__ void consume_b(f32x4 (&acc)[2][RT], const ASrc& A, const PB& B, f32x4 (&bq)[NS][2],
                                          int t_begin, int t_end, bool two, const PB* NEXT = nullptr,
                                          const int* next_n0 = nullptr) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    auto a_ptr = [&](int t, int& lda) -> clp {
        if (t < B.T0) { lda = A.lda0; return A.a0 + 16 * t + 4 * q; }
        lda = A.lda1;
        return A.a1 + 16 * (t - B.T0) + 4 * q;
    };
    const float* np0 = nullptr; const float* np1 = nullptr;
    int nT = 1;
    if (NEXT) {
        const int ntl = (NEXT->N + 15) >> 4;
        nT = NEXT->T;
        np0 = NEXT->pk + ((int64_t)min(next_n0[0] >> 4, ntl - 1) * nT * 64 + lane) * 4;
        np1 = NEXT->pk + ((int64_t)min(next_n0[1] >> 4, ntl - 1) * nT * 64 + lane) * 4;
    }
    f32x4 a_cur[RT], a_nxt[RT];
    {
        int lda; clp ap = a_ptr(min(t_begin, t_end - 1), lda);
#pragma unroll
        for (int r = 0; r < RT; ++r) a_cur[r] = lds_ld4(ap + (r * 16 + i) * lda);
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int t = t_begin + j;
        {   // unconditional (clamped) read of the next activation fragment: hides the LDS latency
            int lda; clp ap = a_ptr(min(t + 1, t_end - 1), lda);
#pragma unroll
            for (int r = 0; r < RT; ++r) a_nxt[r] = lds_ld4(ap + (r * 16 + i) * lda);
        }
        if (t < t_end) {
            if (two) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int r = 0; r < RT; ++r) acc[c][r] = mfma4(a_cur[r][e], bq[j][c][e], acc[c][r]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < RT; ++r) acc[0][r] = mfma4(a_cur[r][e], bq[j][0][e], acc[0][r]);
            }
        }
        if (NEXT) {
            const int tn = min(j, nT - 1);
            bq[j][0] = g_ld4(np0 + tn * 256);
            bq[j][1] = g_ld4(np1 + tn * 256);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) a_cur[r] = a_nxt[r];
    }
}

// acc[ct][rt] += A * W'[tiles n0]^T over k-steps [t_begin, t_end), just-in-time (issue + consume)
template <int RT>
__device__ __forceinline__ void wave_gemm_any(f32x4 (&acc)[2][RT], const ASrc& A, const PB& B, const int (&n0)[2],
                                              int t_begin, int t_end) {
    const bool two = n0[1] < B.N;
    for (int tb = t_begin; tb < t_end; tb += TQ) {
        f32x4 bq[TQ][2];
        issue_b<TQ>(bq, B, n0, tb);
        consume_b<RT, TQ>(acc, A, B, bq, tb, t_end, two);
    }
}

template <int RT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[2][RT]) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int RT, class Epi>
__device__ __forceinline__ void run_epilogue(const f32x4 (&acc)[2][RT], const int (&n0)[2], int N, Epi&& epi) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (n0[c] < N) {
            const int col = n0[c] + i;
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) epi(r * 16 + q * 4 + k, col, c, acc[c][r][k]);
        }
    }
}

// out[rows x N] = A * W'^T with A resident in LDS; epilogue per element (col may be >= N: skip there)
template <int RT, class Epi>
__device__ __forceinline__ void layer_nt(const ASrc& A, const PB& B, Epi&& epi) {
    const int wave = wave_id();
    const int ntiles = (B.N + 15) >> 4;
    for (int base = 0; base < ntiles; base += 8) {
        const int n0[2] = {16 * (base + wave), 16 * (base + wave + 4)};
        if (n0[0] >= B.N) continue;
        f32x4 acc[2][RT];
        zero_acc<RT>(acc);
        wave_gemm_any<RT>(acc, A, B, n0, 0, B.T);
        run_epilogue<RT>(acc, n0, B.N, [&](int row, int col, int, float v) { epi(row, col, v); });
    }
}

// Copy a [nrows x ncols] global tile (row stride ld_src) into an LDS image whose rows hold
// round_up(ncols,16) floats (zero filled), rows >= nrows zero filled.  Work split without
// divisions: a wave takes rows wave, wave+4, ...; lanes take 4 columns each.
__device__ __forceinline__ void stage_rows(lp dst, int ld_dst, const float* __restrict__ src, int64_t ld_src,
                                           int nrows, int rows_pad, int ncols) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int cpad = round_up(ncols, 16);
    const bool vec = ((ld_src & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    for (int r = wave; r < rows_pad; r += 4) {
        for (int c = lane * 4; c < cpad; c += 256) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < nrows && c < ncols) {
                const float* p = src + (int64_t)r * ld_src + c;
                if (vec && c + 3 < ncols) {
                    v = g_ld4(p);
                } else {
                    v.x = g_ld(p);
                    if (c + 1 < ncols) v.y = g_ld(p + 1);
                    if (c + 2 < ncols) v.z = g_ld(p + 2);
                    if (c + 3 < ncols) v.w = g_ld(p + 3);
                }
            }
            lds_st4(dst + r * ld_dst + c, v);
        }
    }
}

// LDS tile [nrows x ncols] -> global (row stride = ncols), coalesced
__device__ __forceinline__ void store_rows(float* __restrict__ dst, clp src, int ld_src, int nrows, int ncols) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const bool vec = ((ncols & 3) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    for (int r = wave; r < nrows; r += 4) {
        if (vec) {
            for (int c = lane * 4; c < ncols; c += 256) g_st4(dst + (int64_t)r * ncols + c, lds_ld4(src + r * ld_src + c));
        } else {
            for (int c = lane; c < ncols; c += 64) g_st(dst + (int64_t)r * ncols + c, lds_ld(src + r * ld_src + c));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_prepare: blocks [0, scan_blocks): nan_flags[slot] = 1 if data slot has a NaN (flags are zero
// on entry: plan creation zeroes them and k_reduce re-zeroes them after their last reader);
// blocks [scan_blocks, ...): repack of the weights into fragment order (one thread per element;
// padding elements are never written: the pack is zeroed once at plan creation).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_prepare(const DevPlan* __restrict__ P, mmn_batch b, int scan_blocks,
                                                int blocks_per_slot) {
    const DevPlan& p = *P;
    if ((int)blockIdx.x < scan_blocks) {
        const int t = blockIdx.x / blocks_per_slot;          // sequence position
        const int part = blockIdx.x - t * blocks_per_slot;
        const int slot = b.seq_data[t];
        const int F = p.m.enc[b.seq_enc[t]].n_features;
        const float* x = b.x[slot];
        const int64_t ld = b.ldx[slot];
        const int f4 = (F + 3) >> 2;
        const int64_t total = (int64_t)b.batch * f4;
        const bool vec = ((ld & 3) == 0) && ((F & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
        bool bad = false;
        const int64_t stride = (int64_t)blocks_per_slot * NT;
        if (vec) {
            // four independent 16-byte reads per thread in flight (a load -> test loop waits for every read in turn:
            // at the MIMIC shape that was four dependent HBM round trips per thread)
            for (int64_t idx = (int64_t)part * NT + threadIdx.x; idx < total; idx += 4 * stride) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t id = idx + u * stride;
                    const int64_t ic = id < total ? id : idx;                 // clamp: re-reads a valid element
                    const int64_t r = ic / f4;
                    v[u] = g_ld4(x + r * ld + ((int)(ic - r * f4) << 2));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) bad |= (v[u].x != v[u].x) | (v[u].y != v[u].y) | (v[u].z != v[u].z) | (v[u].w != v[u].w);
            }
        } else {
            for (int64_t idx = (int64_t)part * NT + threadIdx.x; idx < total; idx += stride) {
                const int64_t r = idx / f4;
                const int c = (int)(idx - r * f4) << 2;
                const float* q = x + r * ld + c;
                for (int k = 0; k < 4 && c + k < F; ++k) { const float v = g_ld(q + k); bad |= (v != v); }
            }
        }
        if (__any(bad) && (threadIdx.x & 63) == 0) const_cast<int32_t*>(b.nan_flags)[slot] = 1;
        return;
    }
    // ---- repack: one thread per pack element; the task table is read into LDS once per block
    // (searching it in global memory was ~6 dependent round trips per thread)
    constexpr int MAXPT = 2 * MMN_MAX_ENCODERS * MMN_MAX_LAYERS + MMN_MAX_ENCODERS + MMN_MAX_DECODERS +
                          2 * MMN_MAX_DECODERS * (MMN_MAX_DEC_HIDDEN + 1);
    __shared__ PackTask stk[MAXPT];
    const int npt = min(p.n_ptasks, MAXPT);
    {
        const int nw = npt * (int)(sizeof(PackTask) / 4);
        const int32_t* src = reinterpret_cast<const int32_t*>(p.ptasks);
        int32_t* dst = reinterpret_cast<int32_t*>(stk);
        for (int k = threadIdx.x; k < nw; k += NT) dst[k] = src[k];
    }
    __syncthreads();
    const int64_t g = (int64_t)(blockIdx.x - scan_blocks) * NT + threadIdx.x;
    if (g >= p.n_pack_elems) {                              // tail threads: gather the biases
        const int64_t j = g - p.n_pack_elems;
        if (j < p.n_bias) g_st(p.biasbuf + j, g_ld(p.bias_src[j]));
        return;
    }
    int lo = 0, hi = npt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (stk[mid].start <= g) lo = mid; else hi = mid - 1;
    }
    const PackTask tk = stk[lo];
    const int local = (int)(g - tk.start);
    const int c = local & 3, lane = (local >> 2) & 63, ft = local >> 8;
    const int ct = ft / tk.T, t = ft - ct * tk.T;
    const int n = 16 * ct + (lane & 15);
    const int kk = 16 * t + 4 * (lane >> 4) + c - tk.kk_off;
    if (n >= tk.N || kk < 0) return;
    const int len0p = round_up(tk.len0, 16);
    int col;
    if (kk < len0p) { if (kk >= tk.len0) return; col = tk.col0 + kk; }
    else { const int k1 = kk - len0p; if (k1 >= tk.len1) return; col = tk.col1 + k1; }
    const float v = tk.mode ? g_ld(tk.src + (int64_t)col * tk.ld_src + n) : g_ld(tk.src + (int64_t)n * tk.ld_src + col);
    g_st(tk.dst + local, v);
}

// ------------------------------------------------------------------------------------------------
// The chain kernels read the plan (model descriptor, buffer pointers, offsets) from an LDS copy:
// one coalesced global read at kernel start instead of a dependent global round trip at every
// "p.m.enc[e].layer[l].w" (pointer chase -> weights -> bias was 3 serial round trips per layer).
// ------------------------------------------------------------------------------------------------
typedef const MMN_AS3 DevPlan LPlan;
constexpr int PLAN_FLOATS = (int)((sizeof(DevPlan) + 15) / 16) * 4;

__device__ __forceinline__ void copy_plan_to_lds(const DevPlan* P, lp sPlan) {
    for (int idx = threadIdx.x; idx < PLAN_FLOATS / 4; idx += NT)
        lds_st4(sPlan + 4 * idx, g_ld4(reinterpret_cast<const float*>(P) + 4 * idx));
}

// ------------------------------------------------------------------------------------------------
// decoder evaluation of one state tile (used by k_chain_fwd)
// ------------------------------------------------------------------------------------------------
struct DecodeCtx {
    LPlan* p;
    lp sZ;
    f32x4 wd[4];           // this wave's decoder-weight fragments (constant for the whole kernel)
    float bd0, bd1;        // this thread's decoder bias pair
    int y;                 // this thread's target (row, d)
    int row0, nrows, tile;
    float cL;
    int want_grads;
};

__device__ __forceinline__ void load_decoder_frags(LPlan& p, f32x4 (&wd)[4]) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15, q = lane >> 4;
    const int S = p.S, Tdec = p.S16 >> 4, KD = (Tdec + 3) >> 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int t = wave * KD + j;
        if (j < KD && t < Tdec && i < 2 * p.D) {
            const float* w = p.m.dec[i >> 1].w + (i & 1) * S;
            const int k = 16 * t + 4 * q;
            if (k < S) v.x = g_ld(w + k);
            if (k + 1 < S) v.y = g_ld(w + k + 1);
            if (k + 2 < S) v.z = g_ld(w + k + 2);
            if (k + 3 < S) v.w = g_ld(w + k + 3);
        }
        wd[j] = v;
    }
}

template <int RT>
__device__ __forceinline__ void decode_state(const DecodeCtx& c, clp sS, int grid_row) {
    constexpr int TB = 16 * RT;
    LPlan& p = *c.p;
    const int ldS = p.ldS, D = p.D, R = p.R;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15, q = lane >> 4;
    const int Tdec = p.S16 >> 4, KD = (Tdec + 3) >> 2;
    // z[TB x 16] = sS[TB x S] * Wdec[16 x S]^T, contraction split over the four waves
    f32x4 z[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) z[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = wave * KD + j;
        if (j < KD && t < Tdec) {
            const f32x4 bb = c.wd[j];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const f32x4 a = lds_ld4(sS + (r * 16 + i) * ldS + 16 * t + 4 * q);
                z[r] = mfma4(a.x, bb.x, z[r]);
                z[r] = mfma4(a.y, bb.y, z[r]);
                z[r] = mfma4(a.z, bb.z, z[r]);
                z[r] = mfma4(a.w, bb.w, z[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) lds_st(c.sZ + (wave * TB + r * 16 + q * 4 + k) * 16 + i, z[r][k]);
    __syncthreads();
    const int t = threadIdx.x;
    const int row = t & (TB - 1), d = t / TB;
    float lossv = 0.f;
    int correct = 0, tp = 0, tn = 0, fp = 0, fn = 0;
    if (d < D && row < c.nrows) {
        float za = c.bd0, zb = c.bd1;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            za += lds_ld(c.sZ + (w * TB + row) * 16 + 2 * d);
            zb += lds_ld(c.sZ + (w * TB + row) * 16 + 2 * d + 1);
        }
        const int64_t grow = (int64_t)c.row0 + row;
        const int y = c.y;
        const float o0 = 1.0f / (1.0f + expf(-za));
        const float o1 = 1.0f / (1.0f + expf(-zb));
        const float mx = fmaxf(o0, o1);
        const float lse = mx + logf(expf(o0 - mx) + expf(o1 - mx));
        lossv = lse - (y ? o1 : o0);
        const int pred = o1 > o0 ? 1 : 0;          // torch.max: first index wins ties
        correct = pred == y;
        tp = pred & y; tn = (1 - pred) & (1 - y); fp = pred & (1 - y); fn = (1 - pred) & y;
        if (c.want_grads) {
            const float g0 = expf(o0 - lse) - (y == 0 ? 1.0f : 0.0f);
            const float g1 = expf(o1 - lse) - (y == 1 ? 1.0f : 0.0f);
            f32x2 dzv;
            dzv.x = c.cL * g0 * o0 * (1.0f - o0);
            dzv.y = c.cL * g1 * o1 * (1.0f - o1);
            g_st2(p.dz + ((int64_t)grid_row * p.maxB + grow) * (2 * D) + 2 * d, dzv);
        } else {                                    // forward-only: the decoder OUTPUTS take dz's place (test()/predict())
            f32x2 ov; ov.x = o0; ov.y = o1;
            g_st2(p.dz + ((int64_t)grid_row * p.maxB + grow) * (2 * D) + 2 * d, ov);
        }
    }
#pragma unroll
    for (int off = TB / 2; off >= 1; off >>= 1) lossv += __shfl_xor(lossv, off);
    const unsigned long long mc = __ballot(correct), mtp = __ballot(tp), mtn = __ballot(tn),
                             mfp = __ballot(fp), mfn = __ballot(fn);
    if (row == 0 && d < D) {
        const int sh = lane & ~(TB - 1);
        const unsigned long long msk = (TB == 32) ? 0xFFFFFFFFull : 0xFFFFull;
        const int64_t cell = (int64_t)c.tile * (R * D) + grid_row * D + d;
        g_st(p.lossp + cell, lossv);
        // per-tile integer partials, packed 5 x 8 bits... no: plain stores (contended atomics here
        // stalled every wave's next vmcnt wait by up to 14 us)
        int32_t* cp = p.cnt + cell * 5;
        g_sti(cp + 0, __popcll((mc >> sh) & msk));
        g_sti(cp + 1, __popcll((mtp >> sh) & msk));
        g_sti(cp + 2, __popcll((mtn >> sh) & msk));
        g_sti(cp + 3, __popcll((mfp >> sh) & msk));
        g_sti(cp + 4, __popcll((mfn >> sh) & msk));
    }
    __syncthreads();
}

// LDS carve of the two chain kernels (floats)
struct ChainLds { int sPlan, sS0, sS1, sDiff, sH0, sH1, sX, sZ, sRed, total; };
__host__ __device__ inline ChainLds chain_lds(int TB, int ldS, int ldH) {
    ChainLds L;
    int o = 0;
    L.sPlan = o; o += PLAN_FLOATS;
    L.sS0 = o; o += TB * ldS;
    L.sS1 = o; o += TB * ldS;
    L.sDiff = o; o += TB * ldS;
    L.sH0 = o; o += TB * ldH;
    L.sH1 = o; o += TB * ldH;
    L.sX = o; o += TB * LDX;
    L.sZ = o; o += 4 * TB * 16;      // also holds the dz tile (TB x LDZ) in the backward kernel
    L.sRed = o; o += 64;
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------------
// k_chain_fwd
// ------------------------------------------------------------------------------------------------
// Registers that hold one encoder's inputs ahead of time: its x tile, the weight fragments of up
// to two hidden layers (<= 4 k-steps each) and of the state update (<= TQ k-steps), and biases.
template <int NS> struct LayerRegs { f32x4 b[NS][2]; float bias[2]; };
template <int RT> struct EncRegs {
    f32x4 x[2 * RT];
    LayerRegs<4> h0, h1;
    LayerRegs<TQ> last;
};
struct EncInfo { int t, e, slot, Lh, F, HL, akind; bool fast; };

// next executed sequence position >= t (NaN-skipped slots are passed over), or n_seq
__device__ __forceinline__ int next_exec(const mmn_batch& b, int t) {
    while (t < b.n_seq && !slot_present(b, b.seq_data[t])) ++t;
    return t;
}

__device__ __forceinline__ EncInfo enc_info(LPlan& p, const mmn_batch& b, int t) {
    EncInfo I;
    I.t = t; I.slot = b.seq_data[t]; I.e = b.seq_enc[t];
    const auto& enc = p.m.enc[I.e];
    const int S = p.S;
    I.Lh = enc.n_layers - 1; I.F = enc.n_features; I.akind = enc.activation;
    I.HL = enc.layer[I.Lh].in_dim - S;
    // eligibility of the register-prefetch path (everything else takes the just-in-time path)
    const float* x = b.x[I.slot];
    bool ok = I.Lh <= 2 && I.F <= XCH && (I.F & 3) == 0 && (b.ldx[I.slot] & 3) == 0 &&
              (reinterpret_cast<uintptr_t>(x) & 15) == 0 && S <= 128 && (p.S16 + round_up(I.HL, 16)) <= 16 * TQ;
    for (int l = 0; l < I.Lh && l < 2; ++l) {
        const auto& lin = enc.layer[l];
        ok = ok && lin.in_dim <= 64 && lin.out_dim <= 128;
    }
    I.fast = ok;
    return I;
}

template <int RT>
__device__ __forceinline__ void issue_encoder(EncRegs<RT>& R, LPlan& p, const mmn_batch& b, const EncInfo& I, int row0,
                                              int nrows, bool with_last = true) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15;
    const auto& enc = p.m.enc[I.e];
    const int S = p.S;
    const int n0[2] = {16 * wave, 16 * (wave + 4)};
    // x tile [TB x 128]: thread -> (row, 4 columns), clamped + masked
    const float* xg = b.x[I.slot] + (int64_t)row0 * b.ldx[I.slot];
    const int64_t ldx = b.ldx[I.slot];
#pragma unroll
    for (int k = 0; k < 2 * RT; ++k) {
        const int idx = threadIdx.x + NT * k;
        const int row = idx >> 5, c = (idx & 31) << 2;
        const bool ok = row < nrows && c < I.F;
        const f32x4 v = g_ld4(xg + (ok ? (int64_t)row * ldx + c : 0));
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        R.x[k] = ok ? v : z;
    }
    if (I.Lh >= 1) {
        const auto& lin = enc.layer[0];
        const PB B = make_pb(p.pack + p.pkf_off[I.e][0], lin.out_dim, lin.in_dim, 0);
        issue_b<4>(R.h0.b, B, n0, 0);
#pragma unroll
        for (int c = 0; c < 2; ++c) R.h0.bias[c] = g_ld(lin.b + min(n0[c] + i, lin.out_dim - 1));
    }
    if (I.Lh >= 2) {
        const auto& lin = enc.layer[1];
        const PB B = make_pb(p.pack + p.pkf_off[I.e][1], lin.out_dim, lin.in_dim, 0);
        issue_b<4>(R.h1.b, B, n0, 0);
#pragma unroll
        for (int c = 0; c < 2; ++c) R.h1.bias[c] = g_ld(lin.b + min(n0[c] + i, lin.out_dim - 1));
    }
    {
        const auto& lin = enc.layer[I.Lh];
        const PB B = make_pb(p.pack + p.pkf_off[I.e][I.Lh], S, S, I.HL);
        if (with_last) issue_b<TQ>(R.last.b, B, n0, 0);
#pragma unroll
        for (int c = 0; c < 2; ++c) R.last.bias[c] = g_ld(lin.b + min(n0[c] + i, S - 1));
    }
}

template <int RT>
__global__ __launch_bounds__(NT) void k_chain_fwd(const DevPlan* __restrict__ P, mmn_batch b, float cL,
                                                  int want_grads) {
    constexpr int TB = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    // the carve needs ldS / ldH before the LDS copy of the plan exists: two scalar loads
    const int ldS = P->ldS, ldH = P->ldH;
    const ChainLds L = chain_lds(TB, ldS, ldH);
    copy_plan_to_lds(P, smem + L.sPlan);
    lp sS[2] = {smem + L.sS0, smem + L.sS1};
    lp sH[2] = {smem + L.sH0, smem + L.sH1};
    const lp sX = smem + L.sX;
    const lp sRed = smem + L.sRed;
    const int tile = blockIdx.x;
    const int row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15;
    int stamp_k = 0;
    const int stamp_block = 7;
    __syncthreads();
    LPlan& p = *(LPlan*)(smem + L.sPlan);
    const int S = p.S, E = p.E;
    STAMP();

    for (int idx = threadIdx.x; idx < TB * ldS; idx += NT) {
        const int k = idx % ldS;
        lds_st(sS[0] + idx, k < S ? g_ld(p.m.init_state + k) : 0.f);   // state.py:29-32 (tile, never materialised)
        lds_st(sS[1] + idx, 0.f);
    }
    for (int idx = threadIdx.x; idx < TB * ldH; idx += NT) { lds_st(sH[0] + idx, 0.f); lds_st(sH[1] + idx, 0.f); }
    for (int idx = threadIdx.x; idx < TB * LDX; idx += NT) lds_st(sX + idx, 0.f);
    if (tile == 0 && threadIdx.x == 0) {                   // which state rows exist this step
        g_sti(p.exec_flags, 1);
        for (int e = 0; e < E; ++e) g_sti(p.exec_flags + e + 1, 0);
        int prev = 0;
        for (int t = 0; t < b.n_seq; ++t) {
            if (!slot_present(b, b.seq_data[t])) continue;
            const int e = b.seq_enc[t];
            g_sti(p.exec_flags + e + 1, 1);
            g_sti(p.prev_row + e, prev);
            prev = e + 1;
        }
    }
    DecodeCtx dc;
    dc.p = &p; dc.sZ = smem + L.sZ; dc.row0 = row0; dc.nrows = nrows; dc.tile = tile; dc.cL = cL;
    dc.want_grads = want_grads;
    load_decoder_frags(p, dc.wd);
    {
        const int row = threadIdx.x & (TB - 1), d = threadIdx.x / TB;
        const bool ok = d < p.D && row < nrows;
        dc.y = ok ? (int)*(const MMN_AS1 int64_t*)(b.y + ((int64_t)row0 + row) * p.D + d) : 0;
        const float* bd = p.m.dec[ok ? d : 0].b;
        dc.bd0 = g_ld(bd); dc.bd1 = g_ld(bd + 1);
    }
    // prefetch the first encoder's inputs behind the decode of the init state
    EncRegs<RT> R;
    int tn = next_exec(b, 0);
    bool have = false;
    if (tn < b.n_seq) {
        const EncInfo I = enc_info(p, b, tn);
        if (I.fast) { issue_encoder<RT>(R, p, b, I, row0, nrows); have = true; }
    }
    __syncthreads();

    int cur = 0;
    STAMP();
    decode_state<RT>(dc, sS[cur], 0);
    STAMP();

    while (tn < b.n_seq) {
        const EncInfo I = enc_info(p, b, tn);
        const int e = I.e, Lh = I.Lh, F = I.F, HL = I.HL, akind = I.akind;
        const auto& enc = p.m.enc[e];
        const float* xg = b.x[I.slot] + (int64_t)row0 * b.ldx[I.slot];
        const int64_t ldx = b.ldx[I.slot];
        const int n0[2] = {16 * wave, 16 * (wave + 4)};
        const int t_next = next_exec(b, tn + 1);
        STAMP();
        float scacc = 0.f;
        const clp sC = sS[cur];
        const lp sN = sS[cur ^ 1];

        if (I.fast) {
            if (!have) issue_encoder<RT>(R, p, b, I, row0, nrows);   // previous encoder was not prefetchable
            // ---- x registers -> LDS image (whole [TB x 128] image, zero padded)
#pragma unroll
            for (int k = 0; k < 2 * RT; ++k) {
                const int idx = threadIdx.x + NT * k;
                lds_st4(sX + (idx >> 5) * LDX + ((idx & 31) << 2), R.x[k]);
            }
            __syncthreads();
            STAMP();   // F1: x in LDS
            // ---- hidden layers from prefetched fragments (mlp_encoder.py:75-76)
            if (Lh >= 1) {
                const auto& lin = enc.layer[0];
                const int N = lin.out_dim;
                const lp out = sH[(Lh - 1) & 1];
                if (n0[0] < N) {
                    f32x4 acc[2][RT];
                    zero_acc<RT>(acc);
                    const ASrc A{sX, LDX, sX, LDX};
                    const PB B = make_pb(p.pack + p.pkf_off[e][0], N, lin.in_dim, 0);
                    consume_b<RT, 4>(acc, A, B, R.h0.b, 0, B.T, n0[1] < N);
                    STAMP();   // F2: h0 consumed (only waves with work stamp; wave 0 always has)
                    run_epilogue<RT>(acc, n0, N, [&](int row, int col, int c, float v) {
                        if (col < N) lds_st(out + row * ldH + col, act_fwd(v + R.h0.bias[c], akind));
                    });
                }
                __syncthreads();
                STAMP();   // F3: h0 epilogue + sync
                if (want_grads) store_rows(p.hid + p.hid_off[e][0] + (int64_t)row0 * N, out, ldH, nrows, N);
                STAMP();   // F4: hid0 stored
            }
            if (Lh >= 2) {
                const auto& lin = enc.layer[1];
                const int N = lin.out_dim;
                const clp in = sH[1];
                const lp out = sH[0];
                if (n0[0] < N) {
                    f32x4 acc[2][RT];
                    zero_acc<RT>(acc);
                    const ASrc A{in, ldH, in, ldH};
                    const PB B = make_pb(p.pack + p.pkf_off[e][1], N, lin.in_dim, 0);
                    consume_b<RT, 4>(acc, A, B, R.h1.b, 0, B.T, n0[1] < N);
                    STAMP();   // F5: h1 consumed
                    run_epilogue<RT>(acc, n0, N, [&](int row, int col, int c, float v) {
                        if (col < N) lds_st(out + row * ldH + col, act_fwd(v + R.h1.bias[c], akind));
                    });
                }
                __syncthreads();
                if (want_grads) store_rows(p.hid + p.hid_off[e][1] + (int64_t)row0 * N, out, ldH, nrows, N);
            }
            STAMP();
            // ---- state update s' = W [h ; s] + b (mlp_encoder.py:78), state columns first
            {
                const auto& lin = enc.layer[Lh];
                f32x4 acc[2][RT];
                zero_acc<RT>(acc);
                const float bias0 = R.last.bias[0], bias1 = R.last.bias[1];
                // the NEXT encoder's state-update fragments are requested slot by slot underneath
                // this layer's MFMAs; its x tile / hidden fragments / biases right after
                EncInfo J = I;
                bool next_fast = false;
                if (t_next < b.n_seq) { J = enc_info(p, b, t_next); next_fast = J.fast; }
                const PB Bn = make_pb(p.pack + p.pkf_off[J.e][J.Lh], S, S, J.HL);
                if (n0[0] < S) {
                    const ASrc A{sC, ldS, Lh > 0 ? sH[0] : sX, Lh > 0 ? ldH : LDX};
                    const PB B = make_pb(p.pack + p.pkf_off[e][Lh], S, S, HL);
                    consume_b<RT, TQ>(acc, A, B, R.last.b, 0, B.T, n0[1] < S);
                }
                (void)Bn;
                STAMP();
                have = false;
                if (next_fast) { issue_encoder<RT>(R, p, b, J, row0, nrows, true); have = true; }
                STAMP();   // F8: next encoder issued
                if (n0[0] < S) {
                    run_epilogue<RT>(acc, n0, S, [&](int row, int col, int c, float v) {
                        if (col < S) {
                            const float ns = v + (c ? bias1 : bias0);
                            const float dlt = ns - lds_ld(sC + row * ldS + col);
                            if (row < nrows) scacc += dlt * dlt;              // multimodn.py:174
                            lds_st(sN + row * ldS + col, ns);
                        }
                    });
                }
            }
        } else {
            have = false;
            // ---- just-in-time path: any shape
            for (int l = 0; l < Lh; ++l) {
                const auto& lin = enc.layer[l];
                const int N = lin.out_dim;
                const lp out = sH[(Lh - 1 - l) & 1];
                const float* bias = lin.b;
                auto epi = [&](int row, int col, int, float v) {
                    if (col < N) lds_st(out + row * ldH + col, act_fwd(v + g_ld(bias + col), akind));
                };
                if (l == 0) {
                    const int ntiles = (N + 15) >> 4;
                    for (int base = 0; base < ntiles; base += 8) {
                        const int m0[2] = {16 * (base + wave), 16 * (base + wave + 4)};
                        f32x4 acc[2][RT];
                        zero_acc<RT>(acc);
                        const PB B = make_pb(p.pack + p.pkf_off[e][0], N, lin.in_dim, 0);
                        for (int xc = 0; xc < F; xc += XCH) {
                            const int kw = min(XCH, F - xc);
                            stage_rows(sX, LDX, xg + xc, ldx, nrows, TB, kw);
                            __syncthreads();
                            if (m0[0] < N) {
                                const ASrc A{sX - xc, LDX, sX - xc, LDX};      // step t reads image column 16 t - xc
                                wave_gemm_any<RT>(acc, A, B, m0, xc >> 4, (xc + kw + 15) >> 4);
                            }
                            __syncthreads();
                        }
                        if (m0[0] < N) run_epilogue<RT>(acc, m0, N, epi);
                    }
                } else {
                    const clp in = sH[(Lh - l) & 1];
                    const ASrc A{in, ldH, in, ldH};
                    const PB B = make_pb(p.pack + p.pkf_off[e][l], N, lin.in_dim, 0);
                    layer_nt<RT>(A, B, [&](int row, int col, float v) { epi(row, col, 0, v); });
                }
                __syncthreads();
                if (want_grads) store_rows(p.hid + p.hid_off[e][l] + (int64_t)row0 * N, out, ldH, nrows, N);
            }
            STAMP();
            {
                const auto& lin = enc.layer[Lh];
                const float* bias = lin.b;
                auto epi = [&](int row, int col, int, float v) {
                    if (col < S) {
                        const float ns = v + g_ld(bias + col);
                        const float dlt = ns - lds_ld(sC + row * ldS + col);
                        if (row < nrows) scacc += dlt * dlt;              // multimodn.py:174
                        lds_st(sN + row * ldS + col, ns);
                    }
                };
                if (Lh > 0) {
                    const ASrc A{sC, ldS, sH[0], ldH};
                    const PB B = make_pb(p.pack + p.pkf_off[e][Lh], S, S, HL);
                    layer_nt<RT>(A, B, [&](int row, int col, float v) { epi(row, col, 0, v); });
                } else {
                    const int ntiles = (S + 15) >> 4;
                    for (int base = 0; base < ntiles; base += 8) {
                        const int m0[2] = {16 * (base + wave), 16 * (base + wave + 4)};
                        f32x4 acc[2][RT];
                        zero_acc<RT>(acc);
                        const PB B = make_pb(p.pack + p.pkf_off[e][Lh], S, S, HL);   // HL == F here
                        if (m0[0] < S) {
                            const ASrc A{sC, ldS, sC, ldS};
                            wave_gemm_any<RT>(acc, A, B, m0, 0, B.T0);
                        }
                        for (int xc = 0; xc < F; xc += XCH) {
                            const int kw = min(XCH, F - xc);
                            stage_rows(sX, LDX, xg + xc, ldx, nrows, TB, kw);
                            __syncthreads();
                            if (m0[0] < S) {
                                const ASrc A{sC, ldS, sX - xc, LDX};
                                wave_gemm_any<RT>(acc, A, B, m0, B.T0 + (xc >> 4), B.T0 + ((xc + kw + 15) >> 4));
                            }
                            __syncthreads();
                        }
                        if (m0[0] < S) run_epilogue<RT>(acc, m0, S, epi);
                    }
                }
            }
            STAMP();
        }
        STAMP();
        scacc = wave_sum(scacc);
        if (lane == 0) lds_st(sRed + wave, scacc);
        __syncthreads();
        if (threadIdx.x == 0)
            g_st(p.scp + (int64_t)tile * E + e, ((lds_ld(sRed) + lds_ld(sRed + 1)) + lds_ld(sRed + 2)) + lds_ld(sRed + 3));
        store_rows(p.states + ((int64_t)e * p.maxB + row0) * S, sN, ldS, nrows, S);   // also forward-only: get_states()
        cur ^= 1;
        STAMP();
        decode_state<RT>(dc, sS[cur], e + 1);
        STAMP();
        tn = t_next;
    }
}

// ------------------------------------------------------------------------------------------------
// k_chain_bwd
// ------------------------------------------------------------------------------------------------
template <int TB>
__device__ __forceinline__ void load_dz_tile(LPlan& p, lp sDz, int grid_row, int row0, int nrows) {
    const int D2 = 2 * p.D;
    for (int idx = threadIdx.x; idx < TB * 16; idx += NT) {
        const int row = idx >> 4, n = idx & 15;
        float v = 0.f;
        if (row < nrows && n < D2) v = g_ld(p.dz + ((int64_t)grid_row * p.maxB + row0 + row) * D2 + n);
        lds_st(sDz + row * LDZ + n, v);
    }
}

// dpre = dh .* act'(h): sBuf (raw dh) -> sBuf and global dpre, h read from global hid (coalesced)
__device__ __forceinline__ void apply_act_grad(lp sBuf, int ld, const float* __restrict__ hid_g,
                                               float* __restrict__ dpre_g, int nrows, int rows_pad, int H, int akind) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    for (int r = wave; r < rows_pad; r += 4) {
        for (int c = lane; c < H; c += 64) {
            float dp = 0.f;
            if (r < nrows) {
                dp = lds_ld(sBuf + r * ld + c) * act_grad_from_out(g_ld(hid_g + (int64_t)r * H + c), akind);
                g_st(dpre_g + (int64_t)r * H + c, dp);
            }
            lds_st(sBuf + r * ld + c, dp);
        }
    }
}

template <int RT>
__global__ __launch_bounds__(NT) void k_chain_bwd(const DevPlan* __restrict__ P, mmn_batch b, float cS) {
    constexpr int TB = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    const int ldS = P->ldS, ldH = P->ldH;
    const ChainLds L = chain_lds(TB, ldS, ldH);
    copy_plan_to_lds(P, smem + L.sPlan);
    __syncthreads();
    LPlan& p = *(LPlan*)(smem + L.sPlan);
    const int S = p.S, E = p.E;
    lp sG[2] = {smem + L.sS0, smem + L.sS1};
    const lp sDiff = smem + L.sDiff;
    lp sH[2] = {smem + L.sH0, smem + L.sH1};
    const lp sDz = smem + L.sZ;
    const int tile = blockIdx.x;
    const int row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = wave_id();

    for (int idx = threadIdx.x; idx < TB * ldS; idx += NT) { lds_st(sG[0] + idx, 0.f); lds_st(sG[1] + idx, 0.f); lds_st(sDiff + idx, 0.f); }
    for (int idx = threadIdx.x; idx < TB * ldH; idx += NT) { lds_st(sH[0] + idx, 0.f); lds_st(sH[1] + idx, 0.f); }
    for (int idx = threadIdx.x; idx < TB * LDZ; idx += NT) lds_st(sDz + idx, 0.f);
    __syncthreads();

    const ASrc Adz{sDz, LDZ, sDz, LDZ};
    const PB Bdz = make_pb(p.pack + p.pkd_off, S, 2 * p.D, 0);        // W' = Wdec^T [S x 2D]
    int cur = 0;

    for (int t = b.n_seq - 1; t >= 0; --t) {
        const int slot = b.seq_data[t];
        if (!slot_present(b, slot)) continue;
        const int e = b.seq_enc[t];
        int tp = t - 1;
        while (tp >= 0 && !slot_present(b, b.seq_data[tp])) --tp;
        const int prev_row = tp >= 0 ? b.seq_enc[tp] + 1 : 0;
        const auto& enc = p.m.enc[e];
        const int nl = enc.n_layers, Lh = nl - 1;
        const int HL = enc.layer[nl - 1].in_dim - S;
        const int akind = enc.activation;
        const lp G = sG[cur];
        const lp Gn = sG[cur ^ 1];

        // diff = s_out - s_in (coalesced), dz tile of grid row e+1
        {
            const float* so = p.states + ((int64_t)e * p.maxB + row0) * S;
            const float* si = prev_row ? p.states + ((int64_t)(prev_row - 1) * p.maxB + row0) * S : nullptr;
            for (int r = wave; r < nrows; r += 4)
                for (int c = lane; c < S; c += 64) {
                    const float a = g_ld(so + (int64_t)r * S + c);
                    const float d = si ? g_ld(si + (int64_t)r * S + c) : g_ld(p.m.init_state + c);
                    lds_st(sDiff + r * ldS + c, a - d);
                }
            load_dz_tile<TB>(p, sDz, e + 1, row0, nrows);
        }
        __syncthreads();
        // G_out = carry + decoder grad of row e+1 + cS * diff
        layer_nt<RT>(Adz, Bdz, [&](int row, int col, float v) {
            if (col < S) lds_st(G + row * ldS + col, lds_ld(G + row * ldS + col) + v + cS * lds_ld(sDiff + row * ldS + col));
        });
        __syncthreads();
        store_rows(p.dS + ((int64_t)e * p.maxB + row0) * S, G, ldS, nrows, S);
        // [dh | carry] = G_out * W_last : W' = W_last^T [(HL+S) x S]; no grad flows to x (Lh == 0)
        {
            const ASrc A{G, ldS, G, ldS};
            const lp dh = sH[0];
            if (Lh > 0) {
                const PB Bh = make_pb(p.pack + p.pkh_off[e], HL, S, 0);
                layer_nt<RT>(A, Bh, [&](int row, int col, float v) {
                    if (col < HL) lds_st(dh + row * ldH + col, v);
                });
            }
            const PB Bc = make_pb(p.pack + p.pkb_off[e][nl - 1], S, S, 0);
            layer_nt<RT>(A, Bc, [&](int row, int col, float v) {
                if (col < S) lds_st(Gn + row * ldS + col, v - cS * lds_ld(sDiff + row * ldS + col));
            });
        }
        __syncthreads();
        // hidden layers, last to first: dpre_l = dh_l .* act'(h_l);  dh_{l-1} = dpre_l * W_l
        for (int l = Lh - 1; l >= 0; --l) {
            const int Hl = enc.layer[l].out_dim;
            const lp cbuf = sH[(Lh - 1 - l) & 1];
            apply_act_grad(cbuf, ldH, p.hid + p.hid_off[e][l] + (int64_t)row0 * Hl,
                           p.dpre + p.hid_off[e][l] + (int64_t)row0 * Hl, nrows, TB, Hl, akind);
            __syncthreads();
            if (l == 0) break;
            const int Hp = enc.layer[l].in_dim;
            const lp nbuf = sH[(Lh - l) & 1];
            const ASrc A{cbuf, ldH, cbuf, ldH};
            const PB B = make_pb(p.pack + p.pkb_off[e][l], Hp, Hl, 0);              // W_l^T [Hp x Hl]
            layer_nt<RT>(A, B, [&](int row, int col, float v) {
                if (col < Hp) lds_st(nbuf + row * ldH + col, v);
            });
            __syncthreads();
        }
        cur ^= 1;
    }
    // row 0: decoders on the init state; dS0 = d loss / d tiled init state
    const lp G = sG[cur];
    load_dz_tile<TB>(p, sDz, 0, row0, nrows);
    __syncthreads();
    layer_nt<RT>(Adz, Bdz, [&](int row, int col, float v) {
        if (col < S) lds_st(G + row * ldS + col, lds_ld(G + row * ldS + col) + v);
    });
    __syncthreads();
    store_rows(p.dS + ((int64_t)E * p.maxB + row0) * S, G, ldS, nrows, S);
}

// ------------------------------------------------------------------------------------------------
// Generic tier: k_gen_fwd / k_gen_bwd.  Any mix of MLPEncoder and MIMIC_MLPEncoder
// (multimodn/encoders/mlp_encoder.py:9-47: Dropout(cat[x, state]) -> Linear+act ... -> Linear+act =
// new state) with ClassDecoder and MLPDecoder heads (multimodn/decoders/decoders.py:22-46: hidden
// Linear+act layers, then sigmoid(Linear -> 2)).  Sequential form of the chain, one 16- or 32-row
// tile per workgroup; every product is an LDS activation tile times fragment-ordered weights, like
// the sequential tier above.  With a MIMIC encoder nothing of the encoder is independent of the
// state (it enters the FIRST layer), so the re-association the 8-wave kernels live on does not exist
// here: every layer of every encoder is on the dependent chain.
// What the backward half and k_wgrad read is written by the forward half when want_grads:
//   states[e], hid[e][l] (outputs of the non-final layers), gact.xin[e] = the (masked) cat[x, state]
//   a MIMIC encoder's first Linear saw, gact.dhid[r][d][l] = decoder d's hidden activations on grid
//   row r, dz[r]; the backward half adds dS[e] (MIMIC: d loss / d PRE-activation of the state layer),
//   dpre[e][l], gdpre[r][d][l].
// ------------------------------------------------------------------------------------------------
// copy a global [nrows x ncols] tile times an optional multiplier tile into an LDS image (zero padded)
__device__ __forceinline__ void stage_rows_mul(lp dst, int ld_dst, const float* __restrict__ src, int64_t ld_src,
                                               const float* __restrict__ mul, int64_t ld_mul, int nrows, int rows_pad,
                                               int ncols) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int cpad = round_up(ncols, 16);
    for (int r = wave; r < rows_pad; r += 4)
        for (int c = lane; c < cpad; c += 64) {
            float v = 0.f;
            if (r < nrows && c < ncols) {
                v = g_ld(src + (int64_t)r * ld_src + c);
                if (mul) v *= g_ld(mul + (int64_t)r * ld_mul + c);
            }
            lds_st(dst + r * ld_dst + c, v);
        }
}

// out[rows x N] = [sState (B.T0 k-steps, may be 0)] ++ [x columns streamed from global, XCH at a time] times W'^T
template <int RT, class Epi>
__device__ __forceinline__ void layer_state_x(clp sState, int ldS, const PB& B, lp sX, const float* __restrict__ xg,
                                              int64_t ldx, const float* __restrict__ mg, int64_t ldm, int F, int nrows,
                                              Epi&& epi) {
    constexpr int TB = 16 * RT;
    const int wave = wave_id();
    const int ntiles = (B.N + 15) >> 4;
    for (int base = 0; base < ntiles; base += 8) {
        const int m0[2] = {16 * (base + wave), 16 * (base + wave + 4)};
        f32x4 acc[2][RT];
        zero_acc<RT>(acc);
        if (B.T0 > 0 && m0[0] < B.N) {
            const ASrc A{sState, ldS, sState, ldS};
            wave_gemm_any<RT>(acc, A, B, m0, 0, B.T0);
        }
        for (int xc = 0; xc < F; xc += XCH) {
            const int kw = min(XCH, F - xc);
            if (mg) stage_rows_mul(sX, LDX, xg + xc, ldx, mg + xc, ldm, nrows, TB, kw);
            else stage_rows(sX, LDX, xg + xc, ldx, nrows, TB, kw);
            __syncthreads();
            if (m0[0] < B.N) {
                const ASrc A{sState, ldS, sX - xc, LDX};       // step t reads image column 16 (t - T0) - xc
                wave_gemm_any<RT>(acc, A, B, m0, B.T0 + (xc >> 4), B.T0 + ((xc + kw + 15) >> 4));
            }
            __syncthreads();
        }
        if (m0[0] < B.N) run_epilogue<RT>(acc, m0, B.N, epi);
    }
}

// out[rows x N] = A * W'^T with BOTH operands in LDS (W' = a fragment-ordered pack copied from global memory once
// per workgroup): one column tile per wave at a time, no global round trip
template <int RT, class Epi>
__device__ __forceinline__ void layer_nt_l(const ASrc& A, clp wl, int N, int len0, int len1, Epi&& epi) {
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    const int T0 = (len0 + 15) >> 4, T = T0 + ((len1 + 15) >> 4);
    const int ntiles = (N + 15) >> 4;
    for (int tile = wave; tile < ntiles; tile += 4) {
        f32x4 acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        clp bp = wl + ((int64_t)tile * T * 64 + lane) * 4;
        for (int t = 0; t < T; ++t) {
            const f32x4 bb = lds_ld4(bp + t * 256);
            const int lda = t < T0 ? A.lda0 : A.lda1;
            clp ap = (t < T0 ? A.a0 + 16 * t : A.a1 + 16 * (t - T0)) + 4 * q;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const f32x4 a = lds_ld4(ap + (r * 16 + i) * lda);
                acc[r] = mfma4(a.x, bb.x, acc[r]);
                acc[r] = mfma4(a.y, bb.y, acc[r]);
                acc[r] = mfma4(a.z, bb.z, acc[r]);
                acc[r] = mfma4(a.w, bb.w, acc[r]);
            }
        }
        const int col = 16 * tile + i;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) epi(r * 16 + q * 4 + k, col, acc[r][k]);
    }
}

// a decoder layer: operands from the LDS copy when the plan says they fit, else from global memory
template <int RT, class Epi>
__device__ __forceinline__ void dec_layer(LPlan& p, clp sW, int64_t pk_off, int64_t region_off, const ASrc& A, int N, int K,
                                          Epi&& epi) {
    if (p.dec_lds) {
        layer_nt_l<RT>(A, sW + (pk_off - region_off), N, K, 0, epi);
    } else {
        const PB B = make_pb(p.pack + pk_off, N, K, 0);
        layer_nt<RT>(A, B, epi);
    }
}

// (eight independent 16-byte requests per thread in flight, then their stores: a plain load -> store loop waits
//  for every load in turn, ~16 dependent round trips for the decoders' 66 KB at the MIMIC shape)
__device__ __forceinline__ void copy_pack_to_lds(lp dst, const float* __restrict__ src, int nfloats) {
    for (int base = threadIdx.x * 4; base < nfloats; base += NT * 4 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = base + k * NT * 4;
            v[k] = g_ld4(src + (idx < nfloats ? idx : 0));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = base + k * NT * 4;
            if (idx < nfloats) lds_st4(dst + idx, v[k]);
        }
    }
}

// LDS tile [nrows x ncols] -> global rows of stride ld_dst
__device__ __forceinline__ void store_rows_ld(float* __restrict__ dst, int64_t ld_dst, clp src, int ld_src, int nrows, int ncols) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    for (int r = wave; r < nrows; r += 4)
        for (int c = lane; c < ncols; c += 64) g_st(dst + (int64_t)r * ld_dst + c, lds_ld(src + r * ld_src + c));
}

// apply_act_grad with the activations / pre-activation gradients in global rows of stride ldg
__device__ __forceinline__ void apply_act_grad_ld(lp sBuf, int ld, const float* __restrict__ hid_g, float* __restrict__ dpre_g,
                                                  int64_t ldg, int nrows, int rows_pad, int H, int akind) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    for (int r = wave; r < rows_pad; r += 4)
        for (int c = lane; c < H; c += 64) {
            float dp = 0.f;
            if (r < nrows) {
                dp = lds_ld(sBuf + r * ld + c) * act_grad_from_out(g_ld(hid_g + (int64_t)r * ldg + c), akind);
                g_st(dpre_g + (int64_t)r * ldg + c, dp);
            }
            lds_st(sBuf + r * ld + c, dp);
        }
}

struct GenDecodeCtx {
    LPlan* p;
    lp sZ; lp sH0; lp sH1;
    clp sW;                // LDS copy of the decoders' forward operands (p->dec_lds)
    int y;                 // this thread's target (row, d)
    int row0, nrows, tile;
    float cL;
    int want_grads;
};

// all D decoders on one state tile: hidden layers (decoders.py:42-43), output Linear + sigmoid (:44-45),
// CrossEntropy over the sigmoid outputs, argmax, confusion counts (multimodn.py:141-157,176-191)
template <int RT>
__device__ __forceinline__ void gen_decode(const GenDecodeCtx& c, clp sS, int grid_row) {
    constexpr int TB = 16 * RT;
    LPlan& p = *c.p;
    const int ldS = p.ldS, ldH = p.ldH, D = p.D, R = p.R, S = p.S;
    const int lane = threadIdx.x & 63;
    for (int d = 0; d < D; ++d) {
        const auto& dec = p.m.dec[d];
        const int nh = dec.n_hidden, hk = dec.hidden_activation;
        clp in = sS; int ldi = ldS, K = S;
        for (int l = 0; l < nh; ++l) {
            const auto& lin = dec.hidden[l];
            const int N = lin.out_dim;
            const lp out = (l & 1) ? c.sH1 : c.sH0;
            const float* bias = lin.b;
            const ASrc A{in, ldi, in, ldi};
            dec_layer<RT>(p, c.sW, p.pkdf_off[d][l], p.dec_f_off, A, N, K, [&](int row, int col, float v) {
                if (col < N) lds_st(out + row * ldH + col, act_fwd(v + g_ld(bias + col), hk));
            });
            __syncthreads();
            if (c.want_grads)
                store_rows_ld(p.gact + p.dh_base + (int64_t)grid_row * p.dh_row_stride + (int64_t)c.row0 * p.dcols + p.dh_off[d][l],
                              p.dcols, out, ldH, c.nrows, N);
            in = out; ldi = ldH; K = N;
        }
        const float* bias = dec.b;
        const ASrc A{in, ldi, in, ldi};
        dec_layer<RT>(p, c.sW, p.pkdf_off[d][nh], p.dec_f_off, A, 2, K, [&](int row, int col, float v) {
            if (col < 2) lds_st(c.sZ + row * 16 + 2 * d + col, v + g_ld(bias + col));
        });
        __syncthreads();
    }
    const int t = threadIdx.x;
    const int row = t & (TB - 1), d = t / TB;
    float lossv = 0.f;
    int correct = 0, tp = 0, tn = 0, fp = 0, fn = 0;
    if (d < D && row < c.nrows) {
        const float za = lds_ld(c.sZ + row * 16 + 2 * d), zb = lds_ld(c.sZ + row * 16 + 2 * d + 1);
        const int64_t grow = (int64_t)c.row0 + row;
        const int y = c.y;
        const float o0 = 1.0f / (1.0f + expf(-za));
        const float o1 = 1.0f / (1.0f + expf(-zb));
        const float mx = fmaxf(o0, o1);
        const float lse = mx + logf(expf(o0 - mx) + expf(o1 - mx));
        lossv = lse - (y ? o1 : o0);
        const int pred = o1 > o0 ? 1 : 0;          // torch.max: first index wins ties
        correct = pred == y;
        tp = pred & y; tn = (1 - pred) & (1 - y); fp = pred & (1 - y); fn = (1 - pred) & y;
        f32x2 ov;
        if (c.want_grads) {
            const float g0 = expf(o0 - lse) - (y == 0 ? 1.0f : 0.0f);
            const float g1 = expf(o1 - lse) - (y == 1 ? 1.0f : 0.0f);
            ov.x = c.cL * g0 * o0 * (1.0f - o0);
            ov.y = c.cL * g1 * o1 * (1.0f - o1);
        } else {                                    // forward-only: the decoder OUTPUTS take dz's place (test()/predict())
            ov.x = o0; ov.y = o1;
        }
        g_st2(p.dz + ((int64_t)grid_row * p.maxB + grow) * (2 * D) + 2 * d, ov);
    }
#pragma unroll
    for (int off = TB / 2; off >= 1; off >>= 1) lossv += __shfl_xor(lossv, off);
    const unsigned long long mc = __ballot(correct), mtp = __ballot(tp), mtn = __ballot(tn),
                             mfp = __ballot(fp), mfn = __ballot(fn);
    if (row == 0 && d < D) {
        const int sh = lane & ~(TB - 1);
        const unsigned long long msk = (TB == 32) ? 0xFFFFFFFFull : 0xFFFFull;
        const int64_t cell = (int64_t)c.tile * (R * D) + grid_row * D + d;
        g_st(p.lossp + cell, lossv);
        int32_t* cp = p.cnt + cell * 5;
        g_sti(cp + 0, __popcll((mc >> sh) & msk));
        g_sti(cp + 1, __popcll((mtp >> sh) & msk));
        g_sti(cp + 2, __popcll((mtn >> sh) & msk));
        g_sti(cp + 3, __popcll((mfp >> sh) & msk));
        g_sti(cp + 4, __popcll((mfn >> sh) & msk));
    }
    __syncthreads();
}

template <int RT>
__global__ __launch_bounds__(NT) void k_gen_fwd(const DevPlan* __restrict__ P, mmn_batch b, float cL, int want_grads) {
    constexpr int TB = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    const int ldS = P->ldS, ldH = P->ldH;
    const ChainLds L = chain_lds(TB, ldS, ldH);
    copy_plan_to_lds(P, smem + L.sPlan);
    lp sS[2] = {smem + L.sS0, smem + L.sS1};
    const lp sM = smem + L.sDiff;                          // masked state tile (MIMIC encoders under dropout)
    lp sH[2] = {smem + L.sH0, smem + L.sH1};
    const lp sX = smem + L.sX;
    const lp sRed = smem + L.sRed;
    const int tile = blockIdx.x;
    const int row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = wave_id();
    __syncthreads();
    LPlan& p = *(LPlan*)(smem + L.sPlan);
    const int S = p.S, E = p.E;

    for (int idx = threadIdx.x; idx < TB * ldS; idx += NT) {
        const int k = idx % ldS;
        lds_st(sS[0] + idx, k < S ? g_ld(p.m.init_state + k) : 0.f);   // state.py:29-32 (tile, never materialised)
        lds_st(sS[1] + idx, 0.f);
        lds_st(sM + idx, 0.f);
    }
    for (int idx = threadIdx.x; idx < TB * ldH; idx += NT) { lds_st(sH[0] + idx, 0.f); lds_st(sH[1] + idx, 0.f); }
    for (int idx = threadIdx.x; idx < TB * LDX; idx += NT) lds_st(sX + idx, 0.f);
    for (int idx = threadIdx.x; idx < 4 * TB * 16; idx += NT) lds_st(smem + L.sZ + idx, 0.f);
    if (tile == 0 && threadIdx.x == 0) {                   // which state rows exist this step
        g_sti(p.exec_flags, 1);
        for (int e = 0; e < E; ++e) g_sti(p.exec_flags + e + 1, 0);
        int prev = 0;
        for (int t = 0; t < b.n_seq; ++t) {
            if (!slot_present(b, b.seq_data[t])) continue;
            const int e = b.seq_enc[t];
            g_sti(p.exec_flags + e + 1, 1);
            g_sti(p.prev_row + e, prev);
            prev = e + 1;
        }
    }
    GenDecodeCtx dc;
    dc.p = &p; dc.sZ = smem + L.sZ; dc.sH0 = sH[0]; dc.sH1 = sH[1];
    dc.sW = smem + L.total;
    if (p.dec_lds) copy_pack_to_lds(smem + L.total, p.pack + p.dec_f_off, p.dec_f_floats);
    dc.row0 = row0; dc.nrows = nrows; dc.tile = tile; dc.cL = cL; dc.want_grads = want_grads;
    {
        const int row = threadIdx.x & (TB - 1), d = threadIdx.x / TB;
        const bool ok = d < p.D && row < nrows;
        dc.y = ok ? (int)*(const MMN_AS1 int64_t*)(b.y + ((int64_t)row0 + row) * p.D + d) : 0;
    }
    __syncthreads();

    int cur = 0;
    gen_decode<RT>(dc, sS[cur], 0);

    for (int tn = next_exec(b, 0); tn < b.n_seq; tn = next_exec(b, tn + 1)) {
        const int slot = b.seq_data[tn], e = b.seq_enc[tn];
        const auto& enc = p.m.enc[e];
        const int nl = enc.n_layers, F = enc.n_features, akind = enc.activation;
        const float* xg = b.x[slot] + (int64_t)row0 * b.ldx[slot];
        const int64_t ldx = b.ldx[slot];
        float scacc = 0.f;
        const clp sC = sS[cur];
        const lp sN = sS[cur ^ 1];
        // epilogue of the layer that produces the new state (act_out: MIMIC applies the activation there too)
        auto state_epi = [&](const float* bias, int act_out) {
            return [&, bias, act_out](int row, int col, int, float v) {
                if (col < S) {
                    const float ns = act_fwd(v + g_ld(bias + col), act_out);
                    const float dlt = ns - lds_ld(sC + row * ldS + col);
                    if (row < nrows) scacc += dlt * dlt;              // multimodn.py:174
                    lds_st(sN + row * ldS + col, ns);
                }
            };
        };
        if (enc.kind == MMN_ENC_MIMIC) {
            // ---- layer 0 reads Dropout(cat[x, state]) (mlp_encoder.py:40-41)
            const int FS = F + S;
            const float* mk = b.drop_mask[e] ? b.drop_mask[e] + (int64_t)row0 * FS : nullptr;
            clp sIn = sC;
            if (mk || want_grads) {
                float* xin = p.gact + p.xin_off[e] + (int64_t)row0 * FS;
                for (int r = wave; r < TB; r += 4)
                    for (int c = lane; c < FS; c += 64) {
                        float v = 0.f;
                        if (r < nrows) {
                            v = c < F ? g_ld(xg + (int64_t)r * ldx + c) : lds_ld(sC + r * ldS + (c - F));
                            if (mk) v *= g_ld(mk + (int64_t)r * FS + c);
                            if (want_grads) g_st(xin + (int64_t)r * FS + c, v);
                        }
                        if (mk && c >= F) lds_st(sM + r * ldS + (c - F), v);
                    }
                if (mk) sIn = sM;
                __syncthreads();
            }
            for (int l = 0; l < nl; ++l) {
                const auto& lin = enc.layer[l];
                const int N = lin.out_dim;
                const bool last = l == nl - 1;
                const lp out = sH[l & 1];
                const float* bias = lin.b;
                auto hid_epi = [&](int row, int col, int, float v) {
                    if (col < N) lds_st(out + row * ldH + col, act_fwd(v + g_ld(bias + col), akind));
                };
                if (l == 0) {
                    const PB B = make_pb(p.pack + p.pkf_off[e][0], N, S, F);           // state columns first in the pack
                    if (last) layer_state_x<RT>(sIn, ldS, B, sX, xg, ldx, mk, FS, F, nrows, state_epi(bias, akind));
                    else layer_state_x<RT>(sIn, ldS, B, sX, xg, ldx, mk, FS, F, nrows, hid_epi);
                } else {
                    const clp in = sH[(l - 1) & 1];
                    const ASrc A{in, ldH, in, ldH};
                    const PB B = make_pb(p.pack + p.pkf_off[e][l], N, lin.in_dim, 0);
                    if (last) {
                        auto se = state_epi(bias, akind);
                        layer_nt<RT>(A, B, [&](int row, int col, float v) { se(row, col, 0, v); });
                    } else {
                        layer_nt<RT>(A, B, [&](int row, int col, float v) { hid_epi(row, col, 0, v); });
                    }
                }
                __syncthreads();
                if (!last && want_grads) store_rows(p.hid + p.hid_off[e][l] + (int64_t)row0 * N, out, ldH, nrows, N);
            }
        } else {
            // ---- MLPEncoder (mlp_encoder.py:74-80): hidden layers on x only, then Linear(cat[h, state])
            const int Lh = nl - 1;
            const int HL = enc.layer[Lh].in_dim - S;
            for (int l = 0; l < Lh; ++l) {
                const auto& lin = enc.layer[l];
                const int N = lin.out_dim;
                const lp out = sH[(Lh - 1 - l) & 1];
                const float* bias = lin.b;
                auto epi = [&](int row, int col, int, float v) {
                    if (col < N) lds_st(out + row * ldH + col, act_fwd(v + g_ld(bias + col), akind));
                };
                if (l == 0) {
                    const PB B = make_pb(p.pack + p.pkf_off[e][0], N, 0, lin.in_dim);
                    layer_state_x<RT>(sC, ldS, B, sX, xg, ldx, nullptr, 0, F, nrows, epi);
                } else {
                    const clp in = sH[(Lh - l) & 1];
                    const ASrc A{in, ldH, in, ldH};
                    const PB B = make_pb(p.pack + p.pkf_off[e][l], N, lin.in_dim, 0);
                    layer_nt<RT>(A, B, [&](int row, int col, float v) { epi(row, col, 0, v); });
                }
                __syncthreads();
                if (want_grads) store_rows(p.hid + p.hid_off[e][l] + (int64_t)row0 * N, out, ldH, nrows, N);
            }
            const float* bias = enc.layer[Lh].b;
            const PB B = make_pb(p.pack + p.pkf_off[e][Lh], S, S, HL);
            if (Lh > 0) {
                const ASrc A{sC, ldS, sH[0], ldH};
                auto se = state_epi(bias, MMN_ACT_IDENTITY);
                layer_nt<RT>(A, B, [&](int row, int col, float v) { se(row, col, 0, v); });
            } else {
                layer_state_x<RT>(sC, ldS, B, sX, xg, ldx, nullptr, 0, F, nrows, state_epi(bias, MMN_ACT_IDENTITY));
            }
        }
        scacc = wave_sum(scacc);
        if (lane == 0) lds_st(sRed + wave, scacc);
        __syncthreads();
        if (threadIdx.x == 0)
            g_st(p.scp + (int64_t)tile * E + e, ((lds_ld(sRed) + lds_ld(sRed + 1)) + lds_ld(sRed + 2)) + lds_ld(sRed + 3));
        store_rows(p.states + ((int64_t)e * p.maxB + row0) * S, sN, ldS, nrows, S);   // also forward-only: get_states()
        cur ^= 1;
        gen_decode<RT>(dc, sS[cur], e + 1);
    }
}

// G += d loss / d state through all D decoders of one grid row; stores the hidden layers' dpre
template <int RT>
__device__ __forceinline__ void gen_decoder_back(LPlan& p, clp sW, lp G, lp sDz, lp sDd, lp sH0, lp sH1, int grid_row,
                                                 int row0, int nrows) {
    constexpr int TB = 16 * RT;
    const int ldS = p.ldS, ldH = p.ldH, S = p.S, D = p.D;
    load_dz_tile<TB>(p, sDz, grid_row, row0, nrows);
    __syncthreads();
    for (int d = 0; d < D; ++d) {
        const auto& dec = p.m.dec[d];
        const int nh = dec.n_hidden, hk = dec.hidden_activation;
        for (int idx = threadIdx.x; idx < TB * 2; idx += NT) {        // this decoder's dz pair -> columns 0, 1 of its own tile
            const int row = idx >> 1, c = idx & 1;
            lds_st(sDd + row * LDZ + c, lds_ld(sDz + row * LDZ + 2 * d + c));
        }
        __syncthreads();
        const ASrc Ad{sDd, LDZ, sDd, LDZ};
        if (nh == 0) {
            dec_layer<RT>(p, sW, p.pkdb_off[d][0], p.dec_b_off, Ad, S, 2, [&](int row, int col, float v) {   // W_f^T [S x 2]
                if (col < S) lds_st(G + row * ldS + col, lds_ld(G + row * ldS + col) + v);
            });
            __syncthreads();
            continue;
        }
        lp cbuf = sH0, nbuf = sH1;
        {
            const int Hl = dec.hidden[nh - 1].out_dim;
            dec_layer<RT>(p, sW, p.pkdb_off[d][nh], p.dec_b_off, Ad, Hl, 2, [&](int row, int col, float v) {   // W_f^T [H_last x 2]
                if (col < Hl) lds_st(cbuf + row * ldH + col, v);
            });
            __syncthreads();
        }
        for (int l = nh - 1; l >= 0; --l) {
            const int Hl = dec.hidden[l].out_dim, Kin = dec.hidden[l].in_dim;
            const int64_t off = (int64_t)grid_row * p.dh_row_stride + (int64_t)row0 * p.dcols + p.dh_off[d][l];
            apply_act_grad_ld(cbuf, ldH, p.gact + p.dh_base + off, p.gdpre + off, p.dcols, nrows, TB, Hl, hk);
            __syncthreads();
            const ASrc A{cbuf, ldH, cbuf, ldH};
            if (l == 0) {                                                             // W_l^T [in x out]
                dec_layer<RT>(p, sW, p.pkdb_off[d][l], p.dec_b_off, A, Kin, Hl, [&](int row, int col, float v) {
                    if (col < S) lds_st(G + row * ldS + col, lds_ld(G + row * ldS + col) + v);
                });
            } else {
                dec_layer<RT>(p, sW, p.pkdb_off[d][l], p.dec_b_off, A, Kin, Hl, [&](int row, int col, float v) {
                    if (col < Kin) lds_st(nbuf + row * ldH + col, v);
                });
                const lp t = cbuf; cbuf = nbuf; nbuf = t;
            }
            __syncthreads();
        }
    }
}

template <int RT>
__global__ __launch_bounds__(NT) void k_gen_bwd(const DevPlan* __restrict__ P, mmn_batch b, float cS) {
    constexpr int TB = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    const int ldS = P->ldS, ldH = P->ldH;
    const ChainLds L = chain_lds(TB, ldS, ldH);
    copy_plan_to_lds(P, smem + L.sPlan);
    __syncthreads();
    LPlan& p = *(LPlan*)(smem + L.sPlan);
    const int S = p.S, E = p.E;
    lp sG[2] = {smem + L.sS0, smem + L.sS1};
    const lp sDiff = smem + L.sDiff;
    lp sH[2] = {smem + L.sH0, smem + L.sH1};
    const lp sDz = smem + L.sZ;
    const lp sDd = smem + L.sX;                            // one decoder's dz pair, [TB x LDZ]
    const int tile = blockIdx.x;
    const int row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = wave_id();

    for (int idx = threadIdx.x; idx < TB * ldS; idx += NT) { lds_st(sG[0] + idx, 0.f); lds_st(sG[1] + idx, 0.f); lds_st(sDiff + idx, 0.f); }
    for (int idx = threadIdx.x; idx < TB * ldH; idx += NT) { lds_st(sH[0] + idx, 0.f); lds_st(sH[1] + idx, 0.f); }
    for (int idx = threadIdx.x; idx < 4 * TB * 16; idx += NT) lds_st(sDz + idx, 0.f);
    for (int idx = threadIdx.x; idx < TB * LDX; idx += NT) lds_st(sDd + idx, 0.f);
    const clp sW = smem + L.total;                         // LDS copy of the decoders' backward operands
    if (p.dec_lds) copy_pack_to_lds(smem + L.total, p.pack + p.dec_b_off, p.dec_b_floats);
    __syncthreads();
    int cur = 0;

    for (int t = b.n_seq - 1; t >= 0; --t) {
        const int slot = b.seq_data[t];
        if (!slot_present(b, slot)) continue;
        const int e = b.seq_enc[t];
        int tp = t - 1;
        while (tp >= 0 && !slot_present(b, b.seq_data[tp])) --tp;
        const int prev_row = tp >= 0 ? b.seq_enc[tp] + 1 : 0;
        const auto& enc = p.m.enc[e];
        const int nl = enc.n_layers, F = enc.n_features, akind = enc.activation;
        const lp G = sG[cur];
        const lp Gn = sG[cur ^ 1];
        // diff = s_out - s_in
        {
            const float* so = p.states + ((int64_t)e * p.maxB + row0) * S;
            const float* si = prev_row ? p.states + ((int64_t)(prev_row - 1) * p.maxB + row0) * S : nullptr;
            for (int r = wave; r < nrows; r += 4)
                for (int c = lane; c < S; c += 64) {
                    const float a = g_ld(so + (int64_t)r * S + c);
                    const float d = si ? g_ld(si + (int64_t)r * S + c) : g_ld(p.m.init_state + c);
                    lds_st(sDiff + r * ldS + c, a - d);
                }
        }
        // G_out = carry + decoder grads of row e+1 + cS * diff
        gen_decoder_back<RT>(p, sW, G, sDz, sDd, sH[0], sH[1], e + 1, row0, nrows);
        for (int r = wave; r < TB; r += 4)
            for (int c = lane; c < S; c += 64) lds_st(G + r * ldS + c, lds_ld(G + r * ldS + c) + cS * lds_ld(sDiff + r * ldS + c));
        __syncthreads();
        if (enc.kind == MMN_ENC_MIMIC) {
            const int FS = F + S;
            const float* mk = b.drop_mask[e] ? b.drop_mask[e] + (int64_t)row0 * FS + F : nullptr;   // the state columns
            // the state layer carries the activation too: dpre_last = G .* act'(s_out); that is what k_wgrad multiplies
            apply_act_grad(G, ldS, p.states + ((int64_t)e * p.maxB + row0) * S, p.dS + ((int64_t)e * p.maxB + row0) * S,
                           nrows, TB, S, akind);
            __syncthreads();
            clp cbuf = G; int ldc = ldS;
            for (int l = nl - 1; l >= 1; --l) {
                const int Hl = enc.layer[l].out_dim, Hp = enc.layer[l].in_dim;
                const lp nbuf = sH[l & 1];
                const ASrc A{cbuf, ldc, cbuf, ldc};
                const PB B = make_pb(p.pack + p.pkb_off[e][l], Hp, Hl, 0);            // W_l^T [in x out]
                layer_nt<RT>(A, B, [&](int row, int col, float v) {
                    if (col < Hp) lds_st(nbuf + row * ldH + col, v);
                });
                __syncthreads();
                apply_act_grad(nbuf, ldH, p.hid + p.hid_off[e][l - 1] + (int64_t)row0 * Hp,
                               p.dpre + p.hid_off[e][l - 1] + (int64_t)row0 * Hp, nrows, TB, Hp, akind);
                __syncthreads();
                cbuf = nbuf; ldc = ldH;
            }
            // carry = (dpre_0 * W_0[:, F:F+S]) .* mask_state - cS * diff ; no grad flows to x
            const int H0 = enc.layer[0].out_dim;
            const ASrc A{cbuf, ldc, cbuf, ldc};
            const PB B = make_pb(p.pack + p.pkb_off[e][0], S, H0, 0);
            layer_nt<RT>(A, B, [&](int row, int col, float v) {
                if (col < S) {
                    const float m = (mk && row < nrows) ? g_ld(mk + (int64_t)row * FS + col) : 1.0f;
                    lds_st(Gn + row * ldS + col, v * m - cS * lds_ld(sDiff + row * ldS + col));
                }
            });
            __syncthreads();
        } else {
            const int Lh = nl - 1;
            const int HL = enc.layer[nl - 1].in_dim - S;
            store_rows(p.dS + ((int64_t)e * p.maxB + row0) * S, G, ldS, nrows, S);
            {
                const ASrc A{G, ldS, G, ldS};
                const lp dh = sH[0];
                if (Lh > 0) {
                    const PB Bh = make_pb(p.pack + p.pkh_off[e], HL, S, 0);
                    layer_nt<RT>(A, Bh, [&](int row, int col, float v) {
                        if (col < HL) lds_st(dh + row * ldH + col, v);
                    });
                }
                const PB Bc = make_pb(p.pack + p.pkb_off[e][nl - 1], S, S, 0);
                layer_nt<RT>(A, Bc, [&](int row, int col, float v) {
                    if (col < S) lds_st(Gn + row * ldS + col, v - cS * lds_ld(sDiff + row * ldS + col));
                });
            }
            __syncthreads();
            for (int l = Lh - 1; l >= 0; --l) {
                const int Hl = enc.layer[l].out_dim;
                const lp cbuf = sH[(Lh - 1 - l) & 1];
                apply_act_grad(cbuf, ldH, p.hid + p.hid_off[e][l] + (int64_t)row0 * Hl,
                               p.dpre + p.hid_off[e][l] + (int64_t)row0 * Hl, nrows, TB, Hl, akind);
                __syncthreads();
                if (l == 0) break;
                const int Hp = enc.layer[l].in_dim;
                const lp nbuf = sH[(Lh - l) & 1];
                const ASrc A{cbuf, ldH, cbuf, ldH};
                const PB B = make_pb(p.pack + p.pkb_off[e][l], Hp, Hl, 0);
                layer_nt<RT>(A, B, [&](int row, int col, float v) {
                    if (col < Hp) lds_st(nbuf + row * ldH + col, v);
                });
                __syncthreads();
            }
        }
        cur ^= 1;
    }
    // row 0: decoders on the init state; dS0 = d loss / d tiled init state
    const lp G = sG[cur];
    gen_decoder_back<RT>(p, sW, G, sDz, sDd, sH[0], sH[1], 0, row0, nrows);
    store_rows(p.dS + ((int64_t)E * p.maxB + row0) * S, G, ldS, nrows, S);
}

// Everything the fast kernels need to know about the model, as a KERNEL ARGUMENT: scalar loads from the kernarg
// segment that the compiler may hoist and keep in SGPRs.  (Through the LDS copy of the plan every "p.m.dec[d]..."
// is a dependent ds_read that cannot be hoisted over the LDS stores around it: measured, the descriptor reads alone
// made one decoder pass cost 9 us for 1.5 us of MFMA work.)  Same member names as DevPlan so that the code reads alike.
constexpr int GF_MAXE = 8, GF_MAXD = 8, GF_MAXL = 3;
// One (decoder, column tile) product of a decoder phase, fully resolved on the host: a wave reads its items from an
// LDS copy of this table (a few 16-byte reads) instead of chasing "dec[d].hidden[ph]..." through scalar loads at every
// phase (measured: ~1.2 us of dependent descriptor loads per phase against 0.3-1.3 us of MFMA work).
struct GfItem {
    int32_t a_col;          // A operand: first column inside the wide activation tile, or -1 = the state tile
    int32_t T;              // k-steps
    int32_t w_off;          // this tile's fragments: float offset inside the LDS copy of the decoder operands
    int32_t bias_off;       // offset of the layer's bias inside the LDS bias copy
    int32_t out_col;        // hidden layer: first column of the layer's slot in the wide tile; output layer: 2 d
    int32_t n_valid;        // valid output columns of the layer (N)
    int32_t col0;           // first column of this tile inside the layer
    int32_t kind_hk;        // bit 8: output layer; low bits: hidden activation
};
constexpr int GF_PHASES = MMN_MAX_DEC_HIDDEN + 1, GF_SLOTS = 4;
struct GfItemTable { int32_t cnt[GF_PHASES][4]; GfItem item[GF_PHASES][4][GF_SLOTS]; };
// backward: phase ph (1 .. deepest-1) = dh of layer ph-1 from dpre of layer ph; top[]: the decoders' K = 2 output
// products (plain FMAs); gd[]: the decoders whose first hidden layer feeds the state gradient (one product stacked over K)
struct GfBwdTable {
    int32_t cnt[GF_PHASES][4]; GfItem item[GF_PHASES][4][GF_SLOTS];
    int32_t n_top, n_gd, pad0, pad1;
    int32_t top[GF_MAXD][8];       // d, n_hidden, hidden_activation, width of the last hidden layer (or S), its first column, dwf_off
    int32_t gd[GF_MAXD][4];        // first column of dpre_0, k-steps, operand offset (tile 0), unused
};
struct GfLinear { int32_t out_dim, in_dim; };
struct GfEncoder { int32_t n_layers, n_features, activation, pad; GfLinear layer[GF_MAXL]; };
struct GfDecoder { int32_t n_hidden, hidden_activation; GfLinear hidden[MMN_MAX_DEC_HIDDEN]; };
struct GfModel { const float* init_state; GfEncoder enc[GF_MAXE]; GfDecoder dec[GF_MAXD]; };
struct GenArgs {
    GfModel m;
    int32_t S, E, D, R, ldS, ldH, maxB, dcols, dec_maxnh, n_bias, dec_f_floats, dec_b_floats;
    const float* pack; const float* biasbuf;
    float* states; float* hid; float* dpre; float* dz; float* dS; float* gact; float* gdpre;
    float* lossp; float* scp; int32_t* cnt; int32_t* exec_flags; int32_t* prev_row; long long* stamps;
    int64_t dec_f_off, dec_b_off, dh_base, dh_row_stride;
    int64_t pkf_off[GF_MAXE][GF_MAXL], pkb_off[GF_MAXE][GF_MAXL], hid_off[GF_MAXE][GF_MAXL], xin_off[GF_MAXE];
    int64_t pkdf_off[GF_MAXD][MMN_MAX_DEC_HIDDEN + 1], pkdb_off[GF_MAXD][MMN_MAX_DEC_HIDDEN + 1];
    int32_t dh_off[GF_MAXD][MMN_MAX_DEC_HIDDEN];
    int32_t ebias_off[GF_MAXE][GF_MAXL], dbias_off[GF_MAXD][MMN_MAX_DEC_HIDDEN + 1], dwf_off[GF_MAXD];
    const GfItemTable* items_f;        // forward decoder phases (device copy inside the workspace)
    const GfBwdTable* items_b;         // backward decoder phases
};
typedef const GenArgs GPlan;

// ------------------------------------------------------------------------------------------------
// Fast form of the generic tier (k_genf_fwd / k_genf_bwd): all encoders MIMIC_MLPEncoder with <= 3
// layers, hidden widths <= 64, features and state <= 128, and the decoders' operands resident in LDS.
// The sequential form above pays one dependent global round trip (~2 us on the busy chip) per LAYER:
// the weight fragments, the bias in the epilogue, the activations for act'.  With ~60 layer
// evaluations per tile and direction that is the whole run time.  Here every global read of one
// encoder step is issued in ONE batch, a whole step ahead of its use:
//   forward : the next encoder's x tile, dropout multipliers and weight fragments (registers) are
//             requested before the decoders of the current state run; all biases and the decoders'
//             operands sit in LDS (one coalesced copy per workgroup)
//   backward: s_out / s_in / hidden activations / dz / the decoders' hidden activations of the grid
//             row and the encoder's W^T fragments are requested together at the top of the step and
//             parked in LDS; act' then reads LDS
// ------------------------------------------------------------------------------------------------
struct GenFastLds { ChainLds c; int sW, sBias, sItems, sActD, ldD, sOut, sActE, total; };
__host__ __device__ inline GenFastLds gen_fast_lds(int TB, int ldS, int ldH, int wfloats, int nbias, int dcols, bool bwd) {
    GenFastLds L;
    L.c = chain_lds(TB, ldS, ldH);
    // (no LDS copy of the plan here: the descriptor is a kernel argument)
    L.c.sS0 -= PLAN_FLOATS; L.c.sS1 -= PLAN_FLOATS; L.c.sDiff -= PLAN_FLOATS; L.c.sH0 -= PLAN_FLOATS; L.c.sH1 -= PLAN_FLOATS;
    L.c.sX -= PLAN_FLOATS; L.c.sZ -= PLAN_FLOATS; L.c.sRed -= PLAN_FLOATS; L.c.total -= PLAN_FLOATS;
    int o = L.c.total;
    L.sW = o; o += round_up(wfloats, 4);
    L.sBias = o; o += round_up(nbias, 4);
    L.sItems = o; o += (int)((bwd ? sizeof(GfBwdTable) : sizeof(GfItemTable)) / 4);
    L.ldD = pick_ld(dcols > 0 ? dcols : 16);               // all decoders' hidden layers side by side: [TB x ldD]
    L.sActD = o; o += TB * L.ldD;
    L.sOut = o; L.sActE = o;
    if (bwd) {
        L.sOut = o; o += TB * ldS;
        L.sActE = o; o += 2 * TB * ldH;
    }
    L.total = o;
    return L;
}

// one wave, one 16-column tile: acc += A[rows x 16 T] * W'[tile]^T, both operands in LDS, next k-step's
// fragments requested before this one's MFMAs
template <int RT>
__device__ __forceinline__ void wave_tile_l(f32x4 (&acc)[RT], clp a, int lda, clp w_tile, int T) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    clp ap = a + i * lda + 4 * q;
    clp bp = w_tile + lane * 4;
    f32x4 bb = lds_ld4(bp);
    f32x4 av[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) av[r] = lds_ld4(ap + r * 16 * lda);
    for (int t = 0; t < T; ++t) {
        const int tn = min(t + 1, T - 1);
        const f32x4 bn = lds_ld4(bp + tn * 256);
        f32x4 an[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) an[r] = lds_ld4(ap + 16 * tn + r * 16 * lda);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            acc[r] = mfma4(av[r].x, bb.x, acc[r]);
            acc[r] = mfma4(av[r].y, bb.y, acc[r]);
            acc[r] = mfma4(av[r].z, bb.z, acc[r]);
            acc[r] = mfma4(av[r].w, bb.w, acc[r]);
        }
        bb = bn;
#pragma unroll
        for (int r = 0; r < RT; ++r) av[r] = an[r];
    }
}

template <int RT> struct GenEncRegs {
    f32x4 x[2 * RT], mx[2 * RT], ms[2 * RT];   // thread -> row (tid + NT k) >> 5, 4 columns at ((tid + NT k) & 31) << 2
    f32x4 b0[TQ][2], b1[4][2], b2[4][2];       // this wave's weight fragments of the (up to) three layers
};

// Unconditional 4-float read from a clamped (always valid) address.  RAW values: nothing at the request site may
// depend on the loaded data (a select on it makes hipcc wait for the load right there, and the prefetch is gone -
// measured: every "prefetched" tile was waited for at its request); out-of-range lanes are zeroed by sel4 at the
// point of USE.
__device__ __forceinline__ f32x4 ld4_masked(const float* __restrict__ base, int64_t off, int c, int limit, bool row_ok) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = row_ok && (c + j) < limit;
        v[j] = g_ld(base + (ok ? off + j : 0));
    }
    return v;
}
__device__ __forceinline__ f32x4 sel4(f32x4 v, int c, int limit, bool row_ok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (row_ok && (c + j) < limit) ? v[j] : 0.f;
    return v;
}

template <int RT>
__device__ __forceinline__ void issue_gen_encoder(GenEncRegs<RT>& R, GPlan& p, const mmn_batch& b, int e, int slot, int row0,
                                                  int nrows) {
    const int wave = wave_id();
    const auto& enc = p.m.enc[e];
    const int S = p.S, F = enc.n_features, FS = F + S, nl = enc.n_layers;
    const float* xg = b.x[slot] + (int64_t)row0 * b.ldx[slot];
    const int64_t ldx = b.ldx[slot];
    const float* mk = b.drop_mask[e] ? b.drop_mask[e] + (int64_t)row0 * FS : nullptr;
#pragma unroll
    for (int k = 0; k < 2 * RT; ++k) {
        const int idx = threadIdx.x + NT * k;
        const int row = idx >> 5, c = (idx & 31) << 2;
        const bool rok = row < nrows;
        R.x[k] = ld4_masked(xg, (int64_t)row * ldx + c, c, F, rok);
        if (mk) {
            R.mx[k] = ld4_masked(mk, (int64_t)row * FS + c, c, F, rok);
            R.ms[k] = ld4_masked(mk, (int64_t)row * FS + F + c, c, S, rok);
        } else {
            R.mx[k] = f32x4{1.f, 1.f, 1.f, 1.f};
            R.ms[k] = f32x4{1.f, 1.f, 1.f, 1.f};
        }
    }
    const int n0[2] = {16 * wave, 16 * (wave + 4)};
    issue_b<TQ>(R.b0, make_pb(p.pack + p.pkf_off[e][0], enc.layer[0].out_dim, S, F), n0, 0);
    if (nl >= 2) issue_b<4>(R.b1, make_pb(p.pack + p.pkf_off[e][1], enc.layer[1].out_dim, enc.layer[1].in_dim, 0), n0, 0);
    if (nl >= 3) issue_b<4>(R.b2, make_pb(p.pack + p.pkf_off[e][2], enc.layer[2].out_dim, enc.layer[2].in_dim, 0), n0, 0);
}

// one layer from prefetched fragments: this wave's (up to) two column tiles
template <int RT, int NS, class Epi>
__device__ __forceinline__ void layer_regs(const ASrc& A, const PB& B, f32x4 (&bq)[NS][2], Epi&& epi) {
    const int wave = wave_id();
    const int n0[2] = {16 * wave, 16 * (wave + 4)};
    if (n0[0] < B.N) {
        f32x4 acc[2][RT];
        zero_acc<RT>(acc);
        consume_b<RT, NS>(acc, A, B, bq, 0, B.T, n0[1] < B.N);
        run_epilogue<RT>(acc, n0, B.N, epi);
    }
}

// All D decoders on one state tile, layer by layer ACROSS the decoders: phase ph evaluates hidden layer ph of every
// decoder that has one and the output Linear of every decoder with exactly ph hidden layers; the (decoder, column
// tile) pairs of a phase are dealt to the four waves, one barrier per phase.  Operands and biases in LDS; the hidden
// activations of all decoders live side by side in sDA [TB x ldD] and leave as ONE contiguous chunk.
struct GenfCtx {
    lp sZ; clp sW; clp sItems;
    int y, row0, nrows, tile;
    float cL;
    int want_grads;
};

template <int RT>
__device__ __forceinline__ void genf_decode(GPlan& p, const GenfCtx& c, clp sBias, lp sDA, int ldD, clp sS, int grid_row,
                                            int& stamp_k) {
    constexpr int TB = 16 * RT;
    const int stamp_block = 7;
    const int ldS = p.ldS, D = p.D, R = p.R, S = p.S;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15, q = lane >> 4;
    const MMN_AS3 GfItemTable* tab = (const MMN_AS3 GfItemTable*)c.sItems;
    for (int ph = 0; ph <= p.dec_maxnh; ++ph) {
        const int n_it = __builtin_amdgcn_readfirstlane(tab->cnt[ph][wave]);
        for (int sl = 0; sl < n_it; ++sl) {
            const MMN_AS3 int32_t* ip = (const MMN_AS3 int32_t*)&tab->item[ph][wave][sl];
            const int a_col = __builtin_amdgcn_readfirstlane(ip[0]), T = __builtin_amdgcn_readfirstlane(ip[1]);
            const int w_off = __builtin_amdgcn_readfirstlane(ip[2]), bias_off = __builtin_amdgcn_readfirstlane(ip[3]);
            const int out_col = __builtin_amdgcn_readfirstlane(ip[4]), N = __builtin_amdgcn_readfirstlane(ip[5]);
            const int col0 = __builtin_amdgcn_readfirstlane(ip[6]), kh = __builtin_amdgcn_readfirstlane(ip[7]);
            const bool fin = (kh & 256) != 0;
            const int hk = kh & 255;
            clp a = a_col < 0 ? sS : (clp)(sDA + a_col);
            const int lda = a_col < 0 ? ldS : ldD;
            f32x4 acc[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            wave_tile_l<RT>(acc, a, lda, c.sW + w_off, T);
            const int col = col0 + i;
            const float bv = col < N ? lds_ld(sBias + bias_off + col) : 0.f;
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = r * 16 + q * 4 + k;
                    if (fin) {
                        if (col < 2) lds_st(c.sZ + row * 16 + out_col + col, acc[r][k] + bv);
                    } else {                                // padding columns of the slot stay finite (zero)
                        lds_st(sDA + row * ldD + out_col + col, col < N ? act_fwd(acc[r][k] + bv, hk) : 0.f);
                    }
                }
        }
        __syncthreads();
        STAMP();   // d.ph
    }
    if (c.want_grads && p.dcols > 0) {                      // one contiguous [nrows x dcols] chunk
        float* dst = p.gact + p.dh_base + (int64_t)grid_row * p.dh_row_stride + (int64_t)c.row0 * p.dcols;
        const int c4n = p.dcols >> 2;
        for (int idx = threadIdx.x; idx < c.nrows * c4n; idx += NT) {
            const int row = idx / c4n, c4 = idx - row * c4n;
            g_st4(dst + (int64_t)row * p.dcols + 4 * c4, lds_ld4(sDA + row * ldD + 4 * c4));
        }
    }
    STAMP();   // d.store
    const int t = threadIdx.x;
    const int row = t & (TB - 1), d = t / TB;
    float lossv = 0.f;
    int correct = 0, tp = 0, tn = 0, fp = 0, fn = 0;
    if (d < D && row < c.nrows) {
        const float za = lds_ld(c.sZ + row * 16 + 2 * d), zb = lds_ld(c.sZ + row * 16 + 2 * d + 1);
        const int64_t grow = (int64_t)c.row0 + row;
        const int y = c.y;
        const float o0 = 1.0f / (1.0f + expf(-za));
        const float o1 = 1.0f / (1.0f + expf(-zb));
        const float mx = fmaxf(o0, o1);
        const float lse = mx + logf(expf(o0 - mx) + expf(o1 - mx));
        lossv = lse - (y ? o1 : o0);
        const int pred = o1 > o0 ? 1 : 0;          // torch.max: first index wins ties
        correct = pred == y;
        tp = pred & y; tn = (1 - pred) & (1 - y); fp = pred & (1 - y); fn = (1 - pred) & y;
        f32x2 ov;
        if (c.want_grads) {
            const float g0 = expf(o0 - lse) - (y == 0 ? 1.0f : 0.0f);
            const float g1 = expf(o1 - lse) - (y == 1 ? 1.0f : 0.0f);
            ov.x = c.cL * g0 * o0 * (1.0f - o0);
            ov.y = c.cL * g1 * o1 * (1.0f - o1);
        } else {
            ov.x = o0; ov.y = o1;
        }
        g_st2(p.dz + ((int64_t)grid_row * p.maxB + grow) * (2 * D) + 2 * d, ov);
    }
#pragma unroll
    for (int off = TB / 2; off >= 1; off >>= 1) lossv += __shfl_xor(lossv, off);
    const unsigned long long mc = __ballot(correct), mtp = __ballot(tp), mtn = __ballot(tn),
                             mfp = __ballot(fp), mfn = __ballot(fn);
    if (row == 0 && d < D) {
        const int sh = lane & ~(TB - 1);
        const unsigned long long msk = (TB == 32) ? 0xFFFFFFFFull : 0xFFFFull;
        const int64_t cell = (int64_t)c.tile * (R * D) + grid_row * D + d;
        g_st(p.lossp + cell, lossv);
        int32_t* cp = p.cnt + cell * 5;
        g_sti(cp + 0, __popcll((mc >> sh) & msk));
        g_sti(cp + 1, __popcll((mtp >> sh) & msk));
        g_sti(cp + 2, __popcll((mtn >> sh) & msk));
        g_sti(cp + 3, __popcll((mfp >> sh) & msk));
        g_sti(cp + 4, __popcll((mfn >> sh) & msk));
    }
    __syncthreads();
}

template <int RT>
__global__ __launch_bounds__(NT) void k_genf_fwd(const GenArgs ga, const mmn_batch b, float cL, int want_grads) {
    constexpr int TB = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    GPlan& p = ga;
    const GenArgs* P = &ga;
    const int ldS = P->ldS, ldH = P->ldH;
    const GenFastLds GL = gen_fast_lds(TB, ldS, ldH, P->dec_f_floats, P->n_bias, P->dcols, false);
    const ChainLds& L = GL.c;
    lp sS[2] = {smem + L.sS0, smem + L.sS1};
    const lp sM = smem + L.sDiff;
    lp sH[2] = {smem + L.sH0, smem + L.sH1};
    const lp sX = smem + L.sX;
    const lp sRed = smem + L.sRed;
    const lp sBias = smem + GL.sBias;
    const lp sDA = smem + GL.sActD;
    const int tile = blockIdx.x;
    const int row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = wave_id();
    int stamp_k = 0;
    const int stamp_block = 7;
    // the NaN flags ONCE, as a wave-uniform mask: a per-use global load is a dependent round trip at every encoder step
    const unsigned pm = present_mask(b);
    // operands that do not depend on the plan copy: request them first
    copy_pack_to_lds(smem + GL.sW, P->pack + P->dec_f_off, P->dec_f_floats);
    for (int idx = threadIdx.x; idx < P->n_bias; idx += NT) lds_st(sBias + idx, g_ld(P->biasbuf + idx));
    for (int idx = threadIdx.x; idx < (int)(sizeof(GfItemTable) / 4); idx += NT)
        lds_st(smem + GL.sItems + idx, g_ld(reinterpret_cast<const float*>(P->items_f) + idx));
    __syncthreads();
    const int S = p.S, E = p.E;
    STAMP();   // 0: plan, decoder operands, biases in LDS

    for (int idx = threadIdx.x; idx < TB * ldS; idx += NT) {
        const int k = idx % ldS;
        lds_st(sS[0] + idx, k < S ? g_ld(p.m.init_state + k) : 0.f);
        lds_st(sS[1] + idx, 0.f);
        lds_st(sM + idx, 0.f);
    }
    for (int idx = threadIdx.x; idx < TB * ldH; idx += NT) { lds_st(sH[0] + idx, 0.f); lds_st(sH[1] + idx, 0.f); }
    for (int idx = threadIdx.x; idx < TB * LDX; idx += NT) lds_st(sX + idx, 0.f);
    for (int idx = threadIdx.x; idx < 4 * TB * 16; idx += NT) lds_st(smem + L.sZ + idx, 0.f);
    for (int idx = threadIdx.x; idx < TB * GL.ldD; idx += NT) lds_st(sDA + idx, 0.f);
    if (tile == 0 && threadIdx.x == 0) {
        g_sti(p.exec_flags, 1);
        for (int e = 0; e < E; ++e) g_sti(p.exec_flags + e + 1, 0);
        int prev = 0;
        for (int t = 0; t < b.n_seq; ++t) {
            if (!slot_present(pm, b.seq_data[t])) continue;
            const int e = b.seq_enc[t];
            g_sti(p.exec_flags + e + 1, 1);
            g_sti(p.prev_row + e, prev);
            prev = e + 1;
        }
    }
    GenfCtx dc;
    dc.sZ = smem + L.sZ; dc.sW = smem + GL.sW; dc.sItems = smem + GL.sItems;
    dc.row0 = row0; dc.nrows = nrows; dc.tile = tile; dc.cL = cL; dc.want_grads = want_grads;
    {
        const int row = threadIdx.x & (TB - 1), d = threadIdx.x / TB;
        const bool ok = d < p.D && row < nrows;
        dc.y = ok ? (int)*(const MMN_AS1 int64_t*)(b.y + ((int64_t)row0 + row) * p.D + d) : 0;
    }
    GenEncRegs<RT> R;
    int tn = next_exec(b, pm, 0);
    if (tn < b.n_seq) issue_gen_encoder<RT>(R, p, b, b.seq_enc[tn], b.seq_data[tn], row0, nrows);
    __syncthreads();

    int cur = 0;
    STAMP();   // 1: init done, first encoder requested
    genf_decode<RT>(p, dc, sBias, sDA, GL.ldD, sS[cur], 0, stamp_k);
    STAMP();   // 2: decode row 0

    while (tn < b.n_seq) {
        const int e = b.seq_enc[tn];
        const auto& enc = p.m.enc[e];
        const int nl = enc.n_layers, F = enc.n_features, FS = F + S, akind = enc.activation;
        const bool masked = b.drop_mask[e] != nullptr;
        const int t_next = next_exec(b, pm, tn + 1);
        float scacc = 0.f;
        const clp sC = sS[cur];
        const lp sN = sS[cur ^ 1];
        // ---- registers -> LDS images: (masked) x, (masked) state; xin for k_wgrad
        {
            float* xin = p.gact + p.xin_off[e] + (int64_t)row0 * FS;
#pragma unroll
            for (int k = 0; k < 2 * RT; ++k) {
                const int idx = threadIdx.x + NT * k;
                const int row = idx >> 5, c = (idx & 31) << 2;
                const bool rok = row < nrows;
                f32x4 xv = sel4(R.x[k], c, F, rok), sv = {0.f, 0.f, 0.f, 0.f};
                const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};
                const f32x4 mxv = masked ? sel4(R.mx[k], c, F, rok) : one4;
                const f32x4 msv = masked ? sel4(R.ms[k], c, S, rok) : one4;
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] *= mxv[j];
                lds_st4(sX + row * LDX + c, xv);
                if (c < S) {
                    const f32x4 s4 = lds_ld4(sC + row * ldS + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) sv[j] = (c + j < S) ? s4[j] * msv[j] : 0.f;
                    if (masked) lds_st4(sM + row * ldS + c, sv);
                }
                if (want_grads && row < nrows) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (c + j < F) g_st(xin + (int64_t)row * FS + c + j, xv[j]);
                        if (c + j < S) g_st(xin + (int64_t)row * FS + F + c + j, sv[j]);
                    }
                }
            }
        }
        __syncthreads();
        STAMP();   // e.0: x / state images staged (waits for the prefetched x, masks)
        const clp sIn = masked ? (clp)sM : sC;
        auto state_epi = [&](int l) {
            const clp bias = sBias + p.ebias_off[e][l];
            return [&, bias](int row, int col, int, float v) {
                if (col < S) {
                    const float ns = act_fwd(v + lds_ld(bias + col), akind);
                    const float dlt = ns - lds_ld(sC + row * ldS + col);
                    if (row < nrows) scacc += dlt * dlt;              // multimodn.py:174
                    lds_st(sN + row * ldS + col, ns);
                }
            };
        };
        auto hid_epi = [&](int l, lp out) {
            const clp bias = sBias + p.ebias_off[e][l];
            const int N = enc.layer[l].out_dim;
            return [&, bias, N, out](int row, int col, int, float v) {
                if (col < N) lds_st(out + row * ldH + col, act_fwd(v + lds_ld(bias + col), akind));
            };
        };
        {   // layer 0: [state | x] (mlp_encoder.py:40-41)
            const ASrc A{sIn, ldS, sX, LDX};
            const PB B = make_pb(p.pack + p.pkf_off[e][0], enc.layer[0].out_dim, S, F);
            if (nl == 1) layer_regs<RT, TQ>(A, B, R.b0, state_epi(0));
            else layer_regs<RT, TQ>(A, B, R.b0, hid_epi(0, sH[0]));
        }
        __syncthreads();
        STAMP();   // e.1: layer 0
        if (nl >= 2) {
            if (want_grads) store_rows(p.hid + p.hid_off[e][0] + (int64_t)row0 * enc.layer[0].out_dim, sH[0], ldH, nrows, enc.layer[0].out_dim);
            const ASrc A{sH[0], ldH, sH[0], ldH};
            const PB B = make_pb(p.pack + p.pkf_off[e][1], enc.layer[1].out_dim, enc.layer[1].in_dim, 0);
            if (nl == 2) layer_regs<RT, 4>(A, B, R.b1, state_epi(1));
            else layer_regs<RT, 4>(A, B, R.b1, hid_epi(1, sH[1]));
            __syncthreads();
        }
        if (nl >= 3) {
            if (want_grads) store_rows(p.hid + p.hid_off[e][1] + (int64_t)row0 * enc.layer[1].out_dim, sH[1], ldH, nrows, enc.layer[1].out_dim);
            const ASrc A{sH[1], ldH, sH[1], ldH};
            const PB B = make_pb(p.pack + p.pkf_off[e][2], enc.layer[2].out_dim, enc.layer[2].in_dim, 0);
            layer_regs<RT, 4>(A, B, R.b2, state_epi(2));
            __syncthreads();
        }
        STAMP();   // e.2: layers 1, 2
        // the next encoder's inputs travel underneath the decoders of this state
        if (t_next < b.n_seq) issue_gen_encoder<RT>(R, p, b, b.seq_enc[t_next], b.seq_data[t_next], row0, nrows);
        scacc = wave_sum(scacc);
        if (lane == 0) lds_st(sRed + wave, scacc);
        __syncthreads();
        if (threadIdx.x == 0)
            g_st(p.scp + (int64_t)tile * E + e, ((lds_ld(sRed) + lds_ld(sRed + 1)) + lds_ld(sRed + 2)) + lds_ld(sRed + 3));
        store_rows(p.states + ((int64_t)e * p.maxB + row0) * S, sN, ldS, nrows, S);
        cur ^= 1;
        STAMP();   // e.3: next encoder requested, state-change partial, state stored
        genf_decode<RT>(p, dc, sBias, sDA, GL.ldD, sS[cur], e + 1, stamp_k);
        STAMP();   // e.4: decode
        tn = t_next;
    }
}

// act' applied with the layer's OUTPUT read from an LDS tile: sBuf (raw dh) -> sBuf and global dpre (row stride ldg)
__device__ __forceinline__ void apply_act_grad_l(lp sBuf, int ld, clp hL, int ldh, float* __restrict__ dpre_g, int64_t ldg,
                                                 int nrows, int rows_pad, int H, int akind) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    for (int r = wave; r < rows_pad; r += 4)
        for (int c = lane; c < H; c += 64) {
            float dp = 0.f;
            if (r < nrows) {
                dp = lds_ld(sBuf + r * ld + c) * act_grad_from_out(lds_ld(hL + r * ldh + c), akind);
                g_st(dpre_g + (int64_t)r * ldg + c, dp);
            }
            lds_st(sBuf + r * ld + c, dp);
        }
}

template <int RT> struct GenBwdRegs {
    f32x4 so[2 * RT], si[2 * RT], ms[2 * RT];  // s_out, s_in, state-column dropout multipliers: row (tid + NT k) >> 5, cols ((..) & 31) << 2
    f32x4 h0[RT], h1[RT];                      // hidden activations: row (tid + NT k) >> 4, cols ((..) & 15) << 2
    float dz[RT];                              // dz tile of the grid row: row (tid + NT k) >> 4, col (..) & 15
    f32x4 da[8];                               // the decoders' hidden activations of the grid row: flat chunk
    f32x4 bL[8][2], bM[4][2], b0[4][2];        // W^T fragments: the layer contracting over S; a middle layer; the carry
};

// W^T fragments of one encoder's backward products (consumed late in a step: requested right after the previous
// step's last use of the registers)
template <int RT>
__device__ __forceinline__ void issue_gen_bwd_weights(GenBwdRegs<RT>& R, GPlan& p, int e) {
    const int wave = wave_id();
    const auto& enc = p.m.enc[e];
    const int S = p.S, nl = enc.n_layers;
    const int n0[2] = {16 * wave, 16 * (wave + 4)};
    const int H0 = enc.layer[0].out_dim;
    if (nl == 1) {
        issue_b<8>(R.bL, make_pb(p.pack + p.pkb_off[e][0], S, H0, 0), n0, 0);              // carry contracts over S
    } else {
        issue_b<8>(R.bL, make_pb(p.pack + p.pkb_off[e][nl - 1], enc.layer[nl - 1].in_dim, S, 0), n0, 0);
        if (nl == 3) issue_b<4>(R.bM, make_pb(p.pack + p.pkb_off[e][1], enc.layer[1].in_dim, enc.layer[1].out_dim, 0), n0, 0);
        issue_b<4>(R.b0, make_pb(p.pack + p.pkb_off[e][0], S, H0, 0), n0, 0);
    }
}

// activation tiles of one step (consumed first: requested a whole step ahead, right after the previous step parked its own)
template <int RT>
__device__ __forceinline__ void issue_gen_bwd(GenBwdRegs<RT>& R, GPlan& p, const mmn_batch& b, int e, int prev_row, int grid_row,
                                              int row0, int nrows, bool with_encoder) {
    constexpr int TB = 16 * RT;
    const int S = p.S, D2 = 2 * p.D;
    if (with_encoder) {
        const auto& enc = p.m.enc[e];
        const int F = enc.n_features, FS = F + S, nl = enc.n_layers;
        const float* so = p.states + ((int64_t)e * p.maxB + row0) * S;
        const float* si = prev_row ? p.states + ((int64_t)(prev_row - 1) * p.maxB + row0) * S : nullptr;
        const float* mk = b.drop_mask[e] ? b.drop_mask[e] + (int64_t)row0 * FS + F : nullptr;
#pragma unroll
        for (int k = 0; k < 2 * RT; ++k) {
            const int idx = threadIdx.x + NT * k;
            const int row = idx >> 5, c = (idx & 31) << 2;
            const bool rok = row < nrows;
            R.so[k] = ld4_masked(so, (int64_t)row * S + c, c, S, rok);
            R.si[k] = si ? ld4_masked(si, (int64_t)row * S + c, c, S, rok) : ld4_masked(p.m.init_state, c, c, S, rok);
            R.ms[k] = mk ? ld4_masked(mk, (int64_t)row * FS + c, c, S, rok) : f32x4{1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int k = 0; k < RT; ++k) {
            const int idx = threadIdx.x + NT * k;
            const int row = idx >> 4, c = (idx & 15) << 2;
            const bool rok = row < nrows;
            const int H0 = enc.layer[0].out_dim, H1 = nl >= 3 ? enc.layer[1].out_dim : 0;
            R.h0[k] = nl >= 2 ? ld4_masked(p.hid + p.hid_off[e][0] + (int64_t)row0 * H0, (int64_t)row * H0 + c, c, H0, rok)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
            R.h1[k] = nl >= 3 ? ld4_masked(p.hid + p.hid_off[e][1] + (int64_t)row0 * H1, (int64_t)row * H1 + c, c, H1, rok)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // dz tile and the decoders' hidden activations of this grid row (one contiguous chunk per tile)
#pragma unroll
    for (int k = 0; k < RT; ++k) {
        const int idx = threadIdx.x + NT * k;
        const int row = idx >> 4, n = idx & 15;
        const bool ok = row < nrows && n < D2;
        R.dz[k] = g_ld(p.dz + ((int64_t)grid_row * p.maxB + row0) * D2 + (ok ? row * D2 + n : 0));    // raw: zeroed when parked
    }
    const float* da = p.gact + p.dh_base + (int64_t)grid_row * p.dh_row_stride + (int64_t)row0 * p.dcols;
    const int nda = nrows * p.dcols;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx4 = (threadIdx.x + NT * k) * 4;
        R.da[k] = g_ld4(da + (idx4 < nda ? idx4 : 0));      // raw: zeroed when parked
    }
}

// G += d loss / d state through the decoders of one grid row, layer by layer ACROSS the decoders (mirror of
// genf_decode).  sActD [TB x ldD] holds the decoders' hidden activations on entry and their pre-activation
// gradients on exit: every product's epilogue multiplies by act'(h) in place (each element is touched by one lane).
template <int RT>
__device__ __forceinline__ void genf_decoder_back(GPlan& p, clp sW, clp sBias, clp sItems, lp G, clp sDz, lp sActD, int ldD,
                                                  int grid_row, int row0, int nrows) {
    constexpr int TB = 16 * RT;
    const int ldS = p.ldS, S = p.S;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15, q = lane >> 4;
    const MMN_AS3 GfBwdTable* tab = (const MMN_AS3 GfBwdTable*)sItems;
    {   // output Linear of every decoder (K = 2): dh = dz0 W[0,:] + dz1 W[1,:], times act'(h) in place; decoders without
        // hidden layers add straight into G (one thread owns (row, col) for all of them)
        const int n_top = tab->n_top;
        const int row = threadIdx.x & (TB - 1), c0 = threadIdx.x / TB, cstep = NT / TB;
        for (int k = 0; k < n_top; ++k) {
            const int d = tab->top[k][0], nh = tab->top[k][1], hk = tab->top[k][2], W = tab->top[k][3], colb = tab->top[k][4];
            clp wf = sBias + tab->top[k][5];
            const float dz0 = lds_ld(sDz + row * LDZ + 2 * d), dz1 = lds_ld(sDz + row * LDZ + 2 * d + 1);
            for (int c = c0; c < W; c += cstep) {
                const float v = fmaf(dz1, lds_ld(wf + W + c), dz0 * lds_ld(wf + c));
                if (nh == 0) {
                    lds_st(G + row * ldS + c, lds_ld(G + row * ldS + c) + v);
                } else {
                    const lp hp = sActD + row * ldD + colb + c;
                    lds_st(hp, row < nrows ? v * act_grad_from_out(lds_ld(hp), hk) : 0.f);
                }
            }
        }
    }
    __syncthreads();
    for (int ph = p.dec_maxnh - 1; ph >= 1; --ph) {
        const int n_it = __builtin_amdgcn_readfirstlane(tab->cnt[ph][wave]);
        for (int sl = 0; sl < n_it; ++sl) {
            const MMN_AS3 int32_t* ip = (const MMN_AS3 int32_t*)&tab->item[ph][wave][sl];
            const int a_col = __builtin_amdgcn_readfirstlane(ip[0]), T = __builtin_amdgcn_readfirstlane(ip[1]);
            const int w_off = __builtin_amdgcn_readfirstlane(ip[2]);
            const int out_col = __builtin_amdgcn_readfirstlane(ip[4]), N = __builtin_amdgcn_readfirstlane(ip[5]);
            const int col0 = __builtin_amdgcn_readfirstlane(ip[6]), hk = __builtin_amdgcn_readfirstlane(ip[7]) & 255;
            f32x4 acc[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            wave_tile_l<RT>(acc, sActD + a_col, ldD, sW + w_off, T);
            const int col = col0 + i;
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = r * 16 + q * 4 + k;
                    const lp hp = sActD + row * ldD + out_col + col;
                    lds_st(hp, (col < N && row < nrows) ? acc[r][k] * act_grad_from_out(lds_ld(hp), hk) : 0.f);
                }
        }
        __syncthreads();
    }
    {   // G += [dpre_0 of every decoder] * [W_0^T of every decoder]: one product per state column tile, stacked over K
        const int n_gd = tab->n_gd;
        const int ntl = (S + 15) >> 4;
        for (int tile = wave; tile < ntl && n_gd > 0; tile += 4) {
            f32x4 acc[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < n_gd; ++k) {
                const int a_col = __builtin_amdgcn_readfirstlane(tab->gd[k][0]), T = __builtin_amdgcn_readfirstlane(tab->gd[k][1]);
                const int w_off = __builtin_amdgcn_readfirstlane(tab->gd[k][2]);
                wave_tile_l<RT>(acc, sActD + a_col, ldD, sW + w_off + tile * T * 256, T);
            }
            const int col = 16 * tile + i;
            if (col < S) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const lp gp = G + (r * 16 + q * 4 + k) * ldS + col;
                        lds_st(gp, lds_ld(gp) + acc[r][k]);
                    }
            }
        }
    }
    __syncthreads();
    if (p.dcols > 0) {                                     // all dpre of this grid row: one contiguous [nrows x dcols] chunk
        float* dst = p.gdpre + (int64_t)grid_row * p.dh_row_stride + (int64_t)row0 * p.dcols;
        const int c4n = p.dcols >> 2;
        for (int idx = threadIdx.x; idx < nrows * c4n; idx += NT) {
            const int row = idx / c4n, c4 = idx - row * c4n;
            g_st4(dst + (int64_t)row * p.dcols + 4 * c4, lds_ld4(sActD + row * ldD + 4 * c4));
        }
    }
}

template <int RT>
__global__ __launch_bounds__(NT) void k_genf_bwd(const GenArgs ga, const mmn_batch b, float cS) {
    constexpr int TB = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    GPlan& p = ga;
    const GenArgs* P = &ga;
    const int ldS = P->ldS, ldH = P->ldH;
    const GenFastLds GL = gen_fast_lds(TB, ldS, ldH, P->dec_b_floats, P->n_bias, P->dcols, true);
    const ChainLds& L = GL.c;
    copy_pack_to_lds(smem + GL.sW, P->pack + P->dec_b_off, P->dec_b_floats);
    for (int idx = threadIdx.x; idx < P->n_bias; idx += NT) lds_st(smem + GL.sBias + idx, g_ld(P->biasbuf + idx));
    for (int idx = threadIdx.x; idx < (int)(sizeof(GfBwdTable) / 4); idx += NT)
        lds_st(smem + GL.sItems + idx, g_ld(reinterpret_cast<const float*>(P->items_b) + idx));
    __syncthreads();
    const int S = p.S, E = p.E, dcols = p.dcols, dc4 = dcols >> 2;
    lp sG[2] = {smem + L.sS0, smem + L.sS1};
    const lp sDiff = smem + L.sDiff;
    lp sH[2] = {smem + L.sH0, smem + L.sH1};
    const lp sDz = smem + L.sZ;
    const clp sBias = smem + GL.sBias;                     // (the raw output Linears of the decoders live behind the biases)
    const clp sItems = smem + GL.sItems;
    const int ldD = GL.ldD;
    const lp sMs = smem + L.sX;                            // state-column dropout multipliers, [TB x LDX]
    const clp sW = smem + GL.sW;
    const lp sActD = smem + GL.sActD;
    const lp sOut = smem + GL.sOut;
    lp sActE[2] = {smem + GL.sActE, smem + GL.sActE + TB * ldH};
    const int tile = blockIdx.x;
    const int row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = wave_id();

    for (int idx = threadIdx.x; idx < TB * ldS; idx += NT) { lds_st(sG[0] + idx, 0.f); lds_st(sG[1] + idx, 0.f); lds_st(sDiff + idx, 0.f); lds_st(sOut + idx, 0.f); }
    for (int idx = threadIdx.x; idx < TB * ldH; idx += NT) { lds_st(sH[0] + idx, 0.f); lds_st(sH[1] + idx, 0.f); lds_st(sActE[0] + idx, 0.f); lds_st(sActE[1] + idx, 0.f); }
    for (int idx = threadIdx.x; idx < 4 * TB * 16; idx += NT) lds_st(sDz + idx, 0.f);
    for (int idx = threadIdx.x; idx < TB * LDX; idx += NT) lds_st(sMs + idx, 0.f);
    __syncthreads();
    int cur = 0;
    GenBwdRegs<RT> R;
    const unsigned pm = present_mask(b);                   // the NaN flags once (see k_genf_fwd)
    int stamp_k = 0;
    const int stamp_block = 7;
    STAMP();   // 0: operands, biases, item table in LDS; tiles zeroed

    // registers -> LDS: the grid row's dz tile and decoder activations (every step), the encoder's tiles (encoder steps)
    auto park = [&](bool with_encoder, int nl, int h0w, int h1w) {
        if (with_encoder) {
#pragma unroll
            for (int k = 0; k < 2 * RT; ++k) {
                const int idx = threadIdx.x + NT * k;
                const int row = idx >> 5, c = (idx & 31) << 2;
                if (c < S) {                                   // (the state tiles are only round_up(S, 16) + 4 floats wide)
                    const bool rok = row < nrows;
                    const f32x4 sov = sel4(R.so[k], c, S, rok), siv = sel4(R.si[k], c, S, rok);
                    lds_st4(sOut + row * ldS + c, sov);
                    f32x4 df;
#pragma unroll
                    for (int j = 0; j < 4; ++j) df[j] = sov[j] - siv[j];
                    lds_st4(sDiff + row * ldS + c, df);
                    lds_st4(sMs + row * LDX + c, sel4(R.ms[k], c, S, rok));
                }
            }
#pragma unroll
            for (int k = 0; k < RT; ++k) {
                const int idx = threadIdx.x + NT * k;
                const int row = idx >> 4, c = (idx & 15) << 2;
                if (c + 4 <= ldH) {                            // (hidden tiles: ldH >= round_up(width, 16) + 4)
                    if (nl >= 2) lds_st4(sActE[0] + row * ldH + c, sel4(R.h0[k], c, h0w, row < nrows));
                    if (nl >= 3) lds_st4(sActE[1] + row * ldH + c, sel4(R.h1[k], c, h1w, row < nrows));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RT; ++k) {
            const int idx = threadIdx.x + NT * k;
            lds_st(sDz + (idx >> 4) * LDZ + (idx & 15), ((idx >> 4) < nrows && (idx & 15) < 2 * p.D) ? R.dz[k] : 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i4 = threadIdx.x + NT * k;           // 16-byte chunk of the [TB x dcols] tile
            if (i4 < TB * dc4) {
                const int row = i4 / dc4, c4 = i4 - row * dc4;
                lds_st4(sActD + row * ldD + 4 * c4, row < nrows ? R.da[k] : f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
    };

    // previous executed position before t (or -1)
    auto prev_exec = [&](int t) { int u = t - 1; while (u >= 0 && !slot_present(pm, b.seq_data[u])) --u; return u; };
    int t = prev_exec(b.n_seq);
    if (t >= 0) {                                          // the first step's requests (every later step's travel a step ahead)
        const int tp0 = prev_exec(t);
        issue_gen_bwd<RT>(R, p, b, b.seq_enc[t], tp0 >= 0 ? b.seq_enc[tp0] + 1 : 0, b.seq_enc[t] + 1, row0, nrows, true);
    } else {
        issue_gen_bwd<RT>(R, p, b, 0, 0, 0, row0, nrows, false);
    }
    for (; t >= 0;) {
        const int e = b.seq_enc[t];
        const int tp = prev_exec(t);
        const int prev_row = tp >= 0 ? b.seq_enc[tp] + 1 : 0;
        const auto& enc = p.m.enc[e];
        const int nl = enc.n_layers, akind = enc.activation;
        const lp G = sG[cur];
        const lp Gn = sG[cur ^ 1];
        park(true, nl, enc.layer[0].out_dim, nl >= 3 ? enc.layer[1].out_dim : 0);
        __syncthreads();
        // The NEXT step's activation tiles and THIS step's W^T fragments travel underneath the decoder pass below
        // (vmcnt retires in order and hipcc waits with vmcnt(0): whatever is requested last must be consumed last,
        // so the fragments are requested here and not a step ahead - measured: requested at the end of the previous
        // step they made the park above wait a full round trip)
        issue_gen_bwd_weights<RT>(R, p, e);
        if (tp >= 0) {
            const int tpp = prev_exec(tp);
            issue_gen_bwd<RT>(R, p, b, b.seq_enc[tp], tpp >= 0 ? b.seq_enc[tpp] + 1 : 0, b.seq_enc[tp] + 1, row0, nrows, true);
        } else {
            issue_gen_bwd<RT>(R, p, b, 0, 0, 0, row0, nrows, false);
        }
        STAMP();   // e.0: the step's global reads have arrived and are parked in LDS
        // G_out = carry + decoder grads of row e+1 + cS * diff
        genf_decoder_back<RT>(p, sW, sBias, sItems, G, sDz, sActD, ldD, e + 1, row0, nrows);
        STAMP();   // e.1: decoders
        for (int r = wave; r < TB; r += 4)
            for (int c = lane; c < S; c += 64) lds_st(G + r * ldS + c, lds_ld(G + r * ldS + c) + cS * lds_ld(sDiff + r * ldS + c));
        __syncthreads();
        // the state layer carries the activation: dpre_last = G .* act'(s_out) = what k_wgrad multiplies (dS[e])
        apply_act_grad_l(G, ldS, sOut, ldS, p.dS + ((int64_t)e * p.maxB + row0) * S, S, nrows, TB, S, akind);
        __syncthreads();
        clp cbuf = G; int ldc = ldS;
        for (int l = nl - 1; l >= 1; --l) {
            const int Hl = enc.layer[l].out_dim, Hp = enc.layer[l].in_dim;
            const lp nbuf = sH[l & 1];
            const ASrc A{cbuf, ldc, cbuf, ldc};
            const PB B = make_pb(p.pack + p.pkb_off[e][l], Hp, Hl, 0);                // W_l^T [in x out]
            const clp hl = sActE[l - 1];
            auto epi = [&](int row, int col, int, float v) {       // dpre_{l-1} = dh .* act'(h_{l-1}), in the epilogue
                if (col < Hp) lds_st(nbuf + row * ldH + col, row < nrows ? v * act_grad_from_out(lds_ld(hl + row * ldH + col), akind) : 0.f);
            };
            if (l == nl - 1) layer_regs<RT, 8>(A, B, R.bL, epi);
            else layer_regs<RT, 4>(A, B, R.bM, epi);
            __syncthreads();
            store_rows(p.dpre + p.hid_off[e][l - 1] + (int64_t)row0 * Hp, nbuf, ldH, nrows, Hp);
            cbuf = nbuf; ldc = ldH;
        }
        {   // carry = (dpre_0 * W_0[:, F:F+S]) .* mask_state - cS * diff ; no grad flows to x
            const int H0 = enc.layer[0].out_dim;
            const ASrc A{cbuf, ldc, cbuf, ldc};
            const PB B = make_pb(p.pack + p.pkb_off[e][0], S, H0, 0);
            auto epi = [&](int row, int col, int, float v) {
                if (col < S) lds_st(Gn + row * ldS + col, v * lds_ld(sMs + row * LDX + col) - cS * lds_ld(sDiff + row * ldS + col));
            };
            if (nl == 1) layer_regs<RT, 8>(A, B, R.bL, epi);
            else layer_regs<RT, 4>(A, B, R.b0, epi);
        }
        __syncthreads();
        STAMP();   // e.2: encoder
        cur ^= 1;
        t = tp;
    }
    // row 0: decoders on the init state; dS0 = d loss / d tiled init state
    const lp G = sG[cur];
    park(false, 0, 0, 0);
    __syncthreads();
    genf_decoder_back<RT>(p, sW, sBias, sItems, G, sDz, sActD, ldD, 0, row0, nrows);
    store_rows(p.dS + ((int64_t)E * p.maxB + row0) * S, G, ldS, nrows, S);
}

// ------------------------------------------------------------------------------------------------
// Parallel-phase chain kernels (16 rows per workgroup; used whenever their LDS budget fits).
//
// The reference walks the encoders strictly in sequence (multimodn.py:159-191), but only the
// state update is sequential: the hidden MLP of every encoder sees x alone (mlp_encoder.py:75-76),
// so u_e = W_x h_e + b can be formed for ALL encoders up front -- one wave per encoder, no barrier
// inside -- and the chain shrinks to s' = W_s s + u_e (one S x S product and one barrier per
// encoder).  All E+1 state tiles stay in LDS, so the D decoders are evaluated on all of them in one
// batched phase at the end.  Backward mirrors it: decoder-gradient terms and state differences are
// formed in parallel first, the chain is G_in = W_s^T G_out - c dS, and the hidden-layer backward
// of every encoder runs in parallel afterwards.  At 16 rows per CU every phase of the sequential
// form was latency-bound (one wave per SIMD, ~12 us per encoder for ~1.5 us of MFMA work).
// ------------------------------------------------------------------------------------------------
struct ParLds { int sPlan, sSt, sU, sG, sW, wstride, oX, oH0, oH1, oDz, sZ, sRed, total; };
__host__ __device__ inline ParLds par_lds(int R, int E, int ldS, int ldH, int ldX, bool bwd) {
    ParLds L;
    int o = 0;
    L.sPlan = o; o += PLAN_FLOATS;
    L.sSt = o; o += R * 16 * ldS;          // fwd: state tiles 0..E        bwd: decoder-grad -> G_out tiles
    L.sU = o; o += E * 16 * ldS;           // fwd: u_e tiles               bwd: state differences
    L.sG = o; o += bwd ? 2 * 16 * ldS : 0; // bwd: carried gradient, ping-pong
    L.oX = 0; L.oH0 = bwd ? 0 : 16 * ldX; L.oH1 = L.oH0 + 16 * ldH; L.oDz = L.oH1 + 16 * ldH;
    L.wstride = L.oDz + (bwd ? 16 * LDZ : 0);
    L.sW = o; o += 4 * L.wstride;          // per-wave scratch: x tile / hidden ping-pong / dz tile
    L.sZ = o; o += bwd ? 0 : R * 16 * 16;
    L.sRed = o; o += 64 + 4 * E;
    L.total = o;
    return L;
}

__device__ __forceinline__ bool row_executed(const mmn_batch& b, int r) {
    if (r == 0) return true;
    for (int t = 0; t < b.n_seq; ++t)
        if (b.seq_enc[t] == r - 1) return slot_present(b, b.seq_data[t]);
    return false;
}

// one wave stores its LDS tile [nrows x N] to global (row stride N), 16 bytes per lane when possible
__device__ __forceinline__ void wave_store_tile(float* __restrict__ dst, clp src, int ld_src, int nrows, int N) {
    const int lane = threadIdx.x & 63;
    if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const int n4 = N >> 2;
        for (int idx = lane; idx < nrows * n4; idx += 64) {
            const int r = idx / n4, c = (idx - r * n4) << 2;
            g_st4(dst + (int64_t)r * N + c, lds_ld4(src + r * ld_src + c));
        }
    } else {
        for (int idx = lane; idx < nrows * N; idx += 64) {
            const int r = idx / N, c = idx - r * N;
            g_st(dst + (int64_t)r * N + c, lds_ld(src + r * ld_src + c));
        }
    }
}

// one wave computes out[16 x N] = A[16 x K] W'^T alone, two column tiles at a time
template <class Epi>
__device__ __forceinline__ void wave_layer(const ASrc& A, const PB& B, int t_begin, int t_end, Epi&& epi) {
    const int ntiles = (B.N + 15) >> 4;
    for (int tp = 0; tp < ntiles; tp += 2) {
        const int n0[2] = {16 * tp, 16 * (tp + 1)};
        f32x4 acc[2][1];
        zero_acc<1>(acc);
        wave_gemm_any<1>(acc, A, B, n0, t_begin, t_end);
        run_epilogue<1>(acc, n0, B.N, [&](int row, int col, int, float v) { epi(row, col, v); });
    }
}

__global__ __launch_bounds__(NT) void k_chain_fwd_par(const DevPlan* __restrict__ P, mmn_batch b, float cL,
                                                      int want_grads) {
    constexpr int TB = 16;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    const int ldS = P->ldS, ldH = P->ldH, ldX = P->ldX, E = P->E;
    const int R = E + 1;
    const ParLds L = par_lds(R, E, ldS, ldH, ldX, false);
    copy_plan_to_lds(P, smem + L.sPlan);
    for (int idx = threadIdx.x; idx < L.total - L.sSt; idx += NT) lds_st(smem + L.sSt + idx, 0.f);
    __syncthreads();
    LPlan& p = *(LPlan*)(smem + L.sPlan);
    const int S = p.S, D = p.D;
    const int tile = blockIdx.x, row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15, q = lane >> 4;
    const lp St = smem + L.sSt;
    const lp Ut = smem + L.sU;
    const lp sZ = smem + L.sZ;
    const lp sRed = smem + L.sRed;
    const lp sXw = smem + L.sW + wave * L.wstride + L.oX;
    lp sHw[2] = {smem + L.sW + wave * L.wstride + L.oH0, smem + L.sW + wave * L.wstride + L.oH1};

    int stamp_k = 0;
    const int stamp_block = 7;
    STAMP();   // P0: plan copied, LDS zeroed
    for (int r = wave; r < TB; r += 4)                     // state row 0 = init state (state.py:29-32)
        for (int c = lane; c < S; c += 64) lds_st(St + r * ldS + c, g_ld(p.m.init_state + c));
    if (tile == 0 && threadIdx.x == 0) {                   // which state rows exist this step
        g_sti(p.exec_flags, 1);
        for (int e = 0; e < E; ++e) g_sti(p.exec_flags + e + 1, 0);
        int prev = 0;
        for (int t = 0; t < b.n_seq; ++t) {
            if (!slot_present(b, b.seq_data[t])) continue;
            const int e = b.seq_enc[t];
            g_sti(p.exec_flags + e + 1, 1);
            g_sti(p.prev_row + e, prev);
            prev = e + 1;
        }
    }
    // decoder weight fragments for the whole contraction (S <= 128 -> <= 8 k-steps)
    const int Tdec = p.S16 >> 4;
    f32x4 wd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < Tdec && i < 2 * D) {
            const float* w = p.m.dec[i >> 1].w + (i & 1) * S;
            const int k = 16 * j + 4 * q;
            if (k < S) v.x = g_ld(w + k);
            if (k + 1 < S) v.y = g_ld(w + k + 1);
            if (k + 2 < S) v.z = g_ld(w + k + 2);
            if (k + 3 < S) v.w = g_ld(w + k + 3);
        }
        wd[j] = v;
    }

    STAMP();   // P1: decoder frags requested
    // chain step 0's W_s fragments are requested now and land during phase A
    f32x4 wsA[8][2], wsB[8][2];
    const int n0c[2] = {16 * wave, 16 * (wave + 4)};
    int t_first = next_exec(b, 0);
    if (t_first < b.n_seq) {
        const int e0 = b.seq_enc[t_first];
        const int Lh0 = p.m.enc[e0].n_layers - 1;
        const PB B0 = make_pb(p.pack + p.pkf_off[e0][Lh0], S, S, p.m.enc[e0].layer[Lh0].in_dim - S);
        issue_b<8>(wsA, B0, n0c, 0);
    }
    // ---- phase A: u_e = W_x h_e + b for every executed encoder, one wave per encoder
    for (int t = wave; t < b.n_seq; t += 4) {
        const int slot = b.seq_data[t];
        if (!slot_present(b, slot)) continue;              // multimodn.py:168-169
        const int e = b.seq_enc[t];
        const auto& enc = p.m.enc[e];
        const int Lh = enc.n_layers - 1, F = enc.n_features, akind = enc.activation;
        const int HL = enc.layer[Lh].in_dim - S;
        const float* xg = b.x[slot] + (int64_t)row0 * b.ldx[slot];
        const int64_t ldx = b.ldx[slot];
        const bool xvec = ((ldx & 3) == 0) && ((F & 3) == 0) && ((reinterpret_cast<uintptr_t>(xg) & 15) == 0);
        bool fastA = xvec && F <= 128 && Lh <= 2 && HL <= 64 && S <= 128;
        for (int l = 0; l < Lh && l < 2; ++l) fastA = fastA && enc.layer[l].in_dim <= 64 && enc.layer[l].out_dim <= 32;
        const lp U = Ut + e * TB * ldS;
        if (fastA) {
            // every operand of this wave's encoder is requested before anything is consumed:
            // one memory round trip for the whole phase instead of one per layer and tile pair
            const int f4 = round_up(F, 16) >> 2;           // float4 per image row (<= 32)
            f32x4 xr[8];
            const int nx = (TB * f4 + 63) >> 6;             // float4 per lane actually needed (wave-uniform)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                xr[k] = z;
                if (k < nx) {
                    const int idx = lane + 64 * k;
                    const int row = idx / f4, c = (idx - row * f4) << 2;
                    const bool ok = idx < TB * f4 && row < nrows && c < F;
                    const f32x4 v = g_ld4(xg + (ok ? (int64_t)row * ldx + c : 0));
                    xr[k] = ok ? v : z;
                }
            }
            const int pair0[2] = {0, 16};
            f32x4 hq0[4][2], hq1[4][2], uq[4][4][2];
            float hb0 = 0.f, hb1 = 0.f, hb0b = 0.f, hb1b = 0.f;
            const int i16 = lane & 15;
            if (Lh >= 1) {
                const auto& lin = enc.layer[0];
                const PB Bq = make_pb(p.pack + p.pkf_off[e][0], lin.out_dim, lin.in_dim, 0);
                if (Bq.T <= 2) issue_b<2>(reinterpret_cast<f32x4 (&)[2][2]>(hq0), Bq, pair0, 0); else issue_b<4>(hq0, Bq, pair0, 0);
                hb0 = g_ld(lin.b + min(i16, lin.out_dim - 1));
                hb0b = g_ld(lin.b + min(16 + i16, lin.out_dim - 1));
            }
            if (Lh >= 2) {
                const auto& lin = enc.layer[1];
                const PB Bq = make_pb(p.pack + p.pkf_off[e][1], lin.out_dim, lin.in_dim, 0);
                if (Bq.T <= 2) issue_b<2>(reinterpret_cast<f32x4 (&)[2][2]>(hq1), Bq, pair0, 0); else issue_b<4>(hq1, Bq, pair0, 0);
                hb1 = g_ld(lin.b + min(i16, lin.out_dim - 1));
                hb1b = g_ld(lin.b + min(16 + i16, lin.out_dim - 1));
            }
            const auto& llin = enc.layer[Lh];
            const PB BU = make_pb(p.pack + p.pkf_off[e][Lh], S, S, HL);
            float ub[4][2];
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const int nn[2] = {32 * pr, 32 * pr + 16};
                if (nn[0] < S) {
                    if (BU.T - BU.T0 <= 2) issue_b<2>(reinterpret_cast<f32x4 (&)[2][2]>(uq[pr]), BU, nn, BU.T0);
                    else issue_b<4>(uq[pr], BU, nn, BU.T0);
                }
                ub[pr][0] = g_ld(llin.b + min(nn[0] + i16, S - 1));
                ub[pr][1] = g_ld(llin.b + min(nn[1] + i16, S - 1));
            }
            STAMP();   // A0: all operand loads issued
            // x -> LDS image
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = lane + 64 * k;
                if (idx < TB * f4) {
                    const int row = idx / f4, c = (idx - row * f4) << 2;
                    lds_st4(sXw + row * ldX + c, xr[k]);
                }
            }
            STAMP();   // A1: x landed and written to LDS
            clp in = sXw;
            int ldin = ldX;
            if (Lh >= 1) {
                const auto& lin = enc.layer[0];
                const int N = lin.out_dim;
                const lp out = sHw[0];
                const PB B = make_pb(p.pack + p.pkf_off[e][0], N, lin.in_dim, 0);
                f32x4 acc[2][1];
                zero_acc<1>(acc);
                consume_b<1, 4>(acc, ASrc{in, ldin, in, ldin}, B, hq0, 0, B.T, 16 < N);
                STAMP();   // A2: h0 MFMAs
                run_epilogue<1>(acc, pair0, N, [&](int row, int col, int c, float v) {
                    if (col < N) lds_st(out + row * ldH + col, act_fwd(v + (c ? hb0b : hb0), akind));
                });
                STAMP();   // A3: h0 epilogue
                if (want_grads) wave_store_tile(p.hid + p.hid_off[e][0] + (int64_t)row0 * N, out, ldH, nrows, N);
                in = out; ldin = ldH;
                STAMP();   // A4: hid0 stored
            }
            if (Lh >= 2) {
                const auto& lin = enc.layer[1];
                const int N = lin.out_dim;
                const lp out = sHw[1];
                const PB B = make_pb(p.pack + p.pkf_off[e][1], N, lin.in_dim, 0);
                f32x4 acc[2][1];
                zero_acc<1>(acc);
                consume_b<1, 4>(acc, ASrc{in, ldin, in, ldin}, B, hq1, 0, B.T, 16 < N);
                run_epilogue<1>(acc, pair0, N, [&](int row, int col, int c, float v) {
                    if (col < N) lds_st(out + row * ldH + col, act_fwd(v + (c ? hb1b : hb1), akind));
                });
                if (want_grads) wave_store_tile(p.hid + p.hid_off[e][1] + (int64_t)row0 * N, out, ldH, nrows, N);
                in = out; ldin = ldH;
            }
            STAMP();   // A5: h1 done
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const int nn[2] = {32 * pr, 32 * pr + 16};
                STAMP();   // A6..: one u pair
                if (nn[0] < S) {
                    f32x4 acc[2][1];
                    zero_acc<1>(acc);
                    consume_b<1, 4>(acc, ASrc{in, ldin, in, ldin}, BU, uq[pr], BU.T0, BU.T, nn[1] < S);
                    run_epilogue<1>(acc, nn, S, [&](int row, int col, int c, float v) {
                        if (col < S) lds_st(U + row * ldS + col, v + ub[pr][c]);
                    });
                }
            }
            continue;
        }
        {   // x tile -> this wave's LDS image
            const int f4 = round_up(F, 16) >> 2;
            for (int idx = lane; idx < TB * f4; idx += 64) {
                const int row = idx / f4, c = (idx - row * f4) << 2;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < nrows && c < F) {
                    const float* src = xg + (int64_t)row * ldx + c;
                    if (xvec) v = g_ld4(src);
                    else {
                        v.x = g_ld(src);
                        if (c + 1 < F) v.y = g_ld(src + 1);
                        if (c + 2 < F) v.z = g_ld(src + 2);
                        if (c + 3 < F) v.w = g_ld(src + 3);
                    }
                }
                lds_st4(sXw + row * ldX + c, v);
            }
        }
        clp in = sXw;
        int ldin = ldX;
        for (int l = 0; l < Lh; ++l) {                     // mlp_encoder.py:75-76
            const auto& lin = enc.layer[l];
            const int N = lin.out_dim;
            const lp out = sHw[l & 1];
            const float* bias = lin.b;
            const PB B = make_pb(p.pack + p.pkf_off[e][l], N, lin.in_dim, 0);
            const ASrc A{in, ldin, in, ldin};
            wave_layer(A, B, 0, B.T, [&](int row, int col, float v) {
                if (col < N) lds_st(out + row * ldH + col, act_fwd(v + g_ld(bias + col), akind));
            });
            if (want_grads) wave_store_tile(p.hid + p.hid_off[e][l] + (int64_t)row0 * N, out, ldH, nrows, N);
            in = out;
            ldin = ldH;
        }
        {   // h (or x) part of the state update, bias folded in (mlp_encoder.py:78)
            const auto& lin = enc.layer[Lh];
            const float* bias = lin.b;
            const PB B = make_pb(p.pack + p.pkf_off[e][Lh], S, S, HL);
            const ASrc A{in, ldin, in, ldin};
            wave_layer(A, B, B.T0, B.T, [&](int row, int col, float v) {
                if (col < S) lds_st(U + row * ldS + col, v + g_ld(bias + col));
            });
        }
    }
    STAMP();   // P2: phase A done (wave 0)
    __syncthreads();
    STAMP();   // P3: all waves done

    // ---- phase B: the sequential part, s' = W_s s + u_e; W_s fragments run one step ahead
    int cur = 0;
    auto chain_step = [&](f32x4 (&wc)[8][2], f32x4 (&wn)[8][2], int t, int t_nxt) {
        const int e = b.seq_enc[t];
        const auto& enc = p.m.enc[e];
        const int Lh = enc.n_layers - 1;
        const int HL = enc.layer[Lh].in_dim - S;
        const clp sC = St + cur * TB * ldS;
        const lp sN = St + (e + 1) * TB * ldS;
        const clp U = Ut + e * TB * ldS;
        const PB B = make_pb(p.pack + p.pkf_off[e][Lh], S, S, HL);
        float scacc = 0.f;
        f32x4 acc[2][1];
        zero_acc<1>(acc);
        if (n0c[0] < S) consume_b<1, 8>(acc, ASrc{sC, ldS, sC, ldS}, B, wc, 0, B.T0, n0c[1] < S);
        STAMP();   // B1: MFMAs
        if (t_nxt < b.n_seq) {                             // next step's fragments, behind this epilogue
            const int e2 = b.seq_enc[t_nxt];
            const int Lh2 = p.m.enc[e2].n_layers - 1;
            issue_b<8>(wn, make_pb(p.pack + p.pkf_off[e2][Lh2], S, S, p.m.enc[e2].layer[Lh2].in_dim - S), n0c, 0);
        }
        if (n0c[0] < S) {
            run_epilogue<1>(acc, n0c, S, [&](int row, int col, int, float v) {
                if (col < S) {
                    const float ns = v + lds_ld(U + row * ldS + col);
                    const float dlt = ns - lds_ld(sC + row * ldS + col);
                    if (row < nrows) scacc += dlt * dlt;              // multimodn.py:174
                    lds_st(sN + row * ldS + col, ns);
                }
            });
        }
        STAMP();   // B2: next issued + epilogue
        scacc = wave_sum(scacc);
        if (lane == 0) lds_st(sRed + 64 + 4 * e + wave, scacc);
        __syncthreads();
        STAMP();   // B3: barrier
        store_rows(p.states + ((int64_t)e * p.maxB + row0) * S, sN, ldS, nrows, S);   // also forward-only: get_states()
        STAMP();   // B4: state tile stored
        cur = e + 1;
    };
    {
        bool flip = false;
        for (int t = t_first; t < b.n_seq;) {
            const int t_nxt = next_exec(b, t + 1);
            if (!flip) chain_step(wsA, wsB, t, t_nxt); else chain_step(wsB, wsA, t, t_nxt);
            flip = !flip;
            t = t_nxt;
        }
    }

    // ---- phase C: all decoders on all state rows (decoders.py:19-20, multimodn.py:141-157,176-191)
    for (int r = wave; r < R; r += 4) {
        if (!row_executed(b, r)) continue;
        const clp sS = St + r * TB * ldS;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < Tdec) {
                const f32x4 a = lds_ld4(sS + i * ldS + 16 * j + 4 * q);
                z = mfma4(a.x, wd[j].x, z);
                z = mfma4(a.y, wd[j].y, z);
                z = mfma4(a.z, wd[j].z, z);
                z = mfma4(a.w, wd[j].w, z);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) lds_st(sZ + (r * TB + q * 4 + k) * 16 + i, z[k]);
    }
    __syncthreads();
    STAMP();   // decoder logits
    const int total = R * D * TB;
    for (int base = 0; base < total; base += NT) {
        const int idx = base + threadIdx.x;
        const bool valid = idx < total;
        const int r = valid ? idx / (D * TB) : 0;
        const int rem = idx - r * D * TB;
        const int d = valid ? rem / TB : 0, row = rem & (TB - 1);
        const bool live = valid && row < nrows && row_executed(b, r);
        float lossv = 0.f;
        int correct = 0, tp = 0, tn = 0, fp = 0, fn = 0;
        if (live) {
            const float* bd = p.m.dec[d].b;
            const float za = g_ld(bd) + lds_ld(sZ + (r * TB + row) * 16 + 2 * d);
            const float zb = g_ld(bd + 1) + lds_ld(sZ + (r * TB + row) * 16 + 2 * d + 1);
            const int64_t grow = (int64_t)row0 + row;
            const int y = (int)*(const MMN_AS1 int64_t*)(b.y + grow * D + d);
            const float o0 = 1.0f / (1.0f + expf(-za));
            const float o1 = 1.0f / (1.0f + expf(-zb));
            const float mx = fmaxf(o0, o1);
            const float lse = mx + logf(expf(o0 - mx) + expf(o1 - mx));
            lossv = lse - (y ? o1 : o0);
            const int pred = o1 > o0 ? 1 : 0;      // torch.max: first index wins ties
            correct = pred == y;
            tp = pred & y; tn = (1 - pred) & (1 - y); fp = pred & (1 - y); fn = (1 - pred) & y;
            if (want_grads) {
                const float g0 = expf(o0 - lse) - (y == 0 ? 1.0f : 0.0f);
                const float g1 = expf(o1 - lse) - (y == 1 ? 1.0f : 0.0f);
                f32x2 dzv;
                dzv.x = cL * g0 * o0 * (1.0f - o0);
                dzv.y = cL * g1 * o1 * (1.0f - o1);
                g_st2(p.dz + ((int64_t)r * p.maxB + grow) * (2 * D) + 2 * d, dzv);
            } else {                                // forward-only: the decoder OUTPUTS take dz's place
                f32x2 ov; ov.x = o0; ov.y = o1;
                g_st2(p.dz + ((int64_t)r * p.maxB + grow) * (2 * D) + 2 * d, ov);
            }
        }
#pragma unroll
        for (int off = TB / 2; off >= 1; off >>= 1) lossv += __shfl_xor(lossv, off);
        const unsigned long long mc = __ballot(correct), mtp = __ballot(tp), mtn = __ballot(tn),
                                 mfp = __ballot(fp), mfn = __ballot(fn);
        if (valid && row == 0) {
            const int sh = lane & ~(TB - 1);
            const int64_t cell = (int64_t)tile * (R * D) + r * D + d;
            g_st(p.lossp + cell, lossv);
            int32_t* cp = p.cnt + cell * 5;
            g_sti(cp + 0, __popcll((mc >> sh) & 0xFFFFull));
            g_sti(cp + 1, __popcll((mtp >> sh) & 0xFFFFull));
            g_sti(cp + 2, __popcll((mtn >> sh) & 0xFFFFull));
            g_sti(cp + 3, __popcll((mfp >> sh) & 0xFFFFull));
            g_sti(cp + 4, __popcll((mfn >> sh) & 0xFFFFull));
        }
    }
    STAMP();   // decoder grid done
    // ---- epilogue: state-change partials and (for backward) the state tiles
    for (int e = threadIdx.x; e < E; e += NT)
        g_st(p.scp + (int64_t)tile * E + e,
             ((lds_ld(sRed + 64 + 4 * e) + lds_ld(sRed + 65 + 4 * e)) + lds_ld(sRed + 66 + 4 * e)) + lds_ld(sRed + 67 + 4 * e));
    STAMP();   // end
}

__global__ __launch_bounds__(NT) void k_chain_bwd_par(const DevPlan* __restrict__ P, mmn_batch b, float cS) {
    constexpr int TB = 16;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    const int ldS = P->ldS, ldH = P->ldH, ldX = P->ldX, E = P->E;
    const int R = E + 1;
    const ParLds L = par_lds(R, E, ldS, ldH, ldX, true);
    copy_plan_to_lds(P, smem + L.sPlan);
    for (int idx = threadIdx.x; idx < L.total - L.sSt; idx += NT) lds_st(smem + L.sSt + idx, 0.f);
    __syncthreads();
    LPlan& p = *(LPlan*)(smem + L.sPlan);
    const int S = p.S, D = p.D;
    const int tile = blockIdx.x, row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = wave_id();
    const lp DG = smem + L.sSt;          // decoder-grad tiles, then G_out tiles in place
    const lp Df = smem + L.sU;           // s_out - s_in per encoder
    lp sG[2] = {smem + L.sG, smem + L.sG + TB * ldS};
    lp sHw[2] = {smem + L.sW + wave * L.wstride + L.oH0, smem + L.sW + wave * L.wstride + L.oH1};
    const lp sDzw = smem + L.sW + wave * L.wstride + L.oDz;

    // ---- phase A': decoder gradient of every state row and the state differences, one wave each
    {
        const PB Bdz = make_pb(p.pack + p.pkd_off, S, 2 * D, 0);      // W' = Wdec^T [S x 2D]
        for (int r = wave; r < R; r += 4) {
            if (!row_executed(b, r)) continue;
            for (int idx = lane; idx < TB * 16; idx += 64) {
                const int row = idx >> 4, n = idx & 15;
                float v = 0.f;
                if (row < nrows && n < 2 * D) v = g_ld(p.dz + ((int64_t)r * p.maxB + row0 + row) * (2 * D) + n);
                lds_st(sDzw + row * LDZ + n, v);
            }
            const lp out = DG + r * TB * ldS;
            const ASrc A{sDzw, LDZ, sDzw, LDZ};
            wave_layer(A, Bdz, 0, Bdz.T, [&](int row, int col, float v) {
                if (col < S) lds_st(out + row * ldS + col, v);
            });
            if (r >= 1) {
                const int e = r - 1;
                int prev_row = 0;
                for (int t = 0, pr = 0; t < b.n_seq; ++t) {
                    if (!slot_present(b, b.seq_data[t])) continue;
                    if (b.seq_enc[t] == e) prev_row = pr;
                    pr = b.seq_enc[t] + 1;
                }
                const float* so = p.states + ((int64_t)e * p.maxB + row0) * S;
                const float* si = prev_row ? p.states + ((int64_t)(prev_row - 1) * p.maxB + row0) * S : nullptr;
                const lp dd = Df + e * TB * ldS;
                for (int rr = 0; rr < nrows; ++rr)
                    for (int c = lane; c < S; c += 64) {
                        const float a = g_ld(so + (int64_t)rr * S + c);
                        const float d0 = si ? g_ld(si + (int64_t)rr * S + c) : g_ld(p.m.init_state + c);
                        lds_st(dd + rr * ldS + c, a - d0);
                    }
            }
        }
    }
    __syncthreads();

    // ---- phase B': G_out(e) = carry + DG[e+1] + cS d_e ;  carry' = G_out W_s - cS d_e
    int cur = 0;
    for (int t = b.n_seq - 1; t >= 0; --t) {
        if (!slot_present(b, b.seq_data[t])) continue;
        const int e = b.seq_enc[t];
        const int Lh = p.m.enc[e].n_layers - 1;
        const lp Go = DG + (e + 1) * TB * ldS;
        const clp dd = Df + e * TB * ldS;
        const clp G = sG[cur];
        const lp Gn = sG[cur ^ 1];
        for (int rr = wave; rr < TB; rr += 4)
            for (int c = lane; c < S; c += 64)
                lds_st(Go + rr * ldS + c, lds_ld(G + rr * ldS + c) + lds_ld(Go + rr * ldS + c) + cS * lds_ld(dd + rr * ldS + c));
        __syncthreads();
        const int n0[2] = {16 * wave, 16 * (wave + 4)};
        if (n0[0] < S) {
            const PB Bc = make_pb(p.pack + p.pkb_off[e][Lh], S, S, 0);
            f32x4 acc[2][1];
            zero_acc<1>(acc);
            const ASrc A{Go, ldS, Go, ldS};
            wave_gemm_any<1>(acc, A, Bc, n0, 0, Bc.T);
            run_epilogue<1>(acc, n0, S, [&](int row, int col, int, float v) {
                if (col < S) lds_st(Gn + row * ldS + col, v - cS * lds_ld(dd + row * ldS + col));
            });
        }
        __syncthreads();
        cur ^= 1;
    }
    {   // row 0: dS0 = carry + DG[0]
        const clp G = sG[cur];
        for (int rr = wave; rr < TB; rr += 4)
            for (int c = lane; c < S; c += 64) lds_st(DG + rr * ldS + c, lds_ld(G + rr * ldS + c) + lds_ld(DG + rr * ldS + c));
    }
    __syncthreads();

    // ---- phase C': hidden-layer backward of every executed encoder, one wave per encoder
    for (int t = wave; t < b.n_seq; t += 4) {
        if (!slot_present(b, b.seq_data[t])) continue;
        const int e = b.seq_enc[t];
        const auto& enc = p.m.enc[e];
        const int Lh = enc.n_layers - 1, akind = enc.activation;
        if (Lh == 0) continue;
        const int HL = enc.layer[Lh].in_dim - S;
        const clp Go = DG + (e + 1) * TB * ldS;
        {
            const PB Bh = make_pb(p.pack + p.pkh_off[e], HL, S, 0);
            const ASrc A{Go, ldS, Go, ldS};
            const lp dh = sHw[0];
            wave_layer(A, Bh, 0, Bh.T, [&](int row, int col, float v) {
                if (col < HL) lds_st(dh + row * ldH + col, v);
            });
        }
        for (int l = Lh - 1; l >= 0; --l) {
            const int Hl = enc.layer[l].out_dim;
            const lp cbuf = sHw[(Lh - 1 - l) & 1];
            const float* hid_g = p.hid + p.hid_off[e][l] + (int64_t)row0 * Hl;
            float* dpre_g = p.dpre + p.hid_off[e][l] + (int64_t)row0 * Hl;
            for (int r = 0; r < TB; ++r)
                for (int c = lane; c < Hl; c += 64) {
                    float dp = 0.f;
                    if (r < nrows) {
                        dp = lds_ld(cbuf + r * ldH + c) * act_grad_from_out(g_ld(hid_g + (int64_t)r * Hl + c), akind);
                        g_st(dpre_g + (int64_t)r * Hl + c, dp);
                    }
                    lds_st(cbuf + r * ldH + c, dp);
                }
            if (l == 0) break;
            const int Hp = enc.layer[l].in_dim;
            const lp nbuf = sHw[(Lh - l) & 1];
            const ASrc A{cbuf, ldH, cbuf, ldH};
            const PB B = make_pb(p.pack + p.pkb_off[e][l], Hp, Hl, 0);
            wave_layer(A, B, 0, B.T, [&](int row, int col, float v) {
                if (col < Hp) lds_st(nbuf + row * ldH + col, v);
            });
        }
    }
    // ---- stores: dS[e] = G_out(e), dS[E] = dS0 (DG tiles are final since the barrier above)
    for (int r = 0; r < R; ++r) {
        if (!row_executed(b, r)) continue;
        const int idx = r == 0 ? E : r - 1;
        store_rows(p.dS + ((int64_t)idx * p.maxB + row0) * S, DG + r * TB * ldS, ldS, nrows, S);
    }
}

// ------------------------------------------------------------------------------------------------
// Fast tier: 8-wave parallel-phase chain kernels for MIMIC-like shapes
//   E <= 8, D <= 8, S <= 128, <= 2 hidden layers of width <= 32, n_features <= 128 (16-byte
//   aligned rows), h width into the state update <= 64.
// Same phase structure as the 4-wave parallel kernels above, but (a) the model descriptor arrives
// through kernel arguments (scalar loads, no per-lane LDS reads of the plan), (b) 512 threads =
// 2 waves per SIMD, so one wave's s_waitcnt / barrier time is covered by its partner (the 4-wave
// form was parked 52 % of its cycles), with the work cut finer: one 16-column tile per wave in
// the chain, two waves per encoder in the parallel phases, (c) every operand is requested ahead.
// ------------------------------------------------------------------------------------------------
constexpr int NT8 = 512;

struct ParEnc {
    int32_t F, Lh, HL, akind;
    int32_t in[3], out[3];              // layer dims; layer Lh is the state update (in = HL + S, out = S)
    int32_t pad0[2];
    int64_t pkf[3], pkb[3], pkh;        // float offsets into the pack
    int64_t hid[2];                     // float offsets into hid / dpre
    const float* bias[3];
};
struct ParArgs {
    int32_t S, E, D, R, S16, ldS, ldH, ldX, maxB, needs_zero, pad0[2];
    const float* init; const float* pack;
    int64_t pkd;
    float* states; float* hid; float* dpre; float* dz; float* dS;
    float* sin;                         // per-sample mode: the state that fed encoder e, [E][maxB][S]
    float* lossp; float* scp; int32_t* cnt; int32_t* exec_flags; int32_t* prev_row;
    long long* stamps;
    const float* dec_w[MMN_MAX_DECODERS]; const float* dec_b[MMN_MAX_DECODERS];
    ParEnc enc[8];
};

struct Par8Lds { int sSt, sU, sG, sW, wstride, oX, oH0, oH1, sDz, sZ, sRed, total; };
__host__ __device__ inline Par8Lds par8_lds(int R, int E, int ldS, int ldH, int ldX, bool bwd) {
    Par8Lds L;
    int o = 0;
    L.sSt = o; o += R * 16 * ldS;          // fwd: state tiles 0..E        bwd: decoder-grad -> G_out tiles
    L.sU = o; o += E * 16 * ldS;           // fwd: u_e tiles               bwd: state differences
    L.sG = o; o += bwd ? 2 * 16 * ldS : 0;
    L.oX = 0; L.oH0 = bwd ? 0 : 16 * ldX; L.oH1 = L.oH0 + 16 * ldH;
    L.wstride = L.oH1 + 16 * ldH;
    L.sW = o; o += 4 * L.wstride;          // scratch per wave PAIR: x tile / hidden ping-pong
    L.sDz = o; o += bwd ? 8 * 16 * LDZ : 0;  // dz tile per wave
    L.sZ = o; o += bwd ? 0 : R * 16 * 16;
    L.sRed = o; o += 8 * 8 + 16;
    L.total = o;
    return L;
}

// NS k-steps of ONE column tile
// (tile, T, t_begin are wave-uniform: each fragment address is a scalar base plus the lane's 16 bytes,
// so a request costs two scalar ops and one vector-memory instruction)
template <int NS>
__device__ __forceinline__ void issue_t(f32x4 (&bq)[NS], const float* pk, int T, int ntiles, int tile, int t_begin) {
    const int lane = threadIdx.x & 63;
    const int tl = __builtin_amdgcn_readfirstlane(min(tile, ntiles - 1));
    const int Tu = __builtin_amdgcn_readfirstlane(T), tb = __builtin_amdgcn_readfirstlane(t_begin);
    const float* p0 = pk + (int64_t)tl * Tu * 256;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float* pt = p0 + min(tb + j, Tu - 1) * 256;    // scalar
        bq[j] = g_ld4(pt + lane * 4);
    }
}
// acc += A[16 x 16*(t_end-t_begin)] x fragment registers; A image row stride lda, step t reads a + 16 (t - t_base)
template <int NS>
__device__ __forceinline__ void consume_t(f32x4& acc, clp a, int lda, int t_base, const f32x4 (&bq)[NS], int t_begin, int t_end) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    clp ap = a + i * lda + 4 * q;
    f32x4 av[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) av[j] = lds_ld4(ap + 16 * (min(t_begin + j, t_end - 1) - t_base));
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        if (t_begin + j < t_end) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma4(av[j][e], bq[j][e], acc);
        }
    }
}
#define STAMP8() stamp(nullptr, a.stamps, stamp_k, stamp_block)

__device__ __forceinline__ int enc_of(const mmn_batch& b, int t) { return b.seq_enc[t]; }

__global__ __launch_bounds__(NT8) void k_fwd8(const ParArgs a, const mmn_batch b, float cL, int want_grads) {
    constexpr int TB = 16;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    const unsigned pm = present_mask(b);
    const int S = a.S, E = a.E, D = a.D, R = a.R, ldS = a.ldS, ldH = a.ldH, ldX = a.ldX;
    const Par8Lds L = par8_lds(R, E, ldS, ldH, ldX, false);
    const int tile = blockIdx.x, row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15, q = lane >> 4;
    const int g = wave >> 1, half = wave & 1;
    const lp St = smem + L.sSt;
    const lp Ut = smem + L.sU;
    const lp sZ = smem + L.sZ;
    const lp sRed = smem + L.sRed;
    const lp sXg = smem + L.sW + g * L.wstride + L.oX;
    lp sHg[2] = {smem + L.sW + g * L.wstride + L.oH0, smem + L.sW + g * L.wstride + L.oH1};
    int stamp_k = 0;
    const int stamp_block = 7;
    STAMP8();
    if (a.needs_zero) {                                    // K-padding columns must be finite
        for (int idx = threadIdx.x; idx < L.total; idx += NT8) lds_st(smem + idx, 0.f);
        __syncthreads();
    }
    for (int r = wave; r < TB; r += 8)                     // state row 0 = init state (state.py:29-32)
        for (int c = lane; c < S; c += 64) lds_st(St + r * ldS + c, g_ld(a.init + c));
    if (tile == 0 && threadIdx.x == 0) {                   // which state rows exist this step
        g_sti(a.exec_flags, 1);
        for (int e = 0; e < E; ++e) g_sti(a.exec_flags + e + 1, 0);
        int prev = 0;
        for (int t = 0; t < b.n_seq; ++t) {
            if (!slot_present(pm, b.seq_data[t])) continue;
            const int e = b.seq_enc[t];
            g_sti(a.exec_flags + e + 1, 1);
            g_sti(a.prev_row + e, prev);
            prev = e + 1;
        }
    }
    const int ntS = (S + 15) >> 4, T0 = a.S16 >> 4;
    // decoder fragments: wave r evaluates state row r (R <= 8)
    f32x4 wd[8];
    {
        const bool rowok = i < 2 * D;
        const float* w = a.dec_w[rowok ? (i >> 1) : 0] + (rowok ? (i & 1) * S : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {                      // clamped address + select: no branch around a load
            const int k = 16 * j + 4 * q;
            const bool ok = rowok && k < S && j < T0;
            const f32x4 v = g_ld4(w + (ok ? k : 0));
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            wd[j] = ok ? v : z;
        }
    }
    // targets / decoder biases of this thread's (row r, decoder d, batch row) triple, first pass
    int y_pre = 0;
    float bd0_pre = 0.f, bd1_pre = 0.f;
    {
        const int idx = threadIdx.x;
        const bool valid = idx < R * D * TB;
        const int r = valid ? idx / (D * TB) : 0;
        const int rem = idx - r * D * TB;
        const int d = valid ? rem / TB : 0, row = rem & (TB - 1);
        const bool ok = valid && row < nrows;
        y_pre = (int)*(const MMN_AS1 int64_t*)(b.y + (ok ? ((int64_t)row0 + row) * D + d : 0));
        bd0_pre = g_ld(a.dec_b[d]);
        bd1_pre = g_ld(a.dec_b[d] + 1);
    }
    // chain step 0's W_s fragments (tile = wave) land during phase A
    f32x4 wsA[8], wsB[8];
    const int t_first = next_exec(b, pm, 0);
    if (t_first < b.n_seq && wave < ntS) {
        const ParEnc& pe = a.enc[b.seq_enc[t_first]];
        issue_t<8>(wsA, a.pack + pe.pkf[pe.Lh], T0 + ((pe.HL + 15) >> 4), ntS, wave, 0);
    }
    STAMP8();

    // ---- phase A: u_e = W_x h_e + b for every executed encoder; wave pair g takes encoder t = tb + g
    for (int tb = 0; tb < b.n_seq; tb += 4) {
        const int t = tb + g;
        const bool act = t < b.n_seq && slot_present(pm, b.seq_data[min(t, b.n_seq - 1)]);
        const int e = act ? b.seq_enc[t] : 0;
        const ParEnc& pe = a.enc[e];
        const int Lh = pe.Lh, F = pe.F, HL = pe.HL, akind = pe.akind;
        int Lmax = 0;                                       // barrier count must be uniform over the workgroup
        for (int k = 0; k < 4; ++k)
            if (tb + k < b.n_seq && slot_present(pm, b.seq_data[tb + k])) Lmax = max(Lmax, a.enc[b.seq_enc[tb + k]].Lh);
        const int f4 = round_up(F, 16) >> 2;               // float4 per image row (<= 32)
        const int TU = (HL + 15) >> 4;
        const float* pkU = a.pack + pe.pkf[Lh];
        // -- request everything this wave will consume in this round
        f32x4 xr[4], h0q[8], h1q[2], uq[2][4][2];
        float hb0 = 0.f, hb1 = 0.f, ub[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        if (act) {
            const int slot = b.seq_data[t];
            const float* xg = b.x[slot] + (int64_t)row0 * b.ldx[slot];
            const int64_t ldx = b.ldx[slot];
#pragma unroll
            for (int k = 0; k < 4; ++k) {                  // this half's 8 rows of the x tile
                const int idx = lane + 64 * k;
                const int row = 8 * half + idx / f4, c = (idx % f4) << 2;
                const bool ok = idx < 8 * f4 && row < nrows && c < F;
                const f32x4 v = g_ld4(xg + (ok ? (int64_t)row * ldx + c : 0));
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                xr[k] = ok ? v : z;
            }
            if (Lh >= 1) {
                issue_t<8>(h0q, a.pack + pe.pkf[0], (pe.in[0] + 15) >> 4, (pe.out[0] + 15) >> 4, half, 0);
                hb0 = g_ld(pe.bias[0] + min(16 * half + i, pe.out[0] - 1));
            }
            if (Lh >= 2) {
                issue_t<2>(h1q, a.pack + pe.pkf[1], (pe.in[1] + 15) >> 4, (pe.out[1] + 15) >> 4, half, 0);
                hb1 = g_ld(pe.bias[1] + min(16 * half + i, pe.out[1] - 1));
            }
            const PB BU = make_pb(pkU, S, S, HL);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int nn[2] = {16 * (4 * half + 2 * pr), 16 * (4 * half + 2 * pr + 1)};
                if (nn[0] < S) {
                    issue_b<4>(uq[pr], BU, nn, BU.T0);
                    ub[pr][0] = g_ld(pe.bias[Lh] + min(nn[0] + i, S - 1));
                    ub[pr][1] = g_ld(pe.bias[Lh] + min(nn[1] + i, S - 1));
                }
            }
            // -- x -> the pair's LDS image
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k;
                if (idx < 8 * f4) lds_st4(sXg + (8 * half + idx / f4) * ldX + ((idx % f4) << 2), xr[k]);
            }
        }
        __syncthreads();
        if (Lmax >= 1) {                                   // hidden layer 0 (mlp_encoder.py:75-76), tile = half
            if (act && Lh >= 1 && 16 * half < pe.out[0]) {
                const int N = pe.out[0], T = (pe.in[0] + 15) >> 4;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                consume_t<8>(acc, sXg, ldX, 0, h0q, 0, T);
                const int col = 16 * half + i;
                if (col < N) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) lds_st(sHg[0] + (4 * q + k) * ldH + col, act_fwd(acc[k] + hb0, akind));
                }
            }
            __syncthreads();
            if (act && Lh >= 1 && want_grads) {            // each half stores 8 rows of the tile
                const int N = pe.out[0];
                const int r0 = 8 * half, nr = max(0, min(8, nrows - r0));
                wave_store_tile(a.hid + pe.hid[0] + (int64_t)(row0 + r0) * N, sHg[0] + r0 * ldH, ldH, nr, N);
            }
        }
        if (Lmax >= 2) {
            if (act && Lh >= 2 && 16 * half < pe.out[1]) {
                const int N = pe.out[1], T = (pe.in[1] + 15) >> 4;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                consume_t<2>(acc, sHg[0], ldH, 0, h1q, 0, T);
                const int col = 16 * half + i;
                if (col < N) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) lds_st(sHg[1] + (4 * q + k) * ldH + col, act_fwd(acc[k] + hb1, akind));
                }
            }
            __syncthreads();
            if (act && Lh >= 2 && want_grads) {
                const int N = pe.out[1];
                const int r0 = 8 * half, nr = max(0, min(8, nrows - r0));
                wave_store_tile(a.hid + pe.hid[1] + (int64_t)(row0 + r0) * N, sHg[1] + r0 * ldH, ldH, nr, N);
            }
        }
        if (act) {                                         // u_e = W_x h + b (mlp_encoder.py:78), this half's 4 tiles
            clp in = Lh == 0 ? (clp)sXg : (clp)sHg[Lh - 1];
            const int ldin = Lh == 0 ? ldX : ldH;
            const lp U = Ut + e * TB * ldS;
            const PB BU = make_pb(pkU, S, S, HL);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int nn[2] = {16 * (4 * half + 2 * pr), 16 * (4 * half + 2 * pr + 1)};
                if (nn[0] < S) {
                    f32x4 acc[2][1];
                    zero_acc<1>(acc);
                    consume_b<1, 4>(acc, ASrc{in, ldin, in, ldin}, BU, uq[pr], BU.T0, BU.T0 + TU, nn[1] < S);
                    run_epilogue<1>(acc, nn, S, [&](int row, int col, int c, float v) {
                        if (col < S) lds_st(U + row * ldS + col, v + ub[pr][c]);
                    });
                }
            }
        }
        __syncthreads();                                    // scratch is reused by the next round
    }
    STAMP8();

    // ---- phase B: s' = W_s s + u_e; one column tile per wave, fragments one step ahead
    int cur = 0;
    auto chain_step = [&](f32x4 (&wc)[8], f32x4 (&wn)[8], int t, int t_nxt) {
        const int e = b.seq_enc[t];
        const clp sC = St + cur * TB * ldS;
        const lp sN = St + (e + 1) * TB * ldS;
        const clp U = Ut + e * TB * ldS;
        float scacc = 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (wave < ntS) consume_t<8>(acc, sC, ldS, 0, wc, 0, T0);
        if (t_nxt < b.n_seq && wave < ntS) {
            const ParEnc& p2 = a.enc[b.seq_enc[t_nxt]];
            issue_t<8>(wn, a.pack + p2.pkf[p2.Lh], T0 + ((p2.HL + 15) >> 4), ntS, wave, 0);
        }
        const int col = 16 * wave + i;
        if (wave < ntS && col < S) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = 4 * q + k;
                const float ns = acc[k] + lds_ld(U + row * ldS + col);
                const float dlt = ns - lds_ld(sC + row * ldS + col);
                if (row < nrows) scacc += dlt * dlt;              // multimodn.py:174
                lds_st(sN + row * ldS + col, ns);
            }
        }
        scacc = wave_sum(scacc);
        if (lane == 0) lds_st(sRed + 8 * e + wave, scacc);
        __syncthreads();
        {                                                   // each wave stores two rows of the new state tile
            for (int r = wave; r < nrows; r += 8)               // (also forward-only: get_states())
                for (int c = lane * 4; c < S; c += 256) {
                    if (((S & 3) == 0)) g_st4(a.states + ((int64_t)e * a.maxB + row0 + r) * S + c, lds_ld4(sN + r * ldS + c));
                    else for (int k = 0; k < 4 && c + k < S; ++k) g_st(a.states + ((int64_t)e * a.maxB + row0 + r) * S + c + k, lds_ld(sN + r * ldS + c + k));
                }
        }
        cur = e + 1;
    };
    {
        bool flip = false;
        for (int t = t_first; t < b.n_seq;) {
            const int t_nxt = next_exec(b, pm, t + 1);
            if (!flip) chain_step(wsA, wsB, t, t_nxt); else chain_step(wsB, wsA, t, t_nxt);
            flip = !flip;
            t = t_nxt;
        }
    }
    STAMP8();

    // ---- phase C: all decoders on all state rows (decoders.py:19-20, multimodn.py:141-157,176-191)
    if (wave < R && row_executed(b, pm, wave)) {
        const clp sS = St + wave * TB * ldS;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        consume_t<8>(z, sS, ldS, 0, wd, 0, T0);
#pragma unroll
        for (int k = 0; k < 4; ++k) lds_st(sZ + (wave * TB + q * 4 + k) * 16 + i, z[k]);
    }
    __syncthreads();
    const int total = R * D * TB;
    for (int base = 0; base < total; base += NT8) {
        const int idx = base + threadIdx.x;
        const bool valid = idx < total;
        const int r = valid ? idx / (D * TB) : 0;
        const int rem = idx - r * D * TB;
        const int d = valid ? rem / TB : 0, row = rem & (TB - 1);
        const bool live = valid && row < nrows && row_executed(b, pm, r);
        float lossv = 0.f;
        int correct = 0, tp = 0, tn = 0, fp = 0, fn = 0;
        if (live) {
            const float* bd = a.dec_b[d];
            const float za = (base == 0 ? bd0_pre : g_ld(bd)) + lds_ld(sZ + (r * TB + row) * 16 + 2 * d);
            const float zb = (base == 0 ? bd1_pre : g_ld(bd + 1)) + lds_ld(sZ + (r * TB + row) * 16 + 2 * d + 1);
            const int64_t grow = (int64_t)row0 + row;
            const int y = base == 0 ? y_pre : (int)*(const MMN_AS1 int64_t*)(b.y + grow * D + d);
            const float o0 = 1.0f / (1.0f + expf(-za));
            const float o1 = 1.0f / (1.0f + expf(-zb));
            const float mx = fmaxf(o0, o1);
            const float lse = mx + logf(expf(o0 - mx) + expf(o1 - mx));
            lossv = lse - (y ? o1 : o0);
            const int pred = o1 > o0 ? 1 : 0;      // torch.max: first index wins ties
            correct = pred == y;
            tp = pred & y; tn = (1 - pred) & (1 - y); fp = pred & (1 - y); fn = (1 - pred) & y;
            if (want_grads) {
                const float g0 = expf(o0 - lse) - (y == 0 ? 1.0f : 0.0f);
                const float g1 = expf(o1 - lse) - (y == 1 ? 1.0f : 0.0f);
                f32x2 dzv;
                dzv.x = cL * g0 * o0 * (1.0f - o0);
                dzv.y = cL * g1 * o1 * (1.0f - o1);
                g_st2(a.dz + ((int64_t)r * a.maxB + grow) * (2 * D) + 2 * d, dzv);
            } else {                                // forward-only: the decoder OUTPUTS take dz's place
                f32x2 ov; ov.x = o0; ov.y = o1;
                g_st2(a.dz + ((int64_t)r * a.maxB + grow) * (2 * D) + 2 * d, ov);
            }
        }
#pragma unroll
        for (int off = TB / 2; off >= 1; off >>= 1) lossv += __shfl_xor(lossv, off);
        const unsigned long long mc = __ballot(correct), mtp = __ballot(tp), mtn = __ballot(tn),
                                 mfp = __ballot(fp), mfn = __ballot(fn);
        if (valid && row == 0) {
            const int sh = lane & ~(TB - 1);
            const int64_t cell = (int64_t)tile * (R * D) + r * D + d;
            g_st(a.lossp + cell, lossv);
            int32_t* cp = a.cnt + cell * 5;
            g_sti(cp + 0, __popcll((mc >> sh) & 0xFFFFull));
            g_sti(cp + 1, __popcll((mtp >> sh) & 0xFFFFull));
            g_sti(cp + 2, __popcll((mtn >> sh) & 0xFFFFull));
            g_sti(cp + 3, __popcll((mfp >> sh) & 0xFFFFull));
            g_sti(cp + 4, __popcll((mfn >> sh) & 0xFFFFull));
        }
    }
    for (int e = threadIdx.x; e < E; e += NT8) {            // state-change partials, fixed order
        float s = 0.f;
        for (int w = 0; w < 8; ++w) s += lds_ld(sRed + 8 * e + w);
        g_st(a.scp + (int64_t)tile * E + e, s);
    }
    STAMP8();
}

__global__ __launch_bounds__(NT8) void k_bwd8(const ParArgs a, const mmn_batch b, float cS) {
    constexpr int TB = 16;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    const unsigned pm = present_mask(b);
    const int S = a.S, E = a.E, D = a.D, R = a.R, ldS = a.ldS, ldH = a.ldH, ldX = a.ldX;
    const Par8Lds L = par8_lds(R, E, ldS, ldH, ldX, true);
    const int tile = blockIdx.x, row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15, q = lane >> 4;
    const int g = wave >> 1, half = wave & 1;
    const lp DG = smem + L.sSt;          // decoder-grad tiles, then G_out tiles in place
    const lp Df = smem + L.sU;           // s_out - s_in per encoder
    lp sHg[2] = {smem + L.sW + g * L.wstride + L.oH0, smem + L.sW + g * L.wstride + L.oH1};
    const lp sDzw = smem + L.sDz + wave * 16 * LDZ;
    const int ntS = (S + 15) >> 4, T0 = a.S16 >> 4;
    if (a.needs_zero) {
        for (int idx = threadIdx.x; idx < L.total; idx += NT8) lds_st(smem + idx, 0.f);
        __syncthreads();
    }
    // first chain step's W_s^T fragments (tile = wave)
    f32x4 wcA[8], wcB[8];
    int t_last = b.n_seq - 1;
    while (t_last >= 0 && !slot_present(pm, b.seq_data[t_last])) --t_last;
    if (t_last >= 0 && wave < ntS) {
        const ParEnc& pe = a.enc[b.seq_enc[t_last]];
        issue_t<8>(wcA, a.pack + pe.pkb[pe.Lh], T0, ntS, wave, 0);
    }

    // ---- phase A': wave r owns state row r: d_e = s_out - s_in (e = r - 1), then
    //      DG[r] = dz[r] Wdec + cS d_e  (everything the chain adds to the carried gradient at row r)
    if (wave < R && row_executed(b, pm, wave)) {
        const int r = wave;
        const lp out = DG + r * TB * ldS;
        // requests first, all branch-free
        float dzr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * k;
            const int row = idx >> 4, n = idx & 15;
            const bool ok = row < nrows && n < 2 * D;
            const float v = g_ld(a.dz + (ok ? ((int64_t)r * a.maxB + row0 + row) * (2 * D) + n : 0));
            dzr[k] = ok ? v : 0.f;
        }
        const int e = r - 1;
        f32x4 so4[8], si4[8];
        const int s4 = a.S16 >> 2;                          // float4 per state row (<= 32)
        if (r >= 1) {
            int prev_row = 0;
            for (int u = 0, pr = 0; u < b.n_seq; ++u) {
                if (!slot_present(pm, b.seq_data[u])) continue;
                if (b.seq_enc[u] == e) prev_row = pr;
                pr = b.seq_enc[u] + 1;
            }
            const float* so = a.states + ((int64_t)e * a.maxB + row0) * S;
            const float* si = prev_row ? a.states + ((int64_t)(prev_row - 1) * a.maxB + row0) * S : a.init;
            const int64_t si_ld = prev_row ? S : 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = lane + 64 * k;
                const int row = idx / s4, c = (idx - row * s4) << 2;
                const bool ok = idx < TB * s4 && row < nrows && c < S;
                so4[k] = g_ld4(so + (ok ? (int64_t)row * S + c : 0));
                si4[k] = g_ld4(si + (ok ? (int64_t)row * si_ld + c : 0));
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (!ok) { so4[k] = z; si4[k] = z; }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * k;
            lds_st(sDzw + (idx >> 4) * LDZ + (idx & 15), dzr[k]);
        }
        const lp dd = Df + (r >= 1 ? e : 0) * TB * ldS;
        if (r >= 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = lane + 64 * k;
                if (idx < TB * s4) lds_st4(dd + (idx / s4) * ldS + ((idx % s4) << 2), so4[k] - si4[k]);
            }
        }
        const PB Bdz = make_pb(a.pack + a.pkd, S, 2 * D, 0);      // W' = Wdec^T [S x 2D]
        wave_layer(ASrc{sDzw, LDZ, sDzw, LDZ}, Bdz, 0, Bdz.T, [&](int row, int col, float v) {
            if (col < S) lds_st(out + row * ldS + col, r >= 1 ? v + cS * lds_ld(dd + row * ldS + col) : v);
        });
    }
    __syncthreads();

    // ---- phase B': carry' = G_out W_s - cS d_e, and the NEXT row's G_out is formed in the same
    //      epilogue (in place, in DG): one barrier per chain step
    {
        auto chain_step = [&](f32x4 (&wc)[8], f32x4 (&wn)[8], int t, int t_prv) {
            const int e = b.seq_enc[t];
            const clp Go = DG + (e + 1) * TB * ldS;
            const clp dd = Df + e * TB * ldS;
            const int r_nxt = t_prv >= 0 ? b.seq_enc[t_prv] + 1 : 0;
            const lp Gx = DG + r_nxt * TB * ldS;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (wave < ntS) consume_t<8>(acc, Go, ldS, 0, wc, 0, T0);
            if (t_prv >= 0 && wave < ntS) {
                const ParEnc& p2 = a.enc[b.seq_enc[t_prv]];
                issue_t<8>(wn, a.pack + p2.pkb[p2.Lh], T0, ntS, wave, 0);
            }
            const int col = 16 * wave + i;
            if (wave < ntS && col < S) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = 4 * q + k;
                    const float carry = acc[k] - cS * lds_ld(dd + row * ldS + col);
                    lds_st(Gx + row * ldS + col, lds_ld(Gx + row * ldS + col) + carry);
                }
            }
            __syncthreads();
        };
        bool flip = false;
        for (int t = t_last; t >= 0;) {
            int t_prv = t - 1;
            while (t_prv >= 0 && !slot_present(pm, b.seq_data[t_prv])) --t_prv;
            if (!flip) chain_step(wcA, wcB, t, t_prv); else chain_step(wcB, wcA, t, t_prv);
            flip = !flip;
            t = t_prv;
        }
    }
    // ---- stores: dS[e] = G_out(e), dS[E] = dS0 (final since the last barrier); fire and forget
    for (int r = 0; r < R; ++r) {
        if (!row_executed(b, pm, r)) continue;
        const int idx = r == 0 ? E : r - 1;
        for (int rr = wave; rr < nrows; rr += 8)
            for (int c = lane * 4; c < S; c += 256) {
                if ((S & 3) == 0) g_st4(a.dS + ((int64_t)idx * a.maxB + row0 + rr) * S + c, lds_ld4(DG + (r * TB + rr) * ldS + c));
                else for (int k = 0; k < 4 && c + k < S; ++k) g_st(a.dS + ((int64_t)idx * a.maxB + row0 + rr) * S + c + k, lds_ld(DG + (r * TB + rr) * ldS + c + k));
            }
    }

    // ---- phase C': hidden-layer backward; wave pair g takes encoder t = tb + g, halves split the tiles
    for (int tb = 0; tb < b.n_seq; tb += 4) {
        const int t = tb + g;
        const bool act0 = t < b.n_seq && slot_present(pm, b.seq_data[min(t, b.n_seq - 1)]);
        const int e = act0 ? b.seq_enc[t] : 0;
        const ParEnc& pe = a.enc[e];
        const int Lh = pe.Lh, HL = pe.HL, akind = pe.akind;
        const bool act = act0 && Lh >= 1;
        int Lmax = 0;
        for (int k = 0; k < 4; ++k)
            if (tb + k < b.n_seq && slot_present(pm, b.seq_data[tb + k])) Lmax = max(Lmax, a.enc[b.seq_enc[tb + k]].Lh);
        if (Lmax == 0) continue;
        const clp Go = DG + (e + 1) * TB * ldS;
        // dh = G_out W_x : tile = half of the <= 2 tiles of HL... HL may be up to 64 -> tiles half, half+2
        f32x4 hq[2][8];
        f32x4 h1q[2];
        if (act) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (16 * (half + 2 * k) < HL) issue_t<8>(hq[k], a.pack + pe.pkh, T0, (HL + 15) >> 4, half + 2 * k, 0);
            if (Lh >= 2 && 16 * half < pe.in[1])
                issue_t<2>(h1q, a.pack + pe.pkb[1], (pe.out[1] + 15) >> 4, (pe.in[1] + 15) >> 4, half, 0);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int tl = half + 2 * k;
                if (16 * tl < HL) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    consume_t<8>(acc, Go, ldS, 0, hq[k], 0, T0);
                    const int col = 16 * tl + i;
                    if (col < HL) {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) lds_st(sHg[0] + (4 * q + kk) * ldH + col, acc[kk]);
                    }
                }
            }
        }
        __syncthreads();
        for (int l = Lmax - 1; l >= 0; --l) {
            // dpre_l = dh_l .* act'(h_l): each half takes 8 rows
            const int lb = Lh - 1 - (Lmax - 1 - l);        // this encoder's layer at this depth (aligned at the top)
            if (act && lb >= 0) {
                const int Hl = pe.out[lb];
                const lp cbuf = sHg[(Lh - 1 - lb) & 1];
                const float* hid_g = a.hid + pe.hid[lb] + (int64_t)row0 * Hl;
                float* dpre_g = a.dpre + pe.hid[lb] + (int64_t)row0 * Hl;
                for (int r = 8 * half; r < 8 * half + 8; ++r)
                    for (int c = lane; c < Hl; c += 64) {
                        float dp = 0.f;
                        if (r < nrows) {
                            dp = lds_ld(cbuf + r * ldH + c) * act_grad_from_out(g_ld(hid_g + (int64_t)r * Hl + c), akind);
                            g_st(dpre_g + (int64_t)r * Hl + c, dp);
                        }
                        lds_st(cbuf + r * ldH + c, dp);
                    }
            }
            __syncthreads();
            if (l == 0) break;
            if (act && lb >= 1) {                           // dh_{lb-1} = dpre_lb W_lb, tile = half
                const int Hp = pe.in[lb], Hl = pe.out[lb];
                if (16 * half < Hp) {
                    const lp cbuf = sHg[(Lh - 1 - lb) & 1];
                    const lp nbuf = sHg[(Lh - lb) & 1];
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    consume_t<2>(acc, cbuf, ldH, 0, h1q, 0, (Hl + 15) >> 4);
                    const int col = 16 * half + i;
                    if (col < Hp) {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) lds_st(nbuf + (4 * q + kk) * ldH + col, acc[kk]);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_fb8: forward AND backward chain of one 16-row tile in ONE launch (8-wave tier, E <= 4).
// The backward chain of a row tile only needs that tile's own forward results, so the state
// tiles, dz and the hidden activations never leave LDS between the two halves: no second kernel
// prologue, no global round trip for states / dz / hid, act' fused into the GEMM epilogues.
// Global stores of states / hid / dz / dS / dpre remain (k_wgrad consumes them), fire and forget.
// ------------------------------------------------------------------------------------------------
struct Fb8Lds { int sSt, sU, sW, wstride, oX, oH0, oH1, sDzA, sZ, sRed, total; };
__host__ __device__ inline Fb8Lds fb8_lds(int R, int ldS, int ldH, int ldX) {
    Fb8Lds L;
    int o = 0;
    L.sSt = o; o += R * 16 * ldS;          // state tiles 0..E (live for the whole kernel)
    L.sU = o; o += R * 16 * ldS;           // forward: u_e tiles; backward: DG -> G_out tiles
    L.oX = 0; L.oH0 = 16 * ldX; L.oH1 = L.oH0 + 16 * ldH;
    L.wstride = L.oH1 + 16 * ldH;
    L.sW = o; o += 4 * L.wstride;          // per wave pair: x tile, h0, h1 (kept for the backward half)
    L.sDzA = o; o += R * 16 * LDZ;         // dz tiles of all rows
    L.sZ = o; o += R * 16 * 16;
    L.sRed = o; o += 8 * 8 + 16;
    L.total = o;
    return L;
}

// Executed-sequence summary, wave-uniform (SGPRs), computed once per kernel from the kernarg
// sequence and the NaN mask: everything the phases ask ("is row r live", "which row fed encoder
// e", "k-th executed encoder") becomes a shift and a mask instead of a loop of scalar loads.
struct ExecInfo {
    unsigned rowmask;                 // bit r: state row r exists this step (bit 0 always)
    int n;                            // executed encoders
    unsigned long long enc, slot;     // 4 bits per executed step k: encoder id / data slot
    unsigned long long prev;          // 4 bits per state row r >= 1: the row that fed encoder r-1
    unsigned long long next;          // 4 bits per state row r: the row it feeds (0 = none)
    __device__ __forceinline__ int e(int k) const { return (int)((enc >> (4 * k)) & 15ull); }
    __device__ __forceinline__ int s(int k) const { return (int)((slot >> (4 * k)) & 15ull); }
    __device__ __forceinline__ int prev_row(int r) const { return (int)((prev >> (4 * r)) & 15ull); }
    __device__ __forceinline__ int next_row(int r) const { return (int)((next >> (4 * r)) & 15ull); }
    __device__ __forceinline__ bool row(int r) const { return (rowmask >> r) & 1u; }
};
__device__ __forceinline__ ExecInfo exec_info(const mmn_batch& b, unsigned pm) {
    ExecInfo x;
    x.rowmask = 1u; x.n = 0; x.enc = 0ull; x.slot = 0ull; x.prev = 0ull; x.next = 0ull;
    int prev = 0;
    for (int t = 0; t < b.n_seq; ++t) {
        const int slot = b.seq_data[t];
        if (!slot_present(pm, slot)) continue;
        const int e = b.seq_enc[t];
        x.rowmask |= 2u << e;
        x.prev |= (unsigned long long)prev << (4 * (e + 1));
        x.next |= (unsigned long long)(e + 1) << (4 * prev);
        x.enc |= (unsigned long long)e << (4 * x.n);
        x.slot |= (unsigned long long)slot << (4 * x.n);
        ++x.n;
        prev = e + 1;
    }
    return x;
}

// per-sample mode: the tile's executed sequence comes packed (4 bits per step: encoder id + 1,
// first step in the low bits, 0 terminates) instead of from the batch-wide sequence + NaN mask
__device__ __forceinline__ ExecInfo exec_from_code(unsigned code) {
    ExecInfo x;
    x.rowmask = 1u; x.n = 0; x.enc = 0ull; x.slot = 0ull; x.prev = 0ull; x.next = 0ull;
    int prev = 0;
    for (int j = 0; j < 8; ++j) {
        const int v = (int)((code >> (4 * j)) & 15u);
        if (!v) break;
        const int e = v - 1;
        x.rowmask |= 2u << e;
        x.prev |= (unsigned long long)prev << (4 * (e + 1));
        x.next |= (unsigned long long)(e + 1) << (4 * prev);
        x.enc |= (unsigned long long)e << (4 * x.n);
        x.slot |= (unsigned long long)e << (4 * x.n);
        ++x.n;
        prev = e + 1;
    }
    return x;
}

// TILED = per-sample mode (mmn_batch.tile_seq): a separate instantiation, so that the ordinary
// batch path carries none of its branches
template <bool TILED>
__global__ __launch_bounds__(NT8) void k_fb8(const ParArgs a, const mmn_batch b, float cL, float cS) {
    constexpr int TB = 16;
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp smem = (lp)smem_generic;
    const unsigned pm = present_mask(b);                   // first load in the queue
    const int S = a.S, E = a.E, D = a.D, R = a.R, ldS = a.ldS, ldH = a.ldH, ldX = a.ldX;
    const Fb8Lds L = fb8_lds(R, ldS, ldH, ldX);
    const lp sDzA = smem + L.sDzA;
    const int tile = blockIdx.x, row0 = tile * TB;
    // per-sample mode (mmn_batch.tile_seq): rows come grouped into tiles of one executed sequence
    constexpr bool tiled = TILED;
    const int nrows = tiled ? g_ldi(b.tile_rows + tile) : min(TB, b.batch - row0);
    const unsigned tcode = tiled ? (unsigned)g_ldi(b.tile_seq + tile) : 0u;
    if (tiled && nrows == 0) {                             // padding tile (wave-uniform): its partials are zeros
        const int RDt = a.R * a.D;
        for (int c = threadIdx.x; c < RDt; c += NT8) {
            g_st(a.lossp + (int64_t)tile * RDt + c, 0.f);
            for (int k = 0; k < 5; ++k) g_sti(a.cnt + ((int64_t)tile * RDt + c) * 5 + k, 0);
        }
        for (int e = threadIdx.x; e < a.E; e += NT8) g_st(a.scp + (int64_t)tile * a.E + e, 0.f);
        // ... and its 16 rows of every k_wgrad A operand (dS, dz, dpre) are zeros: an earlier step may
        // have left real data at these positions
        const int Sx = a.S, D2 = 2 * a.D;
        for (int idx = threadIdx.x; idx < (a.E + 1) * TB * Sx; idx += NT8) {
            const int e = idx / (TB * Sx), rem = idx - e * TB * Sx;
            g_st(a.dS + ((int64_t)e * a.maxB + row0) * Sx + rem, 0.f);
        }
        for (int idx = threadIdx.x; idx < a.R * TB * D2; idx += NT8) {
            const int r = idx / (TB * D2), rem = idx - r * TB * D2;
            g_st(a.dz + ((int64_t)r * a.maxB + row0) * D2 + rem, 0.f);
        }
        for (int e = 0; e < a.E; ++e)
            for (int l = 0; l < a.enc[e].Lh; ++l) {
                const int N = a.enc[e].out[l];
                for (int idx = threadIdx.x; idx < TB * N; idx += NT8) g_st(a.dpre + a.enc[e].hid[l] + (int64_t)row0 * N + idx, 0.f);
            }
        return;
    }
    const int nst = tiled ? TB : nrows;                    // rows that go to HBM: per-sample mode also writes the
                                                           // (zero) padding rows, k_wgrad walks them
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15, q = lane >> 4;
    const int g = wave >> 1, half = wave & 1;
    const lp St = smem + L.sSt;
    const lp Ut = smem + L.sU;
    const lp sZ = smem + L.sZ;
    const lp sRed = smem + L.sRed;
    const lp sXg = smem + L.sW + g * L.wstride + L.oX;
    lp sHg[2] = {smem + L.sW + g * L.wstride + L.oH0, smem + L.sW + g * L.wstride + L.oH1};
    int stamp_k = 0;
    const int stamp_block = 7;
    STAMP8();
    const int ntS = (S + 15) >> 4, T0 = a.S16 >> 4;

    // ---- requests, in the order they are needed (the memory pipe returns in order).  Wave pair g
    //      owns sequence position g (n_seq <= E <= 4); whether that slot is present is only known
    //      once the mask has landed, so the x tile and the pair's weights are requested regardless.
    const int tA = min(g, max(b.n_seq - 1, 0));
    const int slotA = b.seq_data[tA], eA = b.seq_enc[tA];
    const ParEnc& pe = a.enc[eA];
    const int Lh = pe.Lh, F = pe.F, HL = pe.HL, akind = pe.akind;
    const int f4 = round_up(F, 16) >> 2;                   // float4 per image row (<= 32)
    const int TU = (HL + 15) >> 4;
    const float* pkU = a.pack + pe.pkf[Lh];
    const bool inA = g < b.n_seq;
    // Every request below is UNCONDITIONAL (clamped / dummy addresses, never a branch around a
    // load): only then does hipcc keep counted vmcnt waits, i.e. lets a wave consume an early
    // fragment while later prefetches are still in flight.
    const int l0 = min(0, Lh), l1 = min(1, Lh);            // layers this encoder does not have alias lower ones
    f32x4 xr[4], h0q[8], h1q[2], uq[2][4][2];
    float hb0 = 0.f, hb1 = 0.f, ub[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    {
        const float* xg = b.x[slotA] + (int64_t)row0 * b.ldx[slotA];
        const int64_t ldx = b.ldx[slotA];
        // 16-byte loads when every row of the slot is 16-byte aligned (wave-uniform), else four
        // 4-byte loads from clamped addresses; columns >= F are zeroed either way (a row's padding
        // may hold anything)
        const bool xvec = ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(b.x[slotA]) & 15) == 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                      // this half's 8 rows of the x tile
            const int idx = lane + 64 * k;
            const int row = 8 * half + idx / f4, c = (idx % f4) << 2;
            const bool ok = idx < 8 * f4 && row < nrows && c < F;
            const float* px = xg + (ok ? (int64_t)row * ldx + c : 0);
            f32x4 v;
            if (xvec) {
                v = g_ld4(px);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = g_ld(px + ((ok && c + j < F) ? j : 0));
            }
            xr[k] = v;                                     // RAW: a select here would make hipcc wait for the x tile before
                                                           // the requests below are even issued; zeroed at the LDS store
        }
        issue_t<8>(h0q, a.pack + pe.pkf[l0], (pe.in[l0] + 15) >> 4, (pe.out[l0] + 15) >> 4, half, 0);
        hb0 = g_ld(pe.bias[l0] + min(16 * half + i, pe.out[l0] - 1));
        issue_t<2>(h1q, a.pack + pe.pkf[l1], (pe.in[l1] + 15) >> 4, (pe.out[l1] + 15) >> 4, half, 0);
        hb1 = g_ld(pe.bias[l1] + min(16 * half + i, pe.out[l1] - 1));
    }
    float init_v[2];                                       // state row 0 = init state (state.py:29-32)
#pragma unroll
    for (int k = 0; k < 2; ++k) init_v[k] = g_ld(a.init + min(lane + 64 * k, S - 1));
    // targets / decoder biases of this thread's (row r, decoder d, batch row) triple, first pass
    int y_pre = 0;
    float bd0_pre = 0.f, bd1_pre = 0.f;
    {
        const int idx = threadIdx.x;
        const bool valid = idx < R * D * TB;
        const int r = valid ? idx / (D * TB) : 0;
        const int rem = idx - r * D * TB;
        const int d = valid ? rem / TB : 0, row = rem & (TB - 1);
        const bool ok = valid && row < nrows;
        y_pre = (int)*(const MMN_AS1 int64_t*)(b.y + (ok ? ((int64_t)row0 + row) * D + d : 0));
        bd0_pre = g_ld(a.dec_b[d]);
        bd1_pre = g_ld(a.dec_b[d] + 1);
    }

    // ---- LDS init while the requests fly
    if (a.needs_zero) {                                    // K-padding columns must be finite
        for (int idx = threadIdx.x; idx < L.total; idx += NT8) lds_st(smem + idx, 0.f);
        __syncthreads();
    } else {                                               // dz tiles: unwritten entries must read as zero
        for (int idx = threadIdx.x; idx < R * 16 * LDZ; idx += NT8) lds_st(sDzA + idx, 0.f);
    }
    const ExecInfo X = tiled ? exec_from_code(tcode) : exec_info(b, pm);   // waits for the mask only
    const bool act = inA && (tiled ? X.row(eA + 1) : slot_present(pm, slotA));
    int Lmax = 0;                                          // barrier counts must be uniform over the workgroup
    for (int k = 0; k < X.n; ++k) Lmax = max(Lmax, a.enc[X.e(k)].Lh);
    if (tile == 0 && threadIdx.x == 0 && tiled) {          // per-sample: every row may exist somewhere in the batch
        for (int r = 0; r < R; ++r) g_sti(a.exec_flags + r, 1);
        for (int e = 0; e < E; ++e) g_sti(a.prev_row + e, 0);
    }
    if (tile == 0 && threadIdx.x == 0 && !tiled) {         // which state rows exist this step (k_wgrad / k_reduce)
        for (int r = 0; r < R; ++r) g_sti(a.exec_flags + r, X.row(r) ? 1 : 0);
        for (int k = 0; k < X.n; ++k) g_sti(a.prev_row + X.e(k), X.prev_row(X.e(k) + 1));
    }
    // chain weights of the first TWO steps (tile = wave); step k+2 is requested as soon as step k
    // has been consumed, so every fetch has a whole step to land
    f32x4 wsA[8], wsB[8];
    auto issue_fwd = [&](f32x4 (&w)[8], int k) {           // past the last step: 8 reads of one hot KB
        const bool ok = k < X.n;
        const ParEnc& p2 = a.enc[X.e(ok ? k : 0)];
        issue_t<8>(w, a.pack + (ok ? p2.pkf[p2.Lh] : 0), ok ? T0 + ((p2.HL + 15) >> 4) : 1, ok ? ntS : 1, wave, 0);
    };
    {                                                      // x -> the pair's LDS image
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * k;
            const int row = 8 * half + idx / f4, c = (idx % f4) << 2;
            const bool ok = idx < 8 * f4 && row < nrows && c < F;
            f32x4 v = xr[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (ok && c + j < F) ? v[j] : 0.f;
            if (idx < 8 * f4) lds_st4(sXg + row * ldX + c, v);
        }
    }
    for (int r = wave; r < TB; r += 8)
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (lane + 64 * k < S) lds_st(St + r * ldS + lane + 64 * k, init_v[k]);
    __syncthreads();
    STAMP8();
    // second batch of requests (u_e operand, first two chain steps): they stream in underneath the
    // hidden layers instead of queueing in front of the x tile
    {
        const PB BU = make_pb(pkU, S, S, HL);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int nn[2] = {16 * (4 * half + 2 * pr), 16 * (4 * half + 2 * pr + 1)};
            issue_b<4>(uq[pr], BU, nn, BU.T0);             // tiles past S re-read the last tile
            ub[pr][0] = g_ld(pe.bias[Lh] + min(nn[0] + i, S - 1));
            ub[pr][1] = g_ld(pe.bias[Lh] + min(nn[1] + i, S - 1));
        }
    }

    // ---- phase A: u_e = W_x h_e + b of this pair's encoder
    if (Lmax >= 1) {                                       // hidden layer 0 (mlp_encoder.py:75-76), tile = half
        if (act && Lh >= 1 && 16 * half < pe.out[0]) {
            const int N = pe.out[0], T = (pe.in[0] + 15) >> 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            consume_t<8>(acc, sXg, ldX, 0, h0q, 0, T);
            const int col = 16 * half + i;
            if (col < N) {
#pragma unroll
                for (int k = 0; k < 4; ++k) lds_st(sHg[0] + (4 * q + k) * ldH + col, act_fwd(acc[k] + hb0, akind));
            }
        }
        __syncthreads();
        STAMP8();
        if (act && Lh >= 1) {                              // each half stores 8 rows of the tile
            const int N = pe.out[0];
            const int r0 = 8 * half, nr = max(0, min(8, nrows - r0));
            wave_store_tile(a.hid + pe.hid[0] + (int64_t)(row0 + r0) * N, sHg[0] + r0 * ldH, ldH, nr, N);
        }
    }
    if (Lmax >= 2) {
        if (act && Lh >= 2 && 16 * half < pe.out[1]) {
            const int N = pe.out[1], T = (pe.in[1] + 15) >> 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            consume_t<2>(acc, sHg[0], ldH, 0, h1q, 0, T);
            const int col = 16 * half + i;
            if (col < N) {
#pragma unroll
                for (int k = 0; k < 4; ++k) lds_st(sHg[1] + (4 * q + k) * ldH + col, act_fwd(acc[k] + hb1, akind));
            }
        }
        __syncthreads();
        STAMP8();
        if (act && Lh >= 2) {
            const int N = pe.out[1];
            const int r0 = 8 * half, nr = max(0, min(8, nrows - r0));
            wave_store_tile(a.hid + pe.hid[1] + (int64_t)(row0 + r0) * N, sHg[1] + r0 * ldH, ldH, nr, N);
        }
    }
    // the first two chain steps' weights are asked for here: they stream in underneath the u_e GEMM
    // instead of queueing in front of the hidden layers
    issue_fwd(wsA, 0);
    issue_fwd(wsB, 1);
    if (act) {                                             // u_e = W_x h + b (mlp_encoder.py:78), this half's 4 tiles
        clp in = Lh == 0 ? (clp)sXg : (clp)sHg[Lh - 1];
        const int ldin = Lh == 0 ? ldX : ldH;
        const lp U = Ut + eA * TB * ldS;
        const PB BU = make_pb(pkU, S, S, HL);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int nn[2] = {16 * (4 * half + 2 * pr), 16 * (4 * half + 2 * pr + 1)};
            if (nn[0] < S) {
                f32x4 acc[2][1];
                zero_acc<1>(acc);
                consume_b<1, 4>(acc, ASrc{in, ldin, in, ldin}, BU, uq[pr], BU.T0, BU.T0 + TU, nn[1] < S);
                run_epilogue<1>(acc, nn, S, [&](int row, int col, int c, float v) {
                    if (col < S) lds_st(U + row * ldS + col, v + ub[pr][c]);
                });
            }
        }
    }
    // decoder fragments: wave r evaluates state row r (R <= 8); needed in phase C
    f32x4 wd[8];
    {
        const bool rowok = i < 2 * D;
        const float* w = a.dec_w[rowok ? (i >> 1) : 0] + (rowok ? (i & 1) * S : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {                      // clamped address + select: no branch around a load
            const int k = 16 * j + 4 * q;
            const bool ok = rowok && k < S && j < T0;
            wd[j] = g_ld4(w + (ok ? k : 0));               // RAW (see xr); zeroed where they are first used
        }
    }
    __syncthreads();
    STAMP8();

    // ---- phase B: s' = W_s s + u_e; one column tile per wave.  The accumulator starts from the u_e
    //      tile, so the epilogue is the LDS write of the new state tile alone; the state-change sums
    //      and the global copies of the tiles (for k_wgrad) happen off this critical path.
    int cur = 0;
    float scl[4] = {0.f, 0.f, 0.f, 0.f};                    // per-lane state-change partial of executed step k
    // a [16 x S] LDS tile -> global rows: wave w stores rows 2w, 2w+1, 32 lanes x 16 bytes per row
    // (the 8-wave tier has S <= 128, S % 4 == 0)
    auto store_tile = [&](float* gbase, clp tile_lds) {
        const int r = 2 * wave + (lane >> 5), c = (lane & 31) * 4;
        if (r < nst && c < S) g_st4(gbase + (int64_t)(row0 + r) * S + c, lds_ld4(tile_lds + r * ldS + c));
    };
    auto store_state = [&](int e) { store_tile(a.states + (int64_t)e * a.maxB * S, St + (e + 1) * TB * ldS); };
    auto chain_step = [&](f32x4 (&wc)[8], int k) {
        const int e = X.e(k);
        const clp sC = St + cur * TB * ldS;
        const lp sN = St + (e + 1) * TB * ldS;
        const clp U = Ut + e * TB * ldS;
        const int col = 16 * wave + i, colc = min(col, S - 1);
        f32x4 acc, old;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[kk] = lds_ld(U + (4 * q + kk) * ldS + colc);
            old[kk] = lds_ld(sC + (4 * q + kk) * ldS + colc);   // this lane's elements of the state being left
        }
        if (wave < ntS) consume_t<8>(acc, sC, ldS, 0, wc, 0, T0);
        issue_fwd(wc, k + 2);
        if (k > 0) store_state(X.e(k - 1));                 // the previous tile, underneath the MFMAs
        if (tiled) store_tile(a.sin + (int64_t)e * a.maxB * S, sC);   // per-sample: what fed encoder e (k_wgrad's In)
        if (wave < ntS && col < S) {
            float sc = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                lds_st(sN + (4 * q + kk) * ldS + col, acc[kk]);
                const float dlt = acc[kk] - old[kk];
                sc += (4 * q + kk < nrows) ? dlt * dlt : 0.f;     // multimodn.py:174, this lane's share
            }
            scl[k] = sc;
        }
        __syncthreads();
        STAMP8();
        cur = e + 1;
    };
    if (X.n >= 1) {                                        // nested, not a loop: each wait sees one straight path
        chain_step(wsA, 0);
        if (X.n >= 2) {
            chain_step(wsB, 1);
            if (X.n >= 3) {
                chain_step(wsA, 2);
                if (X.n >= 4) chain_step(wsB, 3);
            }
        }
    }
    STAMP8();
    // ---- phase C: all decoders on all state rows (decoders.py:19-20, multimodn.py:141-157,176-191);
    //      waves 0..R-1 take one row each, the others the state-change sums (multimodn.py:174)
    if (wave < R) {
        if (X.row(wave)) {
            const clp sS = St + wave * TB * ldS;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            {                                              // the decoder fragments were requested raw at kernel start
                const bool rowok = i < 2 * D;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = rowok && (16 * j + 4 * q) < S && j < T0;
                    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                    wd[j] = ok ? wd[j] : zero4;
                }
            }
            consume_t<8>(z, sS, ldS, 0, wd, 0, T0);
#pragma unroll
            for (int k = 0; k < 4; ++k) lds_st(sZ + (wave * TB + q * 4 + k) * 16 + i, z[k]);
        }
    }
    // state-change partials: four independent butterflies per wave, then one LDS cell per (step, wave)
    {
        float t4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t4[k] = scl[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) t4[k] += __shfl_xor(t4[k], off);
        }
        if (lane < 4) lds_st(sRed + 8 * lane + wave, lane == 0 ? t4[0] : (lane == 1 ? t4[1] : (lane == 2 ? t4[2] : t4[3])));
    }
    if (X.n >= 1) store_state(X.e(X.n - 1));                // the last state tile
    // ---- backward requests: the Wdec^T fragment of this wave's column tile (one k-step: 2D <= 16),
    //      the W_s^T fragments of the first two reverse steps (tile = wave) and the dh fragments of
    //      this pair's encoder; they land during the decoder grid
    f32x4 wcA[8], wcB[8];
    auto issue_bwd = [&](f32x4 (&w)[8], int j) {            // j-th reverse step = executed step n-1-j
        const bool ok = j < X.n;
        const ParEnc& p2 = a.enc[X.e(ok ? X.n - 1 - j : 0)];
        issue_t<8>(w, a.pack + (ok ? p2.pkb[p2.Lh] : 0), ok ? T0 : 1, ok ? ntS : 1, wave, 0);
    };
    // (requests are spread over the phases that follow: a burst of 35 KB per wave would sit in
    //  front of the vector-memory pipe for ~2 us)
    f32x4 wdzq[1];
    issue_t<1>(wdzq, a.pack + a.pkd, 1, ntS, wave, 0);
    issue_bwd(wcA, 0);
    const bool actC = act && Lh >= 1;                       // E <= 4: one encoder per wave pair
    f32x4 hqC[2][8], h1qC[2];
    auto issue_dh = [&](int k) {                            // dh fragments of this pair's encoder, column tile half + 2k
        const bool okh = Lh >= 1;
        issue_t<8>(hqC[k], a.pack + (okh ? pe.pkh : 0), okh ? T0 : 1, okh ? (HL + 15) >> 4 : 1, half + 2 * k, 0);
    };
    auto issue_dh1 = [&]() {
        const bool ok1 = Lh >= 2;
        issue_t<2>(h1qC, a.pack + (ok1 ? pe.pkb[1] : 0), ok1 ? (pe.out[1] + 15) >> 4 : 1, ok1 ? (pe.in[1] + 15) >> 4 : 1, half, 0);
    };
    __syncthreads();
    STAMP8();
    const int total = R * D * TB;
    for (int base = 0; base < total; base += NT8) {
        const int idx = base + threadIdx.x;
        const bool valid = idx < total;
        const int r = valid ? idx / (D * TB) : 0;
        const int rem = idx - r * D * TB;
        const int d = valid ? rem / TB : 0, row = rem & (TB - 1);
        const bool live = valid && row < nrows && X.row(r);
        float lossv = 0.f;
        int correct = 0, tp = 0, tn = 0, fp = 0, fn = 0;
        if (live) {
            const float* bd = a.dec_b[d];
            const float za = (base == 0 ? bd0_pre : g_ld(bd)) + lds_ld(sZ + (r * TB + row) * 16 + 2 * d);
            const float zb = (base == 0 ? bd1_pre : g_ld(bd + 1)) + lds_ld(sZ + (r * TB + row) * 16 + 2 * d + 1);
            const int64_t grow = (int64_t)row0 + row;
            const int y = base == 0 ? y_pre : (int)*(const MMN_AS1 int64_t*)(b.y + grow * D + d);
            const float o0 = 1.0f / (1.0f + expf(-za));
            const float o1 = 1.0f / (1.0f + expf(-zb));
            const float mx = fmaxf(o0, o1);
            const float lse = mx + logf(expf(o0 - mx) + expf(o1 - mx));
            lossv = lse - (y ? o1 : o0);
            const int pred = o1 > o0 ? 1 : 0;      // torch.max: first index wins ties
            correct = pred == y;
            tp = pred & y; tn = (1 - pred) & (1 - y); fp = pred & (1 - y); fn = (1 - pred) & y;
            const float g0 = expf(o0 - lse) - (y == 0 ? 1.0f : 0.0f);
            const float g1 = expf(o1 - lse) - (y == 1 ? 1.0f : 0.0f);
            f32x2 dzv;
            dzv.x = cL * g0 * o0 * (1.0f - o0);
            dzv.y = cL * g1 * o1 * (1.0f - o1);
            if (tiled && cL < 0.f) { dzv.x = o0; dzv.y = o1; }   // forward-only (per-sample eval): the decoder OUTPUTS
            g_st2(a.dz + ((int64_t)r * a.maxB + grow) * (2 * D) + 2 * d, dzv);
            lds_st(sDzA + (r * TB + row) * LDZ + 2 * d, dzv.x);
            lds_st(sDzA + (r * TB + row) * LDZ + 2 * d + 1, dzv.y);
        } else if (tiled && valid) {                // per-sample: k_wgrad walks every row, dead entries must be 0
            f32x2 zz; zz.x = 0.f; zz.y = 0.f;
            g_st2(a.dz + ((int64_t)r * a.maxB + row0 + row) * (2 * D) + 2 * d, zz);
        }
#pragma unroll
        for (int off = TB / 2; off >= 1; off >>= 1) lossv += __shfl_xor(lossv, off);
        const unsigned long long mc = __ballot(correct), mtp = __ballot(tp), mtn = __ballot(tn),
                                 mfp = __ballot(fp), mfn = __ballot(fn);
        if (valid && row == 0) {
            const int sh = lane & ~(TB - 1);
            const int64_t cell = (int64_t)tile * (R * D) + r * D + d;
            g_st(a.lossp + cell, lossv);
            int32_t* cp = a.cnt + cell * 5;
            g_sti(cp + 0, __popcll((mc >> sh) & 0xFFFFull));
            g_sti(cp + 1, __popcll((mtp >> sh) & 0xFFFFull));
            g_sti(cp + 2, __popcll((mtn >> sh) & 0xFFFFull));
            g_sti(cp + 3, __popcll((mfp >> sh) & 0xFFFFull));
            g_sti(cp + 4, __popcll((mfn >> sh) & 0xFFFFull));
        }
    }
    for (int e = threadIdx.x; e < E; e += NT8)              // encoders that did not run: zero state change
        if (!X.row(e + 1)) g_st(a.scp + (int64_t)tile * E + e, 0.f);
    if ((int)threadIdx.x < X.n) {                           // executed step k: fixed-order sum over the 8 waves
        float sm = 0.f;
        for (int w8 = 0; w8 < 8; ++w8) sm += lds_ld(sRed + 8 * threadIdx.x + w8);
        g_st(a.scp + (int64_t)tile * E + X.e(threadIdx.x), sm);
    }
    __syncthreads();                                        // dz tiles complete; u_e tiles dead
    STAMP8();

    if (tiled && cL < 0.f) return;                          // forward-only step of per-sample mode (mmn_eval_step)
    // ======================= backward half =======================
    const lp DG = smem + L.sU;                              // R tiles: decoder grad -> G_out, in place
    // ---- phase A': DG[r] = dz[r] Wdec + cS (s_r - s_prev(r)) - cS (s_next(r) - s_r): the decoder
    //      gradient plus BOTH state-change terms that touch row r, so the reverse chain only adds
    //      the carry.  Wave w forms column tile w of every executed row from ONE Wdec^T fragment.
    if (wave < ntS) {
        const int col = 16 * wave + i;
        for (int r = 0; r < R; ++r) {
            if (!X.row(r)) continue;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            consume_t<1>(acc, sDzA + r * TB * LDZ, LDZ, 0, wdzq, 0, 1);
            const lp out = DG + r * TB * ldS;
            const clp sr = St + r * TB * ldS;
            const clp sp = St + X.prev_row(r) * TB * ldS;
            const int rn = X.next_row(r);
            const clp sn = St + rn * TB * ldS;
            if (col < S) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = 4 * q + k;
                    const float sv = lds_ld(sr + row * ldS + col);
                    float dlt = r >= 1 ? sv - lds_ld(sp + row * ldS + col) : 0.f;
                    if (rn) dlt -= lds_ld(sn + row * ldS + col) - sv;
                    lds_st(out + row * ldS + col, (row < nrows) ? acc[k] + cS * dlt : 0.f);
                }
            }
        }
    }
    issue_bwd(wcB, 1);
    __syncthreads();
    STAMP8();
    // ---- phase B': G_out(prev row) += G_out(e) W_s; the accumulator starts from the target tile.
    //      G_out(e) is final when its step starts: its rows go out as dS[e] underneath the MFMAs.
    auto store_row = [&](int r, int idx) { store_tile(a.dS + (int64_t)idx * a.maxB * S, DG + r * TB * ldS); };
    {
        auto chain_step = [&](f32x4 (&wc)[8], int j) {
            const int e = X.e(X.n - 1 - j);
            const int r_nxt = X.prev_row(e + 1);            // state that fed encoder e
            const clp Go = DG + (e + 1) * TB * ldS;
            const lp Gx = DG + r_nxt * TB * ldS;
            const int col = 16 * wave + i, colc = min(col, S - 1);
            f32x4 acc;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = lds_ld(Gx + (4 * q + k) * ldS + colc);
            if (wave < ntS) consume_t<8>(acc, Go, ldS, 0, wc, 0, T0);
            issue_bwd(wc, j + 2);
            if (j == 0) issue_dh(0);                        // the hidden-layer backward operands ride along
            if (j == 1) issue_dh(1);
            store_row(e + 1, e);
            if (wave < ntS && col < S) {
#pragma unroll
                for (int k = 0; k < 4; ++k) lds_st(Gx + (4 * q + k) * ldS + col, acc[k]);
            }
            __syncthreads();
            STAMP8();
        };
        if (X.n >= 1) {
            chain_step(wcA, 0);
            if (X.n >= 2) {
                chain_step(wcB, 1);
                if (X.n >= 3) {
                    chain_step(wcA, 2);
                    if (X.n >= 4) chain_step(wcB, 3);
                }
            }
        }
    }
    if (X.n < 1) issue_dh(0);
    if (X.n < 2) issue_dh(1);
    issue_dh1();
    store_row(0, E);                                        // dS0 (final since the last barrier)
    STAMP8();
    // ---- phase C': hidden-layer backward of this pair's encoder; h_0 / h_1 are still in the pair's
    //      LDS scratch, so act' is applied in the GEMM epilogue and dpre overwrites h in place
    if (Lmax >= 1) {
        if (actC) {
            const clp Go = DG + (eA + 1) * TB * ldS;
            const lp hb = sHg[Lh - 1];                      // h_{Lh-1} in, dpre_{Lh-1} out
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int tl = half + 2 * k;
                if (16 * tl < HL) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    consume_t<8>(acc, Go, ldS, 0, hqC[k], 0, T0);
                    const int col = 16 * tl + i;
                    if (col < HL) {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const int row = 4 * q + kk;
                            const float h = lds_ld(hb + row * ldH + col);
                            lds_st(hb + row * ldH + col, row < nrows ? acc[kk] * act_grad_from_out(h, akind) : 0.f);
                        }
                    }
                }
            }
        }
        __syncthreads();
        STAMP8();
        if (actC) {
            const int N = pe.out[Lh - 1];
            const int r0 = 8 * half, nr = max(0, min(8, nst - r0));
            wave_store_tile(a.dpre + pe.hid[Lh - 1] + (int64_t)(row0 + r0) * N, sHg[Lh - 1] + r0 * ldH, ldH, nr, N);
        }
        if (Lmax >= 2) {
            if (actC && Lh >= 2 && 16 * half < pe.in[1]) {   // dpre_0 = (dpre_1 W_1) .* act'(h_0), tile = half
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                consume_t<2>(acc, sHg[1], ldH, 0, h1qC, 0, (pe.out[1] + 15) >> 4);
                const int col = 16 * half + i;
                if (col < pe.in[1]) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int row = 4 * q + kk;
                        const float h = lds_ld(sHg[0] + row * ldH + col);
                        lds_st(sHg[0] + row * ldH + col, row < nrows ? acc[kk] * act_grad_from_out(h, akind) : 0.f);
                    }
                }
            }
            __syncthreads();
            STAMP8();
            if (actC && Lh >= 2) {
                const int N = pe.out[0];
                const int r0 = 8 * half, nr = max(0, min(8, nst - r0));
                wave_store_tile(a.dpre + pe.hid[0] + (int64_t)(row0 + r0) * N, sHg[0] + r0 * ldH, ldH, nr, N);
            }
        }
    }
    if (tiled) {
        // encoders this tile did not execute: their rows of dS / sin / dpre must read as zero in
        // k_wgrad (pair g owns encoder g, as in phase A)
        if (inA && !X.row(eA + 1)) {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const int r = 8 * half + (lane >> 3);                       // 8 rows per half, 8 lanes per row
            for (int c = (lane & 7) * 4; c < S; c += 32) {
                g_st4(a.dS + ((int64_t)eA * a.maxB + row0 + r) * S + c, z4);
                g_st4(a.sin + ((int64_t)eA * a.maxB + row0 + r) * S + c, z4);
            }
            for (int l = 0; l < Lh; ++l) {
                const int N = pe.out[l];
                for (int c = (lane & 7); c < N; c += 8) g_st(a.dpre + pe.hid[l] + (int64_t)(row0 + r) * N + c, 0.f);
            }
        }
    }
    STAMP8();
}


// ------------------------------------------------------------------------------------------------
// k_wgrad: C[M x ncols] = A[rows x M]^T * In[rows x ncols] over a row range -> partial slab.
// No LDS in the main loop: lane (i, q) loads MT consecutive A columns and NT consecutive In columns
// of row r+q, i.e. the 16*MT x 16*NT output tile is made of MT x NT INTERLEAVED 16x16 MFMA tiles
// (tile (c, c') holds rows m0 + MT*i + c and columns n0 + NT*j + c'), so every load instruction
// reads 4 rows x up to 256 contiguous bytes.  The 4 waves of a workgroup split the row range and
// are summed through LDS in a fixed order.
// ------------------------------------------------------------------------------------------------
struct SrcRef { const float* p; int64_t ld; };

// One interleaved operand fragment (V consecutive columns of one row).  FAST (wave-uniform): the
// tile edge is V-aligned, so a lane is wholly inside or wholly outside; then the load is
// unconditional from a clamped address and masked by a select (no branch -> counted vmcnt).
template <int V, bool FAST>
__device__ __forceinline__ void load_frag(float (&dst)[V], const float* __restrict__ base, int64_t row_off, int first,
                                          int limit, bool row_ok) {
    if (FAST) {
        // RAW values from a clamped (always valid) address; nothing here depends on the loaded data,
        // so the load stays in flight until the k-step that consumes it.  Rows past the range are
        // zeroed on the A side at consume time; columns past the tile edge only feed outputs that
        // are never stored.
        (void)row_ok;
        const float* ptr = base + row_off + (first < limit ? first : 0);
        if (V == 4) {
            const f32x4 v = g_ld4(ptr);
            dst[0] = v.x; dst[V > 1 ? 1 : 0] = v.y; dst[V > 2 ? 2 : 0] = v.z; dst[V > 3 ? 3 : 0] = v.w;
        } else if (V == 2) {
            const f32x2 v = g_ld2(ptr);
            dst[0] = v.x; dst[V > 1 ? 1 : 0] = v.y;
        } else {
            dst[0] = g_ld(ptr);
        }
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) dst[k] = 0.f;
        if (!row_ok) return;
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (first + k < limit) dst[k] = g_ld(base + row_off + first + k);
    }
}

template <int MT, int NTL>
__device__ __forceinline__ void wgrad_tile(const WgArgs& w, const WRec& it, const float* Ap, int64_t lda,
                                           SrcRef in, int ncols, int rb, int re, lp sTile, lp sBias) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int i = lane & 15, q = lane >> 4;
    const int M = it.M;
    const bool has_in = it.has_in != 0;
    const bool bias = it.bias != 0;
    f32x4 acc[MT][NTL];
    f32x4 accb[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c) {
        accb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < NTL; ++d) acc[c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // this wave's quarter of the row range (multiples of 4 rows)
    const int rq = round_up((re - rb + 3) / 4, 4);
    const int wb = rb + wave * rq, we = min(re, wb + rq);
    const bool a_fast = ((lda % MT) == 0) && ((it.m0 % MT) == 0) && (((M - it.m0) % MT) == 0) &&
                        ((reinterpret_cast<uintptr_t>(Ap) & (4 * MT - 1)) == 0);
    const bool i_fast = !has_in || (((in.ld % NTL) == 0) && ((it.n0 % NTL) == 0) && (((ncols - it.n0) % NTL) == 0) &&
                                    ((reinterpret_cast<uintptr_t>(in.p) & (4 * NTL - 1)) == 0));
    const float ones = (i == 0) ? 1.0f : 0.0f;
    const int mfirst = it.m0 + MT * i, nfirst = it.n0 + NTL * i;
    const float* inp = has_in ? in.p : Ap;               // bias-only items re-read A (masked to zero)
    const int64_t ldi = has_in ? in.ld : lda;
    const int ilimit = has_in ? ncols : 0;

    // The loop body below has NO branches (bias is a template flag, k-steps past the range run on
    // zeroed A fragments, the accumulators of a bias-only item are simply never stored): with a
    // branch in it hipcc drains every load in flight (vmcnt(0)) at the loop header.
    auto run = [&](auto fast_tag, auto bias_tag) {
        constexpr bool FA = decltype(fast_tag)::value;
        constexpr bool BI = decltype(bias_tag)::value;
        constexpr int DEPTH = 8;                          // k-steps (4 rows each) in flight
        float a[DEPTH][MT], bb[DEPTH][NTL];
        auto ld = [&](float (&av)[MT], float (&bv)[NTL], int r) {
            const int row = r + q;
            const bool ok = row < we;
            const int rc = ok ? row : (we - 1);           // clamp: always a valid row of this range
            load_frag<MT, FA>(av, Ap, (int64_t)rc * lda, mfirst, M, ok);
            load_frag<NTL, FA>(bv, inp, (int64_t)rc * ldi, nfirst, ilimit, ok);
        };
        auto comp = [&](const float (&av_raw)[MT], const float (&bv)[NTL], int r) {
            float av[MT];
            const bool ok = !FA || (r + q < we);           // FAST loads are unmasked: zero the A side of rows past the range
#pragma unroll
            for (int c = 0; c < MT; ++c) av[c] = ok ? av_raw[c] : 0.f;
#pragma unroll
            for (int c = 0; c < MT; ++c)
#pragma unroll
                for (int d = 0; d < NTL; ++d) acc[c][d] = mfma4(av[c], bv[d], acc[c][d]);
            if (BI) {
#pragma unroll
                for (int c = 0; c < MT; ++c) accb[c] = mfma4(av[c], ones, accb[c]);
            }
        };
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) ld(a[k], bb[k], wb + 4 * k);
        for (int r = wb; r < we; r += 4 * DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                comp(a[k], bb[k], r + 4 * k);
                ld(a[k], bb[k], r + 4 * (k + DEPTH));
                __builtin_amdgcn_sched_barrier(0);        // keep the refill right behind its k-step (the
                                                          // scheduler otherwise sinks all loads to the loop end)
            }
        }
    };
    if (wb < we) {
        if (a_fast && i_fast) {
            if (bias) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{});
        } else {
            if (bias) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{});
        }
    }
    if (w.stamps && blockIdx.x == 200 && threadIdx.x == 0) w.stamps[110] = (long long)wall_clock64();
    // fixed-order sum of the four waves' tiles through LDS, then one coalesced slab write
    const lp mine = sTile + wave * (64 * TILE_LD);
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
        for (int d = 0; d < NTL; ++d)
#pragma unroll
            for (int k = 0; k < 4; ++k) lds_st(mine + (MT * (4 * q + k) + c) * TILE_LD + NTL * i + d, acc[c][d][k]);
    if (bias && i == 0) {
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) lds_st(sBias + wave * 64 + MT * (4 * q + k) + c, accb[c][k]);
    }
    __syncthreads();
    if (w.stamps && blockIdx.x == 200 && threadIdx.x == 0) w.stamps[111] = (long long)wall_clock64();
    // flat-gradient-shaped slab: element (m, n) of this task's weight gradient at w_off + row(m) + n
    const int w_ld = it.w_ld, ds = it.dec_stride;
    const int col_off = it.col_off;
    if (has_in) {
        constexpr int TW = 16 * NTL, TH = 16 * MT;
        for (int idx = threadIdx.x; idx < TH * TW; idx += NT) {
            const int ml = idx / TW, nl = idx - ml * TW;
            const int m = it.m0 + ml, n = it.n0 + nl;
            if (m < M && n < ncols) {
                const clp s = sTile + ml * TILE_LD + nl;
                const int64_t rowf = ds ? (int64_t)(m >> 1) * ds + (int64_t)(m & 1) * w_ld : (int64_t)m * w_ld;
                g_st(w.slabs + it.w_off + rowf + col_off + nl,
                     ((lds_ld(s) + lds_ld(s + 64 * TILE_LD)) + lds_ld(s + 2 * 64 * TILE_LD)) + lds_ld(s + 3 * 64 * TILE_LD));
            }
        }
    }
    if (bias) {
        for (int ml = threadIdx.x; ml < 16 * MT; ml += NT) {
            const int m = it.m0 + ml;
            if (m < M)
                g_st(w.slabs + it.b_off + (ds ? (int64_t)(m >> 1) * ds + (m & 1) : (int64_t)m),
                     ((lds_ld(sBias + ml) + lds_ld(sBias + 64 + ml)) + lds_ld(sBias + 128 + ml)) + lds_ld(sBias + 192 + ml));
        }
    }
}

constexpr int WGRAD_LDS_FLOATS = 4 * 64 * TILE_LD + 4 * 64;

__global__ __launch_bounds__(NT) void k_wgrad(const WgArgs w, const mmn_batch b) {
    extern __shared__ __attribute__((aligned(16))) float smem_generic[];
    const lp sTile = (lp)smem_generic;
    const lp sBias = sTile + 4 * 64 * TILE_LD;
    const unsigned pm = present_mask(b);                   // the only dependent global load besides the record
    long long* const stamps = w.stamps;
    if (stamps && blockIdx.x == 200 && threadIdx.x == 0) stamps[100] = (long long)wall_clock64();
    const WRec it = w.recs[blockIdx.x];
    const int rows_per_split = round_up((b.batch + it.nks - 1) / it.nks, 16);
    int rb = it.ks * rows_per_split, re = min(b.batch, rb + rows_per_split);
    const bool tiled = b.tile_seq != nullptr;              // per-sample mode: dead rows are zero in A, no gating
    if ((!tiled && !row_executed(b, pm, it.gate)) || rb >= re) { rb = 0; re = 0; }   // writes zeros
    const float* Ap = (it.a_kind == A_DPRE ? w.dpre : (it.a_kind == A_DS ? w.dS : (it.a_kind == A_DZ ? w.dz : w.gdpre))) + it.a_off;
    SrcRef in{nullptr, 0};
    switch (it.in_kind) {
        case IN_X: {
            int slot = 0;
            for (int t = 0; t < b.n_seq; ++t) if (b.seq_enc[t] == it.in_enc) slot = b.seq_data[t];
            in.p = b.x[slot]; in.ld = b.ldx[slot];
            break;
        }
        case IN_HID: in.p = w.hid + it.in_off; in.ld = it.ldi; break;
        case IN_STATE_ROW: in.p = w.states + it.in_off; in.ld = it.ldi; break;
        case IN_INIT: in.p = w.init; in.ld = 0; break;
        case IN_GEN: in.p = w.gact + it.in_off; in.ld = it.ldi; break;
        case IN_PREV_STATE: {
            if (tiled) { in.p = w.sin + (int64_t)it.in_enc * w.maxB * w.S; in.ld = w.S; break; }
            const int r = prev_row_of(b, pm, it.in_enc);
            if (r == 0) { in.p = w.init; in.ld = 0; }
            else { in.p = w.states + (int64_t)(r - 1) * w.maxB * w.S; in.ld = w.S; }
            break;
        }
        default: break;
    }
    const int ncols = it.ncols;
    const int64_t lda = it.lda;
    const int key = it.mt * 8 + it.nt;
    if (stamps && blockIdx.x == 200 && threadIdx.x == 0) stamps[101] = (long long)wall_clock64();
    switch (key) {
        case 4 * 8 + 4: wgrad_tile<4, 4>(w, it, Ap, lda, in, ncols, rb, re, sTile, sBias); break;
        case 4 * 8 + 2: wgrad_tile<4, 2>(w, it, Ap, lda, in, ncols, rb, re, sTile, sBias); break;
        case 4 * 8 + 1: wgrad_tile<4, 1>(w, it, Ap, lda, in, ncols, rb, re, sTile, sBias); break;
        case 2 * 8 + 4: wgrad_tile<2, 4>(w, it, Ap, lda, in, ncols, rb, re, sTile, sBias); break;
        case 2 * 8 + 2: wgrad_tile<2, 2>(w, it, Ap, lda, in, ncols, rb, re, sTile, sBias); break;
        case 2 * 8 + 1: wgrad_tile<2, 1>(w, it, Ap, lda, in, ncols, rb, re, sTile, sBias); break;
        case 1 * 8 + 4: wgrad_tile<1, 4>(w, it, Ap, lda, in, ncols, rb, re, sTile, sBias); break;
        case 1 * 8 + 2: wgrad_tile<1, 2>(w, it, Ap, lda, in, ncols, rb, re, sTile, sBias); break;
        default:        wgrad_tile<1, 1>(w, it, Ap, lda, in, ncols, rb, re, sTile, sBias); break;
    }
    if (stamps && blockIdx.x == 200 && threadIdx.x == 0) stamps[102] = (long long)wall_clock64();
}

// =====================================================================================
// Per-sample mode: regrouping of the rows of a batch into tiles of one executed sequence
// (mmn_regroup).  Three launches, deterministic (no atomics decide an order):
//   k_ps_code   one wave per sample: which modalities are present (no NaN in the row), packed
//               executed sequence
//   k_ps_layout ONE workgroup: stable counting sort of the rows by sequence code (<= 65 distinct
//               codes), groups padded to whole tiles -> source row of every position, per-tile
//               row count and sequence
//   k_ps_gather one wave per position: the sample's features (zeros where missing / padding),
//               per ENCODER, and its targets
// =====================================================================================
constexpr int PS_MAX_ROWS = 16384;

__global__ __launch_bounds__(NT) void k_ps_code(const mmn_batch b, const int64_t* __restrict__ seq, int E,
                                                int32_t* __restrict__ codes, int32_t* __restrict__ pmask) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (NT / 64) + wave_id();
    if (row >= b.batch) return;
    unsigned present = 0;
    for (int k = 0; k < E; ++k) {
        const float* x = b.x[k] + (int64_t)row * b.ldx[k];
        bool bad = false;
        for (int c = lane; c < b.seq_data[k]; c += 64) { const float v = g_ld(x + c); bad |= (v != v); }   // seq_data[k] carries slot k's width here
        if (!__any(bad)) present |= 1u << k;
    }
    if (lane == 0) {
        unsigned code = 0; int n = 0;
        for (int k = 0; k < E; ++k) {
            if (!((present >> k) & 1u)) continue;
            const int e = seq ? (int)seq[(int64_t)row * E + k] : k;
            code |= (unsigned)((e & 15) + 1) << (4 * n);
            ++n;
        }
        codes[row] = (int32_t)code;
        pmask[row] = (int32_t)present;
    }
}

constexpr int PS_MAXG = 96;                   // distinct executed sequences of E <= 4 encoders: 65

// Stable counting sort by sequence code, one workgroup.  The codes of a batch take at most 65 distinct
// values (ordered subsets of <= 4 encoders): (1) collect the distinct codes in an LDS hash set and
// rank them (group id = rank of the code, so the layout does not depend on the hash order);
// (2) per 64-row chunk, ballots give every row its rank inside (chunk, group) and the chunk's count
// per group; (3) prefix over chunks and padded group bases; (4) scatter.
__global__ __launch_bounds__(1024) void k_ps_layout(const int32_t* __restrict__ codes, int B, int rows_out,
                                                    int32_t* __restrict__ src_of, int32_t* __restrict__ tile_rows,
                                                    int32_t* __restrict__ tile_seq) {
    extern __shared__ __attribute__((aligned(16))) unsigned ps_smem[];
    const int n_chunks = (B + 63) >> 6;
    int* cnt = reinterpret_cast<int*>(ps_smem);                 // [n_chunks][PS_MAXG] -> exclusive prefix over chunks
    __shared__ int hkey[256];                                   // hash set of codes (-1 = empty)
    __shared__ int gcode[PS_MAXG], gtot[PS_MAXG], gbase[PS_MAXG];
    __shared__ int n_groups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < rows_out; i += 1024) src_of[i] = -1;
    for (int i = tid; i < rows_out / 16; i += 1024) { tile_rows[i] = 0; tile_seq[i] = 0; }
    for (int i = tid; i < 256; i += 1024) hkey[i] = -1;
    for (int i = tid; i < n_chunks * PS_MAXG; i += 1024) cnt[i] = 0;
    if (tid == 0) n_groups = 0;
    __syncthreads();
    // (1) distinct codes
    for (int i = tid; i < B; i += 1024) {
        const int c = codes[i];
        unsigned h = ((unsigned)c * 2654435761u) >> 24;
        for (int probe = 0; probe < 256; ++probe) {
            const int old = atomicCAS(&hkey[h], -1, c);
            if (old == -1 || old == c) break;
            h = (h + 1) & 255u;
        }
    }
    __syncthreads();
    if (tid < 256 && hkey[tid] != -1) {                        // compact, then rank
        const int slot = atomicAdd(&n_groups, 1);
        if (slot < PS_MAXG) gcode[slot] = hkey[tid];
    }
    __syncthreads();
    const int G = min(n_groups, PS_MAXG);
    __shared__ int sorted_code[PS_MAXG];
    if (tid < G) {
        const int mine = gcode[tid];
        int rank = 0;
        for (int j = 0; j < G; ++j) rank += gcode[j] < mine;
        sorted_code[rank] = mine;
    }
    __syncthreads();
    // (2) per chunk: group id of every row, rank inside (chunk, group), chunk counts
    int my_gid[16], my_rank[16];
    int nmine = 0;
    for (int ch = wave; ch < n_chunks; ch += 16) {
        const int row = ch * 64 + lane;
        int gid = -1;
        if (row < B) {
            const int c = codes[row];
            int lo = 0, hi = G - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted_code[mid] < c) lo = mid + 1; else hi = mid; }
            gid = lo;
        }
        // lanes of this chunk with MY group id: AND of one ballot per id bit (7 ballots, not one per group)
        unsigned long long same = __ballot(gid >= 0);
#pragma unroll
        for (int bit = 0; bit < 7; ++bit) {
            const unsigned long long mb = __ballot((gid >> bit) & 1);
            same &= ((gid >> bit) & 1) ? mb : ~mb;
        }
        const int rk = __popcll(same & ((1ull << lane) - 1ull));
        if (gid >= 0 && rk == 0) cnt[ch * PS_MAXG + gid] = __popcll(same);     // the group's first lane in the chunk
        my_gid[nmine] = gid; my_rank[nmine] = rk; ++nmine;
    }
    __syncthreads();
    // (3) exclusive prefix over the chunks per group, group totals, padded bases
    if (tid < G) {
        int run = 0;
        for (int ch = 0; ch < n_chunks; ++ch) { const int v = cnt[ch * PS_MAXG + tid]; cnt[ch * PS_MAXG + tid] = run; run += v; }
        gtot[tid] = run;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int g = 0; g < G; ++g) { gbase[g] = run; run += (gtot[g] + 15) / 16 * 16; }
    }
    __syncthreads();
    // (4) scatter
    nmine = 0;
    for (int ch = wave; ch < n_chunks; ch += 16) {
        const int row = ch * 64 + lane;
        const int gid = my_gid[nmine], rk = my_rank[nmine]; ++nmine;
        if (row < B && gid >= 0) {
            const int pos = gbase[gid] + cnt[ch * PS_MAXG + gid] + rk;
            if (pos < rows_out) src_of[pos] = row;
        }
    }
    for (int g = wave; g < G; g += 16) {                        // tile tables straight from the group totals
        const int ntile = (gtot[g] + 15) >> 4;
        for (int t = lane; t < ntile; t += 64) {
            const int tile = (gbase[g] >> 4) + t;
            if (tile < rows_out / 16) { tile_rows[tile] = min(16, gtot[g] - 16 * t); tile_seq[tile] = sorted_code[g]; }
        }
    }
}

__global__ __launch_bounds__(NT) void k_ps_gather(const mmn_batch b, const int64_t* __restrict__ seq, int E, int D,
                                                  const int32_t* __restrict__ src_of, const int32_t* __restrict__ pmask,
                                                  mmn_batch out, int rows_out) {
    const int lane = threadIdx.x & 63;
    const int pos = blockIdx.x * (NT / 64) + wave_id();
    if (pos >= rows_out) return;
    const int src = g_ldi(src_of + pos);
    const unsigned pm = src >= 0 ? (unsigned)g_ldi(pmask + src) : 0u;
    for (int e = 0; e < E; ++e) {
        int slot = -1;                                                   // the present slot that feeds encoder e
        if (src >= 0)
            for (int k = 0; k < E; ++k)
                if (((pm >> k) & 1u) && (seq ? (int)seq[(int64_t)src * E + k] : k) == e) slot = k;
        float* dst = const_cast<float*>(out.x[e]) + (int64_t)pos * out.ldx[e];
        const int F = out.seq_data[e];                                   // width of encoder e's features
        const float* sx = slot >= 0 ? b.x[slot] + (int64_t)src * b.ldx[slot] : nullptr;
        for (int c = lane; c < F; c += 64) g_st(dst + c, slot >= 0 ? g_ld(sx + c) : 0.f);
    }
    int64_t* yo = const_cast<int64_t*>(out.y) + (int64_t)pos * D;
    for (int d = lane; d < D; d += 64) yo[d] = src >= 0 ? b.y[(int64_t)src * D + d] : 0;
}

// =====================================================================================
// k_adam: optimizer.step() of the training loop (multimodn.py:204) for torch.optim.Adam as the
// reference pipelines build it, over the FLAT parameter / gradient / moment buffers: one launch
// for the whole model instead of a multi-tensor apply (one block per 64K-element chunk) plus a
// foreach add for the step counters.  HBM-bound: 16 B read + 12 B written per parameter.
//   - per-tensor step counters live on the device (hipGraph replay).  Every block keeps its OWN
//     copy of the n_seg counters (row blockIdx of `steps`; row 0 is the one callers read) and
//     advances it itself, so no block ever waits on another: no ticket, no fence, one memory
//     round trip per launch;
//   - a tensor whose gradient is None this step (encoder skipped on a NaN batch) is left
//     untouched: no moment decay, no step increment - torch.optim.Adam's behaviour.
// =====================================================================================
constexpr int ADAM_NT = 256;
constexpr int ADAM_MAX_SEG = 512;

struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    float* steps; const int* seg_start; const int* seg_skip;
    int n, n_seg;
    double lr, b1, b2;
    float eps, wd;
    int maximize;
    // optional (data-parallel tail): tensor i has no gradient this step when state row gate_segs[i].gate
    // does not exist (its encoder was skipped on a NaN batch) -> left untouched, like seg_skip
    const Seg* gate_segs; const int32_t* exec_flags;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float omb1, float b2, float omb2,
                                          float eps, float wd, float step_size, float bc2s, int maximize) {
    if (maximize) g = -g;
    if (wd != 0.f) g = fmaf(wd, p, g);
    m = fmaf(g - m, omb1, m);                             // lerp(m, g, 1 - beta1)
    v = fmaf(b2, v, omb2 * g * g);
    const float denom = sqrtf(v) / bc2s + eps;
    p -= step_size * (m / denom);
}

// beta^t for an integer step count, in double (torch evaluates the bias corrections in double)
__device__ __forceinline__ double ipow(double b, unsigned t) {
    double r = 1.0;
    while (t) {
        if (t & 1u) r *= b;
        b *= b;
        t >>= 1;
    }
    return r;
}

__device__ __forceinline__ void adam_block(const AdamArgs& a) {
    __shared__ int s_start[ADAM_MAX_SEG + 1];
    __shared__ float s_ss[ADAM_MAX_SEG], s_bc2s[ADAM_MAX_SEG];
    const int tid = threadIdx.x;
    float* my_steps = a.steps + (size_t)blockIdx.x * a.n_seg;
    const int base = (blockIdx.x * ADAM_NT + tid) * 4;
    const bool full = base + 3 < a.n;
    // the data loads do not depend on the segment table: issue them first
    float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = p4, m4 = p4, v4 = p4;
    if (full) {
        p4 = *reinterpret_cast<const float4*>(a.p + base);
        g4 = *reinterpret_cast<const float4*>(a.g + base);
        m4 = *reinterpret_cast<const float4*>(a.m + base);
        v4 = *reinterpret_cast<const float4*>(a.v + base);
    }
    for (int i = tid; i <= a.n_seg; i += ADAM_NT) s_start[i] = a.seg_start[i];
    for (int i = tid; i < a.n_seg; i += ADAM_NT) {
        int skip = a.seg_skip ? a.seg_skip[i] : 0;
        if (a.gate_segs && !a.exec_flags[a.gate_segs[i].gate]) skip = 1;
        const float t0 = my_steps[i];
        const unsigned t = (unsigned)t0 + 1u;
        if (!skip) my_steps[i] = t0 + 1.f;
        const double bc1 = 1.0 - ipow(a.b1, t);
        const double bc2 = 1.0 - ipow(a.b2, t);
        s_ss[i] = skip ? -1.f : (float)(a.lr / bc1);
        s_bc2s[i] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float omb1 = (float)(1.0 - a.b1), b2 = (float)a.b2, omb2 = (float)(1.0 - a.b2);
    if (base < a.n) {
        int lo = 0, hi = a.n_seg - 1;                      // last segment whose start <= base
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_start[mid] <= base) lo = mid; else hi = mid - 1;
        }
        int seg = lo;
        if (full) {
            float* pp = &p4.x; const float* gp = &g4.x; float* mp = &m4.x; float* vp = &v4.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                while (base + j >= s_start[seg + 1]) ++seg;
                const float ss = s_ss[seg];
                if (ss >= 0.f) adam_elem(pp[j], gp[j], mp[j], vp[j], omb1, b2, omb2, a.eps, a.wd, ss, s_bc2s[seg], a.maximize);
            }
            *reinterpret_cast<float4*>(a.p + base) = p4;
            *reinterpret_cast<float4*>(a.m + base) = m4;
            *reinterpret_cast<float4*>(a.v + base) = v4;
        } else {
            for (int i = base; i < a.n; ++i) {
                while (i >= s_start[seg + 1]) ++seg;
                const float ss = s_ss[seg];
                if (ss >= 0.f) {
                    float pv = a.p[i], mv = a.m[i], vv = a.v[i];
                    adam_elem(pv, a.g[i], mv, vv, omb1, b2, omb2, a.eps, a.wd, ss, s_bc2s[seg], a.maximize);
                    a.p[i] = pv; a.m[i] = mv; a.v[i] = vv;
                }
            }
        }
    }
}

__global__ __launch_bounds__(ADAM_NT) void k_adam(const AdamArgs a) { adam_block(a); }

// ------------------------------------------------------------------------------------------------
// k_reduce: slabs -> gradient tensors (one thread per element); last block: tile partials -> stats
// (+ optional loss combination / epoch accumulation, + re-zeroing of the NaN flags)
// ------------------------------------------------------------------------------------------------
// Epoch accumulators (f64 sums of the f32 step values; tp/tn/fp/fn kept in fp32 like the reference's
// torch.zeros tensors, multimodn.py:112-115,209-212).  Split in two so that k_reduce can request the
// old sums at kernel start and only add + store once the step's stats exist: `st` may be an LDS copy.
struct EpochPre { double v[8]; };
template <class PlanLike>
__device__ __forceinline__ EpochPre epoch_prefetch(const PlanLike& p) {
    const int R = p.R, D = p.D, E = p.E, RD = R * D;
    const double* ep = p.epoch;
    const int c = threadIdx.x;
    EpochPre q;
#pragma unroll
    for (int k = 0; k < 8; ++k) q.v[k] = 0.0;
    if (c < RD) {
        q.v[0] = ep[c];
        q.v[1] = ep[RD + E + c];
        for (int k = 1; k < 5; ++k) q.v[1 + k] = ep[RD + E + k * RD + c];
    }
    if (c < E) q.v[6] = ep[RD + c];
    if (c < R) q.v[7] = ep[RD + E + 5 * RD + c];
    return q;
}
// needs blockDim.x >= max(R*D, 64); `st` holds the step's stats block (first RD + E + 5 RD + R entries)
template <class PlanLike>
__device__ __forceinline__ void epoch_apply(const PlanLike& p, const float* st, const EpochPre& q, float alpha, float beta) {
    const int R = p.R, D = p.D, E = p.E, RD = R * D;
    double* ep = p.epoch;
    const int c = threadIdx.x, lane = threadIdx.x & 63;
    if (c < 64) {                                          // fixed-order sums of grid and state change
        float se = 0.f, ss = 0.f;
        for (int k = lane; k < RD; k += 64) se += st[k];
        for (int e = lane; e < E; e += 64) ss += st[RD + e];
        se = wave_sum(se); ss = wave_sum(ss);
        if (lane == 0) {
            const float ge = se / (float)(D * R);          // multimodn.py:194
            const float gs = ss / (float)E;                // multimodn.py:196
            float* tail = p.stats + RD + E + 5 * RD + R;
            tail[0] = ge * alpha + gs * beta;              // multimodn.py:199-202
            tail[1] = ge; tail[2] = gs; tail[3] = 0.f;
            ep[RD + E + 5 * RD + R] += 1.0;                // n_steps
        }
    }
    if (c < RD) {
        ep[c] = q.v[0] + (double)st[c];                                    // err_loss_epoch (f64 += f32)
        ep[RD + E + c] = q.v[1] + (double)st[RD + E + c];                  // n_correct
        for (int k = 1; k < 5; ++k)                                        // tp/tn/fp/fn kept in fp32
            ep[RD + E + k * RD + c] = (double)((float)q.v[1 + k] + st[RD + E + k * RD + c]);
    }
    if (c < E) ep[RD + c] = q.v[6] + (double)st[RD + c];
    if (c < R) ep[RD + E + 5 * RD + c] = q.v[7] + (double)st[RD + E + 5 * RD + c];
}

constexpr int NTR = 1024;     // k_reduce block size
constexpr int MAXSEG = 2 * MMN_MAX_ENCODERS * MMN_MAX_LAYERS + 1 + 2 * MMN_MAX_DECODERS * (MMN_MAX_DEC_HIDDEN + 1);
static_assert(MAXSEG <= ADAM_MAX_SEG, "k_reduce's fused Adam shares the segment tables");
static_assert(NTR == 4 * ADAM_NT, "k_reduce and k_adam must use the same number of workgroups (step-counter rows)");

// Everything k_reduce reads, in the kernel arguments (no dependent load of the plan first).
struct RdArgs {
    const Seg* segs; const float* slabs; const float* lossp; const float* scp; const int32_t* cnt;
    const int32_t* exec_flags; float* stats; double* epoch;
    int64_t n_grad_elems;
    int32_t n_segs, R, D, E, S, pad;
    int64_t nA, nB;                     // gradient elements of (init state + encoders) / of the decoders
    int32_t ksA, ksB;                   // partial slabs per element in the two regions (ks, R * ks)
};

// adam_on: optimizer.step() (multimodn.py:204) fused behind the gradient sum - the thread that has
// just formed gradient element idx applies Adam to parameter idx (the flat gradient order IS the
// flat parameter order; the host checks it), with k_adam's arithmetic and per-workgroup step rows.
// A tensor whose encoder did not run this step has no gradient: it is left untouched.
__global__ __launch_bounds__(NTR) void k_reduce(const RdArgs r, const AdamArgs ad, int adam_on, int batch, int batch_global,
                                                int n_tiles, int grad_blocks, int want_grads, int accumulate,
                                                float alpha, float beta, int32_t* nan_flags,
                                                const int32_t* tile_rows, const int32_t* tile_seq) {
    __shared__ int s_exec[MMN_MAX_ENCODERS + 1];
    if ((int)blockIdx.x < grad_blocks) {
        if (!want_grads) return;
        const int tid = threadIdx.x;
        const int64_t idx = (int64_t)blockIdx.x * NTR + tid;
        const bool mine = idx < r.n_grad_elems;
        // the parameter / moment loads depend on nothing: first in the queue
        float pv = 0.f, mv = 0.f, vv = 0.f;
        if (adam_on) {
            const int64_t ci = mine ? idx : 0;
            pv = g_ld(ad.p + ci); mv = g_ld(ad.m + ci); vv = g_ld(ad.v + ci);
        }
        // ... and so do the partial sums: the slabs are flat-gradient shaped, element idx of partial k
        // sits at region base + k * region size + idx
        float sum = 0.f;
        {
            const int64_t ci = mine ? idx : 0;
            const bool inA = ci < r.nA;
            const float* src = inA ? r.slabs + ci : r.slabs + r.nA * r.ksA + (ci - r.nA);
            const int64_t pstride = inA ? r.nA : r.nB;
            const int np = inA ? r.ksA : r.ksB;
            int k = 0;
            for (; k + 8 <= np; k += 8) {                  // 8 independent loads in flight, fixed order
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = g_ld(src + (int64_t)(k + j) * pstride);
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += v[j];
            }
            for (; k < np; ++k) sum += g_ld(src + (int64_t)k * pstride);
        }
        // the segment table goes to LDS once per block: no dependent global reads before the partials
        __shared__ Seg sseg[MAXSEG];
        __shared__ float s_ss[MAXSEG], s_bc2s[MAXSEG];
        const int nseg = min(r.n_segs, MAXSEG);
        {
            const int nw = nseg * (int)(sizeof(Seg) / 4);
            const int32_t* src = reinterpret_cast<const int32_t*>(r.segs);
            int32_t* dst = reinterpret_cast<int32_t*>(sseg);
            for (int k = tid; k < nw; k += NTR) dst[k] = g_ldi(src + k);
        }
        if (tid <= r.E) s_exec[tid] = g_ldi(r.exec_flags + tid);
        float* my_steps = ad.steps + (size_t)blockIdx.x * (adam_on ? ad.n_seg : 0);
        float t0 = 0.f;
        int host_skip = 0;
        if (adam_on && tid < nseg) {
            t0 = g_ld(my_steps + tid);
            host_skip = ad.seg_skip ? g_ldi(ad.seg_skip + tid) : 0;
        }
        __syncthreads();
        if (adam_on) {
            if (tid < nseg) {
                const bool skip = host_skip || !s_exec[sseg[tid].gate];
                const unsigned t = (unsigned)t0 + 1u;
                const double bc1 = 1.0 - ipow(ad.b1, t);
                const double bc2 = 1.0 - ipow(ad.b2, t);
                s_ss[tid] = skip ? -1.f : (float)(ad.lr / bc1);
                s_bc2s[tid] = (float)sqrt(bc2);
                if (!skip) g_st(my_steps + tid, t0 + 1.f);
            }
            __syncthreads();
        }
        if (!mine) return;
        int lo = 0, hi = nseg - 1;
        while (lo < hi) {                                   // last segment with start <= idx
            const int mid = (lo + hi + 1) >> 1;
            if (sseg[mid].start <= idx) lo = mid; else hi = mid - 1;
        }
        const Seg sg = sseg[lo];
        const int local = (int)(idx - sg.start);
        if (sg.dst) g_st(sg.dst + local, sum);
        if (adam_on) {
            const float ss = s_ss[lo];
            if (ss >= 0.f) {
                adam_elem(pv, sum, mv, vv, (float)(1.0 - ad.b1), (float)ad.b2, (float)(1.0 - ad.b2), ad.eps, ad.wd, ss,
                          s_bc2s[lo], ad.maximize);
                g_st(ad.p + idx, pv); g_st(ad.m + idx, mv); g_st(ad.v + idx, vv);
            }
        }
        return;
    }
    // stats block: 8 threads per quantity (loss cells, state change, 5 counters per cell) sum
    // disjoint tile chunks with independent loads, then a fixed-order 8-way sum
    const int R = r.R, D = r.D, E = r.E, S = r.S;
    const int RD = R * D;
    float* st = r.stats;
    const float Bg = (float)batch_global;
    if ((int)threadIdx.x < R) s_exec[threadIdx.x] = g_ldi(r.exec_flags + threadIdx.x);
    EpochPre epre;
    if (accumulate) epre = epoch_prefetch(r);              // the old epoch sums travel while the partials are summed
    __shared__ float s_st[MMN_MAX_ENCODERS * 0 + 1024];    // this step's stats, for the epoch accumulation below
    __syncthreads();
    __shared__ float spart[NTR / 8][8];
    const int nq = RD + E + 5 * RD;
    for (int qb = 0; qb < nq; qb += NTR / 8) {
        const int qd = qb + (threadIdx.x >> 3), ch = threadIdx.x & 7;
        float fs = 0.f;
        if (qd < nq) {
            const int per = (n_tiles + 7) >> 3;
            const int t0 = ch * per, t1 = min(n_tiles, t0 + per);
            if (qd < RD) {
                if (s_exec[qd / D]) {
#pragma unroll 8
                    for (int t = t0; t < t1; ++t) fs += g_ld(r.lossp + (int64_t)t * RD + qd);
                }
            } else if (qd < RD + E) {
                const int e = qd - RD;
                if (s_exec[e + 1]) {
#pragma unroll 8
                    for (int t = t0; t < t1; ++t) fs += g_ld(r.scp + (int64_t)t * E + e);
                }
            } else {
                const int c = qd - RD - E;
                const int k = c / RD, cell = c - k * RD;
                if (s_exec[cell / D]) {
                    const int32_t* src = r.cnt + (int64_t)cell * 5 + k;
                    int v = 0;
#pragma unroll 8
                    for (int t = t0; t < t1; ++t) v += g_ldi(src + (int64_t)t * RD * 5);
                    fs = (float)v;                          // exact: per-chunk counts are far below 2^24
                }
            }
        }
        spart[threadIdx.x >> 3][ch] = fs;
        __syncthreads();
        if (qd < nq && ch == 0) {
            const float* sp = spart[threadIdx.x >> 3];
            const float tot = (((sp[0] + sp[1]) + (sp[2] + sp[3])) + ((sp[4] + sp[5]) + (sp[6] + sp[7])));
            const float val = qd < RD ? tot / Bg : (qd < RD + E ? tot / (Bg * (float)S) : tot);
            st[qd] = val;
            if (qd < 1024) s_st[qd] = val;
        }
        __syncthreads();
    }
    if (tile_seq == nullptr) {
        for (int q = threadIdx.x; q < R; q += NTR) {
            const float val = s_exec[q] ? (float)batch : 0.f;
            st[RD + E + 5 * RD + q] = val;
            if (RD + E + 5 * RD + q < 1024) s_st[RD + E + 5 * RD + q] = val;
        }
    } else {                                               // per-sample: samples that own grid row q
        const int q = threadIdx.x >> 6, ln = threadIdx.x & 63;
        if (q < R) {
            int cntq = 0;
            for (int t = ln; t < n_tiles; t += 64) {
                const int nr = g_ldi(tile_rows + t);
                bool has = q == 0;
                const unsigned code = (unsigned)g_ldi(tile_seq + t);
                for (int j = 0; j < 8; ++j) has |= (int)((code >> (4 * j)) & 15u) == q && q > 0;
                cntq += has ? nr : 0;
            }
            const float tot = wave_sum((float)cntq);       // exact: counts are far below 2^24
            if (ln == 0) { st[RD + E + 5 * RD + q] = tot; if (RD + E + 5 * RD + q < 1024) s_st[RD + E + 5 * RD + q] = tot; }
        }
    }
    if (nan_flags && threadIdx.x < MMN_MAX_ENCODERS) nan_flags[threadIdx.x] = 0;
    if (accumulate) {
        __syncthreads();
        // (stats blocks larger than the LDS mirror - more than 16 encoders x 8 decoders would be needed -
        //  fall back to reading the stats back from global memory)
        const bool fits = RD + E + 5 * RD + R <= 1024;
        if (!fits) __threadfence_block();
        epoch_apply(r, fits ? s_st : st, epre, alpha, beta);
    }
}

__global__ __launch_bounds__(NT) void k_epoch_accumulate(const DevPlan* __restrict__ P, float alpha, float beta) {
    const EpochPre q = epoch_prefetch(*P);
    epoch_apply(*P, P->stats, q, alpha, beta);
}

// Data-parallel tail in ONE launch (after the all-reduce of [grads | stats]): workgroups
// [0, adam_blocks) are k_adam, the last one is k_epoch_accumulate.
__global__ __launch_bounds__(ADAM_NT) void k_adam_accumulate(const AdamArgs a, int adam_blocks, const DevPlan* __restrict__ P,
                                                             float alpha, float beta) {
    if ((int)blockIdx.x < adam_blocks) { adam_block(a); return; }
    const EpochPre q = epoch_prefetch(*P);
    epoch_apply(*P, P->stats, q, alpha, beta);
}

}  // namespace

// ================================================================================================
// host side: plan + C ABI
// ================================================================================================
struct mmn_plan {
    mmn_model m;
    DevPlan host;            // host copy of the device plan
    DevPlan* dev;            // device address (start of workspace)
    int max_batch;
    size_t lds_bytes[3];     // by RT
    size_t par_lds_fwd, par_lds_bwd;
    int par_ok;
    ParArgs pa;              // kernel-argument descriptor of the 8-wave fast tier
    WgArgs wg;               // kernel-argument descriptor of k_wgrad
    RdArgs rd;               // kernel-argument descriptor of k_reduce
    int32_t* ps_scratch;     // per-sample regrouping scratch (3 x max_batch ints)
    std::vector<Seg> segs;   // host copy of the gradient segments (fused-Adam layout check)
    const void* adam_ok_seg_start;   // seg_start array already verified against `segs`
    const void* adam_ok_grads;
    size_t f8_lds_fwd, f8_lds_bwd, fb8_lds_bytes;
    int f8_ok, fb8_ok;
    int grad_blocks;
    int rt_override;
    int generic;             // k_gen_fwd / k_gen_bwd (a MIMIC_MLPEncoder or an MLPDecoder in the model)
    int dec_lds_rt;          // 0: decoder operands from global memory; 1: in LDS with 16-row tiles only; 2: any tile height
    int gen_fast;            // k_genf_fwd / k_genf_bwd apply (0, or the largest RT their LDS carve admits)
    GenArgs ga;              // their kernel-argument descriptor
    size_t gen_lds_fwd[3], gen_lds_bwd[3];   // by RT: chain carve (+ the LDS copy of the decoders' operands when it fits)
};

static thread_local int g_last_hip = 0;
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) { g_last_hip = (int)_e; return MMN_ERR_HIP; } \
    } while (0)

static int validate_model(const mmn_model* m) {
    if (!m) return MMN_ERR_ARG;
    if (m->state_size < 1 || m->n_encoders < 1 || m->n_decoders < 1) return MMN_ERR_ARG;
    if (m->state_size > MMN_MAX_DIM || m->n_encoders > MMN_MAX_ENCODERS || m->n_decoders > MMN_MAX_DECODERS)
        return MMN_ERR_UNSUPPORTED;
    for (int e = 0; e < m->n_encoders; ++e) {
        const mmn_encoder& enc = m->enc[e];
        if (enc.n_layers < 1 || enc.n_layers > MMN_MAX_LAYERS || enc.n_features < 1) return MMN_ERR_ARG;
        if (enc.activation < 0 || enc.activation > 2) return MMN_ERR_UNSUPPORTED;
        if (enc.kind != MMN_ENC_MLP && enc.kind != MMN_ENC_MIMIC) return MMN_ERR_UNSUPPORTED;
        const bool mimic = enc.kind == MMN_ENC_MIMIC;
        int in = enc.n_features + (mimic ? m->state_size : 0);      // mlp_encoder.py:22 (n_concat)
        for (int l = 0; l < enc.n_layers; ++l) {
            const mmn_linear& lin = enc.layer[l];
            const bool last = l == enc.n_layers - 1;
            if (lin.in_dim != in + (last && !mimic ? m->state_size : 0)) return MMN_ERR_ARG;
            if (lin.out_dim < 1 || (last && lin.out_dim != m->state_size)) return MMN_ERR_ARG;
            if (!last && lin.out_dim > MMN_MAX_DIM) return MMN_ERR_UNSUPPORTED;
            in = lin.out_dim;
        }
    }
    for (int d = 0; d < m->n_decoders; ++d) {
        const mmn_decoder& dec = m->dec[d];
        if (dec.n_hidden < 0 || dec.n_hidden > MMN_MAX_DEC_HIDDEN) return MMN_ERR_UNSUPPORTED;
        if (dec.n_hidden > 0 && (dec.hidden_activation < 0 || dec.hidden_activation > 2)) return MMN_ERR_UNSUPPORTED;
        int in = m->state_size;
        for (int l = 0; l < dec.n_hidden; ++l) {
            const mmn_linear& lin = dec.hidden[l];
            if (lin.in_dim != in || lin.out_dim < 1) return MMN_ERR_ARG;
            if (lin.out_dim > MMN_MAX_DIM) return MMN_ERR_UNSUPPORTED;
            in = lin.out_dim;
        }
    }
    return MMN_OK;
}

// models the sequential / parallel / 8-wave tiers do not cover go to k_gen_fwd / k_gen_bwd
static bool model_is_generic(const mmn_model& m) {
    if (const char* g = getenv("MMN_GENERIC")) { if (atoi(g)) return true; }
    for (int e = 0; e < m.n_encoders; ++e) if (m.enc[e].kind == MMN_ENC_MIMIC) return true;
    for (int d = 0; d < m.n_decoders; ++d) if (m.dec[d].n_hidden > 0) return true;
    return false;
}

namespace {
struct Layout {
    size_t off_plan, off_states, off_hid, off_dpre, off_dz, off_dS, off_pack, off_lossp, off_scp, off_cnt,
        off_flags, off_slabs, off_epoch, off_stamps, off_tasks, off_items, off_segs, off_ptasks, off_recs, off_sin, off_ps, total;
    int64_t hid_off[MMN_MAX_ENCODERS][MMN_MAX_LAYERS];
    int64_t pkf_off[MMN_MAX_ENCODERS][MMN_MAX_LAYERS];
    int64_t pkb_off[MMN_MAX_ENCODERS][MMN_MAX_LAYERS];
    int64_t pkh_off[MMN_MAX_ENCODERS];
    int64_t pkd_off;
    int64_t hid_floats, pack_floats, pack_elems;
    std::vector<WTask> tasks;
    std::vector<WItem> items;
    std::vector<WRec> recs;         // items with their task resolved: what k_wgrad reads
    std::vector<Seg> segs;
    std::vector<PackTask> ptasks;   // dst = offset until plan creation patches in the workspace address
    int64_t slab_floats, n_grad_elems, nA, nB;
    int KS, max_tiles, ldS, ldH;
    // generic tier
    int generic;
    size_t off_gact, off_gdpre;
    int64_t xin_off[MMN_MAX_ENCODERS];
    int64_t dh_off[MMN_MAX_DECODERS][MMN_MAX_DEC_HIDDEN];
    int64_t dh_row_stride, dh_base, gact_floats, gdpre_floats;
    int dcols, dec_maxnh;
    int32_t dwf_off[MMN_MAX_DECODERS];
    int64_t pkdf_off[MMN_MAX_DECODERS][MMN_MAX_DEC_HIDDEN + 1];
    int64_t pkdb_off[MMN_MAX_DECODERS][MMN_MAX_DEC_HIDDEN + 1];
    int64_t dec_f_off, dec_b_off, dec_f_floats, dec_b_floats;
    size_t off_biasbuf, off_biassrc, off_gfitems, off_gfitems_b;
    GfItemTable items_f;
    GfBwdTable items_b;
    std::vector<const float*> bias_src;
    int32_t ebias_off[MMN_MAX_ENCODERS][MMN_MAX_LAYERS];
    int32_t dbias_off[MMN_MAX_DECODERS][MMN_MAX_DEC_HIDDEN + 1];
    int gen_fast;
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int pick_width(int n) { return n >= 64 ? 4 : (n >= 32 ? 2 : 1); }

void build_layout(const mmn_model& m, int maxB, Layout& L) {
    const int S = m.state_size, E = m.n_encoders, D = m.n_decoders, R = E + 1;
    L.max_tiles = (maxB + 15) / 16;
    int split_rows = 512;                   // row-range splits: rows per workgroup (a quarter per wave)
    if (const char* sr = getenv("MMN_WGRAD_ROWS")) { const int v = atoi(sr); if (v >= 64) split_rows = v; }
    int ks = (maxB + split_rows - 1) / split_rows;
    if (ks < 1) ks = 1;
    if (ks > 64) ks = 64;
    L.KS = ks;
    L.ldS = pick_ld(S);
    int maxh = 16;
    int64_t ho = 0;
    memset(L.hid_off, 0, sizeof(L.hid_off));
    for (int e = 0; e < E; ++e)
        for (int l = 0; l + 1 < m.enc[e].n_layers; ++l) {
            L.hid_off[e][l] = ho;
            ho += (int64_t)maxB * m.enc[e].layer[l].out_dim;
            maxh = maxh > m.enc[e].layer[l].out_dim ? maxh : m.enc[e].layer[l].out_dim;
        }
    L.hid_floats = ho;
    L.generic = model_is_generic(m) ? 1 : 0;
    // generic tier buffers: xin[e] for MIMIC encoders, then per grid row the decoders' hidden activations
    {
        int64_t go = 0;
        for (int e = 0; e < MMN_MAX_ENCODERS; ++e) L.xin_off[e] = -1;
        memset(L.dh_off, 0, sizeof(L.dh_off));
        if (L.generic)
            for (int e = 0; e < E; ++e)
                if (m.enc[e].kind == MMN_ENC_MIMIC) { L.xin_off[e] = go; go += (int64_t)maxB * (m.enc[e].n_features + S); }
        go = (go + 3) / 4 * 4;                             // 16-byte aligned tile chunks
        L.dh_base = go;
        int64_t ro = 0;                                    // columns: the decoders' hidden layers side by side, so that
        L.dec_maxnh = 0;                                   // one tile's activations of a grid row are one contiguous chunk;
        if (L.generic)                                     // layer-major, every block in its own 16-column slot(s)
            for (int l = 0; l < MMN_MAX_DEC_HIDDEN; ++l)
                for (int d = 0; d < D; ++d) {
                    if (l >= m.dec[d].n_hidden) continue;
                    L.dec_maxnh = L.dec_maxnh > l + 1 ? L.dec_maxnh : l + 1;
                    L.dh_off[d][l] = ro;
                    ro += round_up(m.dec[d].hidden[l].out_dim, 16);
                    maxh = maxh > m.dec[d].hidden[l].out_dim ? maxh : m.dec[d].hidden[l].out_dim;
                }
        L.dcols = (int)ro;
        L.dh_row_stride = ro * maxB;
        L.gact_floats = go + L.dh_row_stride * R;
        L.gdpre_floats = L.dh_row_stride * R;
    }
    L.ldH = pick_ld(maxh);

    // ---- fragment-order repack of every operand of the two chain kernels
    int64_t po = 0;
    auto add_pack = [&](const float* src, int ld, int mode, int N, int len0, int col0, int len1, int col1, int kk_off,
                        int T_override, int64_t dst_off) -> int64_t {
        PackTask t{};
        t.src = src; t.dst = reinterpret_cast<float*>(dst_off); t.ld_src = ld; t.mode = mode; t.N = N;
        t.len0 = len0; t.col0 = col0; t.len1 = len1; t.col1 = col1; t.kk_off = kk_off;
        t.T = T_override > 0 ? T_override : (round_up(len0, 16) + round_up(len1, 16)) / 16;
        t.ntiles = (N + 15) / 16;
        t.start = L.ptasks.empty() ? 0 : L.ptasks.back().start + (int64_t)L.ptasks.back().ntiles * L.ptasks.back().T * 256;
        L.ptasks.push_back(t);
        return (int64_t)t.ntiles * t.T * 256;
    };
    for (int e = 0; e < E; ++e) {
        const int nl = m.enc[e].n_layers;
        if (m.enc[e].kind == MMN_ENC_MIMIC) {
            // MIMIC_MLPEncoder: layer 0 contracts [state columns F..F+S) first, then the x columns]; its backward
            // operand is the carry W_0[:, F:F+S]^T (no grad flows to x); every further layer is plain W / W^T
            const int F = m.enc[e].n_features;
            L.pkh_off[e] = -1;
            for (int l = 0; l < nl; ++l) {
                const mmn_linear& lin = m.enc[e].layer[l];
                L.pkf_off[e][l] = po;
                po += l == 0 ? add_pack(lin.w, lin.in_dim, 0, lin.out_dim, S, F, F, 0, 0, 0, po)
                             : add_pack(lin.w, lin.in_dim, 0, lin.out_dim, lin.in_dim, 0, 0, 0, 0, 0, po);
                L.pkb_off[e][l] = po;
                po += l == 0 ? add_pack(lin.w + F, lin.in_dim, 1, S, lin.out_dim, 0, 0, 0, 0, 0, po)
                             : add_pack(lin.w, lin.in_dim, 1, lin.in_dim, lin.out_dim, 0, 0, 0, 0, 0, po);
            }
            continue;
        }
        for (int l = 0; l < nl; ++l) {
            const mmn_linear& lin = m.enc[e].layer[l];
            const bool last = l == nl - 1;
            const int HL = lin.in_dim - S;
            // forward operand W_l [out x in]; the state update contracts the state columns first
            L.pkf_off[e][l] = po;
            po += last ? add_pack(lin.w, lin.in_dim, 0, S, S, HL, HL, 0, 0, 0, po)
                       : add_pack(lin.w, lin.in_dim, 0, lin.out_dim, lin.in_dim, 0, 0, 0, 0, 0, po);
            // backward operand W_l^T (rows = input index, contraction = output index)
            if (last) {
                L.pkb_off[e][l] = po;                      // carry: state columns [HL, HL+S)
                po += add_pack(lin.w + HL, lin.in_dim, 1, S, S, 0, 0, 0, 0, 0, po);
                L.pkh_off[e] = -1;
                if (nl > 1) {                              // dh: h columns [0, HL) (no grad flows to x)
                    L.pkh_off[e] = po;
                    po += add_pack(lin.w, lin.in_dim, 1, HL, S, 0, 0, 0, 0, 0, po);
                }
            } else if (l >= 1) {
                L.pkb_off[e][l] = po;
                po += add_pack(lin.w, lin.in_dim, 1, lin.in_dim, lin.out_dim, 0, 0, 0, 0, 0, po);
            } else {
                L.pkb_off[e][l] = -1;
            }
        }
    }
    L.pkd_off = po;                                        // Wdec^T: rows = state index, contraction = 2d + c
    {
        int64_t sz = 0;
        for (int d = 0; d < D; ++d) sz = add_pack(m.dec[d].w, S, 1, S, 2, 0, 0, 0, 2 * d, 1, po);
        po += sz;
    }
    memset(L.pkdf_off, 0, sizeof(L.pkdf_off));
    memset(L.pkdb_off, 0, sizeof(L.pkdb_off));
    L.dec_f_off = L.dec_b_off = po; L.dec_f_floats = L.dec_b_floats = 0;
    if (L.generic) {                                       // per-decoder operands of the generic tier: all forward
        L.dec_f_off = po;                                  // operands contiguous, then all backward ones (one LDS copy each)
        for (int d = 0; d < D; ++d) {
            const mmn_decoder& dec = m.dec[d];
            int in = S;
            for (int l = 0; l < dec.n_hidden; ++l) {
                const mmn_linear& lin = dec.hidden[l];
                L.pkdf_off[d][l] = po;
                po += add_pack(lin.w, lin.in_dim, 0, lin.out_dim, lin.in_dim, 0, 0, 0, 0, 0, po);
                in = lin.out_dim;
            }
            L.pkdf_off[d][dec.n_hidden] = po;              // output Linear [2 x in]
            po += add_pack(dec.w, in, 0, 2, in, 0, 0, 0, 0, 0, po);
        }
        L.dec_f_floats = po - L.dec_f_off;
        L.dec_b_off = po;
        for (int d = 0; d < D; ++d) {
            const mmn_decoder& dec = m.dec[d];
            int in = S;
            for (int l = 0; l < dec.n_hidden; ++l) {
                const mmn_linear& lin = dec.hidden[l];
                L.pkdb_off[d][l] = po;
                po += add_pack(lin.w, lin.in_dim, 1, lin.in_dim, lin.out_dim, 0, 0, 0, 0, 0, po);
                in = lin.out_dim;
            }
            L.pkdb_off[d][dec.n_hidden] = po;              // its transpose [in x 2]
            po += add_pack(dec.w, in, 1, in, 2, 0, 0, 0, 0, 0, po);
        }
        L.dec_b_floats = po - L.dec_b_off;
    }
    L.pack_floats = po;
    L.pack_elems = L.ptasks.back().start + (int64_t)L.ptasks.back().ntiles * L.ptasks.back().T * 256;

    // ---- wgrad tasks, work items, slabs, gradient segments
    int64_t gstart = 0;
    // flat gradient order = init state, encoders (layer by layer: weight, bias), decoders (weight, bias):
    // region A = everything before the decoders (ks partials per element), region B = decoders (R * ks)
    int64_t nA = S, nB = (int64_t)D * (2 * S + 2);
    if (L.generic) {
        nB = 0;
        for (int d = 0; d < D; ++d) {
            int in = S;
            for (int l = 0; l < m.dec[d].n_hidden; ++l) {
                nB += (int64_t)m.dec[d].hidden[l].out_dim * (m.dec[d].hidden[l].in_dim + 1);
                in = m.dec[d].hidden[l].out_dim;
            }
            nB += 2 * (in + 1);
        }
    }
    for (int e = 0; e < E; ++e)
        for (int l = 0; l < m.enc[e].n_layers; ++l)
            nA += (int64_t)m.enc[e].layer[l].out_dim * m.enc[e].layer[l].in_dim + m.enc[e].layer[l].out_dim;
    L.nA = nA; L.nB = nB;
    auto add_items = [&](int task, int M, int src, int ncols, bool bias_here, int nks) {
        // tile the [M x ncols] block of one source with interleave widths matched to what is left
        for (int m0 = 0; m0 < M;) {
            const int mt = pick_width(M - m0);
            bool first_n = true;
            int n0 = 0;
            do {
                const int nt = src == 2 ? 1 : pick_width(ncols - n0);
                for (int k = 0; k < nks; ++k)
                    L.items.push_back(WItem{task, m0, n0, k, nks, (int16_t)mt, (int16_t)nt, (int16_t)src,
                                            (int16_t)(bias_here && first_n ? 1 : 0)});
                first_n = false;
                n0 += 16 * nt;
            } while (src != 2 && n0 < ncols);
            m0 += 16 * mt;
        }
    };
    auto add_seg = [&](float* dst, int count, int gate) {   // dst may be NULL: the tensor keeps its place, nothing is stored
        L.segs.push_back(Seg{dst, gstart, 0, 0, count, 0, 1, 0, 0, 0, gate, 0});
        gstart += count;
    };
    // init state: column sums of dS0
    {
        WTask t{};
        t.a_kind = A_DS; t.a_idx = E; t.M = S;
        t.in0_kind = IN_NONE; t.in1_kind = IN_NONE; t.k0 = 0; t.k1 = 0; t.bias = 1; t.ntot = 1; t.gate = 0;
        t.part_base = 0; t.part_stride = nA; t.w_flat = 0; t.b_flat = gstart; t.w_ld = 1; t.dec_stride = 0;
        const int id = (int)L.tasks.size();
        L.tasks.push_back(t);
        add_items(id, S, 2, 0, true, ks);
        add_seg(m.g_init_state, S, 0);
    }
    for (int e = 0; e < E; ++e) {
        const mmn_encoder& enc = m.enc[e];
        const int nl = enc.n_layers;
        for (int l = 0; l < nl; ++l) {
            const mmn_linear& lin = enc.layer[l];
            const bool last = l == nl - 1;
            WTask t{};
            t.M = lin.out_dim;
            if (last) { t.a_kind = A_DS; t.a_idx = e; } else { t.a_kind = A_DPRE; t.a_enc = e; t.a_idx = l; }
            t.in0_enc = e;
            if (l == 0) { t.in0_kind = IN_X; } else { t.in0_kind = IN_HID; t.in0_idx = l - 1; }
            t.k0 = last ? lin.in_dim - S : lin.in_dim;
            t.in1_kind = last ? IN_PREV_STATE : IN_NONE;
            t.k1 = last ? S : 0;
            const bool mimic = enc.kind == MMN_ENC_MIMIC;
            if (mimic) {       // every layer has ONE input: xin[e] = (masked) cat[x, state] for layer 0, else the previous output
                t.k0 = lin.in_dim; t.in1_kind = IN_NONE; t.k1 = 0;
                if (l == 0) { t.in0_kind = IN_GEN; t.in_gen_off = L.xin_off[e]; t.gen_ldi = lin.in_dim; }
            }
            t.bias = 1; t.ntot = lin.in_dim + 1; t.gate = e + 1;
            t.part_base = 0; t.part_stride = nA; t.w_ld = lin.in_dim; t.dec_stride = 0;
            t.w_flat = gstart; t.b_flat = gstart + (int64_t)lin.out_dim * lin.in_dim;
            const int id = (int)L.tasks.size();
            L.tasks.push_back(t);
            // (doubling the row-range splits for the big state-update tiles was measured slower: every
            // extra work item costs ~5 us of fixed prologue/epilogue)
            add_items(id, t.M, 0, t.k0, true, ks);
            if (last && !mimic) add_items(id, t.M, 1, t.k1, false, ks);
            add_seg(lin.gw, lin.out_dim * lin.in_dim, e + 1);
            add_seg(lin.gb, lin.out_dim, e + 1);
        }
    }
    if (L.generic) {
        // generic tier: one task per (grid row, decoder, layer); region B holds the decoders' parameters in flat order
        // (hidden layers first, the output Linear last), R * ks partials per element
        for (int r = 0; r < R; ++r) {
            int64_t fo = 0;
            for (int d = 0; d < D; ++d) {
                const mmn_decoder& dec = m.dec[d];
                const int nh = dec.n_hidden;
                for (int l = 0; l <= nh; ++l) {
                    const bool out_layer = l == nh;
                    const int odim = out_layer ? 2 : dec.hidden[l].out_dim;
                    const int idim = l == 0 ? S : dec.hidden[l - 1].out_dim;
                    WTask t{};
                    t.M = odim;
                    if (out_layer) { t.a_kind = A_DZ; t.a_idx = r; t.a_col = 2 * d; }
                    else { t.a_kind = A_GEN; t.a_gen_off = (int64_t)r * L.dh_row_stride + L.dh_off[d][l]; t.gen_lda = L.dcols; }
                    if (l == 0) { t.in0_kind = IN_STATE_ROW; t.in0_idx = r; }
                    else { t.in0_kind = IN_GEN; t.in_gen_off = L.dh_base + (int64_t)r * L.dh_row_stride + L.dh_off[d][l - 1]; t.gen_ldi = L.dcols; }
                    t.k0 = idim; t.in1_kind = IN_NONE; t.k1 = 0;
                    t.bias = 1; t.ntot = idim + 1; t.gate = r;
                    t.part_base = nA * ks + (int64_t)r * ks * nB; t.part_stride = nB;
                    t.w_flat = fo; t.b_flat = fo + (int64_t)odim * idim; t.w_ld = idim; t.dec_stride = 0;
                    fo += (int64_t)odim * (idim + 1);
                    const int id = (int)L.tasks.size();
                    L.tasks.push_back(t);
                    add_items(id, t.M, 0, t.k0, true, ks);
                }
            }
        }
        for (int d = 0; d < D; ++d) {
            const mmn_decoder& dec = m.dec[d];
            int in = S;
            for (int l = 0; l < dec.n_hidden; ++l) {
                add_seg(dec.hidden[l].gw, dec.hidden[l].out_dim * dec.hidden[l].in_dim, 0);
                add_seg(dec.hidden[l].gb, dec.hidden[l].out_dim, 0);
                in = dec.hidden[l].out_dim;
            }
            add_seg(dec.gw, 2 * in, 0);
            add_seg(dec.gb, 2, 0);
        }
    } else
    // decoders: one task per grid row, all rows share one output of [2D x (S+1)] x (R*ks) partials
    {
        for (int r = 0; r < R; ++r) {
            WTask t{};
            t.a_kind = A_DZ; t.a_idx = r; t.M = 2 * D;
            t.in0_kind = IN_STATE_ROW; t.in0_idx = r; t.k0 = S; t.in1_kind = IN_NONE; t.k1 = 0;
            t.bias = 1; t.ntot = S + 1; t.gate = r;
            // region B: decoder d's weight [2 x S] at d * (2S + 2), its bias [2] right behind
            t.part_base = nA * ks + (int64_t)r * ks * nB; t.part_stride = nB;
            t.w_flat = 0; t.b_flat = 2 * S; t.w_ld = S; t.dec_stride = 2 * S + 2;
            const int id = (int)L.tasks.size();
            L.tasks.push_back(t);
            add_items(id, t.M, 0, S, true, ks);
        }
        for (int d = 0; d < D; ++d) {
            add_seg(m.dec[d].gw, 2 * S, 0);
            add_seg(m.dec[d].gb, 2, 0);
        }
    }
    L.slab_floats = nA * ks + nB * (int64_t)R * ks;
    L.n_grad_elems = gstart;
    // Launch order of the work items.  One workgroup per item; the dispatcher fills the 256 CUs
    // round-robin, so items beyond the first 256 double up on the CUs that got the first ones.  Put
    // the cheapest items at both ends (ascending cost, rotated by the overflow count): the CUs that
    // run two items then run two small ones, not a small one on top of a 64x64 tile.
    {
        auto cost = [](const WItem& w) { return (w.src == 2 ? 0 : (int)w.mt * w.nt) + (w.bias ? (int)w.mt : 0); };
        std::stable_sort(L.items.begin(), L.items.end(), [&](const WItem& x, const WItem& y) { return cost(x) < cost(y); });
        const size_t n = L.items.size();
        if (n > 256) {
            const size_t extra = std::min(n - 256, (size_t)256);
            std::rotate(L.items.begin(), L.items.begin() + extra, L.items.end());
        }
        if (getenv("MMN_VERBOSE")) {
            int hist[40] = {0};
            for (const auto& w : L.items) hist[std::min(cost(w), 39)]++;
            fprintf(stderr, "[mmn] wgrad items: %zu; cost histogram:", n);
            for (int c = 0; c < 40; ++c) if (hist[c]) fprintf(stderr, " %d:%d", c, hist[c]);
            fprintf(stderr, "\n");
        }
    }

    for (const WItem& it : L.items) {
        const WTask& t = L.tasks[it.task];
        WRec r{};
        r.a_kind = t.a_kind;
        if (t.a_kind == A_DPRE) { r.a_off = L.hid_off[t.a_enc][t.a_idx]; r.lda = t.M; }
        else if (t.a_kind == A_DS) { r.a_off = (int64_t)t.a_idx * maxB * S; r.lda = S; }
        else if (t.a_kind == A_GEN) { r.a_off = t.a_gen_off; r.lda = t.gen_lda; }
        else { r.a_off = (int64_t)t.a_idx * maxB * (2 * D) + t.a_col; r.lda = 2 * D; }
        r.M = t.M; r.m0 = it.m0; r.n0 = it.n0; r.mt = it.mt; r.nt = it.nt; r.bias = it.bias; r.ks = it.ks;
        r.has_in = it.src != 2 ? 1 : 0;
        const int kind = it.src == 0 ? t.in0_kind : (it.src == 1 ? t.in1_kind : IN_NONE);
        r.in_kind = kind; r.in_enc = t.in0_enc; r.in_off = 0; r.ldi = 0;
        r.ncols = it.src == 0 ? t.k0 : (it.src == 1 ? t.k1 : 0);
        if (kind == IN_HID) { r.in_off = L.hid_off[t.in0_enc][t.in0_idx]; r.ldi = m.enc[t.in0_enc].layer[t.in0_idx].out_dim; }
        else if (kind == IN_GEN) { r.in_off = t.in_gen_off; r.ldi = t.gen_ldi; }
        else if (kind == IN_STATE_ROW) {
            if (t.in0_idx == 0) r.in_kind = IN_INIT;
            else { r.in_off = (int64_t)(t.in0_idx - 1) * maxB * S; r.ldi = S; }
        }
        r.w_ld = t.w_ld; r.dec_stride = t.dec_stride;
        r.col_off = (it.src == 1 ? t.k0 : 0) + it.n0;
        r.gate = t.gate;
        r.w_off = t.part_base + (int64_t)it.ks * t.part_stride + t.w_flat;
        r.b_off = t.part_base + (int64_t)it.ks * t.part_stride + t.b_flat;
        r.nks = it.nks;
        L.recs.push_back(r);
    }

    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + (bytes ? bytes : 4), 256); return at; };
    L.off_plan = take(sizeof(DevPlan));
    L.off_states = take(sizeof(float) * (size_t)E * maxB * S);
    L.off_hid = take(sizeof(float) * (size_t)L.hid_floats);
    L.off_dpre = take(sizeof(float) * (size_t)L.hid_floats);
    L.off_dz = take(sizeof(float) * (size_t)R * maxB * 2 * D);
    L.off_dS = take(sizeof(float) * (size_t)(E + 1) * maxB * S);
    L.off_pack = take(sizeof(float) * (size_t)L.pack_floats);
    L.off_lossp = take(sizeof(float) * (size_t)L.max_tiles * R * D);
    L.off_scp = take(sizeof(float) * (size_t)L.max_tiles * E);
    L.off_cnt = take(sizeof(int32_t) * (size_t)L.max_tiles * R * D * 5);
    L.off_flags = take(sizeof(int32_t) * (size_t)(R + E + MMN_MAX_ENCODERS));
    L.off_slabs = take(sizeof(float) * (size_t)L.slab_floats);
    L.off_epoch = take(sizeof(double) * mmn_epoch_doubles(&m));
    L.off_stamps = take(sizeof(long long) * 256);
    L.off_tasks = take(sizeof(WTask) * L.tasks.size());
    L.off_items = take(sizeof(WItem) * L.items.size());
    L.off_segs = take(sizeof(Seg) * L.segs.size());
    L.off_ptasks = take(sizeof(PackTask) * L.ptasks.size());
    L.off_recs = take(sizeof(WRec) * L.items.size());
    L.off_sin = take(sizeof(float) * (size_t)E * maxB * S);
    L.off_ps = take(sizeof(int32_t) * 3 * (size_t)maxB);      // per-sample regrouping scratch: codes, masks, source rows
    L.off_gact = take(sizeof(float) * (size_t)L.gact_floats);
    L.off_gdpre = take(sizeof(float) * (size_t)L.gdpre_floats);
    // bias gather table (generic tier only)
    memset(L.ebias_off, 0, sizeof(L.ebias_off));
    memset(L.dbias_off, 0, sizeof(L.dbias_off));
    memset(L.dwf_off, 0, sizeof(L.dwf_off));
    if (L.generic) {
        auto add_bias = [&](const float* b, int n) { const int at = (int)L.bias_src.size(); for (int k = 0; k < n; ++k) L.bias_src.push_back(b + k); return at; };
        for (int e = 0; e < E; ++e)
            for (int l = 0; l < m.enc[e].n_layers; ++l) L.ebias_off[e][l] = add_bias(m.enc[e].layer[l].b, m.enc[e].layer[l].out_dim);
        for (int d = 0; d < D; ++d) {
            for (int l = 0; l < m.dec[d].n_hidden; ++l) L.dbias_off[d][l] = add_bias(m.dec[d].hidden[l].b, m.dec[d].hidden[l].out_dim);
            L.dbias_off[d][m.dec[d].n_hidden] = add_bias(m.dec[d].b, 2);
        }
        for (int d = 0; d < D; ++d) {                      // raw [2 x in] output weights: the K = 2 products of the backward
            const int in = m.dec[d].n_hidden ? m.dec[d].hidden[m.dec[d].n_hidden - 1].out_dim : S;   // half are plain FMAs
            L.dwf_off[d] = add_bias(m.dec[d].w, 2 * in);
        }
    }
    L.off_biasbuf = take(sizeof(float) * L.bias_src.size());
    L.off_biassrc = take(sizeof(const float*) * L.bias_src.size());
    // fast form of the generic tier: every encoder a MIMIC_MLPEncoder whose operands fit the register prefetch
    // (<= 3 layers, hidden widths <= 64, features / state <= 128, layer 0 within TQ k-steps)
    L.gen_fast = L.generic && S <= 128 && L.dcols <= 256 && E <= GF_MAXE && D <= GF_MAXD ? 1 : 0;
    for (int e = 0; e < E && L.gen_fast; ++e) {
        const mmn_encoder& enc = m.enc[e];
        bool ok = enc.kind == MMN_ENC_MIMIC && enc.n_layers <= 3 && enc.n_features <= 128 &&
                  (round_up(S, 16) + round_up(enc.n_features, 16)) / 16 <= TQ;
        for (int l = 0; l + 1 < enc.n_layers; ++l) ok = ok && enc.layer[l].out_dim <= 64;
        if (!ok) L.gen_fast = 0;
    }
    if (const char* gf = getenv("MMN_GEN_FAST")) { if (atoi(gf) == 0) L.gen_fast = 0; }
    memset(&L.items_f, 0, sizeof(L.items_f));
    if (L.gen_fast) {                                      // decoder phases, dealt to the four waves (see genf_decode)
        for (int ph = 0; ph <= L.dec_maxnh && L.gen_fast; ++ph) {
            int g = 0;
            for (int d = 0; d < D && L.gen_fast; ++d) {
                const mmn_decoder& dec = m.dec[d];
                const int nh = dec.n_hidden;
                if (ph > nh) continue;
                const bool fin = ph == nh;
                const int N = fin ? 2 : dec.hidden[ph].out_dim;
                const int K = ph == 0 ? S : dec.hidden[ph - 1].out_dim;
                const int T = (K + 15) / 16, ntl = (N + 15) / 16;
                for (int tile = 0; tile < ntl; ++tile, ++g) {
                    const int w = g & 3;
                    int32_t& n = L.items_f.cnt[ph][w];
                    if (n >= GF_SLOTS) { L.gen_fast = 0; break; }
                    GfItem& it = L.items_f.item[ph][w][n++];
                    it.a_col = ph == 0 ? -1 : (int32_t)L.dh_off[d][ph - 1];
                    it.T = T;
                    it.w_off = (int32_t)(L.pkdf_off[d][ph] - L.dec_f_off) + tile * T * 256;
                    it.bias_off = L.dbias_off[d][ph];
                    it.out_col = fin ? 2 * d : (int32_t)L.dh_off[d][ph];
                    it.n_valid = N;
                    it.col0 = 16 * tile;
                    it.kind_hk = (fin ? 256 : 0) | (dec.hidden_activation & 255);
                }
            }
        }
    }
    memset(&L.items_b, 0, sizeof(L.items_b));
    if (L.gen_fast) {
        GfBwdTable& tb = L.items_b;
        for (int d = 0; d < D; ++d) {                      // every decoder's output Linear
            const mmn_decoder& dec = m.dec[d];
            const int nh = dec.n_hidden;
            int32_t* t = tb.top[tb.n_top++];
            t[0] = d; t[1] = nh; t[2] = dec.hidden_activation;
            t[3] = nh ? dec.hidden[nh - 1].out_dim : S;
            t[4] = nh ? (int32_t)L.dh_off[d][nh - 1] : 0;
            t[5] = L.dwf_off[d];
            if (nh) {
                int32_t* gdv = tb.gd[tb.n_gd++];
                gdv[0] = (int32_t)L.dh_off[d][0];
                gdv[1] = (dec.hidden[0].out_dim + 15) / 16;
                gdv[2] = (int32_t)(L.pkdb_off[d][0] - L.dec_b_off);
            }
        }
        for (int ph = L.dec_maxnh - 1; ph >= 1 && L.gen_fast; --ph) {
            int g = 0;
            for (int d = 0; d < D && L.gen_fast; ++d) {
                const mmn_decoder& dec = m.dec[d];
                if (dec.n_hidden <= ph) continue;
                const int N = dec.hidden[ph - 1].out_dim, K = dec.hidden[ph].out_dim;
                const int T = (K + 15) / 16, ntl = (N + 15) / 16;
                for (int tile = 0; tile < ntl; ++tile, ++g) {
                    const int w = g & 3;
                    int32_t& n = tb.cnt[ph][w];
                    if (n >= GF_SLOTS) { L.gen_fast = 0; break; }
                    GfItem& it = tb.item[ph][w][n++];
                    it.a_col = (int32_t)L.dh_off[d][ph];
                    it.T = T;
                    it.w_off = (int32_t)(L.pkdb_off[d][ph] - L.dec_b_off) + tile * T * 256;
                    it.bias_off = 0;
                    it.out_col = (int32_t)L.dh_off[d][ph - 1];
                    it.n_valid = N;
                    it.col0 = 16 * tile;
                    it.kind_hk = dec.hidden_activation & 255;
                }
            }
        }
    }
    L.off_gfitems = take(sizeof(GfItemTable));
    L.off_gfitems_b = take(sizeof(GfBwdTable));
    L.total = o;
}

int choose_rt(const mmn_plan* p, int batch) {
    if (p->rt_override == 1 || p->rt_override == 2) return p->rt_override;
    return batch <= 16 * 320 ? 1 : 2;        // keep >= one workgroup per CU busy as long as possible
}
}  // namespace

extern "C" {

int mmn_version(void) { return MMN_VERSION; }

const char* mmn_error_string(int code) {
    switch (code) {
        case MMN_OK: return "ok";
        case MMN_ERR_ARG: return "invalid argument";
        case MMN_ERR_UNSUPPORTED: return "unsupported model dimensions";
        case MMN_ERR_WORKSPACE: return "workspace too small or misaligned";
        case MMN_ERR_HIP: return "HIP runtime error";
        case MMN_ERR_SEQUENCE: return "invalid encoder sequence";
        default: return "unknown error";
    }
}

int mmn_last_hip_error(void) { return g_last_hip; }

size_t mmn_stats_floats(const mmn_model* m) {
    if (!m) return 0;
    const size_t R = m->n_encoders + 1, D = m->n_decoders, E = m->n_encoders;
    return R * D + E + 5 * R * D + R + 4;
}

size_t mmn_epoch_doubles(const mmn_model* m) {
    if (!m) return 0;
    const size_t R = m->n_encoders + 1, D = m->n_decoders, E = m->n_encoders;
    return R * D + E + 5 * R * D + R + 1;
}

size_t mmn_workspace_bytes(const mmn_model* m, int max_batch) {
    if (validate_model(m) != MMN_OK || max_batch < 1) return 0;
    Layout L;
    build_layout(*m, max_batch, L);
    const ChainLds cl = chain_lds(16, L.ldS, L.ldH);
    if (sizeof(float) * (size_t)cl.total > 160 * 1024) return 0;
    return L.total;
}

int mmn_plan_create(const mmn_model* m, int max_batch, void* workspace, size_t workspace_bytes, float* stats,
                    mmn_plan** out) {
    if (!out) return MMN_ERR_ARG;
    *out = nullptr;
    int rc = validate_model(m);
    if (rc != MMN_OK) return rc;
    if (max_batch < 1 || !workspace || !stats) return MMN_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return MMN_ERR_WORKSPACE;
    Layout L;
    build_layout(*m, max_batch, L);
    if (workspace_bytes < L.total) return MMN_ERR_WORKSPACE;

    mmn_plan* pl = new (std::nothrow) mmn_plan();
    if (!pl) return MMN_ERR_ARG;
    pl->m = *m;
    pl->max_batch = max_batch;
    const char* env = getenv("MMN_RT");
    pl->rt_override = env ? atoi(env) : 0;
    char* ws = static_cast<char*>(workspace);
    DevPlan& h = pl->host;
    memset(&h, 0, sizeof(h));
    h.m = *m;
    h.S = m->state_size; h.E = m->n_encoders; h.D = m->n_decoders; h.R = h.E + 1;
    h.S16 = round_up(h.S, 16);
    h.ldS = L.ldS; h.ldH = L.ldH;
    h.maxB = max_batch; h.max_tiles = L.max_tiles; h.KS = L.KS;
    memcpy(h.hid_off, L.hid_off, sizeof(h.hid_off));
    memcpy(h.pkf_off, L.pkf_off, sizeof(h.pkf_off));
    memcpy(h.pkb_off, L.pkb_off, sizeof(h.pkb_off));
    memcpy(h.pkh_off, L.pkh_off, sizeof(h.pkh_off));
    h.pkd_off = L.pkd_off;
    h.states = reinterpret_cast<float*>(ws + L.off_states);
    h.hid = reinterpret_cast<float*>(ws + L.off_hid);
    h.dpre = reinterpret_cast<float*>(ws + L.off_dpre);
    h.dz = reinterpret_cast<float*>(ws + L.off_dz);
    h.dS = reinterpret_cast<float*>(ws + L.off_dS);
    h.pack = reinterpret_cast<float*>(ws + L.off_pack);
    h.lossp = reinterpret_cast<float*>(ws + L.off_lossp);
    h.scp = reinterpret_cast<float*>(ws + L.off_scp);
    h.cnt = reinterpret_cast<int32_t*>(ws + L.off_cnt);
    h.exec_flags = reinterpret_cast<int32_t*>(ws + L.off_flags);
    h.prev_row = h.exec_flags + h.R;
    h.nan_flags = h.prev_row + h.E;
    h.slabs = reinterpret_cast<float*>(ws + L.off_slabs);
    h.stats = stats;
    h.epoch = reinterpret_cast<double*>(ws + L.off_epoch);
    {
        const char* se = getenv("MMN_STAMPS");
        h.stamps = (se && atoi(se)) ? reinterpret_cast<long long*>(ws + L.off_stamps) : nullptr;
    }
    h.tasks = reinterpret_cast<WTask*>(ws + L.off_tasks);
    h.items = reinterpret_cast<WItem*>(ws + L.off_items);
    h.segs = reinterpret_cast<Seg*>(ws + L.off_segs);
    h.ptasks = reinterpret_cast<PackTask*>(ws + L.off_ptasks);
    h.recs = reinterpret_cast<WRec*>(ws + L.off_recs);
    h.sin = reinterpret_cast<float*>(ws + L.off_sin);
    h.n_tasks = (int)L.tasks.size(); h.n_items = (int)L.items.size(); h.n_segs = (int)L.segs.size();
    h.n_ptasks = (int)L.ptasks.size();
    h.n_grad_elems = L.n_grad_elems;
    h.n_pack_elems = L.pack_elems;
    for (PackTask& t : L.ptasks) t.dst = h.pack + reinterpret_cast<intptr_t>(t.dst);   // offset -> address
    h.generic = L.generic;
    h.gact = reinterpret_cast<float*>(ws + L.off_gact);
    h.gdpre = reinterpret_cast<float*>(ws + L.off_gdpre);
    memcpy(h.xin_off, L.xin_off, sizeof(h.xin_off));
    memcpy(h.dh_off, L.dh_off, sizeof(h.dh_off));
    h.dh_row_stride = L.dh_row_stride; h.dh_base = L.dh_base; h.dcols = L.dcols; h.dec_maxnh = L.dec_maxnh;
    memcpy(h.dwf_off, L.dwf_off, sizeof(h.dwf_off));
    memcpy(h.pkdf_off, L.pkdf_off, sizeof(h.pkdf_off));
    memcpy(h.pkdb_off, L.pkdb_off, sizeof(h.pkdb_off));
    h.dec_f_off = L.dec_f_off; h.dec_b_off = L.dec_b_off;
    h.dec_f_floats = (int32_t)L.dec_f_floats; h.dec_b_floats = (int32_t)L.dec_b_floats;
    h.biasbuf = reinterpret_cast<float*>(ws + L.off_biasbuf);
    h.bias_src = reinterpret_cast<const float* const*>(ws + L.off_biassrc);
    h.n_bias = (int32_t)L.bias_src.size();
    memcpy(h.ebias_off, L.ebias_off, sizeof(h.ebias_off));
    memcpy(h.dbias_off, L.dbias_off, sizeof(h.dbias_off));
    pl->generic = L.generic;
    pl->dev = reinterpret_cast<DevPlan*>(ws + L.off_plan);
    pl->wg = WgArgs{h.recs, h.dpre, h.dS, h.dz, h.states, h.hid, m->init_state, h.slabs, h.stamps, h.sin, h.maxB, h.S,
                    h.gact, h.gdpre};
    pl->rd = RdArgs{h.segs, h.slabs, h.lossp, h.scp, h.cnt, h.exec_flags, h.stats, h.epoch, L.n_grad_elems,
                    (int32_t)L.segs.size(), h.R, h.D, h.E, h.S, 0, L.nA, L.nB, h.KS, h.R * h.KS};
    pl->segs = L.segs;
    pl->ps_scratch = reinterpret_cast<int32_t*>(ws + L.off_ps);
    pl->adam_ok_seg_start = nullptr; pl->adam_ok_grads = nullptr;
    pl->grad_blocks = (int)((L.n_grad_elems + NTR - 1) / NTR);
    {
        int maxF = 1;
        bool ok = h.S <= 128;
        for (int e = 0; e < h.E; ++e) maxF = maxF > m->enc[e].n_features ? maxF : m->enc[e].n_features;
        ok = ok && maxF <= 256;
        h.ldX = pick_ld(maxF);
        pl->par_lds_fwd = sizeof(float) * (size_t)par_lds(h.R, h.E, h.ldS, h.ldH, h.ldX, false).total;
        pl->par_lds_bwd = sizeof(float) * (size_t)par_lds(h.R, h.E, h.ldS, h.ldH, h.ldX, true).total;
        ok = ok && pl->par_lds_fwd <= 160 * 1024 && pl->par_lds_bwd <= 160 * 1024;
        const char* pe = getenv("MMN_PAR");
        if (pe && atoi(pe) == 0) ok = false;
        if (L.generic) ok = false;                         // the other tiers know MLPEncoder + ClassDecoder only
        pl->par_ok = ok ? 1 : 0;
        h.par_ok = pl->par_ok;
        if (getenv("MMN_VERBOSE"))
            fprintf(stderr, "[mmn] plan: sizeof(DevPlan)=%zu par_lds fwd=%zu bwd=%zu par_ok=%d ldS=%d ldH=%d ldX=%d\n",
                    sizeof(DevPlan), pl->par_lds_fwd, pl->par_lds_bwd, pl->par_ok, h.ldS, h.ldH, h.ldX);
    }
    {   // ---- 8-wave fast tier: eligibility + kernel-argument descriptor
        ParArgs& a = pl->pa;
        memset(&a, 0, sizeof(a));
        bool ok = pl->par_ok && h.E <= 8 && h.D <= MMN_MAX_DECODERS && h.S <= 128 && (h.S % 4) == 0;
        bool aligned16 = (h.S % 16) == 0;
        bool f_al = true;
        for (int e = 0; e < h.E && ok; ++e) {
            const mmn_encoder& enc = m->enc[e];
            const int Lh = enc.n_layers - 1;
            ParEnc& pe = a.enc[e];
            ok = ok && Lh <= 2 && enc.n_features <= 128;
            f_al = f_al && (enc.n_features % 4) == 0;       // k_fwd8 / k_bwd8 read x with 16-byte loads only
            if (!ok) break;
            pe.F = enc.n_features; pe.Lh = Lh; pe.akind = enc.activation;
            pe.HL = enc.layer[Lh].in_dim - h.S;
            ok = ok && pe.HL <= 64;
            aligned16 = aligned16 && (pe.F % 16) == 0 && (pe.HL % 16) == 0;
            for (int l = 0; l <= Lh; ++l) {
                pe.in[l] = enc.layer[l].in_dim; pe.out[l] = enc.layer[l].out_dim;
                pe.pkf[l] = L.pkf_off[e][l]; pe.pkb[l] = L.pkb_off[e][l];
                pe.bias[l] = enc.layer[l].b;
                if (l < Lh) {
                    pe.hid[l] = L.hid_off[e][l];
                    ok = ok && enc.layer[l].out_dim <= 32;
                    aligned16 = aligned16 && (enc.layer[l].out_dim % 16) == 0;
                }
            }
            pe.pkh = L.pkh_off[e];
        }
        a.S = h.S; a.E = h.E; a.D = h.D; a.R = h.R; a.S16 = h.S16; a.ldS = h.ldS; a.ldH = h.ldH; a.ldX = h.ldX;
        a.maxB = h.maxB; a.needs_zero = aligned16 ? 0 : 1;
        a.init = m->init_state; a.pack = h.pack; a.pkd = L.pkd_off;
        a.states = h.states; a.hid = h.hid; a.dpre = h.dpre; a.dz = h.dz; a.dS = h.dS; a.sin = h.sin;
        a.lossp = h.lossp; a.scp = h.scp; a.cnt = h.cnt; a.exec_flags = h.exec_flags; a.prev_row = h.prev_row;
        a.stamps = h.stamps;
        for (int d = 0; d < h.D && d < MMN_MAX_DECODERS; ++d) { a.dec_w[d] = m->dec[d].w; a.dec_b[d] = m->dec[d].b; }
        pl->f8_lds_fwd = sizeof(float) * (size_t)par8_lds(h.R, h.E, h.ldS, h.ldH, h.ldX, false).total;
        pl->f8_lds_bwd = sizeof(float) * (size_t)par8_lds(h.R, h.E, h.ldS, h.ldH, h.ldX, true).total;
        ok = ok && pl->f8_lds_fwd <= 160 * 1024 && pl->f8_lds_bwd <= 160 * 1024;
        const char* fe = getenv("MMN_FAST8");
        if (fe && atoi(fe) == 0) ok = false;
        pl->f8_ok = (ok && f_al) ? 1 : 0;
        pl->fb8_lds_bytes = sizeof(float) * (size_t)fb8_lds(h.R, h.ldS, h.ldH, h.ldX).total;
        {
            const char* fb = getenv("MMN_FUSED");
            pl->fb8_ok = (ok && h.E <= 4 && pl->fb8_lds_bytes <= 160 * 1024 && !(fb && atoi(fb) == 0)) ? 1 : 0;
        }
        if (getenv("MMN_VERBOSE"))
            fprintf(stderr, "[mmn] fast8: ok=%d fused=%d lds fwd=%zu bwd=%zu fused=%zu needs_zero=%d sizeof(ParArgs)=%zu\n",
                    pl->f8_ok, pl->fb8_ok, pl->f8_lds_fwd, pl->f8_lds_bwd, pl->fb8_lds_bytes, a.needs_zero, sizeof(ParArgs));
    }
    pl->lds_bytes[0] = 0;
    for (int rt = 1; rt <= 2; ++rt) pl->lds_bytes[rt] = sizeof(float) * (size_t)chain_lds(16 * rt, h.ldS, h.ldH).total;
    if (pl->lds_bytes[1] > 160 * 1024) { delete pl; return MMN_ERR_UNSUPPORTED; }
    {   // generic tier: LDS budgets.  Sequential form: the decoders' operands ride along when both directions fit next to
        // the 32-row carve (so that the choice does not depend on the batch), else next to the 16-row one with 16-row
        // tiles only.  Fast form: its own carve (operands + biases + parked activations).
        const size_t extra_f = sizeof(float) * (size_t)L.dec_f_floats, extra_b = sizeof(float) * (size_t)L.dec_b_floats;
        const size_t extra = extra_f > extra_b ? extra_f : extra_b;
        int fit = 0, fast = 0;
        auto fast_bytes = [&](int rt, bool bwd) {
            return sizeof(float) * (size_t)gen_fast_lds(16 * rt, h.ldS, h.ldH, (int)(bwd ? L.dec_b_floats : L.dec_f_floats),
                                                        (int)L.bias_src.size(), L.dcols, bwd).total;
        };
        if (L.generic && L.gen_fast) {
            if (fast_bytes(2, false) <= 160 * 1024 && fast_bytes(2, true) <= 160 * 1024 && L.dcols * 32 <= 8192) fast = 2;
            else if (fast_bytes(1, false) <= 160 * 1024 && fast_bytes(1, true) <= 160 * 1024) fast = 1;
        }
        if (!fast && L.generic && !(getenv("MMN_DEC_LDS") && atoi(getenv("MMN_DEC_LDS")) == 0)) {
            if (pl->lds_bytes[2] + extra <= 160 * 1024) fit = 2;
            else if (pl->lds_bytes[1] + extra <= 160 * 1024) fit = 1;
        }
        h.dec_lds = (fit || fast) ? 1 : 0;
        h.gen_fast = fast ? 1 : 0;
        pl->dec_lds_rt = fast ? fast : fit;
        pl->gen_fast = fast;
        for (int rt = 0; rt <= 2; ++rt) {
            pl->gen_lds_fwd[rt] = fast ? (rt ? fast_bytes(rt, false) : 0) : pl->lds_bytes[rt] + (fit ? extra_f : 0);
            pl->gen_lds_bwd[rt] = fast ? (rt ? fast_bytes(rt, true) : 0) : pl->lds_bytes[rt] + (fit ? extra_b : 0);
        }
        if (fast) {                                        // kernel-argument descriptor of the fast kernels
            GenArgs& a = pl->ga;
            memset(&a, 0, sizeof(a));
            a.m.init_state = m->init_state;
            for (int e = 0; e < h.E; ++e) {
                const mmn_encoder& enc = m->enc[e];
                a.m.enc[e].n_layers = enc.n_layers; a.m.enc[e].n_features = enc.n_features; a.m.enc[e].activation = enc.activation;
                for (int l = 0; l < enc.n_layers; ++l) {
                    a.m.enc[e].layer[l].out_dim = enc.layer[l].out_dim; a.m.enc[e].layer[l].in_dim = enc.layer[l].in_dim;
                    a.pkf_off[e][l] = L.pkf_off[e][l]; a.pkb_off[e][l] = L.pkb_off[e][l]; a.hid_off[e][l] = L.hid_off[e][l];
                    a.ebias_off[e][l] = L.ebias_off[e][l];
                }
                a.xin_off[e] = L.xin_off[e];
            }
            for (int d = 0; d < h.D; ++d) {
                const mmn_decoder& dec = m->dec[d];
                a.m.dec[d].n_hidden = dec.n_hidden; a.m.dec[d].hidden_activation = dec.hidden_activation;
                for (int l = 0; l < dec.n_hidden; ++l) {
                    a.m.dec[d].hidden[l].out_dim = dec.hidden[l].out_dim; a.m.dec[d].hidden[l].in_dim = dec.hidden[l].in_dim;
                    a.dh_off[d][l] = (int32_t)L.dh_off[d][l];
                }
                for (int l = 0; l <= dec.n_hidden; ++l) {
                    a.pkdf_off[d][l] = L.pkdf_off[d][l]; a.pkdb_off[d][l] = L.pkdb_off[d][l]; a.dbias_off[d][l] = L.dbias_off[d][l];
                }
                a.dwf_off[d] = L.dwf_off[d];
            }
            a.S = h.S; a.E = h.E; a.D = h.D; a.R = h.R; a.ldS = h.ldS; a.ldH = h.ldH; a.maxB = h.maxB; a.dcols = L.dcols;
            a.dec_maxnh = L.dec_maxnh; a.n_bias = (int32_t)L.bias_src.size();
            a.dec_f_floats = (int32_t)L.dec_f_floats; a.dec_b_floats = (int32_t)L.dec_b_floats;
            a.pack = h.pack; a.biasbuf = h.biasbuf;
            a.states = h.states; a.hid = h.hid; a.dpre = h.dpre; a.dz = h.dz; a.dS = h.dS; a.gact = h.gact; a.gdpre = h.gdpre;
            a.lossp = h.lossp; a.scp = h.scp; a.cnt = h.cnt; a.exec_flags = h.exec_flags; a.prev_row = h.prev_row; a.stamps = h.stamps;
            a.dec_f_off = L.dec_f_off; a.dec_b_off = L.dec_b_off; a.dh_base = L.dh_base; a.dh_row_stride = L.dh_row_stride;
            a.items_f = reinterpret_cast<const GfItemTable*>(ws + L.off_gfitems);
            a.items_b = reinterpret_cast<const GfBwdTable*>(ws + L.off_gfitems_b);
        }
        if (getenv("MMN_VERBOSE")) {
            fprintf(stderr, "[mmn] sizeof(GenArgs)=%zu sizeof(mmn_batch)=%zu sizeof(GfItemTable)=%zu\n", sizeof(GenArgs), sizeof(mmn_batch),
                    sizeof(GfItemTable));
            for (int ph = 0; ph < GF_PHASES; ++ph)
                for (int w = 0; w < 4; ++w)
                    for (int k = 0; k < L.items_f.cnt[ph][w]; ++k) {
                        const GfItem& it = L.items_f.item[ph][w][k];
                        fprintf(stderr, "[mmn] item ph=%d wave=%d: a_col=%d T=%d w_off=%d bias=%d out=%d N=%d col0=%d kh=%d\n", ph, w,
                                it.a_col, it.T, it.w_off, it.bias_off, it.out_col, it.n_valid, it.col0, it.kind_hk);
                    }
        }
        if (getenv("MMN_VERBOSE"))
            fprintf(stderr, "[mmn] generic=%d fast=%d dec_lds=%d lds fwd=%zu/%zu bwd=%zu/%zu dcols=%d n_bias=%zu\n", L.generic, fast,
                    h.dec_lds, pl->gen_lds_fwd[1], pl->gen_lds_fwd[2], pl->gen_lds_bwd[1], pl->gen_lds_bwd[2], L.dcols, L.bias_src.size());
    }
    auto fail = [&](hipError_t e) { g_last_hip = (int)e; delete pl; return MMN_ERR_HIP; };
    hipError_t e;
    if ((e = hipMemcpy(pl->dev, &h, sizeof(h), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(h.tasks, L.tasks.data(), sizeof(WTask) * L.tasks.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(h.items, L.items.data(), sizeof(WItem) * L.items.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(h.recs, L.recs.data(), sizeof(WRec) * L.recs.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(h.segs, L.segs.data(), sizeof(Seg) * L.segs.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(h.ptasks, L.ptasks.data(), sizeof(PackTask) * L.ptasks.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(ws + L.off_gfitems, &L.items_f, sizeof(GfItemTable), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(ws + L.off_gfitems_b, &L.items_b, sizeof(GfBwdTable), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if (!L.bias_src.empty() &&
        (e = hipMemcpy(const_cast<const float**>(h.bias_src), L.bias_src.data(), sizeof(const float*) * L.bias_src.size(),
                       hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemset(h.epoch, 0, sizeof(double) * mmn_epoch_doubles(m))) != hipSuccess) return fail(e);
    if ((e = hipMemset(h.exec_flags, 0, sizeof(int32_t) * (h.R + h.E + MMN_MAX_ENCODERS))) != hipSuccess) return fail(e);
    if ((e = hipMemset(h.pack, 0, sizeof(float) * (size_t)L.pack_floats)) != hipSuccess) return fail(e);
    // activations / gradient operands start finite: per-sample mode multiplies rows nobody wrote by zero rows
    if ((e = hipMemset(h.states, 0, L.off_pack - L.off_states)) != hipSuccess) return fail(e);
    if ((e = hipMemset(h.sin, 0, sizeof(float) * (size_t)h.E * h.maxB * h.S)) != hipSuccess) return fail(e);
    if (L.gact_floats && (e = hipMemset(h.gact, 0, sizeof(float) * (size_t)L.gact_floats)) != hipSuccess) return fail(e);
    if (L.gdpre_floats && (e = hipMemset(h.gdpre, 0, sizeof(float) * (size_t)L.gdpre_floats)) != hipSuccess) return fail(e);
    if (L.generic) {
        const void* gf[4] = {pl->gen_fast ? reinterpret_cast<const void*>(k_genf_fwd<1>) : reinterpret_cast<const void*>(k_gen_fwd<1>),
                             pl->gen_fast ? reinterpret_cast<const void*>(k_genf_fwd<2>) : reinterpret_cast<const void*>(k_gen_fwd<2>),
                             pl->gen_fast ? reinterpret_cast<const void*>(k_genf_bwd<1>) : reinterpret_cast<const void*>(k_gen_bwd<1>),
                             pl->gen_fast ? reinterpret_cast<const void*>(k_genf_bwd<2>) : reinterpret_cast<const void*>(k_gen_bwd<2>)};
        for (int k = 0; k < 4; ++k) {
            const size_t need = (k < 2 ? pl->gen_lds_fwd : pl->gen_lds_bwd)[1 + (k & 1)];
            if (need <= 160 * 1024 &&
                (e = hipFuncSetAttribute(gf[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)need)) != hipSuccess)
                return fail(e);
        }
    }
    const void* fns[4] = {reinterpret_cast<const void*>(k_chain_fwd<1>), reinterpret_cast<const void*>(k_chain_fwd<2>),
                          reinterpret_cast<const void*>(k_chain_bwd<1>), reinterpret_cast<const void*>(k_chain_bwd<2>)};
    for (int k = 0; k < 4; ++k) {
        const size_t need = pl->lds_bytes[1 + (k & 1)];
        if (need <= 160 * 1024 &&
            (e = hipFuncSetAttribute(fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)need)) != hipSuccess)
            return fail(e);
    }
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ps_layout), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(sizeof(int) * (PS_MAX_ROWS / 64) * PS_MAXG))) != hipSuccess) return fail(e);
    if (pl->fb8_ok &&
        ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fb8<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)pl->fb8_lds_bytes)) != hipSuccess ||
         (e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fb8<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)pl->fb8_lds_bytes)) != hipSuccess)) return fail(e);
    if (pl->f8_ok) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd8), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)pl->f8_lds_fwd)) != hipSuccess) return fail(e);
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd8), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)pl->f8_lds_bwd)) != hipSuccess) return fail(e);
    }
    if (pl->par_ok) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fwd_par), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)pl->par_lds_fwd)) != hipSuccess) return fail(e);
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_bwd_par), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)pl->par_lds_bwd)) != hipSuccess) return fail(e);
    }
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(sizeof(float) * WGRAD_LDS_FLOATS))) != hipSuccess) return fail(e);
    *out = pl;
    return MMN_OK;
}

void mmn_plan_destroy(mmn_plan* p) { delete p; }

int32_t* mmn_nan_flags(mmn_plan* p) { return p ? p->host.nan_flags : nullptr; }

static int check_batch(const mmn_plan* p, const mmn_batch* b) {
    if (!p || !b) return MMN_ERR_ARG;
    // (per-sample mode pads every sequence group to whole tiles: there batch_global, the true sample
    //  count, may be smaller than the padded batch)
    if (b->batch < 1 || b->batch > p->max_batch || b->batch_global < 1 || !b->y) return MMN_ERR_ARG;
    if (!b->tile_seq && b->batch_global < b->batch) return MMN_ERR_ARG;
    if (b->n_seq < 0 || b->n_seq > p->m.n_encoders) return MMN_ERR_SEQUENCE;
    unsigned seen_e = 0, seen_k = 0;
    for (int t = 0; t < b->n_seq; ++t) {
        const int e = b->seq_enc[t], k = b->seq_data[t];
        if (e < 0 || e >= p->m.n_encoders || k < 0 || k >= MMN_MAX_ENCODERS) return MMN_ERR_SEQUENCE;
        if ((seen_e >> e) & 1u) return MMN_ERR_SEQUENCE;          // repeated encoder id
        if ((seen_k >> k) & 1u) return MMN_ERR_SEQUENCE;
        seen_e |= 1u << e; seen_k |= 1u << k;
        if (!b->x[k] || b->ldx[k] < p->m.enc[e].n_features) return MMN_ERR_ARG;
    }
    for (int e = 0; e < MMN_MAX_ENCODERS; ++e)             // dropout multipliers exist for MIMIC encoders only
        if (b->drop_mask[e] && (e >= p->m.n_encoders || p->m.enc[e].kind != MMN_ENC_MIMIC)) return MMN_ERR_ARG;
    if (b->tile_seq || b->tile_rows) {                     // per-sample mode
        if (!b->tile_seq || !b->tile_rows || b->nan_flags) return MMN_ERR_ARG;
        if ((b->batch & 15) != 0 || b->n_seq != p->m.n_encoders) return MMN_ERR_ARG;
        for (int t = 0; t < b->n_seq; ++t)
            if (b->seq_data[t] != t || b->seq_enc[t] != t) return MMN_ERR_SEQUENCE;   // slot k = encoder k
        if (!p->fb8_ok) return MMN_ERR_UNSUPPORTED;        // only the fused kernel implements it
    }
    return MMN_OK;
}

// the 8-wave tier additionally needs 16-byte aligned feature rows in this batch
static bool use_fast8(const mmn_plan* p, const mmn_batch* b) {
    if (!p->f8_ok) return false;
    for (int t = 0; t < b->n_seq; ++t) {
        const int k = b->seq_data[t];
        if ((b->ldx[k] & 3) != 0 || (reinterpret_cast<uintptr_t>(b->x[k]) & 15) != 0) return false;
    }
    return true;
}

// the fused kernel reads x rows of any alignment
static bool use_fb8(const mmn_plan* p, const mmn_batch* b) { return p->fb8_ok && b->n_seq >= 1; }

static int rt_for(const mmn_plan* p, const mmn_batch* b) {
    if (p->par_ok) return 1;                 // the parallel-phase kernels use 16-row tiles
    if (p->generic && p->dec_lds_rt == 1) return 1;
    int rt = choose_rt(p, b->batch);
    if (p->lds_bytes[rt] > 160 * 1024) rt = 1;
    return rt;
}

const char* mmn_chain_kernel_name(mmn_plan* p, const mmn_batch* b, int backward) {
    if (!p || !b) return "";
    if (backward == 2) return use_fb8(p, b) ? "k_fb8" : "";
    if (p->generic && p->gen_fast) return backward ? "k_genf_bwd" : "k_genf_fwd";
    if (p->generic) return backward ? "k_gen_bwd" : "k_gen_fwd";
    if (use_fast8(p, b)) return backward ? "k_bwd8" : "k_fwd8";
    if (p->par_ok) return backward ? "k_chain_bwd_par" : "k_chain_fwd_par";
    return backward ? "k_chain_bwd" : "k_chain_fwd";
}

int mmn_prepare(mmn_plan* p, const mmn_batch* b, int want_grads, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    const bool scan = b->nan_flags != nullptr && b->n_seq > 0;
    int bps = 0, scan_blocks = 0;
    if (scan) {
        bps = 256 / b->n_seq;
        if (bps < 1) bps = 1;
        scan_blocks = b->n_seq * bps;
    }
    // the forward operands are needed by eval steps too, so the repack always runs
    (void)want_grads;
    const int tblocks = (int)((p->host.n_pack_elems + p->host.n_bias + NT - 1) / NT);
    if (scan_blocks + tblocks == 0) return MMN_OK;
    mmn_batch bb = *b;
    hipLaunchKernelGGL(k_prepare, dim3(scan_blocks + tblocks), dim3(NT), 0, static_cast<hipStream_t>(stream), p->dev, bb,
                       scan_blocks, bps > 0 ? bps : 1);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

int mmn_nan_scan(mmn_plan* p, const mmn_batch* b, void* stream) { return mmn_prepare(p, b, 0, stream); }

int mmn_chain_fwd(mmn_plan* p, const mmn_batch* b, float err_penalty, float sc_pen_x001, int want_grads,
                  void* stream) {
    (void)sc_pen_x001;
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    if (b->tile_seq) return MMN_ERR_UNSUPPORTED;           // per-sample mode exists in the fused kernel only
    const int rt = rt_for(p, b);
    const int tiles = (b->batch + 16 * rt - 1) / (16 * rt);
    const float cL = err_penalty / ((float)p->m.n_decoders * (float)(p->m.n_encoders + 1) * (float)b->batch_global);
    mmn_batch bb = *b;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->generic && p->gen_fast) {
        if (rt == 1) hipLaunchKernelGGL(k_genf_fwd<1>, dim3(tiles), dim3(NT), p->gen_lds_fwd[1], st, p->ga, bb, cL, want_grads);
        else hipLaunchKernelGGL(k_genf_fwd<2>, dim3(tiles), dim3(NT), p->gen_lds_fwd[2], st, p->ga, bb, cL, want_grads);
    } else if (p->generic) {
        if (rt == 1) hipLaunchKernelGGL(k_gen_fwd<1>, dim3(tiles), dim3(NT), p->gen_lds_fwd[1], st, p->dev, bb, cL, want_grads);
        else hipLaunchKernelGGL(k_gen_fwd<2>, dim3(tiles), dim3(NT), p->gen_lds_fwd[2], st, p->dev, bb, cL, want_grads);
    } else if (use_fast8(p, b)) hipLaunchKernelGGL(k_fwd8, dim3(tiles), dim3(NT8), p->f8_lds_fwd, st, p->pa, bb, cL, want_grads);
    else if (p->par_ok) hipLaunchKernelGGL(k_chain_fwd_par, dim3(tiles), dim3(NT), p->par_lds_fwd, st, p->dev, bb, cL, want_grads);
    else if (rt == 1) hipLaunchKernelGGL(k_chain_fwd<1>, dim3(tiles), dim3(NT), p->lds_bytes[1], st, p->dev, bb, cL, want_grads);
    else hipLaunchKernelGGL(k_chain_fwd<2>, dim3(tiles), dim3(NT), p->lds_bytes[2], st, p->dev, bb, cL, want_grads);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

static float sc_coeff(const mmn_plan* p, const mmn_batch* b, float beta) {
    return beta * 2.0f / ((float)p->m.n_encoders * (float)b->batch_global * (float)p->m.state_size);
}

int mmn_chain_bwd(mmn_plan* p, const mmn_batch* b, float sc_pen_x001, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    if (b->tile_seq) return MMN_ERR_UNSUPPORTED;
    const int rt = rt_for(p, b);
    const int tiles = (b->batch + 16 * rt - 1) / (16 * rt);
    mmn_batch bb = *b;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float cS = sc_coeff(p, b, sc_pen_x001);
    if (p->generic && p->gen_fast) {
        if (rt == 1) hipLaunchKernelGGL(k_genf_bwd<1>, dim3(tiles), dim3(NT), p->gen_lds_bwd[1], st, p->ga, bb, cS);
        else hipLaunchKernelGGL(k_genf_bwd<2>, dim3(tiles), dim3(NT), p->gen_lds_bwd[2], st, p->ga, bb, cS);

    } else if (p->generic) {
        if (rt == 1) hipLaunchKernelGGL(k_gen_bwd<1>, dim3(tiles), dim3(NT), p->gen_lds_bwd[1], st, p->dev, bb, cS);
        else hipLaunchKernelGGL(k_gen_bwd<2>, dim3(tiles), dim3(NT), p->gen_lds_bwd[2], st, p->dev, bb, cS);
    } else if (use_fast8(p, b)) hipLaunchKernelGGL(k_bwd8, dim3(tiles), dim3(NT8), p->f8_lds_bwd, st, p->pa, bb, cS);
    else if (p->par_ok) hipLaunchKernelGGL(k_chain_bwd_par, dim3(tiles), dim3(NT), p->par_lds_bwd, st, p->dev, bb, cS);
    else if (rt == 1) hipLaunchKernelGGL(k_chain_bwd<1>, dim3(tiles), dim3(NT), p->lds_bytes[1], st, p->dev, bb, cS);
    else hipLaunchKernelGGL(k_chain_bwd<2>, dim3(tiles), dim3(NT), p->lds_bytes[2], st, p->dev, bb, cS);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

int mmn_chain_fwd_bwd(mmn_plan* p, const mmn_batch* b, float err_penalty, float sc_pen_x001, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    if (use_fb8(p, b)) {                                   // one launch for both chains
        const int tiles = (b->batch + 15) / 16;
        const float cL = err_penalty / ((float)p->m.n_decoders * (float)(p->m.n_encoders + 1) * (float)b->batch_global);
        mmn_batch bb = *b;
        if (b->tile_seq)
            hipLaunchKernelGGL(k_fb8<true>, dim3(tiles), dim3(NT8), p->fb8_lds_bytes, static_cast<hipStream_t>(stream), p->pa, bb,
                               cL, sc_coeff(p, b, sc_pen_x001));
        else
            hipLaunchKernelGGL(k_fb8<false>, dim3(tiles), dim3(NT8), p->fb8_lds_bytes, static_cast<hipStream_t>(stream), p->pa, bb,
                               cL, sc_coeff(p, b, sc_pen_x001));
        HIP_TRY(hipGetLastError());
        return MMN_OK;
    }
    if ((rc = mmn_chain_fwd(p, b, err_penalty, sc_pen_x001, 1, stream)) != MMN_OK) return rc;
    return mmn_chain_bwd(p, b, sc_pen_x001, stream);
}

int mmn_wgrad(mmn_plan* p, const mmn_batch* b, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    mmn_batch bb = *b;
    hipLaunchKernelGGL(k_wgrad, dim3(p->host.n_items), dim3(NT), sizeof(float) * WGRAD_LDS_FLOATS,
                       static_cast<hipStream_t>(stream), p->wg, bb);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

static AdamArgs adam_args(const mmn_adam* d) {
    AdamArgs a{};
    if (!d) return a;
    a.p = d->params; a.g = d->grads; a.m = d->exp_avg; a.v = d->exp_avg_sq;
    a.steps = d->steps; a.seg_start = d->seg_start; a.seg_skip = d->seg_skip;
    a.n = (int)d->n; a.n_seg = d->n_seg;
    a.lr = d->lr; a.b1 = d->beta1; a.b2 = d->beta2; a.eps = (float)d->eps; a.wd = (float)d->weight_decay;
    a.maximize = d->maximize;
    a.gate_segs = nullptr; a.exec_flags = nullptr;
    return a;
}

static int check_adam(const mmn_adam* d) {
    if (!d || !d->params || !d->grads || !d->exp_avg || !d->exp_avg_sq || !d->steps || !d->seg_start) return MMN_ERR_ARG;
    if (d->n < 1 || d->n > 0x7fffffff - 4 * ADAM_NT || d->n_seg < 1 || d->n_seg > ADAM_MAX_SEG) return MMN_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d->params) | reinterpret_cast<uintptr_t>(d->grads) |
         reinterpret_cast<uintptr_t>(d->exp_avg) | reinterpret_cast<uintptr_t>(d->exp_avg_sq)) & 15)
        return MMN_ERR_ARG;
    return MMN_OK;
}

// The fused reduce+Adam needs the optimizer's flat layout to BE the plan's gradient layout: same
// tensors, same order, gradients at the plan's addresses.  Checked once per (seg_start, grads) pair
// (one small synchronous device read: do the first fused step outside stream capture).
static int adam_fusable(mmn_plan* p, const mmn_adam* d) {
    int rc = check_adam(d);
    if (rc != MMN_OK) return rc;
    if (p->adam_ok_seg_start == d->seg_start && p->adam_ok_grads == d->grads) return MMN_OK;
    if (d->n != p->host.n_grad_elems || d->n_seg != (int)p->segs.size() || p->segs.size() > (size_t)MAXSEG) return MMN_ERR_UNSUPPORTED;
    std::vector<int32_t> st(d->n_seg + 1);
    HIP_TRY(hipMemcpy(st.data(), d->seg_start, sizeof(int32_t) * st.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < d->n_seg; ++i) {
        if ((int64_t)st[i] != p->segs[i].start || p->segs[i].dst != d->grads + st[i]) return MMN_ERR_UNSUPPORTED;
    }
    if ((int64_t)st[d->n_seg] != d->n) return MMN_ERR_UNSUPPORTED;
    p->adam_ok_seg_start = d->seg_start; p->adam_ok_grads = d->grads;
    return MMN_OK;
}

static int launch_reduce(mmn_plan* p, const mmn_batch* b, int want_grads, int accumulate, float alpha, float beta,
                         void* stream, const mmn_adam* adam = nullptr) {
    const int rt = rt_for(p, b);
    const int tiles = (b->batch + 16 * rt - 1) / (16 * rt);
    hipLaunchKernelGGL(k_reduce, dim3(p->grad_blocks + 1), dim3(NTR), 0, static_cast<hipStream_t>(stream), p->rd,
                       adam_args(adam), adam ? 1 : 0, b->batch, b->batch_global, tiles, p->grad_blocks, want_grads, accumulate,
                       alpha, beta, const_cast<int32_t*>(b->nan_flags), b->tile_rows, b->tile_seq);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

int mmn_reduce(mmn_plan* p, const mmn_batch* b, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    return launch_reduce(p, b, 1, 0, 0.f, 0.f, stream);
}

int mmn_epoch_accumulate(mmn_plan* p, float err_penalty, float sc_pen_x001, void* stream) {
    if (!p) return MMN_ERR_ARG;
    hipLaunchKernelGGL(k_epoch_accumulate, dim3(1), dim3(NT), 0, static_cast<hipStream_t>(stream), p->dev, err_penalty,
                       sc_pen_x001);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

int mmn_train_step(mmn_plan* p, const mmn_batch* b, float err_penalty, float sc_pen_x001, int accumulate_epoch,
                   void* stream) {
    int rc = mmn_prepare(p, b, 1, stream);
    if (rc != MMN_OK) return rc;
    if ((rc = mmn_chain_fwd_bwd(p, b, err_penalty, sc_pen_x001, stream)) != MMN_OK) return rc;
    if ((rc = mmn_wgrad(p, b, stream)) != MMN_OK) return rc;
    return launch_reduce(p, b, 1, accumulate_epoch, err_penalty, sc_pen_x001, stream);
}

int mmn_train_step_adam(mmn_plan* p, const mmn_batch* b, float err_penalty, float sc_pen_x001, int accumulate_epoch,
                        const mmn_adam* adam, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    if ((rc = adam_fusable(p, adam)) != MMN_OK) return rc;
    if ((rc = mmn_prepare(p, b, 1, stream)) != MMN_OK) return rc;
    if ((rc = mmn_chain_fwd_bwd(p, b, err_penalty, sc_pen_x001, stream)) != MMN_OK) return rc;
    if ((rc = mmn_wgrad(p, b, stream)) != MMN_OK) return rc;
    return launch_reduce(p, b, 1, accumulate_epoch, err_penalty, sc_pen_x001, stream, adam);
}

int mmn_reduce_adam(mmn_plan* p, const mmn_batch* b, const mmn_adam* adam, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    if ((rc = adam_fusable(p, adam)) != MMN_OK) return rc;
    return launch_reduce(p, b, 1, 0, 0.f, 0.f, stream, adam);
}

int mmn_regroup_rows(int batch, int n_encoders) {
    if (batch < 1 || n_encoders < 1 || n_encoders > 4) return 0;
    int patterns = 0;                                      // ordered subsets of the encoders = distinct executed sequences
    for (int k = 0; k <= n_encoders; ++k) {
        int perm = 1;
        for (int j = 0; j < k; ++j) perm *= (n_encoders - j);
        patterns += perm;
    }
    return (batch + 15) / 16 * 16 + 16 * (patterns < batch ? patterns : batch);
}

int mmn_regroup(mmn_plan* p, const mmn_batch* in, const int64_t* seq, mmn_batch* out, void* stream) {
    if (!p || !in || !out || !in->y || !out->y || !out->tile_rows || !out->tile_seq) return MMN_ERR_ARG;
    const int E = p->m.n_encoders, D = p->m.n_decoders, B = in->batch;
    if (!p->fb8_ok || E > 4) return MMN_ERR_UNSUPPORTED;
    if (B < 1 || B > PS_MAX_ROWS) return MMN_ERR_UNSUPPORTED;
    const int rows = mmn_regroup_rows(B, E);
    if (rows > p->max_batch) return MMN_ERR_ARG;
    mmn_batch bi = *in, bo = *out;
    for (int k = 0; k < E; ++k) {
        if (!in->x[k] || !out->x[k]) return MMN_ERR_ARG;
        // (the kernels take the widths through the otherwise unused seq_data slots of their by-value copies)
        bi.seq_data[k] = p->m.enc[seq ? 0 : k].n_features;   // slot k's width (all equal when an order is given)
        bo.seq_data[k] = p->m.enc[k].n_features;
        if (seq && p->m.enc[k].n_features != p->m.enc[0].n_features) return MMN_ERR_UNSUPPORTED;
        if (in->ldx[k] < bi.seq_data[k] || out->ldx[k] < bo.seq_data[k]) return MMN_ERR_ARG;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t* codes = p->ps_scratch; int32_t* pmask = codes + p->max_batch; int32_t* src_of = pmask + p->max_batch;
    hipLaunchKernelGGL(k_ps_code, dim3((B + NT / 64 - 1) / (NT / 64)), dim3(NT), 0, st, bi, seq, E, codes, pmask);
    hipLaunchKernelGGL(k_ps_layout, dim3(1), dim3(1024), sizeof(int) * (size_t)((B + 63) / 64) * PS_MAXG, st, codes, B, rows, src_of,
                       const_cast<int32_t*>(out->tile_rows), const_cast<int32_t*>(out->tile_seq));
    hipLaunchKernelGGL(k_ps_gather, dim3((rows + NT / 64 - 1) / (NT / 64)), dim3(NT), 0, st, bi, seq, E, D, src_of, pmask, bo, rows);
    HIP_TRY(hipGetLastError());
    out->batch = rows;
    out->batch_global = in->batch_global;
    out->n_seq = E;
    out->nan_flags = nullptr;
    for (int k = 0; k < E; ++k) { out->seq_data[k] = k; out->seq_enc[k] = k; }
    return MMN_OK;
}

int mmn_eval_step(mmn_plan* p, const mmn_batch* b, int accumulate_epoch, void* stream) {
    int rc = mmn_prepare(p, b, 0, stream);
    if (rc != MMN_OK) return rc;
    if (b->tile_seq) {                                     // per-sample mode: the fused kernel, forward half only
        const int tiles = b->batch / 16;
        mmn_batch bb = *b;
        hipLaunchKernelGGL(k_fb8<true>, dim3(tiles), dim3(NT8), p->fb8_lds_bytes, static_cast<hipStream_t>(stream), p->pa, bb,
                           -1.0f, 0.0f);
        HIP_TRY(hipGetLastError());
        return launch_reduce(p, b, 0, accumulate_epoch, 1.0f, 0.0f, stream);
    }
    if ((rc = mmn_chain_fwd(p, b, 1.0f, 0.0f, 0, stream)) != MMN_OK) return rc;
    return launch_reduce(p, b, 0, accumulate_epoch, 1.0f, 0.0f, stream);
}

int mmn_adam_blocks(int64_t n) {
    if (n < 1 || n > 0x7fffffff - 4 * ADAM_NT) return 0;
    return (int)((n + 4 * ADAM_NT - 1) / (4 * ADAM_NT));
}

int mmn_adam_step_accumulate(mmn_plan* p, const mmn_adam* d, float err_penalty, float sc_pen_x001, void* stream) {
    if (!p) return MMN_ERR_ARG;
    const int rc = check_adam(d);
    if (rc != MMN_OK) return rc;
    const int blocks = mmn_adam_blocks(d->n);
    AdamArgs aa = adam_args(d);
    if (adam_fusable(p, d) == MMN_OK) {                    // the optimizer's tensors ARE the plan's: honour skipped encoders
        aa.gate_segs = p->rd.segs; aa.exec_flags = p->rd.exec_flags;
    }
    hipLaunchKernelGGL(k_adam_accumulate, dim3(blocks + 1), dim3(ADAM_NT), 0, static_cast<hipStream_t>(stream), aa,
                       blocks, p->dev, err_penalty, sc_pen_x001);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

int mmn_adam_step(const mmn_adam* d, void* stream) {
    const int rc = check_adam(d);
    if (rc != MMN_OK) return rc;
    hipLaunchKernelGGL(k_adam, dim3(mmn_adam_blocks(d->n)), dim3(ADAM_NT), 0, static_cast<hipStream_t>(stream), adam_args(d));
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

int mmn_epoch_reset(mmn_plan* p, void* stream) {
    if (!p) return MMN_ERR_ARG;
    HIP_TRY(hipMemsetAsync(p->host.epoch, 0, sizeof(double) * mmn_epoch_doubles(&p->m), static_cast<hipStream_t>(stream)));
    return MMN_OK;
}

int mmn_epoch_read(mmn_plan* p, double* out_host, void* stream) {
    if (!p || !out_host) return MMN_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(out_host, p->host.epoch, sizeof(double) * mmn_epoch_doubles(&p->m), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MMN_OK;
}

const float* mmn_debug_buffer(mmn_plan* p, int kind, int index) {
    if (!p) return nullptr;
    const DevPlan& h = p->host;
    switch (kind) {
        case 0: return (index >= 1 && index <= h.E) ? h.states + (int64_t)(index - 1) * h.maxB * h.S : nullptr;
        case 1: return (index >= 0 && index < h.R) ? h.dz + (int64_t)index * h.maxB * 2 * h.D : nullptr;
        case 2: return (index >= 0 && index <= h.E) ? h.dS + (int64_t)index * h.maxB * h.S : nullptr;
        case 3: return reinterpret_cast<const float*>(h.stamps);
        case 4: return reinterpret_cast<const float*>(h.exec_flags);
        case 5: return reinterpret_cast<const float*>(p->ps_scratch + 2 * (size_t)p->max_batch);   // mmn_regroup: source row of every position
        default: return nullptr;
    }
}

}  // extern "C"
