// mmn_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the MultiModN sequential-fusion training
// step and the C ABI declared in include/mmn_hip.h.  Written for wave64 + fp32 MFMA
// (v_mfma_f32_16x16x4_f32: bit-exact fp32 FMA chains, so fp32 parity with the reference's ATen
// path holds to rounding-order effects only).  No CUDA names, no dual paths.
//
// Launch structure of one training step (reference: multimodn/multimodn.py:137-203):
//   k_nan_scan         any(isnan(x_k)) per data slot                              (:168)
//   k_chain_fwd        row-tile parallel: init broadcast, every encoder, state-change partials,
//                      all D decoders on all E+1 states, CE-over-sigmoid, argmax, confusion counts
//   k_chain_bwd        row-tile parallel reverse chain: grads wrt states / pre-activations
//   k_wgrad            grouped split-K "A^T B" GEMM: weight, bias and init-state grads as slabs
//   k_reduce           fixed-order slab reduction -> grads; tile partials -> stats block
//   k_epoch_accumulate loss combination + epoch accumulators                      (:194-212)
//
// Data layout in HBM (all fp32 row-major, B = batch rows):
//   states[e][B][S]     output state of encoder e          hid[e][l][B][H_l]  hidden activations
//   dz[r][B][2D]        d loss / d decoder logits, row r   dS[e][B][S]        d loss / d state_e (+dS0)
//   dpre[e][l][B][H_l]  d loss / d hidden pre-activation   slabs              split-K partial grads
//
// Tiling: one workgroup = 256 threads = 4 waves owns 32 batch rows; the state tile lives in LDS
// for the whole chain.  Every Linear is "tile[32 x K] x W[N x K]^T": W is staged through LDS in
// [<=128 x 64] images (zero padded, row stride 68 floats = 4 mod 64 so that the ds_read_b64
// fragment reads are bank-conflict free), wave w owns output column tiles {w, w+4} x both 16-row
// tiles.  K is walked 8 at a time: one 8-byte fragment read feeds two MFMAs (the contraction
// index is permuted identically for A and B, which is legal because the sum is order-free per
// MFMA pair and both operands use the same permutation).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

#include "mmn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TB = 32;       // batch rows per workgroup tile
constexpr int NT = 256;      // threads per workgroup
constexpr int LDW = 68;      // row stride (floats) of every staged [.. x 64] LDS image
constexpr int LDZ = 20;      // row stride of the dz tile
constexpr int WG_TILE = 64;  // wgrad output tile edge

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// smallest row stride = 4 (mod 64) that holds round_up(k, 8) floats
__host__ __device__ inline int pick_ld(int k) {
    const int k8 = round_up(k, 8);
    int ld = 68;
    while (ld < k8) ld += 64;
    return ld;
}

// ------------------------------------------------------------------------------------------------
// device-side plan (lives at the start of the workspace)
// ------------------------------------------------------------------------------------------------
enum { A_DPRE = 0, A_DS = 1, A_DZ = 2 };
enum { IN_NONE = 0, IN_X = 1, IN_HID = 2, IN_STATE_ROW = 3, IN_PREV_STATE = 4 };

struct WTask {
    int32_t a_kind, a_enc, a_idx;      // A_DPRE: (enc, layer) ; A_DS: idx ; A_DZ: row
    int32_t M;
    int32_t in0_kind, in0_enc, in0_idx, k0;
    int32_t in1_kind, k1;
    int32_t bias, ntot;
    int32_t gate;                      // exec_flags index that must be set, else slab tile = 0
    int32_t pad;
    int64_t slab_base, pstride;
};

struct WItem { int32_t task, m0, n0, ks; };

struct Seg {                            // one gradient tensor
    float* dst;
    int64_t start;                      // first flat element index
    int64_t slab_base, pstride;
    int32_t count, n_partials, kdiv, ntot, coff, row_off;
};

struct DevPlan {
    mmn_model m;
    int32_t S, E, D, R, S8, ldS, ldAct, maxB, max_tiles, KS;
    int64_t hid_off[MMN_MAX_ENCODERS][MMN_MAX_LAYERS];   // float offsets into hid / dpre
    float* states; float* hid; float* dpre; float* dz; float* dS;
    float* lossp; float* scp; int32_t* cntp;
    int32_t* exec_flags;      // [R]   1 if state row r was produced this step
    int32_t* prev_row;        // [E]   state row that fed encoder e this step
    int32_t* nan_flags;       // [MMN_MAX_ENCODERS] scratch for mmn_nan_scan users
    float* slabs; float* stats; double* epoch;
    WTask* tasks; WItem* items; Seg* segs;
    int32_t n_tasks, n_items, n_segs, pad0;
    int64_t n_grad_elems;
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

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
    return b.nan_flags == nullptr || b.nan_flags[slot] == 0;
}

// Cooperative copy of a [nr_valid x nc_valid] global tile (row stride ld_src) into an LDS image
// [nr_pad x nc_pad] (row stride ld_dst), zero filling the padding.  nc_pad % 4 == 0.
__device__ __forceinline__ void stage_tile(float* dst, int ld_dst, const float* __restrict__ src,
                                           int64_t ld_src, int nr_valid, int nr_pad, int nc_valid,
                                           int nc_pad) {
    const int c4n = nc_pad >> 2;
    const bool vec = ((ld_src & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    const int total = nr_pad * c4n;
    for (int idx = threadIdx.x; idx < total; idx += NT) {
        const int r = idx / c4n;
        const int c = (idx - r * c4n) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nr_valid && c < nc_valid) {
            const float* p = src + (int64_t)r * ld_src + c;
            if (vec && c + 3 < nc_valid) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                v.x = p[0];
                if (c + 1 < nc_valid) v.y = p[1];
                if (c + 2 < nc_valid) v.z = p[2];
                if (c + 3 < nc_valid) v.w = p[3];
            }
        }
        *reinterpret_cast<float4*>(dst + r * ld_dst + c) = v;
    }
}

// acc[ci][rt] += A[32 x klen8] * Wimg[n][k]^T.  A row stride lda (even), Wimg row stride ldw.
// Wave w owns image-row tiles {w, w+4} (16 rows each) that lie below n_pad16.
__device__ __forceinline__ void mma_nt(f32x4 (&acc)[2][2], const float* A, int lda, int kbase,
                                       int klen8, const float* Wimg, int ldw, int n_pad16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    if (wave * 16 >= n_pad16) return;
    const bool t1 = (wave + 4) * 16 < n_pad16;
    const float* a0p = A + i * lda + kbase + 2 * q;
    const float* a1p = a0p + 16 * lda;
    const float* b0p = Wimg + (wave * 16 + i) * ldw + 2 * q;
    const float* b1p = b0p + 64 * ldw;
    for (int k = 0; k < klen8; k += 8) {
        const float2 a0 = *reinterpret_cast<const float2*>(a0p + k);
        const float2 a1 = *reinterpret_cast<const float2*>(a1p + k);
        const float2 b0 = *reinterpret_cast<const float2*>(b0p + k);
        acc[0][0] = mfma4(a0.x, b0.x, acc[0][0]);
        acc[0][1] = mfma4(a1.x, b0.x, acc[0][1]);
        acc[0][0] = mfma4(a0.y, b0.y, acc[0][0]);
        acc[0][1] = mfma4(a1.y, b0.y, acc[0][1]);
        if (t1) {
            const float2 b1 = *reinterpret_cast<const float2*>(b1p + k);
            acc[1][0] = mfma4(a0.x, b1.x, acc[1][0]);
            acc[1][1] = mfma4(a1.x, b1.x, acc[1][1]);
            acc[1][0] = mfma4(a0.y, b1.y, acc[1][0]);
            acc[1][1] = mfma4(a1.y, b1.y, acc[1][1]);
        }
    }
}

// acc[rt] += A[32 x nlen8] * Wimg[n][c0 + 16*wave + j]  (contraction over image ROWS n).
__device__ __forceinline__ void mma_nn(f32x4 (&acc)[2], const float* A, int lda, int nbase,
                                       int nlen8, const float* Wimg, int ldw, int c_pad16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    if (wave * 16 >= c_pad16) return;
    const float* a0p = A + i * lda + nbase + 2 * q;
    const float* a1p = a0p + 16 * lda;
    const float* bp = Wimg + (2 * q) * ldw + wave * 16 + i;
    for (int n = 0; n < nlen8; n += 8) {
        const float2 a0 = *reinterpret_cast<const float2*>(a0p + n);
        const float2 a1 = *reinterpret_cast<const float2*>(a1p + n);
        const float bx = bp[n * ldw];
        const float by = bp[(n + 1) * ldw];
        acc[0] = mfma4(a0.x, bx, acc[0]);
        acc[1] = mfma4(a1.x, bx, acc[1]);
        acc[0] = mfma4(a0.y, by, acc[0]);
        acc[1] = mfma4(a1.y, by, acc[1]);
    }
}

struct ASeg {
    const float* g;     // global source already offset to the tile's first row (nullptr: LDS)
    int64_t ldg;
    const float* lds;
    int ldl;
    int K;
    int wcol;           // first W column this segment multiplies
};

// out[32 x N] = sum_seg A_seg[32 x K_seg] * W[:, wcol_seg : wcol_seg + K_seg]^T, epilogue per element.
template <class Epi>
__device__ __forceinline__ void linear_nt(const float* __restrict__ W, int ldw_g, int N,
                                          const ASeg* seg, int nseg, int nrows, float* sX, float* sW,
                                          Epi&& epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    for (int nc = 0; nc < N; nc += 128) {
        const int nval = min(128, N - nc), npad = round_up(nval, 16);
        f32x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nseg; ++s) {
            for (int kc = 0; kc < seg[s].K; kc += 64) {
                const int kw = min(64, seg[s].K - kc), kw8 = round_up(kw, 8);
                stage_tile(sW, LDW, W + (int64_t)nc * ldw_g + seg[s].wcol + kc, ldw_g, nval, npad, kw, kw8);
                const float* A;
                int lda, kbase;
                if (seg[s].g) {
                    stage_tile(sX, LDW, seg[s].g + kc, seg[s].ldg, nrows, TB, kw, kw8);
                    A = sX; lda = LDW; kbase = 0;
                } else {
                    A = seg[s].lds; lda = seg[s].ldl; kbase = kc;
                }
                __syncthreads();
                mma_nt(acc, A, lda, kbase, kw8, sW, LDW, npad);
                __syncthreads();
            }
        }
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            if ((wave + 4 * ci) * 16 < npad) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        epi(rt * 16 + q * 4 + r, nc + (wave + 4 * ci) * 16 + i, acc[ci][rt][r]);
            }
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ------------------------------------------------------------------------------------------------
// k_nan_scan: flags[slot] = 1 if any element of data slot `slot` is NaN (flags pre-zeroed)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_nan_scan(mmn_batch b, const mmn_model* __restrict__ m_dev,
                                                 int32_t* flags, int blocks_per_slot) {
    const int t = blockIdx.x / blocks_per_slot;          // sequence position
    const int part = blockIdx.x - t * blocks_per_slot;
    const int slot = b.seq_data[t];
    const int F = m_dev->enc[b.seq_enc[t]].n_features;
    const float* x = b.x[slot];
    const int64_t ld = b.ldx[slot];
    const int64_t total = (int64_t)b.batch * F;
    bool bad = false;
    for (int64_t idx = (int64_t)part * NT + threadIdx.x; idx < total; idx += (int64_t)blocks_per_slot * NT) {
        const int64_t r = idx / F;
        const int c = (int)(idx - r * F);
        const float v = x[r * ld + c];
        bad |= (v != v);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) flags[slot] = 1;
}

// ------------------------------------------------------------------------------------------------
// decoder evaluation of one state tile (used by k_chain_fwd)
// ------------------------------------------------------------------------------------------------
struct DecodeCtx {
    const DevPlan* p;
    const mmn_batch* b;
    float* sDec; float* sZ;
    int row0, nrows, tile;
    float cL;
    int want_grads;
};

__device__ __forceinline__ void decode_state(const DecodeCtx& c, const float* sS, int grid_row) {
    const DevPlan& p = *c.p;
    const int S8 = p.S8, ldS = p.ldS, D = p.D, R = p.R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    // z[32 x 16] = sS[32 x S] * sDec[16 x S]^T, contraction split over the four waves
    f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
    const int kq = (((S8 >> 3) + 3) >> 2) << 3;
    const int kb = wave * kq, ke = min(S8, kb + kq);
    const float* a0p = sS + i * ldS + 2 * q;
    const float* a1p = a0p + 16 * ldS;
    const float* bp = c.sDec + i * ldS + 2 * q;
    for (int k = kb; k < ke; k += 8) {
        const float2 a0 = *reinterpret_cast<const float2*>(a0p + k);
        const float2 a1 = *reinterpret_cast<const float2*>(a1p + k);
        const float2 bb = *reinterpret_cast<const float2*>(bp + k);
        z0 = mfma4(a0.x, bb.x, z0);
        z1 = mfma4(a1.x, bb.x, z1);
        z0 = mfma4(a0.y, bb.y, z0);
        z1 = mfma4(a1.y, bb.y, z1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c.sZ[(wave * 32 + q * 4 + r) * 16 + i] = z0[r];
        c.sZ[(wave * 32 + 16 + q * 4 + r) * 16 + i] = z1[r];
    }
    __syncthreads();
    const int t = threadIdx.x;
    const int row = t & 31, d = t >> 5;
    float lossv = 0.f;
    int correct = 0, tp = 0, tn = 0, fp = 0, fn = 0;
    if (d < D && row < c.nrows) {
        const float* bd = p.m.dec[d].b;
        float za = bd[0], zb = bd[1];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            za += c.sZ[(w * 32 + row) * 16 + 2 * d];
            zb += c.sZ[(w * 32 + row) * 16 + 2 * d + 1];
        }
        const int64_t grow = (int64_t)c.row0 + row;
        const int y = (int)c.b->y[grow * D + d];
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
            float2 dzv;
            dzv.x = c.cL * g0 * o0 * (1.0f - o0);
            dzv.y = c.cL * g1 * o1 * (1.0f - o1);
            float* dz = p.dz + ((int64_t)grid_row * p.maxB + grow) * (2 * D) + 2 * d;
            *reinterpret_cast<float2*>(dz) = dzv;
        }
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) lossv += __shfl_xor(lossv, off);
    const unsigned long long mc = __ballot(correct), mtp = __ballot(tp), mtn = __ballot(tn),
                             mfp = __ballot(fp), mfn = __ballot(fn);
    if ((t & 31) == 0 && d < D) {
        const int sh = (lane >= 32) ? 32 : 0;
        const int64_t cell = (int64_t)c.tile * (R * D) + grid_row * D + d;
        p.lossp[cell] = lossv;
        int32_t* cp = p.cntp + cell * 5;
        cp[0] = __popc((unsigned)(mc >> sh));
        cp[1] = __popc((unsigned)(mtp >> sh));
        cp[2] = __popc((unsigned)(mtn >> sh));
        cp[3] = __popc((unsigned)(mfp >> sh));
        cp[4] = __popc((unsigned)(mfn >> sh));
    }
    __syncthreads();
}

// LDS carve of the two chain kernels (floats)
struct ChainLds {
    int sS0, sS1, sDiff, sDec, sAct0, sAct1, sX, sW, sZ, sRed, total;
};
__host__ __device__ inline ChainLds chain_lds(int ldS, int ldAct) {
    ChainLds L;
    int o = 0;
    L.sS0 = o; o += TB * ldS;
    L.sS1 = o; o += TB * ldS;
    L.sDiff = o; o += TB * ldS;
    L.sDec = o; o += 16 * ldS;
    L.sAct0 = o; o += TB * ldAct;
    L.sAct1 = o; o += TB * ldAct;
    L.sX = o; o += TB * LDW;
    L.sW = o; o += 128 * LDW;
    L.sZ = o; o += 4 * 32 * 16;
    L.sRed = o; o += 64;
    L.total = o;
    return L;
}

__device__ __forceinline__ void fill_decoder_image(const DevPlan& p, float* sDec) {
    const int S = p.S, ldS = p.ldS, D = p.D;
    for (int idx = threadIdx.x; idx < 16 * ldS; idx += NT) {
        const int n = idx / ldS, k = idx - n * ldS;
        float v = 0.f;
        if (n < 2 * D && k < S) v = p.m.dec[n >> 1].w[(n & 1) * S + k];
        sDec[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// k_chain_fwd
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_chain_fwd(const DevPlan* __restrict__ P, mmn_batch b, float cL,
                                                  int want_grads) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevPlan& p = *P;
    const int S = p.S, E = p.E, ldS = p.ldS, ldA = p.ldAct;
    const ChainLds L = chain_lds(ldS, ldA);
    float* sCur = smem + L.sS0;
    float* sNext = smem + L.sS1;
    float* sDec = smem + L.sDec;
    float* sAct[2] = {smem + L.sAct0, smem + L.sAct1};
    float* sX = smem + L.sX;
    float* sW = smem + L.sW;
    float* sRed = smem + L.sRed;
    const int tile = blockIdx.x;
    const int row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    fill_decoder_image(p, sDec);
    for (int idx = threadIdx.x; idx < TB * ldS; idx += NT) {
        const int k = idx % ldS;
        sCur[idx] = k < S ? p.m.init_state[k] : 0.f;       // state.py:29-32 (tile, never materialised)
        sNext[idx] = 0.f;
    }
    if (tile == 0 && threadIdx.x == 0) {                   // which state rows exist this step
        p.exec_flags[0] = 1;
        for (int e = 0; e < E; ++e) p.exec_flags[e + 1] = 0;
        int prev = 0;
        for (int t = 0; t < b.n_seq; ++t) {
            if (!slot_present(b, b.seq_data[t])) continue;
            const int e = b.seq_enc[t];
            p.exec_flags[e + 1] = 1;
            p.prev_row[e] = prev;
            prev = e + 1;
        }
    }
    __syncthreads();

    DecodeCtx dc{P, &b, sDec, smem + L.sZ, row0, nrows, tile, cL, want_grads};
    decode_state(dc, sCur, 0);

    for (int t = 0; t < b.n_seq; ++t) {
        const int slot = b.seq_data[t];
        if (!slot_present(b, slot)) continue;              // multimodn.py:168-169
        const int e = b.seq_enc[t];
        const mmn_encoder& enc = p.m.enc[e];
        const int nl = enc.n_layers;
        const float* xg = b.x[slot] + (int64_t)row0 * b.ldx[slot];
        // hidden layers: h = act(W h + b) on x only (mlp_encoder.py:75-76)
        for (int l = 0; l + 1 < nl; ++l) {
            const mmn_linear& lin = enc.layer[l];
            const int N = lin.out_dim, N8 = round_up(N, 8);
            ASeg seg[1];
            if (l == 0) seg[0] = ASeg{xg, b.ldx[slot], nullptr, 0, lin.in_dim, 0};
            else seg[0] = ASeg{nullptr, 0, sAct[(l - 1) & 1], ldA, lin.in_dim, 0};
            float* out = sAct[l & 1];
            float* hid_g = want_grads ? p.hid + p.hid_off[e][l] + (int64_t)row0 * N : nullptr;
            const float* bias = lin.b;
            const int akind = enc.activation;
            linear_nt(lin.w, lin.in_dim, N, seg, 1, nrows, sX, sW, [&](int row, int col, float v) {
                if (col < N8) {
                    float h = 0.f;
                    if (col < N) {
                        h = act_fwd(v + bias[col], akind);
                        if (hid_g && row < nrows) hid_g[(int64_t)row * N + col] = h;
                    }
                    out[row * ldA + col] = h;
                }
            });
            __syncthreads();
        }
        // state update: s' = W [h ; s] + b, no activation (mlp_encoder.py:78)
        {
            const mmn_linear& lin = enc.layer[nl - 1];
            const int HL = lin.in_dim - S;
            ASeg seg[2];
            if (nl == 1) seg[0] = ASeg{xg, b.ldx[slot], nullptr, 0, HL, 0};
            else seg[0] = ASeg{nullptr, 0, sAct[(nl - 2) & 1], ldA, HL, 0};
            seg[1] = ASeg{nullptr, 0, sCur, ldS, S, HL};
            float* st_g = want_grads ? p.states + ((int64_t)e * p.maxB + row0) * S : nullptr;
            const float* bias = lin.b;
            float scacc = 0.f;
            linear_nt(lin.w, lin.in_dim, S, seg, 2, nrows, sX, sW, [&](int row, int col, float v) {
                if (col < S) {
                    const float ns = v + bias[col];
                    const float dlt = ns - sCur[row * ldS + col];
                    if (row < nrows) {
                        scacc += dlt * dlt;                               // multimodn.py:174
                        if (st_g) st_g[(int64_t)row * S + col] = ns;
                    }
                    sNext[row * ldS + col] = ns;
                }
            });
            scacc = wave_sum(scacc);
            if (lane == 0) sRed[wave] = scacc;
            __syncthreads();
            if (threadIdx.x == 0) p.scp[(int64_t)tile * E + e] = sRed[0] + sRed[1] + sRed[2] + sRed[3];
            float* tmp = sCur; sCur = sNext; sNext = tmp;
        }
        decode_state(dc, sCur, e + 1);
    }
}

// ------------------------------------------------------------------------------------------------
// k_chain_bwd
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_dz_tile(const DevPlan& p, float* sDz, int grid_row, int row0, int nrows) {
    const int D2 = 2 * p.D;
    for (int idx = threadIdx.x; idx < TB * 16; idx += NT) {
        const int row = idx >> 4, n = idx & 15;
        float v = 0.f;
        if (row < nrows && n < D2) v = p.dz[((int64_t)grid_row * p.maxB + row0 + row) * D2 + n];
        sDz[row * LDZ + n] = v;
    }
}

// sG[row][col] += extra(row, col) + sum_n sDz[row][n] * sDec[n][col]
template <class Extra>
__device__ __forceinline__ void add_decoder_grad(const DevPlan& p, float* sG, const float* sDz,
                                                 const float* sDec, Extra&& extra) {
    const int S = p.S, ldS = p.ldS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int n8 = round_up(2 * p.D, 8);
    for (int cg = 0; cg < S; cg += 64) {
        const int cpad = min(64, round_up(S - cg, 16));
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        mma_nn(acc, sDz, LDZ, 0, n8, sDec + cg, ldS, cpad);
        if (wave * 16 < cpad) {
            const int col = cg + wave * 16 + i;
            if (col < S) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rt * 16 + q * 4 + r;
                        sG[row * ldS + col] += acc[rt][r] + extra(row, col);
                    }
            }
        }
    }
}

__global__ __launch_bounds__(NT) void k_chain_bwd(const DevPlan* __restrict__ P, mmn_batch b, float cS) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevPlan& p = *P;
    const int S = p.S, E = p.E, ldS = p.ldS, ldA = p.ldAct;
    const ChainLds L = chain_lds(ldS, ldA);
    float* sG = smem + L.sS0;
    float* sG2 = smem + L.sS1;
    float* sDiff = smem + L.sDiff;
    float* sDec = smem + L.sDec;
    float* sAct[2] = {smem + L.sAct0, smem + L.sAct1};
    float* sW = smem + L.sW;
    float* sDz = smem + L.sZ;
    const int tile = blockIdx.x;
    const int row0 = tile * TB;
    const int nrows = min(TB, b.batch - row0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;

    fill_decoder_image(p, sDec);
    for (int idx = threadIdx.x; idx < TB * ldS; idx += NT) { sG[idx] = 0.f; sG2[idx] = 0.f; sDiff[idx] = 0.f; }
    __syncthreads();

    for (int t = b.n_seq - 1; t >= 0; --t) {
        const int slot = b.seq_data[t];
        if (!slot_present(b, slot)) continue;
        const int e = b.seq_enc[t];
        int tp = t - 1;
        while (tp >= 0 && !slot_present(b, b.seq_data[tp])) --tp;
        const int prev_row = tp >= 0 ? b.seq_enc[tp] + 1 : 0;
        const mmn_encoder& enc = p.m.enc[e];
        const int nl = enc.n_layers, Lh = nl - 1;
        const mmn_linear& last = enc.layer[nl - 1];
        const int HL = last.in_dim - S;
        const int akind = enc.activation;

        // diff = s_out - s_in, dz tile of grid row e+1
        {
            const float* so = p.states + ((int64_t)e * p.maxB + row0) * S;
            const float* si = prev_row ? p.states + ((int64_t)(prev_row - 1) * p.maxB + row0) * S : nullptr;
            for (int idx = threadIdx.x; idx < TB * S; idx += NT) {
                const int row = idx / S, col = idx - row * S;
                float v = 0.f;
                if (row < nrows) {
                    const float a = so[(int64_t)row * S + col];
                    const float c = si ? si[(int64_t)row * S + col] : p.m.init_state[col];
                    v = a - c;
                }
                sDiff[row * ldS + col] = v;
            }
            load_dz_tile(p, sDz, e + 1, row0, nrows);
            for (int idx = threadIdx.x; idx < TB * ldA; idx += NT) { sAct[0][idx] = 0.f; sAct[1][idx] = 0.f; }
        }
        __syncthreads();
        // G_out = carry + decoder grad of row e+1 + cS * diff
        add_decoder_grad(p, sG, sDz, sDec, [&](int row, int col) { return cS * sDiff[row * ldS + col]; });
        __syncthreads();
        {
            float* dS = p.dS + ((int64_t)e * p.maxB + row0) * S;
            for (int idx = threadIdx.x; idx < nrows * S; idx += NT) {
                const int row = idx / S, col = idx - row * S;
                dS[(int64_t)row * S + col] = sG[row * ldS + col];
            }
        }
        // dcat = G_out * W_last : columns [0,HL) -> dh, [HL, HL+S) -> carry
        {
            const int ldw_g = last.in_dim;
            const int c_begin = Lh == 0 ? HL : 0;                  // no grad flows to x
            const float* hid_g = Lh ? p.hid + p.hid_off[e][Lh - 1] + (int64_t)row0 * HL : nullptr;
            float* dpre_g = Lh ? p.dpre + p.hid_off[e][Lh - 1] + (int64_t)row0 * HL : nullptr;
            float* dact = sAct[0];
            for (int kc = c_begin; kc < HL + S; kc += 64) {
                const int kw = min(64, HL + S - kc), kpad = round_up(kw, 16);
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                for (int nc = 0; nc < S; nc += 128) {
                    const int nlen = min(128, S - nc), nlen8 = round_up(nlen, 8);
                    stage_tile(sW, LDW, last.w + (int64_t)nc * ldw_g + kc, ldw_g, nlen, nlen8, kw, kpad);
                    __syncthreads();
                    mma_nn(acc, sG, ldS, nc, nlen8, sW, LDW, kpad);
                    __syncthreads();
                }
                if (wave * 16 < kpad) {
                    const int col = kc + wave * 16 + i;
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = rt * 16 + q * 4 + r;
                            const float v = acc[rt][r];
                            if (col < HL) {
                                float dp = 0.f;
                                if (row < nrows) {
                                    dp = v * act_grad_from_out(hid_g[(int64_t)row * HL + col], akind);
                                    dpre_g[(int64_t)row * HL + col] = dp;
                                }
                                dact[row * ldA + col] = dp;
                            } else if (col < HL + S) {
                                const int j = col - HL;
                                sG2[row * ldS + j] = v - cS * sDiff[row * ldS + j];
                            }
                        }
                }
            }
        }
        __syncthreads();
        // hidden layers, last to first: dpre_{l-1} = (dpre_l * W_l) .* act'(h_{l-1})
        for (int l = Lh - 1; l >= 1; --l) {
            const mmn_linear& lin = enc.layer[l];
            const int Hl = lin.out_dim, Hp = lin.in_dim;
            const float* cur = sAct[(Lh - 1 - l) & 1];
            float* nxt = sAct[(Lh - l) & 1];
            const float* hid_g = p.hid + p.hid_off[e][l - 1] + (int64_t)row0 * Hp;
            float* dpre_g = p.dpre + p.hid_off[e][l - 1] + (int64_t)row0 * Hp;
            for (int idx = threadIdx.x; idx < TB * ldA; idx += NT) nxt[idx] = 0.f;
            __syncthreads();
            for (int kc = 0; kc < Hp; kc += 64) {
                const int kw = min(64, Hp - kc), kpad = round_up(kw, 16);
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                for (int nc = 0; nc < Hl; nc += 128) {
                    const int nlen = min(128, Hl - nc), nlen8 = round_up(nlen, 8);
                    stage_tile(sW, LDW, lin.w + (int64_t)nc * Hp + kc, Hp, nlen, nlen8, kw, kpad);
                    __syncthreads();
                    mma_nn(acc, cur, ldA, nc, nlen8, sW, LDW, kpad);
                    __syncthreads();
                }
                if (wave * 16 < kpad) {
                    const int col = kc + wave * 16 + i;
                    if (col < Hp) {
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = rt * 16 + q * 4 + r;
                                float dp = 0.f;
                                if (row < nrows) {
                                    dp = acc[rt][r] * act_grad_from_out(hid_g[(int64_t)row * Hp + col], akind);
                                    dpre_g[(int64_t)row * Hp + col] = dp;
                                }
                                nxt[row * ldA + col] = dp;
                            }
                    }
                }
            }
            __syncthreads();
        }
        float* tmp = sG; sG = sG2; sG2 = tmp;      // carry becomes the incoming gradient
        __syncthreads();
    }
    // row 0: decoders on the init state; dS0 = d loss / d tiled init state
    load_dz_tile(p, sDz, 0, row0, nrows);
    __syncthreads();
    add_decoder_grad(p, sG, sDz, sDec, [&](int, int) { return 0.f; });
    __syncthreads();
    float* dS0 = p.dS + ((int64_t)E * p.maxB + row0) * S;
    for (int idx = threadIdx.x; idx < nrows * S; idx += NT) {
        const int row = idx / S, col = idx - row * S;
        dS0[(int64_t)row * S + col] = sG[row * ldS + col];
    }
}

// ------------------------------------------------------------------------------------------------
// k_wgrad: C[M x ntot] = A[rows x M]^T * [in0 | in1 | 1][rows x ntot] over a row range -> slab
// ------------------------------------------------------------------------------------------------
struct SrcRef { const float* p; int64_t ld; };

__device__ __forceinline__ SrcRef resolve_in(const DevPlan& p, const mmn_batch& b, int kind, int enc, int idx) {
    SrcRef s{nullptr, 0};
    switch (kind) {
        case IN_X: {
            int slot = 0;
            for (int t = 0; t < b.n_seq; ++t) if (b.seq_enc[t] == enc) slot = b.seq_data[t];
            s.p = b.x[slot]; s.ld = b.ldx[slot];
            break;
        }
        case IN_HID:
            s.p = p.hid + p.hid_off[enc][idx]; s.ld = p.m.enc[enc].layer[idx].out_dim;
            break;
        case IN_STATE_ROW:
            if (idx == 0) { s.p = p.m.init_state; s.ld = 0; }
            else { s.p = p.states + (int64_t)(idx - 1) * p.maxB * p.S; s.ld = p.S; }
            break;
        case IN_PREV_STATE: {
            const int r = p.prev_row[enc];
            if (r == 0) { s.p = p.m.init_state; s.ld = 0; }
            else { s.p = p.states + (int64_t)(r - 1) * p.maxB * p.S; s.ld = p.S; }
            break;
        }
        default: break;
    }
    return s;
}

__global__ __launch_bounds__(NT) void k_wgrad(const DevPlan* __restrict__ P, mmn_batch b, int rows_per_split) {
    __shared__ __attribute__((aligned(16))) float sA[TB * LDW];
    __shared__ __attribute__((aligned(16))) float sI[TB * LDW];
    const DevPlan& p = *P;
    const WItem it = p.items[blockIdx.x];
    const WTask& tk = p.tasks[it.task];
    const int M = tk.M, ntot = tk.ntot;
    const int m0 = it.m0, n0 = it.n0;
    const int mt = min(WG_TILE, M - m0), nt = min(WG_TILE, ntot - n0);
    float* slab = p.slabs + tk.slab_base + (int64_t)it.ks * tk.pstride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int rb = it.ks * rows_per_split, re = min(b.batch, rb + rows_per_split);

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (p.exec_flags[tk.gate] && rb < re) {
        const float* Ap; int64_t lda;
        if (tk.a_kind == A_DPRE) { Ap = p.dpre + p.hid_off[tk.a_enc][tk.a_idx]; lda = M; }
        else if (tk.a_kind == A_DS) { Ap = p.dS + (int64_t)tk.a_idx * p.maxB * p.S; lda = p.S; }
        else { Ap = p.dz + (int64_t)tk.a_idx * p.maxB * (2 * p.D); lda = 2 * p.D; }
        const SrcRef in0 = resolve_in(p, b, tk.in0_kind, tk.in0_enc, tk.in0_idx);
        const SrcRef in1 = resolve_in(p, b, tk.in1_kind, tk.in0_enc, 0);
        const int k0 = tk.k0, k01 = tk.k0 + tk.k1;
        for (int r = rb; r < re; r += TB) {
            const int nr = min(TB, re - r);
            stage_tile(sA, LDW, Ap + (int64_t)r * lda + m0, lda, nr, TB, mt, WG_TILE);
            for (int idx = threadIdx.x; idx < TB * WG_TILE; idx += NT) {
                const int rr = idx >> 6, c = idx & 63, n = n0 + c;
                float v = 0.f;
                if (rr < nr) {
                    if (n < k0) v = in0.p[(int64_t)(r + rr) * in0.ld + n];
                    else if (n < k01) v = in1.p[(int64_t)(r + rr) * in1.ld + (n - k0)];
                    else if (n == k01 && tk.bias) v = 1.0f;
                }
                sI[rr * LDW + c] = v;
            }
            __syncthreads();
            const float* ap = sA + (2 * q) * LDW + 32 * wm + i;
            const float* ip = sI + (2 * q) * LDW + 32 * wn + i;
#pragma unroll
            for (int k = 0; k < TB; k += 8) {
                const float ax0 = ap[k * LDW], ay0 = ap[(k + 1) * LDW];
                const float ax1 = ap[k * LDW + 16], ay1 = ap[(k + 1) * LDW + 16];
                const float bx0 = ip[k * LDW], by0 = ip[(k + 1) * LDW];
                const float bx1 = ip[k * LDW + 16], by1 = ip[(k + 1) * LDW + 16];
                acc[0][0] = mfma4(ax0, bx0, acc[0][0]);
                acc[0][1] = mfma4(ax0, bx1, acc[0][1]);
                acc[1][0] = mfma4(ax1, bx0, acc[1][0]);
                acc[1][1] = mfma4(ax1, bx1, acc[1][1]);
                acc[0][0] = mfma4(ay0, by0, acc[0][0]);
                acc[0][1] = mfma4(ay0, by1, acc[0][1]);
                acc[1][0] = mfma4(ay1, by0, acc[1][0]);
                acc[1][1] = mfma4(ay1, by1, acc[1][1]);
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wm + 16 * rt + 4 * q + r;
                const int n = n0 + 32 * wn + 16 * ct + i;
                if (m < M && n < ntot) slab[(int64_t)m * ntot + n] = acc[rt][ct][r];
            }
}

// ------------------------------------------------------------------------------------------------
// k_reduce: slabs -> gradient tensors (one thread per element); last block: tile partials -> stats
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_reduce(const DevPlan* __restrict__ P, int batch, int batch_global,
                                               int n_tiles, int grad_blocks, int want_grads) {
    const DevPlan& p = *P;
    if ((int)blockIdx.x < grad_blocks) {
        if (!want_grads) return;
        const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
        if (idx >= p.n_grad_elems) return;
        int lo = 0, hi = p.n_segs - 1;
        while (lo < hi) {                                   // last segment with start <= idx
            const int mid = (lo + hi + 1) >> 1;
            if (p.segs[mid].start <= idx) lo = mid; else hi = mid - 1;
        }
        const Seg& sg = p.segs[lo];
        const int local = (int)(idx - sg.start);
        const int m = local / sg.kdiv, n = local - m * sg.kdiv;
        const float* src = p.slabs + sg.slab_base + (int64_t)(sg.row_off + m) * sg.ntot + n + sg.coff;
        float sum = 0.f;
        for (int k = 0; k < sg.n_partials; ++k) sum += src[(int64_t)k * sg.pstride];
        sg.dst[local] = sum;
        return;
    }
    // stats block
    const int R = p.R, D = p.D, E = p.E, S = p.S;
    const int RD = R * D;
    float* st = p.stats;
    const float Bg = (float)batch_global;
    for (int cell = threadIdx.x; cell < RD; cell += NT) {
        const int r = cell / D;
        float ls = 0.f;
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
        if (p.exec_flags[r]) {
            for (int t = 0; t < n_tiles; ++t) {
                ls += p.lossp[(int64_t)t * RD + cell];
                const int32_t* cp = p.cntp + ((int64_t)t * RD + cell) * 5;
                c0 += cp[0]; c1 += cp[1]; c2 += cp[2]; c3 += cp[3]; c4 += cp[4];
            }
        }
        st[cell] = ls / Bg;
        st[RD + E + 0 * RD + cell] = (float)c0;
        st[RD + E + 1 * RD + cell] = (float)c1;
        st[RD + E + 2 * RD + cell] = (float)c2;
        st[RD + E + 3 * RD + cell] = (float)c3;
        st[RD + E + 4 * RD + cell] = (float)c4;
    }
    for (int e = threadIdx.x; e < E; e += NT) {
        float s = 0.f;
        if (p.exec_flags[e + 1])
            for (int t = 0; t < n_tiles; ++t) s += p.scp[(int64_t)t * E + e];
        st[RD + e] = s / ((float)batch_global * (float)S);
    }
    for (int r = threadIdx.x; r < R; r += NT) st[RD + E + 5 * RD + r] = p.exec_flags[r] ? (float)batch : 0.f;
}

// ------------------------------------------------------------------------------------------------
// k_epoch_accumulate (one workgroup)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_epoch_accumulate(const DevPlan* __restrict__ P, float alpha, float beta) {
    const DevPlan& p = *P;
    const int R = p.R, D = p.D, E = p.E, RD = R * D;
    float* st = p.stats;
    double* ep = p.epoch;
    __shared__ float sred[2];
    if (threadIdx.x == 0) {
        float se = 0.f, ss = 0.f;
        for (int c = 0; c < RD; ++c) se += st[c];
        for (int e = 0; e < E; ++e) ss += st[RD + e];
        const float ge = se / (float)(D * R);              // multimodn.py:194
        const float gs = ss / (float)E;                    // multimodn.py:196
        float* tail = st + RD + E + 5 * RD + R;
        tail[0] = ge * alpha + gs * beta;                  // multimodn.py:199-202
        tail[1] = ge; tail[2] = gs; tail[3] = 0.f;
        ep[RD + E + 5 * RD + R] += 1.0;                    // n_steps
    }
    for (int c = threadIdx.x; c < RD; c += NT) {
        ep[c] += (double)st[c];                                            // err_loss_epoch (f64 += f32)
        ep[RD + E + c] += (double)st[RD + E + c];                          // n_correct
        for (int k = 1; k < 5; ++k) {                                      // tp/tn/fp/fn kept in fp32
            double* a = ep + RD + E + k * RD + c;
            *a = (double)((float)*a + st[RD + E + k * RD + c]);
        }
    }
    for (int e = threadIdx.x; e < E; e += NT) ep[RD + e] += (double)st[RD + e];
    for (int r = threadIdx.x; r < R; r += NT) ep[RD + E + 5 * RD + r] += (double)st[RD + E + 5 * RD + r];
    (void)sred;
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
    size_t lds_bytes;
    int grad_blocks;
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
        int in = enc.n_features;
        for (int l = 0; l < enc.n_layers; ++l) {
            const mmn_linear& lin = enc.layer[l];
            const bool last = l == enc.n_layers - 1;
            if (lin.in_dim != in + (last ? m->state_size : 0)) return MMN_ERR_ARG;
            if (lin.out_dim != (last ? m->state_size : lin.out_dim) || lin.out_dim < 1) return MMN_ERR_ARG;
            if (!last && lin.out_dim > MMN_MAX_DIM) return MMN_ERR_UNSUPPORTED;
            in = lin.out_dim;
        }
    }
    return MMN_OK;
}

namespace {
struct Layout {
    size_t off_plan, off_states, off_hid, off_dpre, off_dz, off_dS, off_lossp, off_scp, off_cntp, off_flags,
        off_slabs, off_epoch, off_tasks, off_items, off_segs, total;
    int64_t hid_off[MMN_MAX_ENCODERS][MMN_MAX_LAYERS];
    int64_t hid_floats;
    std::vector<WTask> tasks;
    std::vector<WItem> items;
    std::vector<Seg> segs;
    int64_t slab_floats, n_grad_elems;
    int KS, max_tiles;
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

void build_layout(const mmn_model& m, int maxB, Layout& L) {
    const int S = m.state_size, E = m.n_encoders, D = m.n_decoders, R = E + 1;
    L.max_tiles = (maxB + TB - 1) / TB;
    int ks = (maxB + 511) / 512;
    if (ks < 1) ks = 1;
    if (ks > 16) ks = 16;
    L.KS = ks;
    int64_t ho = 0;
    for (int e = 0; e < E; ++e)
        for (int l = 0; l + 1 < m.enc[e].n_layers; ++l) {
            L.hid_off[e][l] = ho;
            ho += (int64_t)maxB * m.enc[e].layer[l].out_dim;
        }
    L.hid_floats = ho;

    // ---- wgrad tasks, slabs, gradient segments
    int64_t slab = 0, gstart = 0;
    auto add_items = [&](int task, int M, int ntot) {
        for (int m0 = 0; m0 < M; m0 += WG_TILE)
            for (int n0 = 0; n0 < ntot; n0 += WG_TILE)
                for (int k = 0; k < ks; ++k) L.items.push_back(WItem{task, m0, n0, k});
    };
    auto add_seg = [&](float* dst, int count, int64_t base, int64_t pstride, int nparts, int kdiv, int ntot,
                       int coff, int row_off) {
        if (!dst) return;
        L.segs.push_back(Seg{dst, gstart, base, pstride, count, nparts, kdiv, ntot, coff, row_off});
        gstart += count;
    };
    // init state: colsum of dS0
    {
        WTask t{};
        t.a_kind = A_DS; t.a_idx = E; t.M = S;
        t.in0_kind = IN_NONE; t.in1_kind = IN_NONE; t.k0 = 0; t.k1 = 0; t.bias = 1; t.ntot = 1; t.gate = 0;
        t.slab_base = slab; t.pstride = (int64_t)S;
        const int id = (int)L.tasks.size();
        L.tasks.push_back(t);
        add_items(id, S, 1);
        add_seg(m.g_init_state, S, slab, t.pstride, ks, 1, 1, 0, 0);
        slab += (int64_t)ks * t.pstride;
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
            t.bias = 1; t.ntot = lin.in_dim + 1; t.gate = e + 1;
            t.slab_base = slab; t.pstride = (int64_t)t.M * t.ntot;
            const int id = (int)L.tasks.size();
            L.tasks.push_back(t);
            add_items(id, t.M, t.ntot);
            add_seg(lin.gw, lin.out_dim * lin.in_dim, slab, t.pstride, ks, lin.in_dim, t.ntot, 0, 0);
            add_seg(lin.gb, lin.out_dim, slab, t.pstride, ks, 1, t.ntot, lin.in_dim, 0);
            slab += (int64_t)ks * t.pstride;
        }
    }
    // decoders: one task per grid row, all rows share one output of [2D x (S+1)] x (R*ks) partials
    {
        const int64_t pstride = (int64_t)(2 * D) * (S + 1);
        const int64_t base = slab;
        for (int r = 0; r < R; ++r) {
            WTask t{};
            t.a_kind = A_DZ; t.a_idx = r; t.M = 2 * D;
            t.in0_kind = IN_STATE_ROW; t.in0_idx = r; t.k0 = S; t.in1_kind = IN_NONE; t.k1 = 0;
            t.bias = 1; t.ntot = S + 1; t.gate = r;
            t.slab_base = base + (int64_t)r * ks * pstride; t.pstride = pstride;
            const int id = (int)L.tasks.size();
            L.tasks.push_back(t);
            add_items(id, t.M, t.ntot);
        }
        for (int d = 0; d < D; ++d) {
            add_seg(m.dec[d].gw, 2 * S, base, pstride, R * ks, S, S + 1, 0, 2 * d);
            add_seg(m.dec[d].gb, 2, base, pstride, R * ks, 1, S + 1, S, 2 * d);
        }
        slab += (int64_t)R * ks * pstride;
    }
    L.slab_floats = slab;
    L.n_grad_elems = gstart;

    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    L.off_plan = take(sizeof(DevPlan));
    L.off_states = take(sizeof(float) * (size_t)E * maxB * S);
    L.off_hid = take(sizeof(float) * (size_t)(L.hid_floats ? L.hid_floats : 1));
    L.off_dpre = take(sizeof(float) * (size_t)(L.hid_floats ? L.hid_floats : 1));
    L.off_dz = take(sizeof(float) * (size_t)R * maxB * 2 * D);
    L.off_dS = take(sizeof(float) * (size_t)(E + 1) * maxB * S);
    L.off_lossp = take(sizeof(float) * (size_t)L.max_tiles * R * D);
    L.off_scp = take(sizeof(float) * (size_t)L.max_tiles * E);
    L.off_cntp = take(sizeof(int32_t) * (size_t)L.max_tiles * R * D * 5);
    L.off_flags = take(sizeof(int32_t) * (size_t)(R + E + MMN_MAX_ENCODERS));
    L.off_slabs = take(sizeof(float) * (size_t)L.slab_floats);
    L.off_epoch = take(sizeof(double) * mmn_epoch_doubles(&m));
    L.off_tasks = take(sizeof(WTask) * L.tasks.size());
    L.off_items = take(sizeof(WItem) * L.items.size());
    L.off_segs = take(sizeof(Seg) * L.segs.size());
    L.total = o;
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
    char* ws = static_cast<char*>(workspace);
    DevPlan& h = pl->host;
    memset(&h, 0, sizeof(h));
    h.m = *m;
    h.S = m->state_size; h.E = m->n_encoders; h.D = m->n_decoders; h.R = h.E + 1;
    h.S8 = round_up(h.S, 8);
    h.ldS = pick_ld(h.S);
    int maxh = 8;
    for (int e = 0; e < h.E; ++e)
        for (int l = 0; l + 1 < m->enc[e].n_layers; ++l) maxh = maxh > m->enc[e].layer[l].out_dim ? maxh : m->enc[e].layer[l].out_dim;
    h.ldAct = pick_ld(maxh);
    h.maxB = max_batch; h.max_tiles = L.max_tiles; h.KS = L.KS;
    memcpy(h.hid_off, L.hid_off, sizeof(h.hid_off));
    h.states = reinterpret_cast<float*>(ws + L.off_states);
    h.hid = reinterpret_cast<float*>(ws + L.off_hid);
    h.dpre = reinterpret_cast<float*>(ws + L.off_dpre);
    h.dz = reinterpret_cast<float*>(ws + L.off_dz);
    h.dS = reinterpret_cast<float*>(ws + L.off_dS);
    h.lossp = reinterpret_cast<float*>(ws + L.off_lossp);
    h.scp = reinterpret_cast<float*>(ws + L.off_scp);
    h.cntp = reinterpret_cast<int32_t*>(ws + L.off_cntp);
    h.exec_flags = reinterpret_cast<int32_t*>(ws + L.off_flags);
    h.prev_row = h.exec_flags + h.R;
    h.nan_flags = h.prev_row + h.E;
    h.slabs = reinterpret_cast<float*>(ws + L.off_slabs);
    h.stats = stats;
    h.epoch = reinterpret_cast<double*>(ws + L.off_epoch);
    h.tasks = reinterpret_cast<WTask*>(ws + L.off_tasks);
    h.items = reinterpret_cast<WItem*>(ws + L.off_items);
    h.segs = reinterpret_cast<Seg*>(ws + L.off_segs);
    h.n_tasks = (int)L.tasks.size(); h.n_items = (int)L.items.size(); h.n_segs = (int)L.segs.size();
    h.n_grad_elems = L.n_grad_elems;
    pl->dev = reinterpret_cast<DevPlan*>(ws + L.off_plan);
    pl->grad_blocks = (int)((L.n_grad_elems + NT - 1) / NT);

    const ChainLds cl = chain_lds(h.ldS, h.ldAct);
    pl->lds_bytes = sizeof(float) * (size_t)cl.total;
    if (pl->lds_bytes > 160 * 1024) { delete pl; return MMN_ERR_UNSUPPORTED; }

    auto fail = [&](hipError_t e) { g_last_hip = (int)e; delete pl; return MMN_ERR_HIP; };
    hipError_t e;
    if ((e = hipMemcpy(pl->dev, &h, sizeof(h), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(h.tasks, L.tasks.data(), sizeof(WTask) * L.tasks.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(h.items, L.items.data(), sizeof(WItem) * L.items.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(h.segs, L.segs.data(), sizeof(Seg) * L.segs.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemset(h.epoch, 0, sizeof(double) * mmn_epoch_doubles(m))) != hipSuccess) return fail(e);
    if ((e = hipMemset(h.exec_flags, 0, sizeof(int32_t) * (h.R + h.E + MMN_MAX_ENCODERS))) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes)) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes)) != hipSuccess) return fail(e);
    *out = pl;
    return MMN_OK;
}

void mmn_plan_destroy(mmn_plan* p) { delete p; }

static int check_batch(const mmn_plan* p, const mmn_batch* b) {
    if (!p || !b) return MMN_ERR_ARG;
    if (b->batch < 1 || b->batch > p->max_batch || b->batch_global < b->batch || !b->y) return MMN_ERR_ARG;
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
    return MMN_OK;
}

int mmn_nan_scan(mmn_plan* p, const mmn_batch* b, int32_t* nan_flags_out, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    if (!nan_flags_out) return MMN_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(nan_flags_out, 0, sizeof(int32_t) * MMN_MAX_ENCODERS, st));
    if (b->n_seq == 0) return MMN_OK;
    int bps = 256 / b->n_seq;
    if (bps < 1) bps = 1;
    mmn_batch bb = *b;
    hipLaunchKernelGGL(k_nan_scan, dim3(b->n_seq * bps), dim3(NT), 0, st, bb, &p->dev->m, nan_flags_out, bps);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

int mmn_chain_fwd(mmn_plan* p, const mmn_batch* b, float err_penalty, float sc_pen_x001, int want_grads,
                  void* stream) {
    (void)sc_pen_x001;
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    const int tiles = (b->batch + TB - 1) / TB;
    const float cL = err_penalty / ((float)p->m.n_decoders * (float)(p->m.n_encoders + 1) * (float)b->batch_global);
    mmn_batch bb = *b;
    hipLaunchKernelGGL(k_chain_fwd, dim3(tiles), dim3(NT), p->lds_bytes, static_cast<hipStream_t>(stream), p->dev, bb, cL,
                       want_grads);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

static float sc_coeff(const mmn_plan* p, const mmn_batch* b, float beta) {
    return beta * 2.0f / ((float)p->m.n_encoders * (float)b->batch_global * (float)p->m.state_size);
}

int mmn_chain_bwd(mmn_plan* p, const mmn_batch* b, float sc_pen_x001, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    const int tiles = (b->batch + TB - 1) / TB;
    mmn_batch bb = *b;
    hipLaunchKernelGGL(k_chain_bwd, dim3(tiles), dim3(NT), p->lds_bytes, static_cast<hipStream_t>(stream), p->dev, bb,
                       sc_coeff(p, b, sc_pen_x001));
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

int mmn_wgrad(mmn_plan* p, const mmn_batch* b, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    const int ks = p->host.KS;
    const int rps = round_up((b->batch + ks - 1) / ks, TB);
    mmn_batch bb = *b;
    hipLaunchKernelGGL(k_wgrad, dim3(p->host.n_items), dim3(NT), 0, static_cast<hipStream_t>(stream), p->dev, bb, rps);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

static int launch_reduce(mmn_plan* p, const mmn_batch* b, int want_grads, void* stream) {
    const int tiles = (b->batch + TB - 1) / TB;
    hipLaunchKernelGGL(k_reduce, dim3(p->grad_blocks + 1), dim3(NT), 0, static_cast<hipStream_t>(stream), p->dev, b->batch,
                       b->batch_global, tiles, p->grad_blocks, want_grads);
    HIP_TRY(hipGetLastError());
    return MMN_OK;
}

int mmn_reduce(mmn_plan* p, const mmn_batch* b, void* stream) {
    int rc = check_batch(p, b);
    if (rc != MMN_OK) return rc;
    return launch_reduce(p, b, 1, stream);
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
    int rc = mmn_chain_fwd(p, b, err_penalty, sc_pen_x001, 1, stream);
    if (rc != MMN_OK) return rc;
    if ((rc = mmn_chain_bwd(p, b, sc_pen_x001, stream)) != MMN_OK) return rc;
    if ((rc = mmn_wgrad(p, b, stream)) != MMN_OK) return rc;
    if ((rc = launch_reduce(p, b, 1, stream)) != MMN_OK) return rc;
    if (accumulate_epoch) rc = mmn_epoch_accumulate(p, err_penalty, sc_pen_x001, stream);
    return rc;
}

int mmn_eval_step(mmn_plan* p, const mmn_batch* b, int accumulate_epoch, void* stream) {
    int rc = mmn_chain_fwd(p, b, 1.0f, 0.0f, 0, stream);
    if (rc != MMN_OK) return rc;
    if ((rc = launch_reduce(p, b, 0, stream)) != MMN_OK) return rc;
    if (accumulate_epoch) rc = mmn_epoch_accumulate(p, 1.0f, 0.0f, stream);
    return rc;
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
        default: return nullptr;
    }
}

}  // extern "C"
