// mmn_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the MultiModN sequential-fusion training
// step and the C ABI declared in include/mmn_hip.h.  Written for wave64 + fp32 MFMA
// (v_mfma_f32_16x16x4_f32: bit-exact fp32 FMA chains, so fp32 parity with the reference's ATen
// path holds up to summation order).  No CUDA names, no dual paths.
//
// Launch structure of one training step (reference: multimodn/multimodn.py:137-204):
//   k_prepare          any(isnan(x_k)) per data slot (:168) + repack of every weight matrix the
//                      chain kernels multiply by into MFMA-fragment order (forward W, backward W^T)
//   k_fb9<TILED, ..>   FUSED forward + reverse chain of one 16-row tile, 8 waves (E <= 4, n_features <= 64, hidden <= 32): init
//                      broadcast, hidden MLPs, state updates with the step's operands resident in registers, state-change
//                      partials, all D decoders on all E+1 states (K-split), CE-over-sigmoid, argmax, confusion counts,
//                      then grads wrt states / pre-activations; TILED = per-sample mode
//     k_fb8            the same as one kernel for wider shapes (n_features <= 128, h width <= 64), pair-per-encoder phases
//     k_fwd8 + k_bwd8  the same math as two 8-wave kernels (E <= 8)
//     k_chain_fwd/bwd  sequential form, any shape
//   MIMIC pipelines' modules (MIMIC_MLPEncoder: state enters the FIRST layer, activation on every layer, dropout
//   multipliers from mmn_batch.drop_mask; MLPDecoder heads):
//     k_mfwd + k_dec_fb + k_mbwd   the pipelines' shape class: 8-wave chain kernels with every operand in registers
//                      (mmn_chain_mimic.inc), the decoders' forward + loss + backward in a launch of their own
//     k_genf2_fwd/bwd  other all-MIMIC models (weights by LDS-DMA, descriptor as kernel argument)
//     k_gen_fwd/bwd    sequential form: any mix of encoder / decoder kinds and shapes, and per-sample mode
//   k_wgrad            grouped split-K "A^T B" GEMM: weight, bias and init-state grads as
//                      flat-gradient-shaped partial slabs
//   k_reduce           fixed-order slab reduction -> grads (+ Adam on the element just summed, :204);
//                      tile partials -> stats block (+ loss combination and epoch accumulators, :194-212)
//   k_adam / k_adam_accumulate   optimizer.step() over the flat buffers as its own launch
//                      (data parallel: after the all-reduce, together with the epoch accumulation)
//   k_ps_code / k_ps_layout / k_ps_gather   per-sample mode: regrouping of the rows into tiles of one
//                      executed sequence (mmn_regroup)
// ONE translation unit, split by section into the *.inc files included below (in this order): mmn_plan.inc (plan
// structs, device helpers), mmn_prepare.inc (k_prepare), mmn_chain_seq.inc (sequential chain kernels),
// mmn_generic.inc (generic tier: sequential and batched forms, k_dec_fb), mmn_chain_mimic.inc (k_mfwd, k_mbwd),
// mmn_wave_helpers.inc (single-wave helpers of the 8-wave tiers), mmn_chain_8w.inc (k_fwd8, k_bwd8, k_fb8), mmn_chain_fb9.inc
// (k_fb9), mmn_wgrad.inc, mmn_per_sample.inc (k_ps_*), mmn_adam_reduce.inc (k_adam, k_reduce, the one-shot data-parallel
// tail), mmn_host.inc (layout, plan, C ABI).  36 kernels, none with scratch memory (profiles/r04_kernel_resources.txt).
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
#include "mmn_plan.inc"
#include "mmn_prepare.inc"
#include "mmn_chain_seq.inc"
#include "mmn_generic.inc"
#include "mmn_chain_mimic.inc"
#include "mmn_wave_helpers.inc"
#include "mmn_chain_8w.inc"
#include "mmn_chain_fb9.inc"
#include "mmn_per_sample.inc"
#include "mmn_adam_reduce.inc"
#include "mmn_wgrad.inc"
#include "mmn_epoch_small.inc"
}  // namespace

#include "mmn_host.inc"
