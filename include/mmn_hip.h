/*
 * mmn_hip.h -- C ABI of libmmn_hip.so: the MI355X (gfx950) implementation of MultiModN's
 * sequential-fusion training step.
 *
 * The reference (EPFLiGHT/MultiModN) is pure Python on PyTorch and has NO FFI, operator registry
 * or plugin loader for this path; its boundary is the Python class surface
 * MultiModN(encoders, decoders, state).train_epoch(...) (multimodn/multimodn.py:66-98).  This
 * header is therefore the C ABI a maintainer would bind from that method's inner loop body
 * (multimodn/multimodn.py:119-212); each entry point cites the reference lines it replaces.
 * The binding the reference side would add is shown in INTEGRATION.md (ctypes, as used by
 * multimodn_amd/hip.py).
 *
 * Conventions
 *   - plain C, no torch types: device pointers, sizes, a hipStream_t passed as void*.
 *   - every pointer inside mmn_model / mmn_batch is a DEVICE pointer owned by the caller
 *     (PyTorch's allocator); the library never allocates device memory: the caller provides one
 *     workspace of mmn_workspace_bytes() bytes.
 *   - all launches are asynchronous on `stream`, capture-safe (no sync, no malloc) except
 *     mmn_plan_create (one synchronous hipMemcpy of descriptor tables) and mmn_epoch_read.
 *   - return value: 0 = ok, <0 = error (mmn_error_string()).  Nothing throws.
 *   - arithmetic: fp32 in, fp32 accumulate (v_mfma_f32_16x16x4_f32 = exact fp32 FMA chains);
 *     int64 targets; deterministic (no float atomics: fixed-order two-stage reductions).
 *
 * Environment (diagnostics and tests only; read ONCE per mmn_plan_create, never needed in production):
 *   MMN_FUSED=0 / MMN_FAST8=0               step down the kernel tiers (fused 8-wave k_fb9 / k_fb8 -> 8-wave pair k_fwd8 + k_bwd8 ->
 *                                           sequential k_chain_*); MMN_FB8_LEAN=0: k_fb8 also for the shapes k_fb9 takes
 *   MMN_GENERIC=1                           run MLPEncoder / ClassDecoder models on the generic tier's kernels too
 *   MMN_GEN_FAST=0 / MMN_DEC_LDS=0          generic tier: sequential form only / decoder operands from global memory
 *   MMN_WGRAD_ROWS=n                        rows per k_wgrad row-range split (default 512)
 *   MMN_FB9=0 / MMN_FB9_FULLK=0             k_fb8 instead of k_fb9 / k_fb9 with run-time trip counts
 *   MMN_GEN_BATCHED=0 / MMN_GEN_SPLIT=0 / MMN_MC=0   generic tier: sequential form / decoders inside the chain kernels / k_genf2_* as
 *                                           the chain where k_mfwd / k_mbwd would run
 *   MMN_WGRAD_SMALL=0 / MMN_MC_TILED=0      k_wgrad's two-workgroups-per-CU form also for models without a 64 x 64 gradient tile /
 *                                           per-sample batches of the MIMIC pipelines' shapes on the sequential k_gen_* kernels
 *                                           (MMN_MC_TILED is read per call)
 *   MMN_STAMPS=1                            phase timestamps of one workgroup (mmn_debug_buffer kind 3; tools/stamps*.py)
 *   MMN_VERBOSE=1                           plan summary on stderr
 */
#ifndef MMN_HIP_H
#define MMN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMN_VERSION 113            /* 0.1.13: mmn_model.flags / MMN_MODEL_GENERIC_TIER + mmn_per_sample_supported (per-sample mode for every MLPEncoder shape), mmn_dp_oneshot_detach (a plan that is really "not attached" again: the refused-attach path and detach close the peers' buffers, the plan must not keep their addresses); 0.1.12: mmn_epoch_small_rows / mmn_train_epoch_small (a whole epoch of small batches as one launch); 0.1.11: mmn_wgrad_reduce (the second half of mmn_train_step_ex on its own: k_wgrad with the stats block in its launch, then k_reduce as gradient blocks only); 0.1.10: mmn_regroup_multi (the batches of a captured per-sample group regrouped by one set of launches); 0.1.9: one-shot data-parallel exchange (mmn_dp_xbuf_*, mmn_dp_oneshot_*, mmn_adam_step_accumulate_oneshot, MMN_ERR_PEER); 0.1.8: mmn_source_hash (a library that was not built from the sources next to it does not load); 0.1.7: mmn_eval_step_ex (test(): outputs and row flag of a step collected by the call); 0.1.6: mmn_dp_rescale (uneven data-parallel shards);
                                      0.1.5: mmn_step_opts.next_drop_* (the next step's dropout multipliers in this step's last launch), mmn_dropout_adopt;
                                      0.1.4: mmn_train_step_ex (pre-scan of the next batch, flag sets in the stats block), mmn_pack_invalidate;
                                      0.1.3: + mmn_draw_dropout; 0.1.2: MIMIC_MLPEncoder / MLPDecoder (mmn_encoder.kind, mmn_decoder.hidden, mmn_batch.drop_mask) */
#define MMN_MAX_ENCODERS 16
#define MMN_MAX_DECODERS 8
#define MMN_MAX_LAYERS 8           /* hidden layers + the state-update Linear */
#define MMN_MAX_DIM 128            /* state_size and every hidden width (what the LDS carve of the chain kernels admits; wider
                                      models get MMN_ERR_UNSUPPORTED); n_features is unbounded */
#define MMN_MAX_DEC_HIDDEN 3       /* hidden layers of an MLPDecoder */

/* activation of MLPEncoder hidden layers (multimodn/encoders/mlp_encoder.py:56,75-76) */
#define MMN_ACT_IDENTITY 0
#define MMN_ACT_RELU 1
#define MMN_ACT_SIGMOID 2

enum {
    MMN_OK = 0,
    MMN_ERR_ARG = -1,          /* null pointer / out-of-range size */
    MMN_ERR_UNSUPPORTED = -2,  /* dims beyond MMN_MAX_* or LDS budget */
    MMN_ERR_WORKSPACE = -3,    /* workspace too small / misaligned */
    MMN_ERR_HIP = -4,          /* a HIP runtime call failed (see mmn_last_hip_error) */
    MMN_ERR_SEQUENCE = -5,     /* encoder sequence invalid (repeat / out of range) */
    MMN_ERR_PEER = -6          /* one-shot data-parallel exchange: a peer's data did not arrive within the wait's bound */
};

/* One nn.Linear: weight [out_dim, in_dim] row-major, bias [out_dim]; gw/gb receive the grads.
 * (encoders.{e}.layers.{l}.weight/.bias in the reference state_dict) */
typedef struct mmn_linear {
    const float* w;
    const float* b;
    float* gw;
    float* gb;
    int32_t out_dim;
    int32_t in_dim;
} mmn_linear;

/* MLPEncoder (multimodn/encoders/mlp_encoder.py:49-80): n_layers-1 hidden Linear+activation on x
 * only, then Linear(cat[h, state]) with no activation.  layer[n_layers-1].in_dim = h_dim + S with
 * the h columns FIRST (mlp_encoder.py:78).  n_layers == 1 is SLP/Linear/LogisticEncoder. */
/* kind MMN_ENC_MIMIC = MIMIC_MLPEncoder (multimodn/encoders/mlp_encoder.py:9-47): the FIRST Linear
 * reads Dropout(cat[x, state]) (x columns first, :40-41: layer[0].in_dim = n_features + S), every
 * further Linear reads the previous output, the activation follows EVERY Linear including the last
 * one, whose output is the new state (:42-47).  The Dropout module (layers.0 of the reference) has no
 * parameters: layer[l] here is the reference's layers.{l+1}; its multipliers arrive per step through
 * mmn_batch.drop_mask. */
#define MMN_ENC_MLP 0
#define MMN_ENC_MIMIC 1
typedef struct mmn_encoder {
    int32_t n_features;
    int32_t n_layers;
    int32_t activation;
    int32_t kind;                  /* MMN_ENC_MLP / MMN_ENC_MIMIC */
    mmn_linear layer[MMN_MAX_LAYERS];
} mmn_encoder;

/* ClassDecoder with sigmoid and n_classes = 2 = LogisticDecoder (decoders.py:9-20,49-53):
 * fc.weight [2, S], fc.bias [2]. */
/* n_hidden > 0 = MLPDecoder(state_size, hidden_layers, 2, sigmoid, hidden_activation)
 * (decoders.py:22-46): hidden[l] are layers.0 .. layers.{n_hidden-1} (Linear + hidden_activation),
 * w/b the last Linear [2, hidden[n_hidden-1].out_dim] followed by the sigmoid.  n_hidden == 0 with an
 * MLPDecoder's parameter names is the same arithmetic as ClassDecoder. */
typedef struct mmn_decoder {
    const float* w;
    const float* b;
    float* gw;
    float* gb;
    int32_t n_hidden;
    int32_t hidden_activation;     /* MMN_ACT_* */
    mmn_linear hidden[MMN_MAX_DEC_HIDDEN];
} mmn_decoder;

/* MultiModN(state_size, encoders, decoders, ...) with TrainableInitState
 * (multimodn.py:66-87, state.py:19-32). */
/* mmn_model.flags.  MMN_MODEL_GENERIC_TIER: plan the generic tier's kernels (k_gen_fwd / k_gen_bwd: any mix of encoder / decoder
 * kinds and widths) also for a model the fused MLPEncoder kernels would take.  What per-sample mode (mmn_regroup*, BASELINE
 * configs[4]) needs for MLPEncoder models OUTSIDE the fused kernel's tiled form (n_features > 64 or hidden widths > 32):
 * the reference runs any shape at batch size 1 (multimodn/multimodn.py:509-531); mmn_per_sample_supported(plan) says
 * whether a plan takes per-sample batches as it is. */
#define MMN_MODEL_GENERIC_TIER 1
typedef struct mmn_model {
    int32_t state_size;
    int32_t n_encoders;
    int32_t n_decoders;
    int32_t flags;                 /* MMN_MODEL_* (ABI 113; was `reserved`, 0) */
    const float* init_state;       /* [S]  (init_state.state_value) */
    float* g_init_state;           /* [S] */
    mmn_encoder enc[MMN_MAX_ENCODERS];
    mmn_decoder dec[MMN_MAX_DECODERS];
} mmn_model;

/* One mini-batch as train_epoch sees it after collate + .to(device) (multimodn.py:119,132-135):
 * data slot k -> x[k] [batch, F] fp32 (row stride ldx[k] floats), targets [batch, D] int64.
 * (seq_data[t], seq_enc[t]) is get_encoder_iterable's t-th pair (multimodn.py:509-531): feed data
 * slot seq_data[t] to encoder seq_enc[t].  nan_flags (optional, device, one int per DATA SLOT, as
 * written by mmn_nan_scan): nonzero = that slot's batch contained a NaN, skip its encoder
 * (multimodn.py:168-169).  NULL = the caller already removed skipped slots from the sequence.
 * flags_ready != 0: nan_flags already hold this batch's decision (an earlier step pre-scanned it, see
 * mmn_step_opts.next; with data parallel the flag sets have been through the all-reduce): the step does not scan.
 * batch_global: divisor of every batch mean; > batch for a data-parallel shard. */
typedef struct mmn_batch {
    const float* x[MMN_MAX_ENCODERS];
    int32_t ldx[MMN_MAX_ENCODERS];
    const int64_t* y;
    const int32_t* nan_flags;
    int32_t batch;
    int32_t batch_global;
    int32_t n_seq;
    int32_t flags_ready;
    int32_t seq_data[MMN_MAX_ENCODERS];
    int32_t seq_enc[MMN_MAX_ENCODERS];
    /* Per-sample mode (BASELINE.json configs[4]: per-sample missing modalities and per-sample
     * encoder order; a build-defined extension - the reference defines it only at batch size 1,
     * multimodn.py:168,518-523 - whose result is the mean over the samples of the reference's
     * batch-size-1 result).  Both NULL = ordinary batch.  Otherwise the caller has grouped the rows
     * into 16-row tiles of ONE executed sequence each: tile t covers rows [16 t, 16 t + tile_rows[t])
     * (tile_rows[t] in 0..16, the remaining rows of the tile are padding) and runs the encoders
     * packed in tile_seq[t], 4 bits per step (encoder id + 1, first step in the low bits, 0 ends the
     * list; a missing modality is simply absent from the list).  Then: batch = 16 * number of tiles,
     * batch_global = the true number of samples, data slot k holds encoder k's features
     * (seq_data[t] = seq_enc[t] = t, n_seq = n_encoders), every feature value is finite (missing
     * entries zeroed), nan_flags = NULL.  mmn_train_step / mmn_train_step_adam / mmn_eval_step,
     * fused-kernel shapes only (MMN_ERR_UNSUPPORTED otherwise). */
    const int32_t* tile_rows;
    const int32_t* tile_seq;
    /* MIMIC_MLPEncoder dropout (mlp_encoder.py:34,41), indexed by ENCODER id: drop_mask[e] = device
     * [batch x (n_features_e + S)] multipliers (0 or 1/(1-p), row stride n_features_e + S) applied to
     * cat([x, state]) before the first Linear of encoder e, forward and backward.  NULL = no dropout
     * for that encoder this step (eval mode, p = 0, or an MLPEncoder).  The caller provides them: from its own generator,
     * or with mmn_draw_dropout below (what multimodn_amd does; tests hand in the reference's recorded draws). */
    const float* drop_mask[MMN_MAX_ENCODERS];    /* 16-byte aligned (the backward kernel fetches them as float4s; MMN_ERR_ARG otherwise) */
} mmn_batch;

/* Per-step statistics block, fp32, written by mmn_reduce (local sums, ready for an all-reduce)
 * and consumed by mmn_epoch_accumulate.  R = n_encoders + 1, D = n_decoders.
 *   [0, R*D)                 err_loss grid      (multimodn.py:123,146,181)
 *   [R*D, R*D+E)             state_change       (multimodn.py:124,174)
 *   then 5 blocks of R*D     n_correct, tp, tn, fp, fn as exact fp32 counts (multimodn.py:147,157)
 *   then R                   rows executed this step: batch rows if the state row was produced
 *                            (n_samples_epoch increments, multimodn.py:121,171)
 *   then 4                   loss, global_err_loss, global_state_change, reserved
 *                            (multimodn.py:194-202; filled by mmn_epoch_accumulate)
 *   then 2 * MMN_MAX_ENCODERS   the two NaN-flag sets (mmn_nan_flags_set): one word per data slot, 0 = no NaN, the
 *                            float 1.0 = NaN found.  They live here so that a data-parallel caller's ONE all-reduce of
 *                            [grads | stats] also carries the pre-scanned flags of the next batch (the per-rank flags
 *                            are summed; any non-zero word means "some rank saw a NaN": multimodn.py:168 on the GLOBAL batch). */
size_t mmn_stats_floats(const mmn_model* m);

typedef struct mmn_plan mmn_plan;   /* opaque host handle */

int mmn_version(void);
/* First 16 hex digits of the sha256 over the sources this library was built from (multimodn_amd/build.py::source_hash;
 * "unknown" for a build without -DMMN_SOURCE_HASH).  The Python binding refuses a library whose hash is not that of the
 * sources next to it: a stale build must not ship silently. */
const char* mmn_source_hash(void);
const char* mmn_error_string(int code);
int mmn_last_hip_error(void);

/* Bytes of device workspace needed for batches of up to max_batch rows. */
size_t mmn_workspace_bytes(const mmn_model* m, int max_batch);

/* Build launch tables for (model, max_batch) inside `workspace` (256-byte aligned device memory).
 * `stats` = device fp32 block of mmn_stats_floats(); put it right behind the flat gradient buffer
 * to all-reduce grads + stats in one collective.  Synchronous (descriptor upload). */
int mmn_plan_create(const mmn_model* m, int max_batch, void* workspace, size_t workspace_bytes,
                    float* stats, mmn_plan** out);
void mmn_plan_destroy(mmn_plan* p);

/* Plan-owned NaN flags (device, MMN_MAX_ENCODERS words, zero after plan creation; inside the stats block).  Put this
 * pointer into mmn_batch.nan_flags to keep the NaN-skip decision on the device.  There are TWO sets (which = 0, 1;
 * mmn_nan_flags = set 0) so that one step can consume one set while it pre-scans the next batch into the other. */
int32_t* mmn_nan_flags(mmn_plan* p);
int32_t* mmn_nan_flags_set(mmn_plan* p, int which);

/* The chain kernels read the weights from fragment-order copies inside the workspace.  The library rebuilds them
 * (k_prepare) in the first step after mmn_plan_create, after every training step that does not apply the optimizer
 * itself (mmn_train_step: somebody else's optimizer is about to change the parameters) and after mmn_pack_invalidate();
 * a step with the fused Adam tail (mmn_train_step_adam, mmn_adam_step_accumulate) writes every parameter it updates
 * straight into its copies, so that no repack launch sits in front of the next step.  Call mmn_pack_invalidate whenever
 * anything but the library writes the parameters (load_state_dict, a foreign optimizer, manual edits). */
void mmn_pack_invalidate(mmn_plan* p);
/* Rebuild the copies now (one launch) if they are stale; afterwards they are current.  For callers that capture steps
 * into a hipGraph: a captured step must not depend on whether a repack was due at capture time. */
int mmn_pack_refresh(mmn_plan* p, void* stream);

/* Per-step preparation, one launch: (a) if b->nan_flags != NULL, multimodn.py:168:
 * nan_flags[k] = 1 iff any element of data slot k is NaN, for the slots the sequence names (flags
 * must be zero on entry; mmn_reduce / mmn_train_step / mmn_eval_step re-zero them after their last
 * reader); (b) if want_grads, transposed copies of the weights the backward chain multiplies by
 * (a layout transform with no reference counterpart). */
int mmn_prepare(mmn_plan* p, const mmn_batch* b, int want_grads, void* stream);
/* (a) alone. */
int mmn_nan_scan(mmn_plan* p, const mmn_batch* b, void* stream);

/* Which device kernel mmn_chain_fwd (backward = 0) / mmn_chain_bwd (backward = 1) launches for this
 * plan and batch: "k_fwd8"/"k_bwd8" (8-wave tier for MIMIC-like shapes), "k_chain_fwd"/"k_chain_bwd" (any MLPEncoder /
 * ClassDecoder shape); models with a MIMIC_MLPEncoder or an MLPDecoder: "k_mfwd"/"k_mbwd" (the MIMIC pipelines' shape),
 * "k_genf2_fwd"/"k_genf2_bwd" (other all-MIMIC models), "k_gen_fwd"/"k_gen_bwd" (anything else).  For matching rocprof
 * rows.  backward = 2 asks for the fused forward+backward kernel ("k_fb9", "k_fb8" or ""), 3 for the decoders' own
 * launch ("k_dec_fb" or ""). */
const char* mmn_chain_kernel_name(mmn_plan* p, const mmn_batch* b, int backward);

/* Forward chain, one launch: init-state broadcast (state.py:29-32), every executed encoder
 * (mlp_encoder.py:74-80), state-change partials (multimodn.py:174), all D decoders on all E+1
 * states with CrossEntropy-over-sigmoids, argmax and confusion counts (decoders.py:19-20,
 * multimodn.py:141-157,176-191).  want_grads != 0 also stores what backward needs. */
int mmn_chain_fwd(mmn_plan* p, const mmn_batch* b, float err_penalty, float state_change_penalty_x001,
                  int want_grads, void* stream);

/* Reverse chain (the activation-gradient half of loss.backward(), multimodn.py:203). */
int mmn_chain_bwd(mmn_plan* p, const mmn_batch* b, float state_change_penalty_x001, void* stream);

/* Forward + reverse chain.  ONE launch (k_fb8) when the 8-wave tier applies and E <= 4: the state
 * tiles, dz and hidden activations then stay in LDS between the two halves; otherwise the two
 * launches above.  (mmn_chain_kernel_name(p, b, 2) names the fused kernel, "" if it does not apply; 3 names the decoder
 * kernel of the generic tier's split form, where mmn_chain_fwd is two launches: want_grads | 2 launches only the chain
 * kernel, want_grads | 4 only the decoder kernel - for per-kernel timing.) */
int mmn_chain_fwd_bwd(mmn_plan* p, const mmn_batch* b, float err_penalty, float state_change_penalty_x001,
                      void* stream);

/* Weight/bias/init-state gradients as split-K partial slabs (the other half of :203).
 * ORDER: mmn_wgrad and mmn_reduce take "which state rows exist this step" and "which row fed encoder e" from tables the
 * chain kernel of THIS batch left in the workspace (its tile 0), not from b's own sequence and NaN flags: they are only
 * correct right behind mmn_chain_fwd_bwd (or mmn_chain_fwd + mmn_chain_bwd) of the same batch on the same stream, with
 * no chain launch of another batch - a forward-only step included - in between.  mmn_train_step keeps that order. */
int mmn_wgrad(mmn_plan* p, const mmn_batch* b, void* stream);

/* Fixed-order reduction of slabs -> gw/gb/g_init_state and of per-tile partials -> stats (ORDER: see mmn_wgrad). */
int mmn_reduce(mmn_plan* p, const mmn_batch* b, void* stream);

/* Loss combination (multimodn.py:194-202) and epoch accumulators (multimodn.py:206-212) from
 * `stats` (call after the all-reduce when data-parallel). */
int mmn_epoch_accumulate(mmn_plan* p, float err_penalty, float state_change_penalty_x001, void* stream);

/* Data parallel with uneven shards (ranks feed different numbers of rows, e.g. the last batch of an epoch): every rank
 * runs its step with mmn_batch.batch_global = `nominal_batch` - the same power of two on all ranks - and, after the
 * all-reduce of [grads | stats], calls this in front of mmn_epoch_accumulate / mmn_adam_step_accumulate: the n floats
 * at `grads` (n may be 0: evaluation), the loss cells and the state changes are multiplied by
 * nominal_batch / (rows of grid row 0 in the summed stats = the true global batch), which makes them the reference's
 * means over the global batch (multimodn.py:184-196).  Counters and row counts are sums already. */
int mmn_dp_rescale(mmn_plan* p, float* grads, int64_t n, int nominal_batch, void* stream);

/* prepare + chain_fwd + chain_bwd + wgrad + reduce in order: one full multimodn.py:137-203 body
 * (without optimizer.step).  accumulate_epoch != 0 folds mmn_epoch_accumulate into the reduce
 * launch (single-GPU); pass 0 when an all-reduce of [grads | stats] must happen first. */
int mmn_train_step(mmn_plan* p, const mmn_batch* b, float err_penalty, float state_change_penalty_x001,
                   int accumulate_epoch, void* stream);

/* mmn_train_step / mmn_train_step_adam with the extras that take k_prepare off the critical path:
 *   adam   (may be NULL) as in mmn_train_step_adam;
 *   next   (may be NULL) the batch the NEXT call will run: its NaN scan (multimodn.py:168) rides in THIS step's last
 *          launch and writes next->nan_flags, which must be the flag set this step does not use.  The next call then
 *          passes the same batch with flags_ready = 1.  Data parallel: all-reduce [grads | stats] in between - the flag
 *          sets are part of the stats block.  Ignored for per-sample batches and when next->nan_flags is NULL.
 *   next_drop_p != NULL (needs next): the step's last launch also draws the NEXT step's dropout multipliers - what
 *          mmn_draw_dropout(p, next, next_drop_p, next_drop_seed, next_drop_buf, next_drop_floats, .) would draw in
 *          front of that step, with the same draw index - so no k_dropout launch sits in front of the next step: the next
 *          call adopts them with mmn_dropout_adopt.  The buffer may be the one this step's multipliers live in (its last
 *          reader, the backward kernel, has run). */
typedef struct mmn_step_opts {
    const struct mmn_adam* adam;
    const mmn_batch* next;
    int32_t accumulate_epoch;
    int32_t reserved;
    const float* next_drop_p;
    float* next_drop_buf;
    uint64_t next_drop_seed;
    uint64_t next_drop_floats;
} mmn_step_opts;
int mmn_train_step_ex(mmn_plan* p, const mmn_batch* b, float err_penalty, float state_change_penalty_x001,
                      const mmn_step_opts* opts, void* stream);

/* Per-sample mode helper: regroup the rows of an ordinary batch, ON THE DEVICE, into the tile layout
 * mmn_batch.tile_rows / tile_seq describe (four launches, deterministic: no atomics).
 *   in  : x[k] = data slot k [batch x F] (NaN anywhere in a row = that sample's modality is missing),
 *         y, batch = number of samples (any: one workgroup per 512 rows), batch_global; sequence fields are ignored
 *   seq : device int64 [batch x n_encoders], sample b feeds data slot k to encoder seq[b][k]
 *         (all modalities then need the same width; models of at most 4 encoders), or NULL: slot k feeds encoder k
 *         (models of at most 8 encoders - the reference's feature-wise pipelines have five and six,
 *         pipelines/titanic/titanic_missingness_pipeline.py:26,71: per-sample missing features, default order)
 *   out : caller-allocated device buffers for rows = mmn_regroup_rows(batch, n_encoders) rows:
 *         out->x[e] [rows x F_e] (row stride out->ldx[e]), out->y [rows x D], out->tile_rows and
 *         out->tile_seq [rows / 16]; the call fills them (zeros where a modality is missing and in
 *         padding rows) and completes every other field of *out, so that *out can go straight into
 *         mmn_train_step.  The plan's max_batch must be >= rows.  E <= 4. */
int mmn_regroup_rows(int batch, int n_encoders);
int mmn_regroup(mmn_plan* p, const mmn_batch* in, const int64_t* seq, mmn_batch* out, void* stream);
/* The same with caller-owned scratch (device, int32[2 * batch + rows]; NULL = the plan's): the call then touches nothing
 * of the plan's workspace, so the regrouping of the NEXT batch may run on another stream while this plan's step runs.
 * scratch + 2 * batch is the "source row of every position" table (mmn_debug_buffer kind 5 for the plan's scratch). */
int mmn_regroup_ex(mmn_plan* p, const mmn_batch* in, const int64_t* seq, mmn_batch* out, int32_t* scratch, void* stream);
/* n <= 8 batches regrouped by ONE set of four launches (ABI 110): what a captured group of per-sample steps runs in front of
 * its first step.  ins / seqs / outs / scratches: host arrays of n pointers with mmn_regroup_ex's meaning per batch
 * (seqs or seqs[j] may be NULL; scratches is required for n > 1, each int32[2 * batch + rows]).  One multi-batch call of
 * a plan in flight at a time (its histograms live in the plan). */
int mmn_regroup_multi(mmn_plan* p, int n, const mmn_batch* const* ins, const int64_t* const* seqs, mmn_batch* const* outs,
                      int32_t* const* scratches, void* stream);

/* Forward-only step for test()/predict()/get_states() (multimodn.py:255-492): fwd + reduce
 * (+ accumulate).  Leaves the state rows and the decoder outputs of every grid row in the workspace
 * (mmn_debug_buffer kinds 0 and 1). */
int mmn_eval_step(mmn_plan* p, const mmn_batch* b, int accumulate_epoch, void* stream);
/* mmn_eval_step plus what the reference's test() keeps of the step (multimodn.py:354-357,410-419): the decoders'
 * outputs on grid row `row` (the state after encoder row - 1; [batch x 2D] floats, decoder d at columns 2d, 2d + 1) are
 * copied to out_dst and "that row exists in this step" (0 / 1) to *flag_dst (may be NULL) - device pointers, e.g. the
 * step's slice of an epoch-sized buffer: no host synchronisation, one call per step. */
int mmn_eval_step_ex(mmn_plan* p, const mmn_batch* b, int accumulate_epoch, int row, float* out_dst, int32_t* flag_dst,
                     void* stream);

/* optimizer.step() (multimodn.py:204) for torch.optim.Adam as the reference pipelines build it
 * (titanic_mlp_pipeline.py:74), over FLAT buffers: one launch (k_adam) for the whole model.
 * All pointers are device memory owned by the caller; params/grads/exp_avg/exp_avg_sq hold n
 * floats each and are 16-byte aligned.  The n_seg tensors that make up the flat buffers are
 * [seg_start[i], seg_start[i+1]) (seg_start[0] = 0, seg_start[n_seg] = n).  steps holds
 * mmn_adam_blocks(n) identical rows of n_seg floats: tensor i's step count (float, like torch's
 * state["step"]); every workgroup reads and advances its own row, so a captured launch can be
 * replayed and no workgroup waits on another.  Row 0 is the one to read; to set the counters
 * (load_state_dict) write every row.  seg_skip (may be NULL): seg_skip[i] != 0 -> tensor i has
 * grad None this step and is left untouched (no moment decay, no step increment), torch's
 * behaviour for the encoders the reference skips on a NaN batch (multimodn.py:168).
 * amsgrad / decoupled weight decay are not implemented. */
typedef struct mmn_adam {
    float* params;
    const float* grads;
    float* exp_avg;
    float* exp_avg_sq;
    float* steps;
    const int32_t* seg_start;
    const int32_t* seg_skip;
    int64_t n;
    double lr, beta1, beta2, eps, weight_decay;   /* doubles: torch evaluates 1 - beta^t in double */
    int32_t n_seg;
    int32_t maximize;
} mmn_adam;
int mmn_adam_blocks(int64_t n);      /* rows of `steps`; 0 if n is out of range */
int mmn_adam_step(const mmn_adam* d, void* stream);
/* Data-parallel tail in one launch, after the all-reduce of [grads | stats]: mmn_adam_step plus
 * mmn_epoch_accumulate. */
int mmn_adam_step_accumulate(mmn_plan* p, const mmn_adam* d, float err_penalty, float state_change_penalty_x001,
                             void* stream);

/* One-shot data-parallel exchange fused with that tail (opt-in; the default data-parallel step is ONE RCCL all-reduce of
 * [grads | stats] followed by mmn_adam_step_accumulate).  Every rank owns an exchange buffer that all peer PROCESSES map
 * (hipIpc; peer access over xGMI): mmn_dp_xbuf_bytes(plan) bytes from mmn_dp_xbuf_alloc, whose 64-byte handle the caller
 * hands to the peers (any transport: torch.distributed's object collectives), who map it with mmn_dp_xbuf_open.
 * mmn_dp_oneshot_attach(plan, world, rank, xbufs, spin_ms): xbufs[r] = rank r's buffer as THIS process sees it (its own
 * allocation at xbufs[rank]); world <= 8 (one node).  mmn_adam_step_accumulate_oneshot then replaces the all-reduce AND
 * mmn_adam_step_accumulate: one launch that publishes this rank's [grads | stats] chunk by chunk, waits for the same
 * chunks of the peers (bounded: spin_ms per chunk), adds all ranks' chunks in rank order - bitwise the same sums on every
 * rank -, leaves them in [grads | stats] and applies Adam + the epoch accumulation.  Requires the fusable Adam layout
 * (mmn_adam_fusable) and the same number of such calls on every rank (the step number is a device-side counter: the
 * call may sit in a captured graph).  When a wait runs out, the workgroup leaves its parameters untouched and raises the
 * error word: mmn_dp_oneshot_error (no synchronisation: a host-mapped word) and every later call return MMN_ERR_PEER.
 * Reference: none - the reference is single-device (multimodn/multimodn.py:79-80). */
size_t mmn_dp_xbuf_bytes(const mmn_plan* p);
int mmn_dp_xbuf_alloc(size_t bytes, void** dev_ptr, void* handle64);
int mmn_dp_xbuf_open(const void* handle64, void** dev_ptr);
int mmn_dp_xbuf_close(void* dev_ptr, int own);
int mmn_dp_oneshot_attach(mmn_plan* p, int world, int rank, void* const* xbufs, int spin_ms);
int mmn_dp_oneshot_error(mmn_plan* p);
/* The plan forgets the exchange buffers (and frees its step counters / error words): to be called BEFORE the buffers are
 * closed (mmn_dp_xbuf_close) - when an attach was refused by a peer, or on detach.  Waits for the device first: an exchange
 * kernel may still be running.  mmn_adam_step_accumulate_oneshot returns MMN_ERR_ARG afterwards until the next attach. */
int mmn_dp_oneshot_detach(mmn_plan* p);
/* After MMN_ERR_PEER: what the first wait that ran out was looking at - out8 = [1 + peer rank, chunk, the step it waited for,
 * the flag value it last saw, the peer's flag of the other buffer parity, this rank, world, this rank's own step counter of
 * that chunk] (zeros: no wait has run out).  Diagnostic; synchronises with the device when an error is recorded. */
int mmn_dp_oneshot_diag(mmn_plan* p, unsigned out8[8]);

/* 1 if this plan runs per-sample batches (mmn_batch.tile_rows / tile_seq from mmn_regroup*): the fused kernel's tiled form
 * (MLPEncoder family, n_features <= 64, hidden widths <= 32), or the generic tier's (MIMIC modules, and any model planned
 * with MMN_MODEL_GENERIC_TIER: up to 8 encoders, of which 5 .. 8 in the default encoder order only); the fused kernel: at most 4
 * encoders.  0: mmn_regroup* return MMN_ERR_UNSUPPORTED.  (ABI 113) */
int mmn_per_sample_supported(const mmn_plan* p);
int mmn_adam_step_accumulate_oneshot(mmn_plan* p, const mmn_adam* d, float err_penalty, float state_change_penalty_x001,
                                     void* stream);

/* mmn_train_step with optimizer.step() fused behind the gradient sum (single GPU: no all-reduce in
 * between): the last launch forms each gradient element and immediately applies Adam to that
 * parameter.  The gradients are still written.  Requires the optimizer's flat layout to be the
 * plan's gradient layout (same tensors in the same order, adam->grads = the address given as
 * mmn_model.g_init_state, every other gradient pointer contiguous behind it): otherwise
 * MMN_ERR_UNSUPPORTED and nothing is launched - call mmn_train_step + mmn_adam_step instead.  A
 * tensor whose encoder did not run this step (NaN batch) is left untouched, as torch leaves
 * parameters with grad None.  The first call with a new (seg_start, grads) pair reads seg_start back
 * synchronously: make it outside stream capture.  mmn_reduce_adam is the last launch alone. */
int mmn_train_step_adam(mmn_plan* p, const mmn_batch* b, float err_penalty, float state_change_penalty_x001,
                        int accumulate_epoch, const mmn_adam* adam, void* stream);
/* The layout check alone: MMN_OK if `adam` can be fused with this plan's launches, else MMN_ERR_UNSUPPORTED / MMN_ERR_ARG. */
int mmn_adam_fusable(mmn_plan* p, const mmn_adam* adam);
int mmn_reduce_adam(mmn_plan* p, const mmn_batch* b, const mmn_adam* adam, void* stream);
/* ABI 111.  The second half of a training step on its own (per-kernel timing, tests): what mmn_wgrad + mmn_reduce (+ Adam
 * when opts->adam is set, + the epoch accumulation when opts->accumulate_epoch) do, in the launch layout mmn_train_step_ex
 * uses since round 5 - the stats block (tile partials -> step statistics, Adam's per-tensor coefficients and step counters)
 * rides in the k_wgrad launch, where it costs nothing, and k_reduce is gradient blocks only (multimodn.py:193-204; same
 * results bit for bit).  opts->next / next_drop_* are ignored here.  MMN_SIDE=0: the round-4 layout.  ORDER: as mmn_wgrad. */
int mmn_wgrad_reduce(mmn_plan* p, const mmn_batch* b, float err_penalty, float state_change_penalty_x001,
                     const struct mmn_step_opts* opts, void* stream);

/* nn.Dropout of the MIMIC_MLPEncoders (mlp_encoder.py:34,41) as ONE launch (k_dropout): draws the multipliers of every
 * MIMIC encoder of b's sequence whose drop_p[e] > 0 into `buf` (device, 16-byte aligned, mmn_dropout_floats(p, b->batch)
 * floats: encoder after encoder in encoder-id order, [batch x (n_features_e + S)] each) and points b->drop_mask[e] at
 * them (host fields of *b).  Counter-based generator: Philox4x32-10, key = seed, counter = (index of the 4-float group
 * inside buf, draw index); a multiplier is 1/(1-p) if u >= p else 0, u = (32 random bits >> 8) * 2^-24.  The draw index
 * lives in the plan's workspace and advances by one per call ON THE DEVICE (in the k_prepare launch of the step that
 * follows - mmn_prepare / mmn_train_step* / mmn_eval_step - or in a one-thread launch when another draw comes first),
 * so a captured draw + step pair draws fresh multipliers at every hipGraph replay; mmn_dropout_reset restarts it at 0 (call it
 * when the seed changes).  drop_p: host array of n_encoders probabilities (entries of non-MIMIC encoders are ignored). */
size_t mmn_dropout_floats(mmn_plan* p, int batch);
int mmn_draw_dropout(mmn_plan* p, mmn_batch* b, const float* drop_p, uint64_t seed, float* buf, size_t buf_floats,
                     void* stream);
int mmn_dropout_reset(mmn_plan* p, void* stream);
/* The multipliers of b's step are already in `buf` (an earlier step drew them: mmn_step_opts.next_drop_p with the same
 * drop_p / buffer): point b->drop_mask[e] at them exactly as mmn_draw_dropout would, without a launch.  The step that
 * follows advances the draw index as it does behind mmn_draw_dropout. */
int mmn_dropout_adopt(mmn_plan* p, mmn_batch* b, const float* drop_p, float* buf, size_t buf_floats, void* stream);

/* Epoch accumulators (device, inside the workspace): reset at epoch start, read at epoch end.
 * mmn_epoch_read synchronises the stream.  Layout of `out` (doubles): err_sum[R*D], sc_sum[E],
 * n_correct[R*D], tp[R*D], tn[R*D], fp[R*D], fn[R*D] (the four accumulated in fp32 like the
 * reference's torch.zeros tensors, multimodn.py:112-115,209-212), rows[R], n_steps[1]. */
/* ---- the whole batch loop of one train_epoch call in ONE launch (small models, small batches)  [round 5, ABI 112]
 * Replaces: `for batch in train_loader: ... optimizer.step()` of multimodn/multimodn.py:117-212 as the reference's Titanic
 * pipeline runs it (pipelines/titanic/titanic_mlp_pipeline.py:63-85: batches of 32 rows, 1,379 parameters) - one workgroup,
 * parameters and Adam moments resident in LDS / registers, every batch's forward, loss grid, backward, weight gradients,
 * torch.optim.Adam update and epoch sums, the next batch prefetched under the current one.
 * mmn_epoch_small_rows: the largest batch (rows) this model can run that way, 0 = not at all (MLPEncoder + ClassDecoder
 *   models with <= 8 encoders of <= 4 Linears whose image fits 160 KB of LDS; at most 64 rows).
 * mmn_train_epoch_small: `batches_host` / `batches_dev` = the same n_batches descriptors in host memory (validated here)
 *   and in device memory (read by the kernel; the caller keeps both alive until the stream has passed the launch).
 *   Every batch: default encoder sequence (seq_data[k] = seq_enc[k] = k), batch_global = batch, no tile tables, no
 *   dropout.  NaN batches skip their encoder as in mmn_train_step (decided inside the kernel: nan_flags is not read).
 *   `adam`: the optimizer state in the plan's flat layout (mmn_adam_fusable), updated in place, step counters included;
 *   adam->grads receives the LAST step's gradients.  Epoch sums accumulate onto mmn_epoch_read's doubles; the stats block
 *   holds the last step's values.  The chain kernels' weight copies are invalidated (mmn_pack_invalidate).
 *   The first call of a plan - and the first one after the largest batch of the descriptors changed - uploads the step's tile
 *   table (a few KB: hipMalloc + synchronous hipMemcpy, freed by mmn_plan_destroy): make that call outside a stream capture.
 * Errors: MMN_ERR_UNSUPPORTED (model / batch outside the scope above), MMN_ERR_ARG, MMN_ERR_HIP. */
int mmn_epoch_small_rows(mmn_plan* p);
int mmn_train_epoch_small(mmn_plan* p, const mmn_batch* batches_host, const mmn_batch* batches_dev, int n_batches,
                          float err_penalty, float state_change_penalty_x001, const struct mmn_adam* adam, void* stream);

size_t mmn_epoch_doubles(const mmn_model* m);
int mmn_epoch_reset(mmn_plan* p, void* stream);
int mmn_epoch_read(mmn_plan* p, double* out_host, void* stream);
/* the inverse (same layout; synchronises): carries an epoch's sums over to a new plan when the workspace had to grow */
int mmn_epoch_write(mmn_plan* p, const double* in_host, void* stream);

/* Device pointers (into the workspace) to what the last step left behind; the forward-only
 * consumers test() / predict() / get_states() (multimodn.py:255-492) read them, parity tests too.
 *   kind 0, index r in 1..E : state row r, [batch x S] (state after encoder r-1), written by every step
 *   kind 1, index r in 0..E : [batch x 2D]: after mmn_eval_step the decoder OUTPUTS sigmoid(z) of
 *                             grid row r (decoder d at columns 2d, 2d+1); after a training step dz
 *   kind 2, index e         : dS of encoder e [batch x S] (e = n_encoders -> dS0), training only
 *   kind 3                  : diagnostic timestamps (MMN_STAMPS=1)
 *   kind 4                  : int32[E+1] "state row exists" flags of the last step (row 0 always 1)
 *   kind 5                  : int32[rows] of the last mmn_regroup: source row of every position, -1 = padding
 *   kind 6 / 7, index = encoder * MMN_MAX_LAYERS + hidden layer : an MLPEncoder's hidden activations h_l [batch x H_l] /
 *                             d loss / d pre-activation of that layer (training only): what k_wgrad multiplies
 *   kind 8                  : the epoch accumulators as mmn_epoch_read copies them (mmn_epoch_doubles() doubles, device
 *                             memory): for callers that read them back asynchronously, behind the epoch's last launch
 * Rows of encoders that did not run hold stale data.  Returns NULL if out of range. */
const float* mmn_debug_buffer(mmn_plan* p, int kind, int index);

#ifdef __cplusplus
}
#endif
#endif /* MMN_HIP_H */
