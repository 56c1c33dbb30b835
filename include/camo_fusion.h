/*
 * camo_fusion.h -- C ABI of the MI355X-native fusion hot path.
 *
 * The reference (rajan-dubey8/camouflage-multimodal) is pure Python and has no
 * FFI of its own; its boundary for this path is a Python operator API.  Each
 * entry point below states the reference interface it stands behind (paths
 * relative to the reference root).  Host code (the .py files of camouflage_multimodal_amd)
 * keeps that Python API and reaches these functions through ctypes with raw
 * device pointers; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its comment says "host";
 *    the caller (PyTorch-ROCm, or any HIP program) owns every buffer;
 *  - all tensors are float32, row-major, contiguous; a batch is PACKED:
 *    the RG rows of sample b are rows rg_offsets[b] .. rg_offsets[b+1]-1 of
 *    one [T, rg_dim] matrix (variable Nr per sample, no padding), the KG rows
 *    are [B*Nk, kg_dim];
 *  - calls only ENQUEUE work on `stream` (a hipStream_t passed as void*); no
 *    allocation, no synchronisation, and the environment is never read.  The only
 *    process-global state is the two testing/measurement hooks at the end of this
 *    file (camo_options_t, camo_prof_begin/end); without them the calls are
 *    graph-capturable and thread-compatible;
 *  - return value: 0 on success, a negative CAMO_E_* code otherwise; no C++
 *    exception crosses the boundary; camo_last_error() gives a thread-local
 *    message for the last failing call.
 */
#ifndef CAMO_FUSION_H
#define CAMO_FUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAMO_ABI_VERSION 10

enum {
  CAMO_OK = 0,
  CAMO_E_ARG = -1,         /* bad argument (null pointer, non-positive size ...)        */
  CAMO_E_UNSUPPORTED = -2, /* configuration outside the kernels' envelope               */
  CAMO_E_WORKSPACE = -3,   /* workspace too small                                        */
  CAMO_E_HIP = -4          /* a HIP launch failed (message has hipGetErrorString)        */
};

enum { CAMO_FUSION_CROSS_ATTENTION = 0, CAMO_FUSION_LATE = 1 };
enum { CAMO_PREC_F32 = 0,  /* f32-input MFMA: bit-faithful fp32 FMA chains            */
       CAMO_PREC_BF16 = 1  /* bf16 MFMA operands, fp32 accumulate/activations         */ };

/* Model dimensions = the keys build_multimodal_model() reads
 * (models/multimodal/fusion_model.py:249-259). */
/* Schedule options of ONE caller (engine): which of the library's kernel schedules its calls take where several compute the same
 * function.  Caller-owned, reached through camo_dims_t::options -- the library keeps no option state of its own, so two engines in
 * one process (or two threads) cannot change each other's schedule.  Product callers leave every field at its default
 * (camo_options_init; options == NULL means the same): the schedule then follows from the call's arguments alone.
 * Tests and developer tools set fields by name (camo_options_set) to run two schedules of the same call in one process:
 *   sched16  -1 choose from the configuration, 0 never take the bf16-resident schedule
 *   fused    -1 choose from the configuration, 0 never take the fused row-tile schedule
 *   tail17   -1 the per-sample tail runs as one launch where the fused schedule does (B <= 16; groups up to 48), 0 never
 *   fused_rt -1 wide row tiles by size, 0 never, 1 / 2 / 4 that many 32-row tiles per 8-wave block (csrc/fused_wide.hip)
 *   wide2    -1 the 64-row half-block forward by size (csrc/fused_wide2.hip), 0 never, 1 whenever the shape allows
 *   wide2_bwd  -1 the backward's node-level halves of the RG rows on 64-row blocks by size (csrc/bwd_wide2.hip), 0 never, 1 always,
 *            2 always, with the 32-row second half behind the 64-row first half
 *   fused_one, wide_front_rt, tailw, tailw_bwd, param_space, tn_big, fused_variant, back_lead, tn_balance, tn_kcap, tn_exp, exp: developer A/Bs
 *   fused_save  1 makes inference calls of the fused schedule also write the tensors a backward would read (names for
 *            camo_debug_ws_offset: R16 G16 Q16 Q2_16 KV16 KV2_16 O16 O2_16 Y16 Y2_16 XH16 XH2_16 rstd1 rstd2 mask1 mask2 lse2 X16
 *            Wqkv_rg W1s Ymean H1mean Y2mean H2mean)
 *   tail_skip_arrival  block id + 1 of the NEXT one-launch tail that skips its first arrival (the give-up path's test); one shot:
 *            the call that consumes it writes 0 back. */
typedef struct camo_options {
  int32_t sched16, fused, tail17, fused_rt, wide2, fused_one, wide_front_rt, tailw, tailw_bwd, param_space, tn_big, fused_variant,
          back_lead, tn_balance, tn_kcap, tn_exp, exp, fused_save, tail_skip_arrival, wide2_bwd;
} camo_options_t;

typedef struct camo_dims {
  int32_t rg_dim, kg_dim, hidden_dim, num_heads, num_classes;
  int32_t fusion_type; /* CAMO_FUSION_* */
  float dropout;
  camo_options_t* options;   /* NULL: defaults.  Caller-owned; read at every call that takes these dims */
} camo_dims_t;

/* Parameter table: an array of CAMO_NPARAMS_* device pointers in the order of the
 * reference module's state_dict (fusion_model.py:21-73, :151-162, :208-235).
 * Slots of layers the configuration does not have (rg_proj/kg_proj when the
 * input dim equals hidden_dim -> nn.Identity, fusion_model.py:29-30) are NULL.
 * Gradient tables use the same indices. */
enum {
  CAMO_P_RG_PROJ_W = 0, CAMO_P_RG_PROJ_B, CAMO_P_KG_PROJ_W, CAMO_P_KG_PROJ_B,
  CAMO_P_A1_IN_W, CAMO_P_A1_IN_B, CAMO_P_A1_OUT_W, CAMO_P_A1_OUT_B, /* cross_attn_rg2kg */
  CAMO_P_A2_IN_W, CAMO_P_A2_IN_B, CAMO_P_A2_OUT_W, CAMO_P_A2_OUT_B, /* cross_attn_kg2rg */
  CAMO_P_LN1_W, CAMO_P_LN1_B, CAMO_P_LN2_W, CAMO_P_LN2_B,           /* ln_rg, ln_kg      */
  CAMO_P_F1_W0, CAMO_P_F1_B0, CAMO_P_F1_W3, CAMO_P_F1_B3,           /* ffn_rg.{0,3}      */
  CAMO_P_F2_W0, CAMO_P_F2_B0, CAMO_P_F2_W3, CAMO_P_F2_B3,           /* ffn_kg.{0,3}      */
  CAMO_P_FU_W0, CAMO_P_FU_B0, CAMO_P_FU_W3, CAMO_P_FU_B3,           /* fusion_layer      */
  CAMO_P_HEADS = 28, /* 4 heads x {0.weight,0.bias,3.weight,3.bias}: mask, instance, edge, score */
  CAMO_NPARAMS_CROSS = 44
};
enum {
  CAMO_PL_W0 = 0, CAMO_PL_B0, CAMO_PL_W3, CAMO_PL_B3, CAMO_PL_W6, CAMO_PL_B6, /* LateFusion.fusion.{0,3,6} */
  CAMO_PL_HEADS = 6,
  CAMO_NPARAMS_LATE = 22
};

/* ---- sizes ------------------------------------------------------------- */
int camo_abi_version(void);
const char* camo_last_error(void);

/* Bytes of workspace camo_forward/camo_backward need for a packed batch of B
 * samples, T RG rows in total, Nk KG rows per sample.  The same workspace must
 * be handed, unmodified, from camo_forward to the camo_backward of that batch:
 * it holds the saved activations.  Returns 0 and sets the error on bad input. */
size_t camo_workspace_bytes(const camo_dims_t* dims, int32_t B, int32_t T, int32_t Nk);

/* ---- forward -------------------------------------------------------------
 * Stands behind MultimodalCamouflageDetector.forward (fusion_model.py:237-246)
 * = CrossAttentionFusion.forward (:75-146) or LateFusion.forward (:164-171)
 * followed by the four heads, on inputs already normalised to 3-D
 * (:86-105, done by the host).
 *   rg          [T, rg_dim]           packed RG node embeddings
 *   rg_offsets  int32 [B+1]           device; rg_offsets[0]=0, rg_offsets[B]=T, every Nr_b >= 1
 *   batch_desc  opaque device buffer of camo_batch_desc_bytes(B, T) bytes filled by camo_prepare_batch() from
 *               rg_offsets: row -> sample map, 1/Nr_b, first 32-row tile of every sample.  It depends on the
 *               shape tuple (Nr_0 .. Nr_B-1) only, so a caller builds it once per distinct tuple.
 *   kg          [B*Nk, kg_dim]
 *   max_nr      host value: max_b Nr_b (grid sizing only)
 *   outs        [B, 2*num_classes+2]  = mask logits | instance logits | edge logit | sigmoid(score)
 *   attn_rg2kg  [T, Nk] or NULL       head-averaged attention, row t = RG node t
 *   attn_kg2rg  [T, Nk] or NULL       TRANSPOSED head-averaged map: element (t, j) = weight of
 *                                     KG query j on RG key t (host transposes per sample to [Nk, Nr])
 *   training    0: eval (no dropout); 1: train, dropout p = dims->dropout with the
 *               counter-based mask of (seed, site, element index)
 *   flags       CAMO_FWD_* bits.  CAMO_FWD_INFERENCE: no camo_backward will follow this call (validation,
 *               prediction): nothing is saved for it and the workspace contents are undefined afterwards.
 */
#define CAMO_FWD_INFERENCE 1
#define CAMO_FLAG_ATTN_MAPS 2   /* camo_backward only: the forward call of this workspace was given attention-map pointers */
size_t camo_batch_desc_bytes(int32_t B, int32_t T);
int camo_prepare_batch(const int32_t* rg_offsets, int32_t B, int32_t T, int32_t max_nr,
                       void* batch_desc, size_t batch_desc_bytes, void* stream);

/* ---- minibatch out of a device-resident dataset ---------------------------
 * Stands behind one DataLoader step of the reference's training loop (train_multimodal.py:385-395: WeightedRandomSampler
 * indices -> collate) plus SmartMultimodalDataset.__getitem__'s augmentation (:173-175: with probability 1/2 per sample,
 * N(0, noise_std^2) noise on both streams; the reference uses 0.01), for a dataset that lives in HBM: ONE launch, no host copy.
 *   rg_all [rows, rg_dim], sample_offsets int64 [N+1] (first row of every sample), kg_all [N, kg_floats], y/e/s_all [N]: the dataset;
 *   idx int64 [B] (device): the minibatch's samples; T = sum of their row counts (the host knows the counts: output sizing).
 * Writes the packed rows rg_out [T, rg_dim], kg_out [B, kg_floats], the packed row offsets offsets_out int32 [B+1] (what
 * camo_prepare_batch takes) and the gathered labels.  noise_std = 0: no augmentation.  The noise is a counter hash of
 * (seed, element): like the reference's unseeded draws it is not reproducible against torch's generator, only in distribution.
 * B <= 4096. */
int camo_gather_batch(const float* rg_all, const int64_t* sample_offsets, const float* kg_all,
                      const int64_t* y_all, const float* e_all, const float* s_all,
                      const int64_t* idx, int32_t B, int32_t T, int32_t rg_dim, int32_t kg_floats,
                      float* rg_out, float* kg_out, int32_t* offsets_out, int64_t* y_out, float* e_out, float* s_out,
                      float noise_std, uint64_t seed, void* stream);

int camo_forward(const camo_dims_t* dims, const float* const* params,
                 const float* rg, const int32_t* rg_offsets, const void* batch_desc,
                 const float* kg, int32_t B, int32_t T, int32_t Nk, int32_t max_nr,
                 void* workspace, size_t workspace_bytes,
                 float* outs, float* attn_rg2kg, float* attn_kg2rg,
                 int32_t training, uint64_t seed, int32_t precision, int32_t flags, void* stream);

/* camo_forward_cached: camo_forward for callers that run many forward calls between parameter changes -- validation and
 * prediction loops (train_multimodal.py:304-342 validate_fixed, test_multimodal.py:105-150): the fused schedule's bf16 weight
 * shadows live in the caller's persistent buffer (camo_shadow_bytes() bytes, 256-byte aligned) instead of the per-batch
 * workspace, and a call with shadows_valid != 0 -- the caller's promise that the buffer holds the current parameters, i.e.
 * that nothing wrote them since the call that reported the buffer filled -- skips the shadow launch (3 launches instead of 4
 * per inference call).  shadows_valid == 2: valid, but left by camo_clip_adamw_shadows, whose set lacks the one piece only
 * inference calls build -- the RG rows' folded in-projection Wf = [Wq; Wk'; Wv'] Wrg (csrc/fused_wide2.hip) -- which this call
 * then builds alone (one small launch); afterwards the caller may pass 1.  *shadows_state (may be null) reports what the buffer holds after the call: 0 = untouched (the call
 * took a schedule without shadows: Nk > 16, attention maps, f32 ...; a promise is then simply not used), 1 = the forward
 * shadows (an inference call built them, or used valid ones); 2 is reported by camo_forward_loss_backward only (forward and
 * transposed shadows).  With a shadow buffer the call must be an inference call (flags contain CAMO_FWD_INFERENCE), else
 * CAMO_E_UNSUPPORTED: camo_backward takes no shadow argument, so a saving call's transposed shadows would be out of its reach --
 * the training pair is camo_forward_loss_backward.  shadows == NULL: camo_forward. */
int camo_forward_cached(const camo_dims_t* dims, const float* const* params,
                        const float* rg, const int32_t* rg_offsets, const void* batch_desc,
                        const float* kg, int32_t B, int32_t T, int32_t Nk, int32_t max_nr,
                        void* workspace, size_t workspace_bytes,
                        float* outs, float* attn_rg2kg, float* attn_kg2rg,
                        int32_t training, uint64_t seed, int32_t precision, int32_t flags,
                        void* shadows, int32_t shadows_valid, int32_t* shadows_state, void* stream);

/* ---- backward ------------------------------------------------------------
 * Stands behind loss.backward() through the model (train_multimodal.py:270):
 * given d(loss)/d(outs) it ACCUMULATES (+=) parameter gradients into `grads`
 * (the reference sums gradients over the samples of a minibatch,
 * train_multimodal.py:239-279).  Inputs carry no gradient in the reference, so
 * none is produced.  Arguments as camo_forward; `workspace` is the one that
 * call filled.  d_outs is d(loss)/d(outs); with d_outs_pre_activation = 1 its last
 * column is instead taken w.r.t. the score head's pre-sigmoid value (what camo_loss's
 * d_pre output holds), which saves the conversion pass.  flags: CAMO_FLAG_ATTN_MAPS when the forward call
 * that filled the workspace returned attention maps (it then ran a schedule that materialises them). */
int camo_backward(const camo_dims_t* dims, const float* const* params, float* const* grads,
                  const float* rg, const int32_t* rg_offsets, const void* batch_desc,
                  const float* kg, int32_t B, int32_t T, int32_t Nk, int32_t max_nr,
                  void* workspace, size_t workspace_bytes,
                  const float* outs, const float* d_outs, int32_t d_outs_pre_activation,
                  int32_t training, uint64_t seed, int32_t precision, int32_t flags, void* stream);

/* ---- loss ----------------------------------------------------------------
 * Per-sample loss of train_multimodal.py:256-268 --
 *   3.0*AggressiveFocalLoss(alpha .75, gamma 3)(mask, y)  (:29-57)
 * + 1.0*cross_entropy(instance, y) + 0.5*BCEWithLogits(edge, e) + 0.3*MSE(score, s)
 * each evaluated at batch size 1 as the reference loop does, so the batch loss is
 * the SUM over samples.  Writes loss_terms [B,4] (already weighted), d_outs
 * [B, 2C+2] = d(sum of losses)/d(outs), optionally d_pre (same, but the score column taken
 * w.r.t. the pre-sigmoid value) and pred [B] = argmax(mask logits) (:273).
 *   y int64 [B]; e, s float [B]; d_outs, d_pre, pred may each be NULL. */
int camo_loss(const float* outs, const int64_t* y, const float* e, const float* s,
              int32_t B, int32_t num_classes,
              float* loss_terms, float* d_outs, float* d_pre, int32_t* pred, void* stream);

/* ---- optimizer -----------------------------------------------------------
 * clip_grad_norm_(params, 1.0) + AdamW.step (train_multimodal.py:278-279,
 * :403-407) over flat buffers of n floats.
 * `sumsq` is a caller-owned buffer of CAMO_SUMSQ_FLOATS floats.
 * camo_grad_sumsq: per-block partial sums of g*g into sumsq[1..] (plain stores: no memset, no
 *   atomics, deterministic).  A data-parallel caller all-reduces g (SUM) BEFORE this call -- see ddp.py.
 * camo_clip_adamw: norm = sqrt(sum of the partials) (also written to sumsq[0] for the host);
 *   coef = min(1, max_norm/(norm+1e-6)); g <- g*coef in place, as the reference leaves clipped
 *   grads behind -- or g <- 0 when zero_grads != 0 (the next minibatch's zero_grad() fused in);
 *   decoupled weight decay; Adam moments; bias correction with `step` (1-based). */
#define CAMO_SUMSQ_FLOATS 257
int camo_grad_sumsq(const float* g, size_t n, float* sumsq, void* stream);
int camo_clip_adamw(float* p, float* g, float* m, float* v, size_t n, float* sumsq,
                    float max_norm, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int32_t step, int32_t zero_grads, void* stream);

/* The same update that ALSO leaves the fused schedule's bf16 weight shadows of the updated parameters in `shadows`
 * (camo_shadow_bytes(dims) bytes; 0 = the configuration has no fused schedule): the shadowed matrices are walked in
 * 32 x 64 tiles whose new values go out as fragment-order chunks straight from registers (plain) and through an LDS
 * transpose (the backward's transposed copies); everything else in [p, p + n) is updated elementwise as above.
 * params: the parameter table of camo_forward (pointers into [p, p + n)); bit-identical parameters, moments and
 * gradients to camo_clip_adamw.  Pair with camo_forward_loss_backward(..., shadows, shadows_valid = 1). */
size_t camo_shadow_bytes(const camo_dims_t* dims);
int camo_clip_adamw_shadows(const camo_dims_t* dims, const float* const* params, float* p, float* g, float* m, float* v,
                            size_t n, float* sumsq, float max_norm, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int32_t step, int32_t zero_grads, void* shadows, void* stream);

/* camo_forward_loss_backward: the native training call = camo_forward (training / seed / precision as there),
 * camo_loss on its outputs with the labels y [B] (int64), e [B], s [B], and camo_backward on the loss gradient:
 * same results, fewer launches (the head output layer, the loss and that layer's backward run as one kernel).
 * Writes outs [B, 2C+2], loss_terms [B, 4], pred [B] (may be null) and ACCUMULATES into grads like
 * camo_backward.  Stands behind the body of the per-minibatch loop of train_epoch_fixed
 * (train_multimodal.py:245-270) for one packed minibatch.
 * tail_event (hipEvent_t, may be null): recorded on `stream` as soon as the gradients of the per-sample tail are final
 *   -- parameters CAMO_P_F2_W3 .. the end of the table (pooled KG FFN layer, fusion layer, the four heads; one
 *   contiguous run when the gradients live in one flat buffer in table order) and CAMO_P_F1_W3/B3 -- i.e. before the
 *   node-level backward launches.  A data-parallel caller starts the all-reduce of that run behind the event and the
 *   rest behind the call (ddp.py: BucketedGradAllReducer).  Every schedule records it (at the end when it has no
 *   earlier point: late fusion).
 * shadows (may be null; camo_shadow_bytes() bytes, 256-byte aligned, caller-owned, persistent across steps): where the
 *   fused schedule keeps its bf16 weight shadows instead of the per-batch workspace.  shadows_valid != 0 is the caller's
 *   PROMISE that they hold the current parameters -- i.e. that the last writer of the parameters was
 *   camo_clip_adamw_shadows on this buffer -- and lets the call skip rebuilding them (one launch less per step).  With
 *   shadows_valid == 0 the call rebuilds them there.  Both are ignored on calls that do not take the fused schedule. */
int camo_forward_loss_backward(const camo_dims_t* dims, const float* const* params, float* const* grads, const float* rg,
                               const int32_t* rg_offsets, const void* batch_desc, const float* kg,
                               int32_t B, int32_t T, int32_t Nk, int32_t max_nr, void* workspace, size_t workspace_bytes,
                               const int64_t* y, const float* e, const float* s, float* outs, float* loss_terms, int32_t* pred,
                               int32_t training, uint64_t seed, int32_t precision, void* tail_event, void* shadows,
                               int32_t shadows_valid, void* stream);

/* ---- testing hooks ---------------------------------------------------------
 * Not part of the operator surface; used by tests/ to check kernels in isolation.
 * camo_debug_gemm: one problem of the grouped GEMM,
 *   C[M,N] (+)= epi(A.B + bias) (+res), flags = GF_* bits of csrc/gemm.h
 *   (1 relu, 4 atomic accumulate, 64 A k-major, 128 B k-major).
 * camo_debug_ws_offset: byte offset of a named saved activation inside the
 *   workspace (names: R G Q KV2 KV Q2 P P2 O O2 U U2 Y Y2 H1 H2 comb fused, and the fused schedule's: see below), or -1. */
int camo_debug_gemm(const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                    const float* bias, const float* res, int32_t ldr, float* bias_grad,
                    int32_t M, int32_t N, int32_t K, int32_t flags, int32_t precision, void* stream);
/* camo_debug_gemm16: one problem of the bf16-resident grouped GEMM (csrc/gemm16.h): A16/B16/C16 are bf16
 * bit patterns; flags 64|128 together = dW += A^T.B (rows of A/B up to round_up(K,128) must be readable). */
int camo_debug_gemm16(const void* A16, int32_t lda, const void* B16, int32_t ldb, float* C, int32_t ldc,
                      void* C16, int32_t ldc16, const float* bias, const float* res, int32_t ldr, float* bias_grad,
                      int32_t M, int32_t N, int32_t K, int32_t flags, void* stream);
int64_t camo_debug_ws_offset(const camo_dims_t* dims, int32_t B, int32_t T, int32_t Nk, const char* name);
/* camo_options_init: every field to its default.  camo_options_set: one field by name (CAMO_E_ARG for an unknown name). */
int camo_options_init(camo_options_t* options);
int camo_options_set(camo_options_t* options, const char* name, int32_t value);
/* camo_debug_set_stamps: developer timeline of the fused kernels.  buf = device buffer of 2 * blocks_per_kernel * 8 uint64
 * (or NULL to switch it off): wave 0 of every block stores the 100 MHz wall clock at its phase boundaries. */
int camo_debug_set_stamps(void* buf, int32_t blocks_per_kernel);

/* Opt-in launch timing for bench.py's roofline leg: between camo_prof_begin and camo_prof_end every
 * launch of the path's kernels (the grouped GEMMs, the fused row-tile kernels, the tail GEMMs, weight shadows) is bracketed
 * by two HIP events recorded on the launch stream.  camo_prof_end synchronises on them and returns
 * the summed kernel time, the number of launches and the FLOPs those launches executed.  This is the
 * one piece of process-global state in the library: single-threaded use, not for production loops. */
int camo_prof_begin(int32_t max_launches);
int camo_prof_end(double* gemm_ms, int32_t* gemm_launches, double* gemm_flops);
/* After camo_prof_end: the same three figures per kernel family.  kind: 0 grouped GEMMs (weight gradients in the fused
 * schedule), 1 fused forward front half, 2 fused forward back half, 3 fused backward first half, 4 second half,
 * 5 per-sample tail GEMMs, 6 optimizer, 7 weight shadows + clears, 8 attention kernels of the unfused schedules, 9 other.
 * (camo_prof_end's own outputs are the totals over all kinds.) */
int camo_prof_kind(int32_t kind, double* ms, int32_t* launches, double* flops);

/* Health check of the one-launch per-sample tail (the only kernel of the library whose blocks wait for each other: three
 * all-reduces among its 64 co-resident blocks, each wait bounded).  *count = number of waits that gave up since the
 * library was loaded; 0 in any run with one process per GPU.  A non-zero count means a step produced wrong results (the
 * GPU was shared with another process that kept the launch's blocks from being co-resident).  SYNCHRONOUS (reads a device
 * counter): call it between epochs, not inside a step.  train_multimodal.fit does, and raises. */
int camo_tail_timeouts(uint32_t* count);

/* Data-parallel companion of the rule above (no reference counterpart; the reference is single-process).  A rank whose tail gave
 * up holds garbage gradients that the SUM all-reduce (ddp.py; SURVEY 8e "sum, then clip", train_multimodal.py:238-279) would add
 * into every rank's buffer, and only that rank's norm would turn NaN.  Enqueue this on the launch stream between
 * camo_forward_loss_backward and the all-reduce of the piece that holds element 0 of the flat gradient buffer: if a timeout is
 * pending on this device, flat_grads[0] becomes NaN, the SUM carries it to every rank, every rank's camo_grad_sumsq yields a NaN
 * norm and every rank's optimizer call skips the same step -- replicas stay bit-identical.  One thread, enqueue-only. */
int camo_tail_poison_to_grads(float* flat_grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAMO_FUSION_H */
