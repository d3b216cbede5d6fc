/* libkasf_hip -- C-ABI of the MI355X-native (gfx950) KASportsFormer forward/backward path.
 *
 * The reference (jw0r1n/KASportsFormer) is pure Python/PyTorch: its "FFI" for this path is the
 * nn.Module call  KASportsFormer.forward(x[B,T,17,3]) -> [B,T,17,3]  plus autograd.  This header is
 * the boundary a reference maintainer binds instead (ctypes stub in INTEGRATION.md): plain pointers
 * and sizes, no torch types.  All pointers are DEVICE pointers owned by the caller (torch's caching
 * allocator in the shipped host code); `stream` is a hipStream_t passed as void*.  Nothing here
 * allocates, frees or synchronises in the hot path (graph-capture safe); only kasf_model_create()
 * allocates a few KB of device tables.
 *
 * Every function returns 0 on success or a non-zero code; kasf_last_error() describes it.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   kasf_model_create      <- KASportsFormer.__init__             model/KASportsFormer.py:291-318
 *   kasf_param_*           <- nn.Module.state_dict() layout       model/KASportsFormer.py:296-318 (names identical)
 *   kasf_forward           <- KASportsFormer.forward              model/KASportsFormer.py:320-347
 *   kasf_backward          <- torch.autograd of the above         train_and_evaluate_sp.py:241  (loss.backward())
 *   kasf_loss3             <- mpjpe + 0.5 n_mpjpe + 20 velocity   utils/loss_calc.py:6-27, train_and_evaluate_sp.py:212-222
 *   kasf_adamw_step        <- optim.AdamW(...).step()             train_and_evaluate_sp.py:270-272,243
 *   kasf_gather_clips      <- Dataset.__getitem__ + DataLoader collate data/reader/sp_dataset.py:45-92
 *   kasf_joint_flip        <- joint_flip                            utils/utilities.py:128-135
 *   kasf_tta_merge         <- flip-TTA average + root zeroing        train_and_evaluate_sp.py:46-55
 *   kasf_eval_metrics      <- de-normalise + MPJPE/JPE/accel/P-MPJPE train_and_evaluate_sp.py:57-93, utils/error_calc.py:5-48
 *   kasf_op_*              <- the individual nn.Modules under model/modules/ (unit-test entry points)
 */
#ifndef KASF_H_
#define KASF_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KASF_DTYPE_F32 0   /* parity mode: fp32 storage, exact-f32 MFMA (v_mfma_f32_16x16x4_f32) */
#define KASF_DTYPE_BF16 1  /* fast mode: bf16 activations/weights, fp32 accumulate/statistics/master weights */

#define KASF_FLAG_TRAIN 1       /* BatchNorm batch statistics + running-stat update; keep activations for backward */
#define KASF_FLAG_RETURN_REP 2  /* out is the [B,T,17,512] tanh representation (forward(x, return_rep=True)) */
#define KASF_FLAG_KEEP 4        /* keep activations for kasf_backward WITHOUT batch statistics: autograd through model.eval() (BatchNorm on running stats) */

typedef struct kasf_model kasf_model;

typedef struct kasf_config {
    int32_t n_layers;            /* 26 in the shipped yaml (configs/sportspose-gt-kasportsformer.yaml:71) */
    int32_t n_frames;            /* T in [4, 256]: sizes the temporal BatchNorm1d; 9 / 27 / 81 have tuned temporal kernels, T <= 96 MFMA attention cores */
    int32_t num_heads;           /* 8 in every yaml (:84; head dim 16: MFMA attention kernels); 2 / 4 / 16 run generic kernels (4 = the constructor's default; 2: n_frames <= 157) */
    int32_t neighbour_num;       /* top-k of the temporal GCN adjacency (graph.py:104-112), 1..4; every yaml uses 4 (:89); other values are rejected */
    int32_t use_adaptive_fusion; /* 1: softmax gate, 0: plain mean (KASportsFormer.py:284) */
    int32_t dtype;               /* KASF_DTYPE_* */
} kasf_config;

const char* kasf_last_error(void);
/* Process-wide: 1 = the three branches of a layer run back to back on the caller's stream instead of on three streams (the mode isolated kernel
 * profiles are taken in; it costs a quarter of the training throughput since round 4: the MLP launches of the engine take half the chip -- so that two
 * branches' launches run side by side -- and the grids are the same in both settings, for the sake of the bits).  NOT a determinism switch: gradients are reproducible from run to run either way and
 * the two settings give the same bits (every gradient reduction is a fixed-order sum; the BatchNorm batch statistics, which cross workgroups through
 * atomics, are accumulated EXACTLY -- 52-bit pieces of a fixed-point number, 64-bit integer atomic adds, csrc/k_gcn.hip -- so their totals do not depend on
 * arrival order either: since ABI 7 there is no exception left).  Default 0 (or 1 when KASF_SINGLE_STREAM is set in the environment).
 * kasf_forward / kasf_backward read the setting once per call.  kasf_set_deterministic / kasf_get_deterministic are the round-3 names of the same
 * two functions, kept as aliases. */
void kasf_set_single_stream(int32_t on);
int32_t kasf_get_single_stream(void);
/* Process-wide: from how many tokens per launch (M = batch x frames x 17) the bf16 backward forms the mixers' weight gradients INSIDE the data-gradient kernels
 * (one bf16 partial tile per workgroup, fixed-order reduce) instead of in a separate streaming launch.  Default 40,000 (below it the partial tiles cost more
 * than the second pass over dY they save); tokens < 0 restores the default.  Both forms are bit-reproducible; they differ from each other at bf16 rounding
 * level (tests/test_gpu_determinism.py compares them on one shape). */
void kasf_set_fused_wgrad_min_tokens(int64_t tokens);
int64_t kasf_get_fused_wgrad_min_tokens(void);
/* Process-wide (ABI 8), OPT-IN: 1 (default 0; 1 when KASF_ATTN_BWD_FUSED=1 is in the environment) = in bf16 mode with 8 heads the backward of an attention / bone block whose
 * groups have at most 32 positions (every spatial block; temporal blocks up to n_frames = 32) is ONE launch that re-forms LN(x), q | k | v and the attention output from the
 * block input (csrc/k_attn_bwd_f.hip; reference: modules/selfattention.py:18-57, modules/bone_crossattention.py:19-62, KASportsFormer.py:103-110) followed by one streaming
 * weight-gradient launch: the training forward then saves no q | k | v | o for these blocks and kasf_workspace_bytes shrinks accordingly (T = 27, B = 256: 26.9 -> 14.4 GB).
 * 0 = the four-launch sequence (saved q | k | v | o, attention cores, data gradient + LayerNorm backward with the fused weight gradient, proj weight gradient): 15 % FASTER per
 * training step on MI355X, hence the default.  Both are bit-reproducible; they agree with each other at bf16 rounding level
 * (tests/test_gpu_determinism.py).  kasf_workspace_bytes, kasf_forward and kasf_backward of one step must see the SAME setting (the workspace layout and what the forward saves
 * depend on it): change it only between steps, then re-query kasf_workspace_bytes.  on < 0 restores the default.  `on` is a bit mask of block kinds -- 1 self-attention
 * spatial, 2 self-attention temporal, 4 bone spatial, 8 bone temporal; on = 1 means all four (15), which is also what kasf_get_fused_attn_bwd then returns.  Measured in the
 * three-stream step (B = 256, T = 27): mask 2 runs at the default's speed (4,625-4,646 against 4,635-4,660 clips/s) with 7 GB less HBM traffic per step; 3: -2 %; 10: -5 %. */
void kasf_set_fused_attn_bwd(int32_t on);
int32_t kasf_get_fused_attn_bwd(void);
void kasf_set_deterministic(int32_t on);
int32_t kasf_get_deterministic(void);
int kasf_version(void);

/* Streams: every model forks its attention / graph / bone branches onto ONE process-wide pair of side streams per device (created with the first model of
 * the device, released with the process; device indices 0..63, error 2 beyond).  Events order every fork and join, so any number of models may share the
 * pair; models of one device driven from different host threads therefore serialise on it, and a stream capture of kasf_forward has to be the only work
 * in flight on that device while it is recorded. */
int kasf_model_create(const kasf_config* cfg, kasf_model** out);
/* the same handle without touching a device: answers every layout / size query below (kasf_param_*, kasf_buffer_*, kasf_workspace_bytes, kasf_stage_grad_range ...);
 * what a host-side binding uses to build its module tree before any GPU exists (kasportsformer_amd/model.py does) */
int kasf_model_create_layout_only(const kasf_config* cfg, kasf_model** out);
void kasf_model_destroy(kasf_model* m);

/* 0 = healthy.  Non-zero: a kernel's bounded wait on another workgroup ran out (the affected gradient rows were poisoned with NaN instead of
 * hanging the GPU); blocking device read -- for tests and post-mortems, not for the step loop. */
int kasf_model_status(const kasf_model* m, int32_t* status);

/* ---- parameter / buffer layout: one flat fp32 array each; entries carry the reference's state_dict names ---- */
int64_t kasf_param_count(const kasf_model* m);       /* elements of the flat parameter (and gradient) array, multiple of 4 */
int64_t kasf_param_live_count(const kasf_model* m);  /* [0, live) receive gradients; [live, count) are the never-used norm1_limb */
int32_t kasf_param_entries(const kasf_model* m);
int kasf_param_entry(const kasf_model* m, int32_t idx, char* name, int32_t name_cap, int64_t* offset, int32_t* ndim, int64_t shape[4]);
int64_t kasf_buffer_count(const kasf_model* m);      /* BatchNorm running_mean / running_var, fp32 */
int32_t kasf_buffer_entries(const kasf_model* m);
int kasf_buffer_entry(const kasf_model* m, int32_t idx, char* name, int32_t name_cap, int64_t* offset, int32_t* ndim, int64_t shape[4]);
/* gradient ranges that become final after each backward stage (data-parallel all-reduce buckets) */
int32_t kasf_backward_stages(const kasf_model* m);   /* n_layers + 2: head, layers (last to first), prologue */
int kasf_stage_grad_range(const kasf_model* m, int32_t stage, int64_t* begin, int64_t* end);

/* ---- kernel-side weight arena (dtype of the model; transposed / layer-scale-folded copies) ---- */
int64_t kasf_packed_bytes(const kasf_model* m);
int kasf_pack_weights(const kasf_model* m, const float* params, void* packed, void* stream);

/* ---- workspace (activations kept for backward + scratch); caller allocates, 256-B aligned ---- */
int64_t kasf_workspace_bytes(const kasf_model* m, int32_t batch, int32_t flags);

/* x [B,T,17,3] fp32 -> out [B,T,17,3] fp32 (or [B,T,17,512] with KASF_FLAG_RETURN_REP).
 * `buffers` (BN running stats) is updated when KASF_FLAG_TRAIN is set.  x and out must not alias.
 * batch >= 1 and batch * n_frames * 17 * 384 < 2^31 (the kernels address activations with 32-bit element offsets): error 2 otherwise. */
int kasf_forward(const kasf_model* m, const float* params, const void* packed, float* buffers, const float* x, float* out, void* workspace,
                 int64_t workspace_bytes, int32_t batch, int32_t flags, void* stream);

/* Gradients of a preceding kasf_forward(KASF_FLAG_TRAIN or KASF_FLAG_KEEP) on the same workspace; `flags` = the forward's flags:
 * KASF_FLAG_TRAIN selects the batch-statistics BatchNorm backward (otherwise running statistics are constants), KASF_FLAG_RETURN_REP means
 * dout is the gradient of the [B,T,17,512] representation (otherwise dout [B,T,17,3] fp32).
 * Accumulates (+=) into grads[0, live); the caller zeroes it.  Stages [stage_begin, stage_end) of
 * kasf_backward_stages() are run; call with (0, stages) for the whole backward, or stage by stage to
 * overlap the gradient all-reduce of finished ranges. */
int kasf_backward(const kasf_model* m, const float* params, const void* packed, const float* dout, float* grads, void* workspace,
                  int64_t workspace_bytes, int32_t batch, int32_t flags, int32_t stage_begin, int32_t stage_end, void* stream);

/* losses: `losses_floats` >= 4 + 4 * batch floats (error 5 otherwise: the capacity is an argument since ABI 6, so that a caller written against the
 * 4-float buffer of ABI <= 4 fails to compile instead of being overrun); on return losses[0..3] = {total, mpjpe, n_mpjpe, velocity} (the rest is
 * scratch: per-clip sums, added in a fixed order so that the result is bit-reproducible); dpred = grad_scale * dTotal/dpred */
int kasf_loss3(const float* pred, const float* target, float* dpred, float* losses, int64_t losses_floats, int32_t batch, int32_t n_frames,
               float lambda_n_mpjpe, float lambda_velocity, float grad_scale, void* stream);

/* torch.optim.AdamW step over n contiguous fp32 elements (n multiple of 4); step_index starts at 1 */
int kasf_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int32_t step_index, float grad_scale, void* stream);

/* ---- batch assembly from a clip set resident in device memory: x_all/y_all [n_clips,T,17,3] fp32 (y_all may be NULL) ----
 * x_out[b] = x_all[index[b]] (same for y), left/right flipped (flip_data, sp_dataset.py:36-40) where flip[b] != 0 (flip may be NULL).
 * An index outside [0,n_clips) yields a zero clip. */
int kasf_gather_clips(const float* x_all, const float* y_all, const int64_t* index, const uint8_t* flip, int64_t n_clips, int32_t batch,
                      int32_t n_frames, float* x_out, float* y_out, void* stream);

/* ---- evaluation side: fp32 [rows = B*T][17][3] poses ---- */
/* dst = joint_flip(src): x negated, left joints [1,2,3,14,15,16] swapped with right [4,5,6,11,12,13]; src != dst */
int kasf_joint_flip(const float* src, float* dst, int64_t rows, void* stream);
/* out = (pred + joint_flip(pred_of_flipped)) / 2 with the root joint zeroed; pred_of_flipped == NULL: root zeroing only */
int kasf_tta_merge(const float* pred, const float* pred_of_flipped, float* out, int64_t rows, void* stream);
/* The reference's per-clip evaluation loop in one launch.  pred [B,T,17,3] (normalised model output, root zeroed here again),
 * label_scaled [B,T,17,3] (mm), factor [B,T], res [B,2] = (w,h) as floats, action [B] ids in [0,n_actions) (or NULL with action_sums NULL).
 * Per-frame outputs: mpjpe [B,T], p_mpjpe [B,T], accel [B,T-2], jpe [B,T,17].  action_sums [n_actions][KASF_EVAL_COLS] fp64 is
 * ACCUMULATED into: columns = sum mpjpe, sum p_mpjpe, sum accel, sum jpe[17], #frames, #accel frames. */
#define KASF_EVAL_COLS 22
int kasf_eval_metrics(const float* pred, const float* label_scaled, const float* factor, const float* res, const int32_t* action, int32_t batch,
                      int32_t n_frames, int32_t n_actions, float* mpjpe, float* p_mpjpe, float* accel, float* jpe, double* action_sums, void* stream);

/* debugging / tests: locate a named activation inside the workspace (see kasf_ws_name()) */
int32_t kasf_ws_entries(const kasf_model* m, int32_t batch, int32_t flags);
int kasf_ws_entry(const kasf_model* m, int32_t batch, int32_t flags, int32_t idx, char* name, int32_t name_cap, int64_t* byte_offset,
                  int64_t* numel, int32_t* elem_kind /* 0: model dtype, 1: fp32, 2: fp64, 3: u32 */);

/* ---- single-operator entry points (tensors of the model dtype are void*; M = tokens) ---- */
/* y = act(LN?(a) W^T + bias): a [M,128], w [N,128] (dtype), N % 128 == 0; ln_g == NULL -> no LayerNorm; act 0 none, 1 tanh */
int kasf_op_linear(int32_t dtype, const void* a, const void* w, const float* bias, void* y, int64_t M, int32_t N, const float* ln_g, const float* ln_b,
                   void* xn_out, int32_t act, void* stream);
/* modules/mlp.py inside a FormerModule: out = x + ls2 * (GELU(LN(x) W1^T + b1) W2^T + b2) */
/* xn_out (optional; bf16 only): also receives LN(x), which kasf_op_mlp_bwd_fused streams instead of recomputing (what training mode does).
 * ABI 7: with dtype = bf16, w2 [128,512] is IEEE FP16 (torch.float16), not bf16: the forward evaluates GELU in packed fp16 and keeps the hidden
 * activation in fp16 for GEMM2 (v_mfma_f32_16x16x32_f16); x, w1 and out stay bf16.  kasf_pack_weights writes that fp16 copy into the arena itself.
 * RANGE (bf16 mode only): the pre-activation z = LN(x) W1^T + b1 and the fc2 weights pass through fp16, whose largest finite value is 65504: |z| > 65504
 * converts to +-inf (GELU(+inf) = inf, GELU(-inf) = -inf . 0 = NaN where the fp32 form returns x and 0), and a fc2 weight beyond 65504 packs to inf.  LayerNorm'd
 * inputs with trained weights are five orders of magnitude below that; a checkpoint that is not should be run with dtype = fp32 (exact erf GELU, no fp16 anywhere). */
int kasf_op_mlp_fwd(int32_t dtype, const void* x, const float* ln_g, const float* ln_b, const void* w1, const float* b1, const void* w2, const float* b2,
                    const float* ls2, void* out, int64_t M, void* xn_out, void* stream);
int kasf_op_mlp_bwd(int32_t dtype, const void* x, const void* g, const float* ln_g, const float* ln_b, const void* w1, const float* b1,
                    const void* w2t_scaled, const void* w1t, void* hbuf, void* dzbuf, void* g_in, float* dgamma, float* dbeta, int64_t M, void* stream);
/* bf16 only: fused MLP backward (hidden-quarter ownership, weights in registers): data gradient AND both weight gradients.
 * g_in = g + LNbwd(dA); dw1 [512,128] += dZ^T LN(x); db1 [512] += colsum(dZ); dw2_unscaled [128,512] += g^T H;
 * gsum [128] += colsum(g); dgamma/dbeta += LayerNorm parameter gradients.  dapart: 4*M*128 elements of scratch (bf16: the four hidden quarters' dA partials;
 * with KASF_MLP_BWD_DZ=1 in the environment, dZ [M,512] in the same bytes);
 * partial: >= 2*64*65536 + 2048 floats of scratch (the tail holds the inter-workgroup hand-off flags; word 2*64*65536 + 1024 is set to 1 if a
 * bounded wait ran out). */
int kasf_op_mlp_bwd_fused(const void* x, const void* xn /* LN(x) from kasf_op_mlp_fwd */, const void* g, const float* ln_g, const void* w1, const float* b1, const void* w2t_scaled,
                          const void* w1t, void* dapart, float* partial, float* dw1, float* dw2_unscaled, float* db1, float* gsum, void* g_in,
                          float* dgamma, float* dbeta, int64_t M, void* stream);
/* dW[N,K] += G^T LN?(X), dbias[N] += colsum(G): G [M,N], X [M,K].  partial: optional fp32 scratch of partial_floats
 * elements (>= 256*128*128 covers every shape of this model): per-split tiles are stored and summed by a second
 * kernel (bitwise reproducible); NULL -> fp32 atomics on dw. */
int kasf_op_wgrad(int32_t dtype, const void* g, int32_t N, const void* x, int32_t K, const float* ln_g, const float* ln_b, float* dw, float* dbias,
                  int64_t M, float* partial, int64_t partial_floats, void* stream);
/* out = [resid] + [out, if accumulate] + LNbwd(dY Wt^T [+ dxn_add]): dY [M,Kd], Wt [128,Kd]; dgamma/dbeta are accumulated into;
 * xn_out (optional, with beta) receives LN(x), the operand of the matching weight-gradient GEMM */
int kasf_op_dgrad_lnbwd(int32_t dtype, const void* dy, int32_t Kd, const void* wt, const void* dxn_add, const void* x, const float* gamma,
                        const void* resid, void* out, int32_t accumulate, float* dgamma, float* dbeta, int64_t M, void* xn_out, const float* beta,
                        void* stream);
/* attention core (selfattention.py:18-41): q [*,ldq], k/v [*,ldkv] token-major; mode 0 spatial, 1 temporal */
int kasf_op_attention_fwd(int32_t dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int32_t batch, int32_t n_frames,
                          int32_t mode, void* stream);
int kasf_op_attention_bwd(int32_t dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq, int64_t lddq,
                          void* dk, void* dv, int64_t lddkv, int32_t batch, int32_t n_frames, int32_t mode, void* stream);
/* the same for num_heads in {2, 4, 8, 16} (head dimension 128 / num_heads; the two entries above are num_heads = 8): bf16 mode runs MFMA cores for 8 and 4 heads
 * (4 = the reference constructor's default, KASportsFormer.py:293) on groups of up to 256 positions, LDS-resident fp32 cores otherwise */
int kasf_op_attention_fwd_heads(int32_t dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int32_t batch, int32_t n_frames,
                                int32_t mode, int32_t num_heads, void* stream);
int kasf_op_attention_bwd_heads(int32_t dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq, int64_t lddq,
                                void* dk, void* dv, int64_t lddkv, int32_t batch, int32_t n_frames, int32_t mode, int32_t num_heads, void* stream);
/* bf16, 8 heads, groups of <= 96 positions: the same with d_o = g_mid . wproj_t_scaled^T formed inside the kernel (what the training step runs:
 * attention.py's proj + layer-scale data gradient folded in); wproj_t_scaled [128 in][128 out] = (ls1 . Wproj)^T packed bf16.
 * Groups of <= 32 positions: form 0 = persistent kernel (the engine's), 1 = one group per workgroup (the comparison point): bit-identical results.
 * Groups of 33..96 positions (temporal attention at T = 81): with o_saved [M,128] (the attention output) and lse [M,8] fp32 (log-sum-exp of the scaled
 * scores per token and head), both as the training forward leaves them, the key-tile-outer kernel runs; with either NULL the self-contained one. */
int kasf_op_attention_bwd_fused_do(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* g_mid, const void* wproj_t_scaled,
                                   void* dq, int64_t lddq, void* dk, void* dv, int64_t lddkv, int32_t batch, int32_t n_frames, int32_t mode, int32_t form,
                                   const void* o_saved, const float* lse, void* stream);
/* fp32 <-> model dtype */
int kasf_op_cast(int32_t dtype, const void* src, void* dst, int64_t n, int32_t to_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KASF_H_ */
