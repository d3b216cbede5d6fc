// Internal host-side launchers (one per kernel family).  Everything takes raw device pointers;
// tensors whose element type follows the model dtype (KASF_F32 / KASF_BF16) are passed as void*.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KASF_F32 0
#define KASF_BF16 1

struct KasfPackDesc {
    int64_t src;        // element offset into the fp32 parameter buffer
    int64_t dst;        // element offset into the packed arena
    int64_t scale;      // element offset of a per-row scale vector in the parameter buffer, or -1
    int rows, cols;     // source is [rows][cols] row-major
    int transpose;      // 1: destination is [cols][rows]
    int fp16;           // 1 (bf16 arenas only): this copy is stored as IEEE fp16 (same element size): fc2.weight for the forward's fp16 GEMM2
};

// element offsets (fp32 parameter buffer) used by the prologue kernels
struct KasfProOff {
    int64_t mlp[51][4];     // [group*3 + channel] -> fc1.weight [16][n], fc1.bias [16], fc2.weight [1][16], fc2.bias [1]
    int64_t embed_w[3];     // joints_embed, bone_embed, limb_embed  weight [128][3]
    int64_t embed_b[3];     //                                        bias   [128]
    int64_t pos[3];         // pos_embed, bone_pos_embed, limb_pos_embed [17][128]
};

int kasf_set_error(int code, const char* msg);

// ---- k_reduce.hip: bit-reproducible per-channel reductions across workgroups ----
// A kernel that ends in "one value per channel per workgroup" stores row blockIdx.x of a scratch matrix (take) instead of adding atomically; add()
// registers the fixed-order sum of a column range of that matrix into a gradient vector; kasf_col_flush runs every registered sum in one launch
// (call it on a stream that is ordered after the producers) and resets the sink.  Launchers take `KasfColSink* sink`: nullptr = the old fp32
// atomics (single-operator entry points without scratch).
#define KASF_COLJOBS_MAX 96
struct KasfColJob {
    const float* part;      // [rows][ld]
    float* dst;             // mode 0: dst[c] += sum_r part[r][c];   mode 1 (fc2 finish): dst[c] = b[c] * sum, dst2[c] += a[c] * sum
    const float *a, *b;
    float* dst2;
    int rows, ncols, ld, mode;
};
struct KasfColSink {
    float* scratch = nullptr;       // device memory, `cap` floats
    int64_t cap = 0, used = 0;
    int njobs = 0;
    bool overflow = false;
    KasfColJob jobs[KASF_COLJOBS_MAX];
    float* take(int rows, int ld);  // rows x ld floats of scratch (nullptr + overflow flag when exhausted: the kernel then falls back to atomics)
    void add(const float* part, int ld, int rows, int ncols, float* dst, int mode = 0, const float* a = nullptr, const float* b = nullptr,
             float* dst2 = nullptr);
};
void kasf_col_flush(hipStream_t s, KasfColSink* const* sinks, int nsinks);

// ---- k_gemm.hip ----
void kasf_launch_linear(int dt, hipStream_t s, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                        int64_t M, int N, const float* ln_g, const float* ln_b, void* xn_out, int act);
void kasf_launch_linear_res(int dt, hipStream_t s, const void* A, const void* W, const float* bias, const float* ls, const void* resid, void* C,
                            int64_t M);
void kasf_launch_dgrad_lnbwd(int dt, hipStream_t s, const void* dY, int Kd, const void* Wt, const void* dxn_add, const void* X, const float* gamma,
                             const void* resid, void* out, int accumulate, float* dgamma, float* dbeta, int64_t M, void* xn_out = nullptr,
                             const float* beta = nullptr, KasfColSink* sink = nullptr);   // xn_out: also write LN(x) (operand of the matching weight gradient)
void kasf_launch_wgrad(int dt, hipStream_t s, const void* G, int64_t ldg, int N, const void* X, int64_t ldx, int K, const float* ln_g,
                       const float* ln_b, float* out, int64_t ldo, float* dbias, int64_t M, float* partial = nullptr, int64_t partial_floats = 0);
// partial: optional fp32 scratch (>= splits*N*K floats): per-split tiles are stored there and summed by a second kernel
// (deterministic); without it the kernel falls back to fp32 atomics on `out`.
void kasf_launch_pack(int dt, hipStream_t s, const float* params, void* arena, const KasfPackDesc* desc, const int* tile_start, int ndesc,
                      int total_tiles);

// ---- k_mlp.hip ----
// out = x + ls2 * (GELU(LN(x) W1^T + b1) W2^T + b2)
// xn_out (optional, bf16 path): also store LN(x) for kasf_launch_mlp_bwd_q
void kasf_launch_mlp_fwd(int dt, hipStream_t s, const void* x, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2,
                         const float* b2, const float* ls2, void* out, int64_t M, void* xn_out = nullptr);
// g_in = g + LNbwd(dA), also writes H = GELU(Z) and dZ ([M x 512] each) for the weight-gradient GEMMs
void kasf_launch_mlp_bwd(int dt, hipStream_t s, const void* x, const void* g, const float* ln_g, const float* ln_b, const void* W1, const float* b1,
                         const void* W2t_scaled, const void* W1t, void* Hbuf, void* dZbuf, void* xn_buf, void* g_in, float* dgamma, float* dbeta,
                         int64_t M, KasfColSink* sink = nullptr);

// ---- k_mlp2.hip / k_mlp3.hip (bf16): hidden-quarter MLP backward with fused weight gradients ----
// dApart: 4*M*128 bf16 scratch; partial: >= KASF_MLP_PARTIAL_FLOATS floats; xn = LN(x) as stored by the forward pass.
// Writes g_in = g + LNbwd(dA), dW1, db1, dgamma / dbeta, the fc2 weight gradient and gsum = colsum(g).
// W2 / b2 / ls2 / dls2 (fp32 masters) given: fc2 is finished as well (dls2, dW2 scaled by ls2, gsum slot = db2 = ls2 . colsum(g)) -- the colsum terms by
// the sink's k_col_finish, so they are complete only after kasf_col_flush.  sink == nullptr: per-channel sums by fp32 atomics (operator tests).
#define KASF_MLP_PARTIAL_FLOATS (2 * 64 * 65536)
void kasf_launch_mlp_bwd_q(hipStream_t s, const void* x, const void* xn, const void* g, const float* ln_g, const void* W1, const float* b1,
                           const void* W2ts, const void* W1t, void* dApart, float* partial, float* dW1, float* dW2, float* db1, float* gsum, void* g_in,
                           float* dgamma, float* dbeta, int64_t M, const float* W2 = nullptr, const float* b2 = nullptr, const float* ls2 = nullptr,
                           float* dls2 = nullptr, KasfColSink* sink = nullptr);
// BatchNorm batch-statistics buffers: KASF_STAT_SLOTS copies of [KASF_MAX_NODES][2] statistics of KASF_STAT_WORDS 64-bit words each (k_gcn.hip: an exact
// fixed-point accumulator fed by integer atomics; producers pick a copy by workgroup index)
#define KASF_STAT_SLOTS 4
#define KASF_STAT_WORDS 5
#define KASF_MAX_NODES 256                 // BatchNorm1d channels = joints (17) or frames: n_frames <= 256
#define KASF_STAT_LD (2 * KASF_MAX_NODES)
// 32-bit words per row of the stored temporal adjacency (bit c of word c >> 5 = "frame c is a neighbour")
inline int kasf_gcn_mask_words(int n_frames) { return n_frames <= 96 ? 3 : (n_frames + 31) / 32; }

// bf16 forward with all weights resident in registers (persistent workgroups, producer / consumer waves; k_mlp3.hip)
void kasf_launch_mlp_fwd_s(hipStream_t s, const void* x, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2,
                           const float* b2, const float* ls2, void* out, int64_t M, void* xn_out, unsigned grid);
void kasf_launch_mlp_bwd_s(hipStream_t s, const void* xn, const void* g, const void* W1, const float* b1, const void* W2ts, const void* W1t, void* dApart,
                           void* p1, void* p2, float* db1, float* db1_rows, int64_t M, int tiles_per_range, int used, bool dz_out = false);     // p1 / p2: bf16 partial tiles per token range; dz_out: dApart receives dZ [M][512] instead of the four dA partials

// ---- k_attn.hip ----
// mode 0: spatial (groups = B*T frames of 17 tokens), mode 1: temporal (groups = B*17 joint tracks of T tokens)
void kasf_launch_attn_fwd(int dt, hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int B, int T,
                          int mode, int heads = 8);
void kasf_launch_attn_bwd(int dt, hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq,
                          int64_t lddq, void* dk, void* dv, int64_t lddkv, int B, int T, int mode, int heads = 8);

// ---- k_attn_mfma.hip (bf16 only; return false when the shape is outside their range) ----
bool kasf_launch_attn_fwd_mfma(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int B, int T, int mode);
bool kasf_launch_attn_bwd_mfma(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq,
                               int64_t lddq, void* dk, void* dv, int64_t lddkv, int B, int T, int mode);
bool kasf_launch_attn_fwd_mfma32(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int B, int T, int mode);   // num_heads = 4
bool kasf_launch_attn_bwd_mfma32(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq, int64_t lddq,
                                 void* dk, void* dv, int64_t lddkv, int B, int T, int mode);

// ---- k_gcn.hip ----
void kasf_gcn_init();   // uploads the skeleton table to constant memory (blocking; call once per process before capture)
void kasf_launch_gcn_agg_fwd(int dt, hipStream_t s, const void* uv, const void* xn, void* y, uint32_t* mask, double* stats, int B, int T, int mode, int kth = 4);
// BatchNorm finalisation (batch sums or running statistics -> per-node affine `coef`, running-statistics update) is part of the apply kernel
void kasf_launch_gcn_apply(int dt, hipStream_t s, const void* x_in, const void* xn, const void* y, const double* stats, const float* bn_w, const float* bn_b,
                           float* run_mean, float* run_var, float* coef, const float* ls1, void* out, int B, int T, int mode, double count, int training,
                           float momentum);
void kasf_launch_gcn_bwd1(int dt, hipStream_t s, const void* g, const void* xn, const void* y, const float* coef, const float* ls1, void* r,
                          float* dls1, double* bstats, int B, int T, int mode, KasfColSink* sink = nullptr);
// BatchNorm-backward finalisation (means of the backward sums, d(bn weight / bias)) is part of bwd2
void kasf_launch_gcn_bwd2(int dt, hipStream_t s, const void* r, const void* y, const float* coef, const uint32_t* mask, void* duv, int B, int T,
                          int mode, const double* bstats, float* d_bn_w, float* d_bn_b, double count, int training = 1);
// training == 0: backward of an evaluation-mode BatchNorm (running statistics are constants: no batch-mean terms)

// ---- k_misc.hip ----
void kasf_launch_prologue_fwd(int dt, hipStream_t s, const float* x, const float* params, const KasfProOff* off, void* xj, void* xb, void* xl,
                              float* bone3, float* limb3, int64_t frames);
void kasf_launch_embed_bwd(int dt, hipStream_t s, const void* g, const float* in3, const float* W, float* dW, float* db, float* dpos, float* din3,
                           int64_t frames, KasfColSink* sink = nullptr);
// grad_base / grad_len: the contiguous range of the flat gradient array that holds the 204 limb-MLP tensors (one scratch row mirrors it)
void kasf_launch_refusion_bwd(hipStream_t s, const float* x, const float* dlimb3, const float* params, float* grads, const KasfProOff* off,
                              int64_t frames, KasfColSink* sink = nullptr, int64_t grad_base = 0, int grad_len = 0);
void kasf_launch_gate_fwd(int dt, hipStream_t s, const void* xa, const void* xg, const void* xb, const float* W, const float* b, void* out,
                          float* alpha, int64_t M, int adaptive);
void kasf_launch_gate_bwd(int dt, hipStream_t s, const void* g, const void* g1, const void* g2, const void* xa, const void* xg, const void* xb,
                          const float* W, const float* alpha, void* ga, void* gg, void* gb, float* dW, float* db, int64_t M, int adaptive,
                          KasfColSink* sink = nullptr);   // g1/g2: optional extra addends of the incoming gradient
void kasf_launch_head_fwd(int dt, hipStream_t s, const void* rep, const float* W, const float* b, float* out, int64_t M);
void kasf_launch_head_bwd(int dt, hipStream_t s, const float* dy, const void* rep, const float* W, void* dpre, float* dW, float* db, int64_t M,
                          KasfColSink* sink = nullptr);
// return_rep=True backward: dpre = drep * (1 - rep^2), drep [M,512] fp32
void kasf_launch_rep_bwd(int dt, hipStream_t s, const float* drep, const void* rep, void* dpre, int64_t M);
void kasf_launch_cast_to_f32(int dt, hipStream_t s, const void* src, float* dst, int64_t n);
void kasf_launch_cast_from_f32(int dt, hipStream_t s, const float* src, void* dst, int64_t n);
void kasf_launch_add_inplace(int dt, hipStream_t s, void* dst, const void* a, int64_t n);   // dst += a
void kasf_launch_add3(int dt, hipStream_t s, void* dst, const void* a, const void* b, const void* c, int64_t n);   // dst = a + b (+ c)
// in: dW = unscaled G = g^T A, db = colsum(g).  out: dls[n] = sum_k W[n][k] G[n][k] + bias[n] db[n];  dW *= ls[n];  db *= ls[n]
void kasf_launch_finalize_ls(hipStream_t s, float* dW, const float* W, const float* bias, const float* ls, float* db, float* dls, int N, int K);
void kasf_launch_loss3(hipStream_t s, const float* pred, const float* tgt, float* dpred, float* losses, int B, int T, float lambda_n, float lambda_v,
                       float grad_scale);
void kasf_launch_adamw(hipStream_t s, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
                       float bc1, float bc2, float grad_scale);

// ---- k_eval.hip ----
void kasf_launch_joint_flip(hipStream_t s, const float* src, float* dst, int64_t rows);
void kasf_launch_tta_merge(hipStream_t s, const float* p, const float* pf, float* out, int64_t rows);
void kasf_launch_eval_metrics(hipStream_t s, const float* pred, const float* label, const float* factor, const float* res, const int* action, int B, int T,
                              int n_actions, float* mpjpe, float* pmpjpe, float* acc, float* jpe, double* action_sums);
void kasf_launch_gather_clips(hipStream_t s, const float* xa, const float* ya, const int64_t* index, const unsigned char* flip, int64_t n_clips, int B,
                              int T, float* xo, float* yo);

// ---- k_gemm2.hip (bf16, persistent, register-resident weights) ----
// bf16 partial tiles a fused data + weight gradient launch left: out[e] += sum over z < nparts of part[z][e], e < elems (elems a multiple of 128)
// ---- persistent launches narrower than the chip (round 4) ----
// The MLP kernels keep their weights in registers: one workgroup owns a whole CU, so a launch that spans all 256 CUs runs ALONE -- the other two branch streams wait, and its
// fill (weights into registers, a three-deep pipeline) and drain are paid with nothing beside them.  Inside the engine's forward / backward these launches therefore take HALF the
// chip below 150,000 tokens: two branches' MLPs run side by side, one's fill under the other's steady state (tools/smallb_probe.py, T = 27 training: +3.1 % step throughput at
// B = 256, +8 % at 128, +9 % at 32; evaluation +12 % at B = 32 ... 0 at 512; T = 81, B = 128 = 176,256 tokens: -0.4 %, hence the threshold; a third of the chip each is
// worse: the branches are not equally long).  The LDS-ring linear kernels do the same below 80,000 tokens (more tiles per workgroup per fill), the data-gradient kernels
// below 150,000 since round 5 (tools/width_sweep.sh, B = 256: 4,617-4,658 clips/s at 50 %, 4,633-4,649 at 66 %, 4,560-4,642 at the full grid on the same boxes; in round 4 it cost 1.5 % there).  The widths are a function of the token count only -- never of the stream mode -- so results stay bit-identical between the one-stream
// and three-stream engines; the operator entry points (tests, bench.py's hot loop) launch at full width.
// Round 6: rounds 4-5 stopped at 150,000 tokens (T = 81, B = 128 = 176,256 tokens measured -0.4 % with the round-4 kernels).  Re-measured after this round's kernel changes, MLP
// forward / backward / data gradient at half width against the full grid: T = 81 B = 128: 1,514-1,518 against 1,456-1,486 clips/s; T = 27 B = 384: 4,754 / 4,758 against 4,614 / 4,622;
// T = 27 B = 512: 4,860 / 4,840 against 4,730 / 4,740; T = 81 B = 256 (352,512 tokens): 1,566 / 1,565 against 1,551 / 1,548 -- half the chip at every size (the linear class keeps 80,000).
// KASF_NARROW_PCTS = "fwd,bwd,dgrad,linear,attn_fwd,attn_bwd,wgrad" (percent of the full grid) and KASF_NARROW_BELOW = tokens override the table: measurement knobs.
#include <cstdlib>
#include <cstdint>
constexpr int64_t KASF_HALF_CHIP_ALWAYS = int64_t(1) << 40;
enum { KASF_NG_MLP_FWD = 0, KASF_NG_MLP_BWD = 1, KASF_NG_DGRAD = 2, KASF_NG_LINEAR = 3, KASF_NG_ATTN_FWD = 4, KASF_NG_ATTN_BWD = 5, KASF_NG_WGRAD = 6 };
inline thread_local int kasf_tls_model_path = 0;        // while the engine is enqueueing (engine.hip): 1 a training step's forward / backward, 2 a forward-only (evaluation) pass
inline int kasf_narrow_grid(int cls, int full, int64_t tokens) {
    static int pcts[7] = {-1, 0, 0, 0, 0, 0, 0};
    static int64_t below[7] = {KASF_HALF_CHIP_ALWAYS, KASF_HALF_CHIP_ALWAYS, KASF_HALF_CHIP_ALWAYS, 80000, 0, 0, 0};      // (round 5: the data-gradient kernels too below 150,000 tokens: +0.7 % at B = 256 with this round's kernels; -1.5 % in round 4)
    if (pcts[0] < 0) {
        const int def[7] = {50, 50, 50, 50, 100, 100, 100};
        int tmp[7];
        for (int k = 0; k < 7; ++k) tmp[k] = def[k];
        if (const char* e = getenv("KASF_NARROW_PCTS")) { int k = 0; while (*e && k < 7) { tmp[k++] = atoi(e); while (*e && *e != ',') ++e; if (*e == ',') ++e; } }
        if (const char* e = getenv("KASF_NARROW_BELOW")) for (int k = 0; k < 7; ++k) below[k] = atoll(e);
        for (int k = 6; k >= 0; --k) pcts[k] = tmp[k];   // pcts[0] last: the table is complete when another thread sees it set
    }
    if (!kasf_tls_model_path || tokens >= (kasf_tls_model_path == 2 && below[cls] == KASF_HALF_CHIP_ALWAYS ? int64_t(150000) : below[cls])) return full;     // forward-only passes keep the
                                                                       // 150,000-token threshold: B = 2,048 evaluation 19,322 / 19,365 clips/s at half width against 19,818 / 19,668 at the full grid, B = 512 even
    const int g = full * pcts[cls] / 100;
    return g < 1 ? 1 : g;
}

// second launch of the bf16 MLP backward in its dZ form (k_gemm2.hip; fin_args: the MlpFinArgs of mlp_fin.h)
void kasf_launch_mlp_dgrad_fin(hipStream_t s, const void* dZ, const void* W1t, const void* X, const float* gamma, const void* g, void* g_in, float* dgamma, float* dbeta,
                               float* gsum, int64_t M, KasfColSink* sink, const void* fin_args, const float* b2, const float* ls2, float* dls2, bool have_w2);
struct KasfBf16Reduce { const void* part; float* out; int nparts; int elems; };
bool kasf_dgrad_wg_supported(int Kd, bool resid, bool accumulate, bool dxn_add, bool dbias, bool proj, int64_t M, int64_t wpart_bytes);
int kasf_launch_dgrad_wg(hipStream_t s, const void* dY, int Kd, const void* Wt, const void* X, const float* gamma, const float* beta, const void* resid, void* out,
                         int accumulate, float* dgamma, float* dbeta, int64_t M, KasfColSink* sink, void* wpart, int64_t wpart_bytes, const void* dxn_add = nullptr,
                         float* dbias = nullptr, const void* proj_o = nullptr, void* proj_part = nullptr, float* proj_brow = nullptr);
// proj_o (bone q linear, Kd = 128): the block's output-projection gradient G = resid^T . proj_o rides along: <= 256 bf16 tiles of 128 x 128 in proj_part and one
// fp32 row of colsum(resid) per tile in proj_brow, in the layout kasf_launch_proj_finish adds up
void kasf_launch_proj_finish(hipStream_t s, const void* proj_part, const float* proj_brow, int nparts, float* dW, const float* W, const float* bias, const float* ls,
                             float* db, float* dls, int nred, const KasfBf16Reduce* red);
void kasf_launch_bf16_reduce(hipStream_t s, int nred, const KasfBf16Reduce* red);      // the fixed-order sum of such partial tiles on its own (no streaming jobs in the block)
bool kasf_launch_dgrad_r(hipStream_t s, const void* dY, int Kd, const void* Wt, const void* dxn_add, const void* X, const float* gamma, const void* resid,
                         void* out, int accumulate, float* dgamma, float* dbeta, int64_t M, void* xn_out, const float* beta, KasfColSink* sink = nullptr);
bool kasf_launch_linear_r(hipStream_t s, const void* A, const void* W, const float* bias, void* C, int64_t M, int N, const float* ln_g, const float* ln_b,
                          void* xn_out);
void kasf_launch_linear_res_r(hipStream_t s, const void* A, const void* W, const float* bias, const float* ls, const void* resid, void* C, int64_t M);

// ---- k_attn_blk.hip (bf16): LN + QKV + 8-head attention + proj + layer-scale + residual of one attention block, forward ----
// q_save / kv_save / o_save: what the backward pass reads (nullptr in evaluation: nothing but x_mid is written).  false: shape not covered.
bool kasf_launch_attn_block_fwd(hipStream_t s, int bone, const void* x, const void* x_limb, const float* ln_g, const float* ln_b, const float* lnl_g,
                                const float* lnl_b, const void* Wq, const void* Wkv, const void* Wproj, const float* bproj, const float* ls1, void* q_save,
                                void* kv_save, void* o_save, void* out, int B, int T, int mode, float* lse_save = nullptr);   // lse_save [M][8]: groups of 33..96 positions in training

// ---- k_attn_bwd_f.hip (bf16, 8 heads, groups of <= 32 positions): the backward of an attention / bone block from x (, x_limb) and g_mid in one launch ----
// Writes the input gradient(s), dq | dk | dv and LN(x) (LN_limb(x_limb)) for the streaming weight-gradient launch, and accumulates G_proj = g_mid^T o.
// Returns the number of per-workgroup G_proj tiles written (0: nothing launched, nothing registered in the sink).  dq: [M][384] (self) / [M][128] (bone), dkv: bone [M][256];
// ppart: 256 x [128][128] bf16; pbrow: 256 x 128 floats (colsum g_mid): the ext_* operands of kasf_launch_wgrad_jobs.
int kasf_launch_attn_block_bwd(hipStream_t s, int bone, const void* x, const void* x_limb, const void* g_mid, const float* ln_g, const float* ln_b, const float* lnl_g,
                               const float* lnl_b, const void* Wf, const void* Wkvf, const void* WT, const void* WprojTs, void* out, void* out_limb, void* dq, void* dkv,
                               void* xn_a, void* xn_b, float* dgamma, float* dbeta, float* dgamma_l, float* dbeta_l, KasfColSink* sink, void* ppart, float* pbrow,
                               int B, int T, int mode);

// ---- k_gemm.hip: several bf16 weight gradients dW_j[N_j][128] += G_j^T X_j in one streaming launch + one finishing launch ----
bool kasf_launch_wgrad_jobs(hipStream_t s, int njobs, const void* const* G, const void* const* X, const int* N, float* const* dW, float* const* dbias,
                            int fin_job, const float* fin_W, const float* fin_bias, const float* fin_ls, float* fin_dls, int64_t M, float* partial,
                            int64_t partial_floats, int nred = 0, const KasfBf16Reduce* red = nullptr, const void* ext_part = nullptr, const float* ext_brow = nullptr,
                            int ext_nparts = 0, float* ext_dW = nullptr, float* ext_db = nullptr);
bool kasf_launch_attn_bwd_fused_do(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* g_mid,
                                   const void* WprojTs, void* dq, int64_t lddq, void* dk, void* dv, int64_t lddkv, int B, int Tn, int mode, int form = 0 /* 0: persistent, 1: one group per workgroup (bit-equal comparison form for the tests) */,
                                   const void* o_saved = nullptr, const float* lse = nullptr);   // o_saved [M][128] + lse [M][8] (both from the forward): groups of 33..96 positions take the key-tile-outer kernel
