// Host-side engine of libkasf_hip: parameter layout (reference state_dict names), packed-weight table,
// workspace plan and the forward / backward launch sequences of the KASportsFormer path
// (reference: model/KASportsFormer.py:204-347).  Pure launch code: no allocation, no synchronisation.
#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "kasf.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;

struct Entry { std::string name; int64_t off; int ndim; int64_t shape[4]; };

enum { KIND_ATT = 0, KIND_GRAPH = 1, KIND_BONE = 2 };
const char* const BLOCK_NAMES[6] = {"att_spatial", "att_temporal", "graph_spatial", "graph_temporal", "bone_spatial", "bone_temporal"};
const int BLOCK_KIND[6] = {KIND_ATT, KIND_ATT, KIND_GRAPH, KIND_GRAPH, KIND_BONE, KIND_BONE};
// modules/bone_refusion.py:34-40 -- sizes of the 17 joint groups
const int LIMB_N[17] = {3, 3, 2, 2, 3, 3, 4, 4, 4, 4, 3, 4, 4, 4, 4, 2, 2};

struct BlockOff {
    int kind, mode, nodes;
    int64_t ls1, ls2, n1w, n1b, n1lw, n1lb, n2w, n2b, fc1w, fc1b, fc2w, fc2b;
    int64_t mix_w, kv_w, proj_w, proj_b, uv_b, bn_w, bn_b;   // mix_w: qkv.weight | qkv_q.weight | U.weight(+V.weight)
    int64_t bn_rm, bn_rv;                                     // offsets into the buffer array
    int64_t p_mix, p_mixT, p_kv, p_kvT, p_proj, p_projTs, p_fc1, p_fc1T, p_fc2, p_fc2Ts;   // packed arena (elements)
};
struct LayerOff { BlockOff blk[6]; int64_t fus_w, fus_b, begin, end; };
struct TopOff { int64_t norm_w, norm_b, fc_w, fc_b, head_w, head_b, p_fc, p_fcT, begin, end; };

// The three branches of a layer run on three streams unless this is set: kasf_set_single_stream(1) (or KASF_SINGLE_STREAM=1 in the environment, read once)
// runs them back to back on the caller's stream: same results bit for bit, the mode the isolated per-kernel profiles are taken in; -23 % throughput since round 4
// (3,389 against 4,428 clips/s: the half-chip MLP grids depend on the token count only, so in this mode they run one at a time on half the chip).
// kasf_forward / kasf_backward read the flag ONCE per call into a local (a toggle from another thread between the fork and the join of a layer would
// otherwise skip one of the two event waits).
static std::atomic<int> g_single_stream{-1};
static bool single_stream() {
    int v = g_single_stream.load(std::memory_order_relaxed);
    if (v < 0) { v = getenv("KASF_SINGLE_STREAM") != nullptr ? 1 : 0; g_single_stream.store(v, std::memory_order_relaxed); }
    return v != 0;
}
// The fused data + weight gradient launches leave one partial dW tile per WORKGROUP whatever the batch: a fixed 25-50 MB of partial traffic per block that only
// pays for itself once the token stream is long enough (measured: T = 27, B = 256 -3 % per step; B = 32 +3 %): below this many tokens the two-kernel sequence runs.
#ifndef KASF_WG_FUSE_MIN_TOKENS
#define KASF_WG_FUSE_MIN_TOKENS 40000
#endif
static std::atomic<int64_t> g_wg_fuse_min_tokens{KASF_WG_FUSE_MIN_TOKENS};      // kasf_set_fused_wgrad_min_tokens(): tests compare the fused with the two-kernel sequence on one shape
#define WG_FUSE_MIN_TOKENS (g_wg_fuse_min_tokens.load(std::memory_order_relaxed))
// Round 6, OPT-IN (kasf_set_fused_attn_bwd(1) or KASF_ATTN_BWD_FUSED=1 in the environment, read once): the backward of an attention / bone block with groups of <= 32 positions
// (bf16, 8 heads) as ONE launch that re-forms q | k | v and the attention output from x (csrc/k_attn_bwd_f.hip) + one streaming weight-gradient launch: the forward then saves
// nothing for these blocks and the workspace has no qkv / kv / o slots (T = 27, B = 256: 26.9 -> 14.4 GB).  Parity-green, bit-reproducible -- and SLOWER than the four-launch
// sequence of rounds 1-5, which therefore stays the default (3,926 against 4,620 clips/s; the phase timers say why).  Forward and backward of one
// step must run under the same setting.
static std::atomic<int> g_fused_attn_bwd{-1};            // bit mask: 1 self-attention spatial, 2 self-attention temporal, 4 bone spatial, 8 bone temporal (the setter's 1 = all four = 15)
static int fused_attn_bwd_mask() {
    int v = g_fused_attn_bwd.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = getenv("KASF_ATTN_BWD_FUSED"); v = e != nullptr ? atoi(e) : 0; if (v == 1) v = 15; if (v < 0 || v > 15) v = 0; g_fused_attn_bwd.store(v, std::memory_order_relaxed); }
    return v;
}
static bool fused_attn_bwd(const kasf_config& cfg, int kind, int mode) {
    const int Lg = mode == 0 ? 17 : cfg.n_frames;
    if (kind == 1 /* KIND_GRAPH */ || cfg.dtype != KASF_BF16 || cfg.num_heads != 8 || Lg > 32) return false;
    return (fused_attn_bwd_mask() >> ((kind == 2 /* KIND_BONE */ ? 2 : 0) + (mode ? 1 : 0))) & 1;
}
constexpr int64_t WG_JOBS_FLOATS = 248 * 128 * 128 + 248 * 128 + 4096;   // the proj job alone: 248 splits of one 128 x 128 tile + their bias rows
constexpr int64_t WG_BF16_BYTES = (int64_t)256 * 384 * 128 * 2 + (int64_t)256 * 128 * 128 * 2 + 256 * 128 * 4;         // <= 256 bf16 partial tiles of the block's fused data + weight gradient launches (qkv: 384 rows; q + kv: 128 + 256)
constexpr int64_t WG_PARTIAL_FLOATS = (KASF_MLP_PARTIAL_FLOATS + 65536 > WG_JOBS_FLOATS + WG_BF16_BYTES / 4 ? KASF_MLP_PARTIAL_FLOATS + 65536 : WG_JOBS_FLOATS + WG_BF16_BYTES / 4);   // per-split weight-gradient tiles (256 workgroups x 128x128, or 64 ranges x (dW1 + dW2) of the MLP) + the per-split rows of a bias gradient

struct BlkWs { int64_t qkv, kv, o, xn, y, mask, x_mid, xn2, x_out, stats, bstats, coef, lse; };
struct LayerWs { BlkWs b[6]; int64_t gate_out, alpha; };
struct WsEntry { std::string name; int64_t off, numel; int kind; };
struct Scratch { int64_t uv, t1, t2, hbuf, dzbuf, d_o, dqkv, rbuf, duv, xn_a, xn_b, wg_part, g_in, col; };   // per-branch (att / graph / bone); col: per-workgroup rows of the column reductions (k_reduce.hip)
struct Plan {
    int64_t total = 0, stats_begin = 0, stats_bytes = 0, bstats_begin = 0, bstats_bytes = 0;
    int64_t x3, bone3, limb3, xj, xb, xl, rep, uv;
    std::vector<LayerWs> layers;
    int64_t g_layer, g_prev, ga, gg, gb, g_limb, g_bone, dlimb3;
    int64_t col_cap = 0;     // floats per Scratch::col
    Scratch sc[3];
    std::vector<WsEntry> entries;
};

}  // namespace

struct kasf_model {
    kasf_config cfg;
    std::vector<Entry> params, buffers;
    std::vector<LayerOff> layers;
    TopOff top;
    KasfProOff pro;
    int64_t n_params = 0, n_live = 0, n_buffers = 0, arena_elems = 0;
    std::vector<KasfPackDesc> pack;
    std::vector<int> pack_tile_start;
    int pack_tiles = 0;
    // device tables
    KasfPackDesc* d_pack = nullptr;
    int* d_tile_start = nullptr;
    KasfProOff* d_pro = nullptr;
    // the three branches of a layer (attention / graph / bone) are independent: branch 0 stays on the caller's stream,
    // branches 1 and 2 fork onto these and join before the gate (pure event fork/join: graph-capture safe)
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    unsigned* d_err = nullptr;       // set by a kernel whose bounded inter-workgroup wait ran out (kasf_model_status)
};

int kasf_set_error(int code, const char* msg) {
    g_err = msg ? msg : "";
    return code;
}

namespace {

#define HIPCHK(x)                                                                           \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) return kasf_set_error(1000 + (int)e_, hipGetErrorString(e_)); \
    } while (0)

struct Alloc {
    std::vector<Entry>* out;
    int64_t cur = 0;
    int64_t add(const std::string& name, std::initializer_list<int64_t> shape) {
        cur = (cur + 3) & ~int64_t(3);
        Entry e;
        e.name = name;
        e.off = cur;
        e.ndim = (int)shape.size();
        int64_t n = 1;
        int i = 0;
        for (int64_t s : shape) { e.shape[i++] = s; n *= s; }
        for (; i < 4; ++i) e.shape[i] = 1;
        out->push_back(e);
        cur += n;
        return e.off;
    }
};

void build_layout(kasf_model* m) {
    const int L = m->cfg.n_layers, T = m->cfg.n_frames;
    Alloc A{&m->params};
    Alloc B{&m->buffers};
    int64_t arena = 0;
    auto packd = [&](int64_t src, int rows, int cols, int64_t scale, int transpose, int fp16 = 0) {
        KasfPackDesc d{src, arena, scale, rows, cols, transpose, fp16};
        m->pack.push_back(d);
        const int64_t at = arena;
        arena += (int64_t)rows * cols;
        return at;
    };
    m->layers.resize(L);
    for (int l = 0; l < L; ++l) {
        LayerOff& lo = m->layers[l];
        lo.begin = (A.cur + 3) & ~int64_t(3);
        for (int b = 0; b < 6; ++b) {
            BlockOff& o = lo.blk[b];
            memset(&o, 0xff, sizeof(o));
            o.kind = BLOCK_KIND[b];
            o.mode = b & 1;
            o.nodes = o.mode == 0 ? 17 : T;
            const std::string p = "layers_with_bone." + std::to_string(l) + "." + BLOCK_NAMES[b] + ".";
            o.ls1 = A.add(p + "layer_scale_1", {128});
            o.ls2 = A.add(p + "layer_scale_2", {128});
            o.n1w = A.add(p + "norm1.weight", {128});
            o.n1b = A.add(p + "norm1.bias", {128});
            if (o.kind == KIND_BONE) {
                o.n1lw = A.add(p + "norm1_limb.weight", {128});
                o.n1lb = A.add(p + "norm1_limb.bias", {128});
            }
            o.n2w = A.add(p + "norm2.weight", {128});
            o.n2b = A.add(p + "norm2.bias", {128});
            if (o.kind == KIND_ATT) {
                o.mix_w = A.add(p + "mixer.qkv.weight", {384, 128});
                o.proj_w = A.add(p + "mixer.proj.weight", {128, 128});
                o.proj_b = A.add(p + "mixer.proj.bias", {128});
                o.p_mix = packd(o.mix_w, 384, 128, -1, 0);
                o.p_mixT = packd(o.mix_w, 384, 128, -1, 1);
            } else if (o.kind == KIND_BONE) {
                o.mix_w = A.add(p + "mixer.qkv_q.weight", {128, 128});
                o.kv_w = A.add(p + "mixer.qkv_kv.weight", {256, 128});
                o.proj_w = A.add(p + "mixer.proj.weight", {128, 128});
                o.proj_b = A.add(p + "mixer.proj.bias", {128});
                o.p_mix = packd(o.mix_w, 128, 128, -1, 0);
                o.p_mixT = packd(o.mix_w, 128, 128, -1, 1);
                o.p_kv = packd(o.kv_w, 256, 128, -1, 0);
                o.p_kvT = packd(o.kv_w, 256, 128, -1, 1);
            } else {
                o.mix_w = A.add(p + "mixer.U.weight", {128, 128});     // U then V: one [256,128] operand
                A.add(p + "mixer.V.weight", {128, 128});
                o.uv_b = A.add(p + "mixer.U.bias", {128});
                A.add(p + "mixer.V.bias", {128});
                o.bn_w = A.add(p + "mixer.batch_norm.weight", {o.nodes});
                o.bn_b = A.add(p + "mixer.batch_norm.bias", {o.nodes});
                o.bn_rm = B.add(p + "mixer.batch_norm.running_mean", {o.nodes});
                o.bn_rv = B.add(p + "mixer.batch_norm.running_var", {o.nodes});
                o.p_mix = packd(o.mix_w, 256, 128, -1, 0);
                o.p_mixT = packd(o.mix_w, 256, 128, -1, 1);
            }
            if (o.kind != KIND_GRAPH) {
                o.p_proj = packd(o.proj_w, 128, 128, -1, 0);
                o.p_projTs = packd(o.proj_w, 128, 128, o.ls1, 1);      // (ls1 . Wproj)^T for the dgrad
            }
            o.fc1w = A.add(p + "mlp.fc1.weight", {512, 128});
            o.fc1b = A.add(p + "mlp.fc1.bias", {512});
            o.fc2w = A.add(p + "mlp.fc2.weight", {128, 512});
            o.fc2b = A.add(p + "mlp.fc2.bias", {128});
            o.p_fc1 = packd(o.fc1w, 512, 128, -1, 0);
            o.p_fc1T = packd(o.fc1w, 512, 128, -1, 1);
            o.p_fc2 = packd(o.fc2w, 128, 512, -1, 0, 1);      // read by k_mlp_fwd_s only: fp16 (ignored by the fp32 arena)
            o.p_fc2Ts = packd(o.fc2w, 128, 512, o.ls2, 1);             // (ls2 . W2)^T
        }
        if (m->cfg.use_adaptive_fusion) {
            const std::string p = "layers_with_bone." + std::to_string(l) + ".fusion_three_channel.";
            lo.fus_w = A.add(p + "weight", {3, 384});
            lo.fus_b = A.add(p + "bias", {3});
        }
        lo.end = (A.cur + 3) & ~int64_t(3);
    }
    TopOff& t = m->top;
    t.begin = (A.cur + 3) & ~int64_t(3);
    const char* emb[3] = {"joints_embed", "bone_embed", "limb_embed"};
    const char* pos[3] = {"pos_embed", "bone_pos_embed", "limb_pos_embed"};
    for (int s = 0; s < 3; ++s) {
        m->pro.embed_w[s] = A.add(std::string(emb[s]) + ".weight", {128, 3});
        m->pro.embed_b[s] = A.add(std::string(emb[s]) + ".bias", {128});
        m->pro.pos[s] = A.add(pos[s], {1, 17, 128});
    }
    t.norm_w = A.add("norm.weight", {128});
    t.norm_b = A.add("norm.bias", {128});
    const char* chn[3] = {"mlp_dir_x", "mlp_dir_y", "mlp_len"};
    for (int i = 0; i < 17; ++i)
        for (int c = 0; c < 3; ++c) {
            const std::string p = "bone_refusion.mlp_layers." + std::to_string(i) + "." + chn[c] + ".";
            m->pro.mlp[i * 3 + c][0] = A.add(p + "fc1.weight", {16, LIMB_N[i]});
            m->pro.mlp[i * 3 + c][1] = A.add(p + "fc1.bias", {16});
            m->pro.mlp[i * 3 + c][2] = A.add(p + "fc2.weight", {1, 16});
            m->pro.mlp[i * 3 + c][3] = A.add(p + "fc2.bias", {1});
        }
    t.fc_w = A.add("rep_logit.fc.weight", {512, 128});
    t.fc_b = A.add("rep_logit.fc.bias", {512});
    t.head_w = A.add("head.weight", {3, 512});
    t.head_b = A.add("head.bias", {3});
    t.p_fc = packd(t.fc_w, 512, 128, -1, 0);
    t.p_fcT = packd(t.fc_w, 512, 128, -1, 1);
    t.end = m->n_live = (A.cur + 3) & ~int64_t(3);
    A.cur = m->n_live;
    for (int l = 0; l < L; ++l)                      // never-used norm1_limb of the non-bone blocks (KASportsFormer.py:73)
        for (int b = 0; b < 4; ++b) {
            const std::string p = "layers_with_bone." + std::to_string(l) + "." + BLOCK_NAMES[b] + ".";
            m->layers[l].blk[b].n1lw = A.add(p + "norm1_limb.weight", {128});
            m->layers[l].blk[b].n1lb = A.add(p + "norm1_limb.bias", {128});
        }
    if (!m->cfg.use_adaptive_fusion)                 // plain-mean fusion (KASportsFormer.py:284): the gate Linear exists but never receives a gradient
        for (int l = 0; l < L; ++l) {
            const std::string p = "layers_with_bone." + std::to_string(l) + ".fusion_three_channel.";
            m->layers[l].fus_w = A.add(p + "weight", {3, 384});
            m->layers[l].fus_b = A.add(p + "bias", {3});
        }
    m->n_params = (A.cur + 3) & ~int64_t(3);
    m->n_buffers = (B.cur + 3) & ~int64_t(3);
    m->arena_elems = arena;
    int tiles = 0;
    for (const KasfPackDesc& d : m->pack) {
        m->pack_tile_start.push_back(tiles);
        tiles += (d.rows / 32) * (d.cols / 32);
    }
    m->pack_tiles = tiles;
}

inline int esize(const kasf_model* m) { return m->cfg.dtype == KASF_F32 ? 4 : 2; }

void build_plan(const kasf_model* m, int B, bool train, bool names, Plan& p) {
    const int L = m->cfg.n_layers, T = m->cfg.n_frames, es = esize(m);
    const int64_t M = (int64_t)B * T * 17, groupsT = (int64_t)B * 17;
    int64_t cur = 0;
    auto take = [&](int64_t numel, int kind, const char* name, int l = -1, int b = -1) {
        const int64_t bytes = numel * (kind == 0 ? es : (kind == 2 ? 8 : 4));
        const int64_t off = cur;
        cur = (cur + bytes + 255) & ~int64_t(255);
        if (names) {
            std::string n;
            if (l >= 0) n = "L" + std::to_string(l) + ".";
            if (b >= 0) n += std::string(BLOCK_NAMES[b]) + ".";
            p.entries.push_back(WsEntry{n + name, off, numel, kind});
        }
        return off;
    };
    const int nl = train ? L : 1;
    p.layers.resize(nl);
    p.stats_begin = cur;
    for (int l = 0; l < nl; ++l)
        for (int b = 2; b < 4; ++b) p.layers[l].b[b].stats = take((int64_t)KASF_STAT_LD * KASF_STAT_SLOTS * KASF_STAT_WORDS, 2, "bn_stats", l, b);
    p.stats_bytes = cur - p.stats_begin;
    p.bstats_begin = cur;
    for (int l = 0; l < nl; ++l)
        for (int b = 2; b < 4; ++b) p.layers[l].b[b].bstats = take((int64_t)KASF_STAT_LD * KASF_STAT_SLOTS * KASF_STAT_WORDS, 2, "bn_bwd_stats", l, b);
    p.bstats_bytes = cur - p.bstats_begin;
    p.x3 = take(M * 3, 1, "x_input");
    p.bone3 = take(M * 3, 1, "bone3");
    p.limb3 = take(M * 3, 1, "limb3");
    p.xj = take(M * 128, 0, "x_joint");
    p.xb = take(M * 128, 0, "x_bone");
    p.xl = take(M * 128, 0, "x_limb");
    for (int l = 0; l < nl; ++l) {
        LayerWs& lw = p.layers[l];
        for (int b = 0; b < 6; ++b) {
            BlkWs& w = lw.b[b];
            const int kind = BLOCK_KIND[b];
            const bool recompute = train && fused_attn_bwd(m->cfg, kind, b & 1);      // the fused backward re-forms q | k | v | o from x: nothing to keep
            w.qkv = w.kv = w.o = -1;
            if (kind == KIND_ATT && !recompute) w.qkv = take(M * 384, 0, "qkv", l, b);
            if (kind == KIND_BONE && !recompute) { w.qkv = take(M * 128, 0, "q", l, b); w.kv = take(M * 256, 0, "kv", l, b); }
            if (kind != KIND_GRAPH && !recompute) w.o = take(M * 128, 0, "o", l, b);
            // temporal attention over 33..96 frames (T = 81): log-sum-exp of the scores per (token, head), left by the fused forward for k_attn_bwd_kt
            w.lse = (train && kind != KIND_GRAPH && (b & 1) && T > 32 && T <= 96 && m->cfg.dtype == KASF_BF16 && m->cfg.num_heads == 8) ? take(M * 8, 1, "lse", l, b) : -1;
            if (kind == KIND_GRAPH) {
                w.xn = take(M * 128, 0, "xn", l, b);
                w.y = take(M * 128, 0, "y", l, b);
                w.coef = take(KASF_MAX_NODES * 8, 1, "bn_coef", l, b);
                w.mask = (b & 1) ? take(groupsT * T * kasf_gcn_mask_words(T), 3, "adj_mask", l, b) : -1;
            }
            w.x_mid = take(M * 128, 0, "x_mid", l, b);
            w.xn2 = (train && m->cfg.dtype == KASF_BF16) ? take(M * 128, 0, "xn2", l, b) : -1;   // LN2(x_mid), streamed by the fused MLP backward
            w.x_out = take(M * 128, 0, "x_out", l, b);
        }
        lw.gate_out = take(M * 128, 0, "gate_out", l);
        lw.alpha = take(M * 4, 1, "alpha", l);
    }
    p.rep = take(M * 512, 0, "rep");
    for (int br = 0; br < 3; ++br) p.sc[br].uv = (br == 1) ? take(M * 256, 0, "scratch_uv") : -1;     // only the graph branch needs it
    if (train) {
        // rows of per-workgroup column sums between a kernel and the stage's k_col_finish: the persistent bf16 kernels need ~1 M floats per layer and branch,
        // the first-generation kernels (fp32 mode, the 512-wide top-level data gradient) one 256-float row per 32 / 64 tokens and launch
        p.col_cap = (int64_t)3 * 1024 * 1024 + (es == 4 ? 56 : 8) * M;
        p.g_layer = take(M * 128, 0, "g_layer");
        p.g_prev = take(M * 128, 0, "g_prev");
        p.ga = take(M * 128, 0, "g_att");
        p.gg = take(M * 128, 0, "g_graph");
        p.gb = take(M * 128, 0, "g_bonebr");
        p.g_limb = take(M * 128, 0, "g_limb");
        p.g_bone = take(M * 128, 0, "g_bone");
        p.dlimb3 = take(M * 3, 1, "dlimb3");
        static const char* const BR[3] = {"att", "graph", "bone"};
        for (int br = 0; br < 3; ++br) {                 // the three branches of a layer run concurrently: private scratch each
            Scratch& s = p.sc[br];
            auto nm = [&](const char* base) { static thread_local std::string t; t = std::string(base) + "." + BR[br]; return t.c_str(); };
            s.t1 = take(M * 128, 0, nm("g_tmp1"));
            s.t2 = take(M * 128, 0, nm("g_tmp2"));
            s.g_in = take(M * 128, 0, nm("g_in"));
            s.hbuf = take(M * 512, 0, nm("mlp_h"));
            s.dzbuf = es == 4 ? take(M * 512, 0, nm("mlp_dz")) : -1;          // bf16 fuses the weight gradients: no dZ round trip
            s.xn_a = take(M * 128, 0, nm("scratch_xn_a"));
            s.wg_part = take(WG_PARTIAL_FLOATS, 1, nm("wgrad_partials"));
            s.col = take(p.col_cap, 1, nm("column_partials"));
            s.d_o = br != 1 ? take(M * 128, 0, nm("d_o")) : -1;
            s.dqkv = br != 1 ? take(M * 384, 0, nm("dqkv")) : -1;
            s.xn_b = br == 2 ? take(M * 128, 0, nm("scratch_xn_b")) : -1;
            s.rbuf = br == 1 ? take(M * 128, 0, nm("gcn_r")) : -1;
            s.duv = br == 1 ? take(M * 256, 0, nm("gcn_duv")) : -1;
        }
    }
    p.total = cur;
}

// ----------------------------------------------------------------------------------------------
struct Ctx {
    const kasf_model* m;
    const float* P;          // fp32 parameters
    const char* A;           // packed arena
    float* buf;              // BN buffers
    float* G;                // gradients (backward)
    char* ws;
    hipStream_t s;
    int B, T, dt, es;
    int64_t M;
    bool train;              // activations are kept for a backward pass
    bool bn_train;           // BatchNorm uses batch statistics and updates the running ones (KASF_FLAG_TRAIN)
    KasfColSink* sink = nullptr;   // backward: where this stream's kernels leave their per-workgroup column sums
    const void* pk(int64_t elem_off) const { return A + elem_off * es; }
    void* w(int64_t byte_off) const { return ws + byte_off; }
};

void block_forward(const Ctx& c, const BlockOff& o, const BlkWs& w, const void* x_in, const void* x_limb, const Plan& p, const Scratch& sc) {
    const float* P = c.P;
    bool mixer_done = false;     // bf16, 8 heads, groups of <= 96 positions: LN + QKV + attention + proj + residual in one kernel (csrc/k_attn_blk.hip)
    const int heads = c.m->cfg.num_heads;
    if (c.dt == KASF_BF16 && o.kind != KIND_GRAPH && heads == 8) {
        const bool bone = o.kind == KIND_BONE;
        mixer_done = kasf_launch_attn_block_fwd(c.s, bone ? 1 : 0, x_in, bone ? x_limb : nullptr, P + o.n1w, P + o.n1b, bone ? P + o.n1lw : nullptr,
                                                bone ? P + o.n1lb : nullptr, c.pk(o.p_mix), bone ? c.pk(o.p_kv) : nullptr, c.pk(o.p_proj), P + o.proj_b,
                                                P + o.ls1, (c.train && w.qkv >= 0) ? c.w(w.qkv) : nullptr, (c.train && bone && w.kv >= 0) ? c.w(w.kv) : nullptr,
                                                (c.train && w.o >= 0) ? c.w(w.o) : nullptr, c.w(w.x_mid), c.B, c.T, o.mode, (c.train && w.lse >= 0) ? (float*)c.w(w.lse) : nullptr);
    }
    if (mixer_done) {
    } else if (o.kind == KIND_ATT) {
        kasf_launch_linear(c.dt, c.s, x_in, 128, c.pk(o.p_mix), 128, nullptr, c.w(w.qkv), 384, c.M, 384, P + o.n1w, P + o.n1b, nullptr, 0);
        const char* q = (const char*)c.w(w.qkv);
        kasf_launch_attn_fwd(c.dt, c.s, q, 384, q + 128 * c.es, q + 256 * c.es, 384, c.w(w.o), c.B, c.T, o.mode, heads);
    } else if (o.kind == KIND_BONE) {
        kasf_launch_linear(c.dt, c.s, x_in, 128, c.pk(o.p_mix), 128, nullptr, c.w(w.qkv), 128, c.M, 128, P + o.n1w, P + o.n1b, nullptr, 0);
        kasf_launch_linear(c.dt, c.s, x_limb, 128, c.pk(o.p_kv), 128, nullptr, c.w(w.kv), 256, c.M, 256, P + o.n1lw, P + o.n1lb, nullptr, 0);
        const char* kv = (const char*)c.w(w.kv);
        kasf_launch_attn_fwd(c.dt, c.s, c.w(w.qkv), 128, kv, kv + 128 * c.es, 256, c.w(w.o), c.B, c.T, o.mode, heads);
    } else {
        kasf_launch_linear(c.dt, c.s, x_in, 128, c.pk(o.p_mix), 128, P + o.uv_b, c.w(sc.uv), 256, c.M, 256, P + o.n1w, P + o.n1b, c.w(w.xn), 0);
        kasf_launch_gcn_agg_fwd(c.dt, c.s, c.w(sc.uv), c.w(w.xn), c.w(w.y), w.mask >= 0 ? (uint32_t*)c.w(w.mask) : nullptr, (double*)c.w(w.stats), c.B,
                                c.T, o.mode, c.m->cfg.neighbour_num);
        const double count = o.mode == 0 ? (double)c.B * c.T * 128 : (double)c.B * 17 * 128;
        kasf_launch_gcn_apply(c.dt, c.s, x_in, c.w(w.xn), c.w(w.y), (const double*)c.w(w.stats), P + o.bn_w, P + o.bn_b, c.buf + o.bn_rm, c.buf + o.bn_rv,
                              (float*)c.w(w.coef), P + o.ls1, c.w(w.x_mid), c.B, c.T, o.mode, count, c.bn_train ? 1 : 0, 0.1f);
    }
    if (o.kind != KIND_GRAPH && !mixer_done)
        kasf_launch_linear_res(c.dt, c.s, c.w(w.o), c.pk(o.p_proj), P + o.proj_b, P + o.ls1, x_in, c.w(w.x_mid), c.M);
    kasf_launch_mlp_fwd(c.dt, c.s, c.w(w.x_mid), P + o.n2w, P + o.n2b, c.pk(o.p_fc1), P + o.fc1b, c.pk(o.p_fc2), P + o.fc2b, P + o.ls2, c.w(w.x_out), c.M,
                        w.xn2 >= 0 ? c.w(w.xn2) : nullptr);
}

// g_out: gradient w.r.t. the block output; writes (or accumulates) the gradient w.r.t. x_in into dst.
// Every per-channel gradient (LayerNorm gamma / beta, biases, layer scales) leaves its kernel as per-workgroup rows in c.sink and is complete only
// after the stage's kasf_col_flush.
void block_backward(const Ctx& c, const BlockOff& o, const BlkWs& w, const void* x_in, const void* x_limb, const void* g_out, void* dst, int accumulate,
                    const Plan& p, const Scratch& sc) {
    const float* P = c.P;
    float* G = c.G;
    void* g_mid = c.w(sc.t2);
    float* part = (float*)c.w(sc.wg_part);
    // ---- MLP half ----
    if (c.dt == KASF_BF16) {
        // hidden-quarter kernel: dgrad + both weight gradients fused, then the 4-way partial sum + LayerNorm backward + the fc2 layer-scale algebra
        kasf_launch_mlp_bwd_q(c.s, c.w(w.x_mid), c.w(w.xn2), g_out, P + o.n2w, c.pk(o.p_fc1), P + o.fc1b, c.pk(o.p_fc2Ts), c.pk(o.p_fc1T), c.w(sc.hbuf),
                              part, G + o.fc1w, G + o.fc2w, G + o.fc1b, G + o.fc2b, g_mid, G + o.n2w, G + o.n2b, c.M, P + o.fc2w, P + o.fc2b,
                              P + o.ls2, G + o.ls2, c.sink);
    } else {
        kasf_launch_mlp_bwd(c.dt, c.s, c.w(w.x_mid), g_out, P + o.n2w, P + o.n2b, c.pk(o.p_fc1), P + o.fc1b, c.pk(o.p_fc2Ts), c.pk(o.p_fc1T), c.w(sc.hbuf),
                            c.w(sc.dzbuf), c.w(sc.xn_a), g_mid, G + o.n2w, G + o.n2b, c.M, c.sink);
        kasf_launch_wgrad(c.dt, c.s, c.w(sc.dzbuf), 512, 512, c.w(sc.xn_a), 128, 128, nullptr, nullptr, G + o.fc1w, 128, G + o.fc1b, c.M, part, WG_PARTIAL_FLOATS);
        kasf_launch_wgrad(c.dt, c.s, g_out, 128, 128, c.w(sc.hbuf), 512, 512, nullptr, nullptr, G + o.fc2w, 512, G + o.fc2b, c.M, part, WG_PARTIAL_FLOATS);
        kasf_launch_finalize_ls(c.s, G + o.fc2w, P + o.fc2w, P + o.fc2b, P + o.ls2, G + o.fc2b, G + o.ls2, 128, 512);
    }
    // ---- mixer half ----
    if (o.kind == KIND_GRAPH) {
        kasf_launch_gcn_bwd1(c.dt, c.s, g_mid, c.w(w.xn), c.w(w.y), (const float*)c.w(w.coef), P + o.ls1, c.w(sc.rbuf), G + o.ls1, (double*)c.w(w.bstats),
                             c.B, c.T, o.mode, c.sink);
        const double count = o.mode == 0 ? (double)c.B * c.T * 128 : (double)c.B * 17 * 128;
        kasf_launch_gcn_bwd2(c.dt, c.s, c.w(sc.rbuf), c.w(w.y), (const float*)c.w(w.coef), w.mask >= 0 ? (const uint32_t*)c.w(w.mask) : nullptr, c.w(sc.duv),
                             c.B, c.T, o.mode, (const double*)c.w(w.bstats), G + o.bn_w, G + o.bn_b, count, c.bn_train ? 1 : 0);
        // bf16: the U | V weight and bias gradients ride in the data-gradient kernel (dY tile in its ring, LN(x) formed by its LayerNorm-backward phase)
        int np = 0;
        if (c.dt == KASF_BF16 && c.sink != nullptr && c.M >= WG_FUSE_MIN_TOKENS)
            np = kasf_launch_dgrad_wg(c.s, c.w(sc.duv), 256, c.pk(o.p_mixT), x_in, P + o.n1w, P + o.n1b, g_mid, dst, accumulate, G + o.n1w, G + o.n1b, c.M, c.sink, part,
                                      (int64_t)WG_PARTIAL_FLOATS * 4, c.w(sc.rbuf), G + o.uv_b);
        if (np > 0) {
            const KasfBf16Reduce red{part, G + o.mix_w, np, 256 * 128};
            kasf_launch_bf16_reduce(c.s, 1, &red);
            return;
        }
        kasf_launch_dgrad_lnbwd(c.dt, c.s, c.w(sc.duv), 256, c.pk(o.p_mixT), c.w(sc.rbuf), x_in, P + o.n1w, g_mid, dst, accumulate, G + o.n1w, G + o.n1b,
                                c.M, nullptr, nullptr, c.sink);
        kasf_launch_wgrad(c.dt, c.s, c.w(sc.duv), 256, 256, c.w(w.xn), 128, 128, nullptr, nullptr, G + o.mix_w, 128, G + o.uv_b, c.M, part, WG_PARTIAL_FLOATS);
        return;
    }
    if (fused_attn_bwd(c.m->cfg, o.kind, o.mode)) {
        // one launch: LN, q | k | v, d_o, the 8 attention cores, the data gradient with its LayerNorm backward + residual, the proj weight gradient (bf16 partial tiles + rows of
        // colsum(g_mid)) and per-workgroup rows of the LayerNorm gamma / beta gradients; it leaves dq | dk | dv and LN(x) for ONE streaming weight-gradient launch, whose finish
        // launch also adds the proj tiles and applies the layer-scale algebra
        const bool bone = o.kind == KIND_BONE;
        char* ppart = (char*)(part + WG_JOBS_FLOATS);
        float* pbrow = (float*)(ppart + (int64_t)256 * 128 * 128 * 2);
        char* dq = (char*)c.w(sc.dqkv);
        char* dkv = dq + c.M * 128 * c.es;
        const int np = (accumulate != 0 || c.sink == nullptr) ? 0 :
            kasf_launch_attn_block_bwd(c.s, bone ? 1 : 0, x_in, bone ? x_limb : nullptr, g_mid, P + o.n1w, P + o.n1b, bone ? P + o.n1lw : nullptr, bone ? P + o.n1lb : nullptr,
                                       c.pk(o.p_mix), bone ? c.pk(o.p_kv) : nullptr, bone ? c.pk(o.p_mixT) : nullptr, c.pk(o.p_projTs), dst, bone ? c.w(p.g_limb) : nullptr,
                                       dq, bone ? dkv : nullptr, c.w(sc.xn_a), bone ? c.w(sc.xn_b) : nullptr, G + o.n1w, G + o.n1b, bone ? G + o.n1lw : nullptr,
                                       bone ? G + o.n1lb : nullptr, c.sink, ppart, pbrow, c.B, c.T, o.mode);
        bool done = false;
        if (np > 0) {
            const void* Gs[2] = {dq, dkv};
            const void* Xs[2] = {c.w(sc.xn_a), bone ? c.w(sc.xn_b) : nullptr};
            const int Ns[2] = {bone ? 128 : 384, 256};
            float* dWs[2] = {G + o.mix_w, bone ? G + o.kv_w : nullptr};
            float* dbs[2] = {nullptr, nullptr};
            done = kasf_launch_wgrad_jobs(c.s, bone ? 2 : 1, Gs, Xs, Ns, dWs, dbs, -1, P + o.proj_w, P + o.proj_b, P + o.ls1, G + o.ls1, c.M, part, WG_JOBS_FLOATS, 0, nullptr,
                                          ppart, pbrow, np, G + o.proj_w, G + o.proj_b);
        }
        if (!done)                                       // the forward kept nothing for the four-launch sequence: there is no other way to finish this block
            kasf_set_error(3, "fused attention-block backward did not launch (column-sum scratch exhausted or unsupported call)");
        return;
    }
    // d_o = g_mid . (ls1 . Wproj);  G_proj = g_mid^T o (unscaled) -> finish: dWproj, dbproj, dls1
    const int heads = c.m->cfg.num_heads;
    const int Lg = o.mode == 0 ? 17 : c.T;
    const bool fdo = c.dt == KASF_BF16 && heads == 8 && Lg <= 96;   // d_o formed inside the attention backward kernel
    if (!fdo) kasf_launch_linear(c.dt, c.s, g_mid, 128, c.pk(o.p_projTs), 128, nullptr, c.w(sc.d_o), 128, c.M, 128, nullptr, nullptr, nullptr, 0);
    // bf16: every weight gradient of the block (proj, qkv | q, kv) goes into ONE streaming launch + one finishing launch at the end of the block
    const bool jobs = c.dt == KASF_BF16;
    if (!jobs) {
        kasf_launch_wgrad(c.dt, c.s, g_mid, 128, 128, c.w(w.o), 128, 128, nullptr, nullptr, G + o.proj_w, 128, G + o.proj_b, c.M, part, WG_PARTIAL_FLOATS);
        kasf_launch_finalize_ls(c.s, G + o.proj_w, P + o.proj_w, P + o.proj_b, P + o.ls1, G + o.proj_b, G + o.ls1, 128, 128);
    }
    if (o.kind == KIND_ATT) {
        const char* q = (const char*)c.w(w.qkv);
        char* dq = (char*)c.w(sc.dqkv);
        if (fdo) kasf_launch_attn_bwd_fused_do(c.s, q, 384, q + 128 * c.es, q + 256 * c.es, 384, g_mid, c.pk(o.p_projTs), dq, 384, dq + 128 * c.es,
                                               dq + 256 * c.es, 384, c.B, c.T, o.mode, 0, c.w(w.o), w.lse >= 0 ? (const float*)c.w(w.lse) : nullptr);
        else kasf_launch_attn_bwd(c.dt, c.s, q, 384, q + 128 * c.es, q + 256 * c.es, 384, c.w(sc.d_o), dq, 384, dq + 128 * c.es, dq + 256 * c.es, 384, c.B,
                                  c.T, o.mode, heads);
        bool fusedwg = false;
        KasfBf16Reduce red[2];
        int nred = 0;
        char* wpart = (char*)(part + WG_JOBS_FLOATS);      // bf16 partial tiles behind the jobs kernel's own fp32 tiles: both live until the block's finish launch
        // the qkv weight gradient rides in the data-gradient kernel (its dY tile and LN(x) are in LDS there): no LN(x) round trip, dqkv read once
        if (jobs && c.M >= WG_FUSE_MIN_TOKENS) {
            const int np = kasf_launch_dgrad_wg(c.s, dq, 384, c.pk(o.p_mixT), x_in, P + o.n1w, P + o.n1b, g_mid, dst, accumulate, G + o.n1w, G + o.n1b, c.M, c.sink,
                                                wpart, WG_BF16_BYTES);
            if (np > 0) { fusedwg = true; red[nred++] = KasfBf16Reduce{wpart, G + o.mix_w, np, 384 * 128}; }
        }
        if (!fusedwg)
            kasf_launch_dgrad_lnbwd(c.dt, c.s, dq, 384, c.pk(o.p_mixT), nullptr, x_in, P + o.n1w, g_mid, dst, accumulate, G + o.n1w, G + o.n1b, c.M,
                                    c.w(sc.xn_a), P + o.n1b, c.sink);
        bool done = false;
        if (jobs) {
            const void* Gs[2] = {g_mid, dq};
            const void* Xs[2] = {c.w(w.o), c.w(sc.xn_a)};
            const int Ns[2] = {128, 384};
            float* dWs[2] = {G + o.proj_w, G + o.mix_w};
            float* dbs[2] = {G + o.proj_b, nullptr};
            done = kasf_launch_wgrad_jobs(c.s, fusedwg ? 1 : 2, Gs, Xs, Ns, dWs, dbs, 0, P + o.proj_w, P + o.proj_b, P + o.ls1, G + o.ls1, c.M, part,
                                          fusedwg ? WG_JOBS_FLOATS : WG_PARTIAL_FLOATS, nred, red);
        }
        if (!done) {
            if (jobs) {
                kasf_launch_wgrad(c.dt, c.s, g_mid, 128, 128, c.w(w.o), 128, 128, nullptr, nullptr, G + o.proj_w, 128, G + o.proj_b, c.M, part, WG_PARTIAL_FLOATS);
                kasf_launch_finalize_ls(c.s, G + o.proj_w, P + o.proj_w, P + o.proj_b, P + o.ls1, G + o.proj_b, G + o.ls1, 128, 128);
            }
            kasf_launch_wgrad(c.dt, c.s, dq, 384, 384, c.w(sc.xn_a), 128, 128, nullptr, nullptr, G + o.mix_w, 128, nullptr, c.M, part, WG_PARTIAL_FLOATS);
        }
    } else {
        const char* kv = (const char*)c.w(w.kv);
        char* dq = (char*)c.w(sc.dqkv);
        char* dkv = dq + c.M * 128 * c.es;
        if (fdo) kasf_launch_attn_bwd_fused_do(c.s, c.w(w.qkv), 128, kv, kv + 128 * c.es, 256, g_mid, c.pk(o.p_projTs), dq, 128, dkv, dkv + 128 * c.es, 256, c.B,
                                               c.T, o.mode, 0, c.w(w.o), w.lse >= 0 ? (const float*)c.w(w.lse) : nullptr);
        else kasf_launch_attn_bwd(c.dt, c.s, c.w(w.qkv), 128, kv, kv + 128 * c.es, 256, c.w(sc.d_o), dq, 128, dkv, dkv + 128 * c.es, 256, c.B, c.T, o.mode, heads);
        bool fusedwg = false;
        KasfBf16Reduce red[2];
        int nred = 0;
        char* wpart = (char*)(part + WG_JOBS_FLOATS);
        bool proj_fused = false;
        const int64_t qb_ = (int64_t)256 * 128 * 128 * 2, kvb_ = (int64_t)256 * 256 * 128 * 2;
        // both fused launches or neither (the q launch registers its dgamma / dbeta rows in the sink; a kv launch declining afterwards would have sent
        // q through the two-kernel sequence a second time)
        if (jobs && c.M >= WG_FUSE_MIN_TOKENS && kasf_dgrad_wg_supported(128, true, accumulate != 0, false, false, true, c.M, qb_) &&
            kasf_dgrad_wg_supported(256, false, true, false, false, false, c.M, kvb_)) {
            // q: data gradient + dW_q + the block's PROJ gradient (g_mid is its residual operand, o one more ring stream); kv: data gradient + dW_kv.  No streaming
            // weight-gradient launch in this block at all: one finish launch adds the three sets of bf16 partial tiles.
            const int64_t qb = (int64_t)256 * 128 * 128 * 2, kvb = (int64_t)256 * 256 * 128 * 2;
            char* ppart = wpart + qb + kvb;                                  // <= 256 bf16 proj tiles (8.4 MB) ...
            float* pbrow = (float*)(ppart + qb);                             // ... and their colsum(g_mid) rows
            const int npq = kasf_launch_dgrad_wg(c.s, dq, 128, c.pk(o.p_mixT), x_in, P + o.n1w, P + o.n1b, g_mid, dst, accumulate, G + o.n1w, G + o.n1b, c.M, c.sink,
                                                 wpart, qb, nullptr, nullptr, c.w(w.o), ppart, pbrow);
            const int npk = kasf_launch_dgrad_wg(c.s, dkv, 256, c.pk(o.p_kvT), x_limb, P + o.n1lw, P + o.n1lb, nullptr, c.w(p.g_limb), 1, G + o.n1lw,
                                                 G + o.n1lb, c.M, c.sink, wpart + qb, kvb);
            if (npq <= 0 || npk <= 0) {                  // cannot happen after the check above; if it ever does the block's remaining gradients are missing: kasf_backward returns non-zero
                kasf_set_error(3, "bone block: a fused data + weight gradient launch declined after kasf_dgrad_wg_supported accepted it");
                return;
            }
            {
                fusedwg = proj_fused = true;
                red[nred++] = KasfBf16Reduce{wpart, G + o.mix_w, npq, 128 * 128};
                red[nred++] = KasfBf16Reduce{wpart + qb, G + o.kv_w, npk, 256 * 128};
                kasf_launch_proj_finish(c.s, ppart, pbrow, npq, G + o.proj_w, P + o.proj_w, P + o.proj_b, P + o.ls1, G + o.proj_b, G + o.ls1, nred, red);
            }
        }
        if (!fusedwg) {
            kasf_launch_dgrad_lnbwd(c.dt, c.s, dq, 128, c.pk(o.p_mixT), nullptr, x_in, P + o.n1w, g_mid, dst, accumulate, G + o.n1w, G + o.n1b, c.M,
                                    c.w(sc.xn_a), P + o.n1b, c.sink);
            kasf_launch_dgrad_lnbwd(c.dt, c.s, dkv, 256, c.pk(o.p_kvT), nullptr, x_limb, P + o.n1lw, nullptr, c.w(p.g_limb), 1, G + o.n1lw, G + o.n1lb, c.M,
                                    c.w(sc.xn_b), P + o.n1lb, c.sink);
        }
        bool done = proj_fused;
        if (jobs && !proj_fused) {
            const void* Gs[3] = {g_mid, dq, dkv};
            const void* Xs[3] = {c.w(w.o), c.w(sc.xn_a), c.w(sc.xn_b)};
            const int Ns[3] = {128, 128, 256};
            float* dWs[3] = {G + o.proj_w, G + o.mix_w, G + o.kv_w};
            float* dbs[3] = {G + o.proj_b, nullptr, nullptr};
            done = kasf_launch_wgrad_jobs(c.s, fusedwg ? 1 : 3, Gs, Xs, Ns, dWs, dbs, 0, P + o.proj_w, P + o.proj_b, P + o.ls1, G + o.ls1, c.M, part,
                                          fusedwg ? WG_JOBS_FLOATS : WG_PARTIAL_FLOATS, nred, red);
        }
        if (!done) {
            if (jobs) {
                kasf_launch_wgrad(c.dt, c.s, g_mid, 128, 128, c.w(w.o), 128, 128, nullptr, nullptr, G + o.proj_w, 128, G + o.proj_b, c.M, part, WG_PARTIAL_FLOATS);
                kasf_launch_finalize_ls(c.s, G + o.proj_w, P + o.proj_w, P + o.proj_b, P + o.ls1, G + o.proj_b, G + o.ls1, 128, 128);
            }
            kasf_launch_wgrad(c.dt, c.s, dq, 128, 128, c.w(sc.xn_a), 128, 128, nullptr, nullptr, G + o.mix_w, 128, nullptr, c.M, part, WG_PARTIAL_FLOATS);
            kasf_launch_wgrad(c.dt, c.s, dkv, 256, 256, c.w(sc.xn_b), 128, 128, nullptr, nullptr, G + o.kv_w, 128, nullptr, c.M, part, WG_PARTIAL_FLOATS);
        }
    }
}

int check_model(const kasf_model* m) {
    if (m == nullptr) return kasf_set_error(2, "null model");
    return 0;
}

}  // namespace

// ================================================================================================ C-ABI
extern "C" {

const char* kasf_last_error(void) { return g_err.c_str(); }
void kasf_set_single_stream(int32_t on) { g_single_stream.store(on ? 1 : 0, std::memory_order_relaxed); }
int32_t kasf_get_single_stream(void) { return single_stream() ? 1 : 0; }
// the round-3 names: the setting never was about determinism (gradients are bit-reproducible either way)
void kasf_set_fused_wgrad_min_tokens(int64_t tokens) { g_wg_fuse_min_tokens.store(tokens < 0 ? (int64_t)KASF_WG_FUSE_MIN_TOKENS : tokens, std::memory_order_relaxed); }
int64_t kasf_get_fused_wgrad_min_tokens(void) { return g_wg_fuse_min_tokens.load(std::memory_order_relaxed); }
void kasf_set_fused_attn_bwd(int32_t on) { g_fused_attn_bwd.store(on < 0 ? -1 : (on == 1 ? 15 : (on > 15 ? 0 : on)), std::memory_order_relaxed); }
int32_t kasf_get_fused_attn_bwd(void) { return fused_attn_bwd_mask(); }
void kasf_set_deterministic(int32_t on) { kasf_set_single_stream(on); }
int32_t kasf_get_deterministic(void) { return kasf_get_single_stream(); }
int kasf_version(void) { return 8; }

int kasf_model_create(const kasf_config* cfg, kasf_model** out) {
    if (cfg == nullptr || out == nullptr) return kasf_set_error(2, "null argument");
    if (cfg->num_heads != 2 && cfg->num_heads != 4 && cfg->num_heads != 8 && cfg->num_heads != 16)
        return kasf_set_error(3, "num_heads must be 2, 4, 8 or 16 (8 = the shipped yaml has the MFMA attention kernels; the others run generic ones)");
    if (cfg->n_layers < 1 || cfg->n_layers > 64) return kasf_set_error(3, "n_layers out of range");
    // any clip length the reference can build (BatchNorm1d(n_frames), top-4 of T similarities needs T >= 4); 9 / 27 / 81 have tuned temporal
    // kernels (and T <= 96 the MFMA attention cores), other lengths run generic ones
    if (cfg->n_frames < 4 || cfg->n_frames > KASF_MAX_NODES) return kasf_set_error(3, "n_frames must be in [4, 256]");
    if (cfg->neighbour_num < 1 || cfg->neighbour_num > 4) return kasf_set_error(3, "neighbour_num must be 1..4 (the temporal GCN kernels keep a row's four largest similarities; every shipped yaml uses 4)");
    // the generic attention backward (everything but 8 heads with n_frames <= 96 in bf16) keeps a track's q, k, v, d_o of one head in LDS:
    // (4 T D + 4 T) floats <= 160 KB, i.e. T (D + 1) <= 10240 -- only num_heads = 2 (D = 64) beyond 157 frames exceeds it
    if ((int64_t)cfg->n_frames * (128 / cfg->num_heads + 1) > 10240)
        return kasf_set_error(3, "num_heads = 2 supports n_frames <= 157 (the attention backward keeps a head's track in LDS)");
    if (cfg->dtype != KASF_F32 && cfg->dtype != KASF_BF16) return kasf_set_error(3, "dtype must be KASF_DTYPE_F32 or KASF_DTYPE_BF16");
    kasf_model* m = new kasf_model();
    m->cfg = *cfg;
    build_layout(m);
    HIPCHK(hipMalloc((void**)&m->d_pack, m->pack.size() * sizeof(KasfPackDesc)));
    HIPCHK(hipMalloc((void**)&m->d_tile_start, m->pack_tile_start.size() * sizeof(int)));
    HIPCHK(hipMalloc((void**)&m->d_pro, sizeof(KasfProOff)));
    HIPCHK(hipMalloc((void**)&m->d_err, 64));
    HIPCHK(hipMemset(m->d_err, 0, 64));
    HIPCHK(hipMemcpy(m->d_pack, m->pack.data(), m->pack.size() * sizeof(KasfPackDesc), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m->d_tile_start, m->pack_tile_start.data(), m->pack_tile_start.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m->d_pro, &m->pro, sizeof(KasfProOff), hipMemcpyHostToDevice));
    // The two side streams are PROCESS-WIDE, one pair per device, created with the first model and never destroyed: every model of the process forks its
    // branches onto the same pair (events order the work, so models may share them).  Round 3 gave every model a pair of its own -- and the SECOND model of
    // a process then ran its step at 25-26 ms where the first ran at 15.8 (T = 27, B = 32; slower than with all three branches on ONE stream, 19.7), the
    // third at 15.8 again, the fourth at 18.4: which hardware queues a new stream lands on next to the existing ones decides how well three streams
    // overlap, and only the first pair was reliably lucky (tools/dp_probe2.py scenarios A / F / G; profiles/r4_stream_placement.jsonl).
    // Consequences of the sharing (documented in kasf.h): models of one device driven from DIFFERENT host threads serialise on the pair (correct -- events order
    // every fork and join -- but not concurrent), and the pair outlives every model: it is released with the process (two streams per device, by design).
    static hipStream_t shared_side[64][2] = {};
    static std::mutex shared_side_mutex;
    {
        std::lock_guard<std::mutex> lock(shared_side_mutex);
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64) { delete m; return kasf_set_error(2, "device index outside [0, 64): no shared side streams for it"); }
        for (int i = 0; i < 2; ++i) {
            if (shared_side[dev][i] == nullptr) HIPCHK(hipStreamCreateWithFlags(&shared_side[dev][i], hipStreamNonBlocking));
            m->side[i] = shared_side[dev][i];
        }
    }
    for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    kasf_gcn_init();
    *out = m;
    return 0;
}

// layout-only handle (no device): used by the CPU tests and by hosts that only need names/offsets
int kasf_model_create_layout_only(const kasf_config* cfg, kasf_model** out) {
    if (cfg == nullptr || out == nullptr) return kasf_set_error(2, "null argument");
    kasf_model* m = new kasf_model();
    m->cfg = *cfg;
    build_layout(m);
    *out = m;
    return 0;
}

void kasf_model_destroy(kasf_model* m) {
    if (m == nullptr) return;
    if (m->d_pack) (void)hipFree(m->d_pack);
    if (m->d_tile_start) (void)hipFree(m->d_tile_start);
    if (m->d_pro) (void)hipFree(m->d_pro);
    if (m->d_err) (void)hipFree(m->d_err);
    for (int i = 0; i < 2; ++i) {
        if (m->ev_join[i]) (void)hipEventDestroy(m->ev_join[i]);
    }
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    delete m;
}

int64_t kasf_param_count(const kasf_model* m) { return m ? m->n_params : 0; }
int64_t kasf_param_live_count(const kasf_model* m) { return m ? m->n_live : 0; }
int32_t kasf_param_entries(const kasf_model* m) { return m ? (int32_t)m->params.size() : 0; }
int64_t kasf_buffer_count(const kasf_model* m) { return m ? m->n_buffers : 0; }
int32_t kasf_buffer_entries(const kasf_model* m) { return m ? (int32_t)m->buffers.size() : 0; }

static int entry_out(const std::vector<Entry>& v, int32_t idx, char* name, int32_t cap, int64_t* offset, int32_t* ndim, int64_t shape[4]) {
    if (idx < 0 || idx >= (int32_t)v.size()) return kasf_set_error(2, "entry index out of range");
    const Entry& e = v[idx];
    if (name != nullptr && cap > 0) { strncpy(name, e.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (offset) *offset = e.off;
    if (ndim) *ndim = e.ndim;
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = e.shape[i];
    return 0;
}
int kasf_param_entry(const kasf_model* m, int32_t idx, char* name, int32_t cap, int64_t* offset, int32_t* ndim, int64_t shape[4]) {
    if (check_model(m)) return 2;
    return entry_out(m->params, idx, name, cap, offset, ndim, shape);
}
int kasf_buffer_entry(const kasf_model* m, int32_t idx, char* name, int32_t cap, int64_t* offset, int32_t* ndim, int64_t shape[4]) {
    if (check_model(m)) return 2;
    return entry_out(m->buffers, idx, name, cap, offset, ndim, shape);
}
int32_t kasf_backward_stages(const kasf_model* m) { return m ? m->cfg.n_layers + 2 : 0; }
int kasf_stage_grad_range(const kasf_model* m, int32_t stage, int64_t* begin, int64_t* end) {
    if (check_model(m)) return 2;
    const int L = m->cfg.n_layers;
    if (stage < 0 || stage > L + 1) return kasf_set_error(2, "stage out of range");
    if (stage == 0) { *begin = 0; *end = 0; }
    else if (stage <= L) { *begin = m->layers[L - stage].begin; *end = m->layers[L - stage].end; }
    else { *begin = m->top.begin; *end = m->top.end; }
    return 0;
}

int kasf_model_status(const kasf_model* m, int32_t* status) {
    if (check_model(m)) return 2;
    if (status == nullptr) return kasf_set_error(2, "null argument");
    unsigned v = 0;
    if (m->d_err != nullptr) HIPCHK(hipMemcpy(&v, m->d_err, sizeof(v), hipMemcpyDeviceToHost));      // blocking: a debugging / test call, not part of the step
    *status = (int32_t)v;
    return 0;
}

int64_t kasf_packed_bytes(const kasf_model* m) { return m ? ((m->arena_elems * esize(m) + 255) & ~int64_t(255)) : 0; }
int kasf_pack_weights(const kasf_model* m, const float* params, void* packed, void* stream) {
    if (check_model(m)) return 2;
    if (m->d_pack == nullptr) return kasf_set_error(4, "layout-only model has no device tables");
    kasf_launch_pack(m->cfg.dtype, (hipStream_t)stream, params, packed, m->d_pack, m->d_tile_start, (int)m->pack.size(), m->pack_tiles);
    HIPCHK(hipGetLastError());
    return 0;
}

int64_t kasf_workspace_bytes(const kasf_model* m, int32_t batch, int32_t flags) {
    if (m == nullptr || batch < 1) return 0;
    Plan p;
    build_plan(m, batch, (flags & (KASF_FLAG_TRAIN | KASF_FLAG_KEEP)) != 0, false, p);
    return p.total;
}
int32_t kasf_ws_entries(const kasf_model* m, int32_t batch, int32_t flags) {
    if (m == nullptr || batch < 1) return 0;
    Plan p;
    build_plan(m, batch, (flags & (KASF_FLAG_TRAIN | KASF_FLAG_KEEP)) != 0, true, p);
    return (int32_t)p.entries.size();
}
int kasf_ws_entry(const kasf_model* m, int32_t batch, int32_t flags, int32_t idx, char* name, int32_t cap, int64_t* byte_offset, int64_t* numel,
                  int32_t* elem_kind) {
    if (check_model(m)) return 2;
    Plan p;
    build_plan(m, batch, (flags & (KASF_FLAG_TRAIN | KASF_FLAG_KEEP)) != 0, true, p);
    if (idx < 0 || idx >= (int32_t)p.entries.size()) return kasf_set_error(2, "workspace entry index out of range");
    const WsEntry& e = p.entries[idx];
    if (name != nullptr && cap > 0) { strncpy(name, e.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (byte_offset) *byte_offset = e.off;
    if (numel) *numel = e.numel;
    if (elem_kind) *elem_kind = e.kind;
    return 0;
}

// Which branch of a layer runs on the CALLER's stream (the other two go to the side streams).  The gate / column-finish kernels are launched on the caller's stream behind the
// joins, and a join costs 12-18 us from the end of a side stream's last kernel to the start of the waiting kernel (event record -> barrier packet -> dispatch; three-stream trace,
// round 5: 1.15 ms of a 57 ms step sit in these hand-overs).  In the BACKWARD pass the bone branch finishes last in 26 of 26 layers (T = 27 and 81, B = 32 ... 256): on the caller's
// stream its last kernel is followed by the stage's finishing kernels without a hand-over, and the shorter branches pay the fork / join latency inside their slack: 4,681 against
// 4,643 clips/s, same box, alternating runs (tools/main_branch_sweep.sh).  In the FORWARD pass the graph branch finishes last (24-26 of 26 layers) -- and moving it to the caller's
// stream measured 1 % SLOWER (4,632 with bone-in-backward against 4,681): which hardware queue a branch's kernels come from matters as much as the hand-over (DESIGN section 6, round 4,
// stream placement), so the forward keeps the attention branch there.  KASF_MAIN_BRANCH_FWD / _BWD = 0 | 1 | 2 override.  Results do not depend on the choice
// (tests/test_gpu_determinism.py: one stream == three streams, bit for bit).
static int main_branch(bool backward) {
    struct MB { int v[2]; };
    static const MB mb = [] {                            // function-local static: initialised once, thread-safe (models may be driven from different host threads)
        const char* b = getenv("KASF_MAIN_BRANCH_BWD");
        const char* f = getenv("KASF_MAIN_BRANCH_FWD");
        return MB{{(f && *f >= '0' && *f <= '2') ? *f - '0' : 0, (b && *b >= '0' && *b <= '2') ? *b - '0' : 2}};
    }();
    return mb.v[backward ? 1 : 0];
}
static inline int side_index(int br, int mainbr) { return br < mainbr ? br : br - 1; }      // 0 / 1: which side stream (and join event) a non-main branch uses

int kasf_forward(const kasf_model* m, const float* params, const void* packed, float* buffers, const float* x, float* out, void* workspace,
                 int64_t workspace_bytes, int32_t batch, int32_t flags, void* stream) {
    struct ModelPath { explicit ModelPath(int v) { kasf_tls_model_path = v; } ~ModelPath() { kasf_tls_model_path = 0; } } model_path((flags & (KASF_FLAG_TRAIN | KASF_FLAG_KEEP)) != 0 ? 1 : 2);      // grid widths of the persistent launches (kernels.h): 1 training step, 2 forward only
    if (check_model(m)) return 2;
    if (!params || !packed || !buffers || !x || !out || !workspace) return kasf_set_error(2, "null pointer argument");
    if (m->d_pro == nullptr) return kasf_set_error(4, "layout-only model cannot run");
    if (batch < 1 || (int64_t)batch * m->cfg.n_frames * 17 * 384 >= ((int64_t)1 << 31))
        return kasf_set_error(2, "batch: 1 <= batch and batch * n_frames * 17 * 384 < 2^31 (32-bit element offsets in the kernels)");
    const bool bn_train = (flags & KASF_FLAG_TRAIN) != 0;
    const bool train = bn_train || (flags & KASF_FLAG_KEEP) != 0;        // keep every layer's activations
    Plan p;
    build_plan(m, batch, train, false, p);
    if (workspace_bytes < p.total) return kasf_set_error(5, "workspace too small (see kasf_workspace_bytes)");
    Ctx c{m, params, (const char*)packed, buffers, nullptr, (char*)workspace, (hipStream_t)stream, batch, m->cfg.n_frames, m->cfg.dtype, esize(m),
          (int64_t)batch * m->cfg.n_frames * 17, train, bn_train, nullptr};
    g_err.clear();
    const bool one_stream = single_stream();
    HIPCHK(hipMemsetAsync(c.w(p.stats_begin), 0, p.stats_bytes, c.s));
    if (train) HIPCHK(hipMemcpyAsync(c.w(p.x3), x, c.M * 3 * sizeof(float), hipMemcpyDeviceToDevice, c.s));
    kasf_launch_prologue_fwd(c.dt, c.s, x, params, m->d_pro, c.w(p.xj), c.w(p.xb), c.w(p.xl), (float*)c.w(p.bone3), (float*)c.w(p.limb3),
                             (int64_t)batch * c.T);
    const void* xcur = c.w(p.xj);
    const int L = m->cfg.n_layers;
    for (int l = 0; l < L; ++l) {
        const LayerOff& lo = m->layers[l];
        const LayerWs& lw = p.layers[train ? l : 0];
        if (!train && l > 0) HIPCHK(hipMemsetAsync(c.w(p.stats_begin), 0, p.stats_bytes, c.s));
        const int mainbr = main_branch(false);
        HIPCHK(hipEventRecord(m->ev_fork, c.s));
        for (int k3 = 0; k3 < 3; ++k3) {
            const int br = (mainbr + k3) % 3;            // (the caller's stream's branch is enqueued first: see kasf_backward)
            Ctx cb = c;
            if (br != mainbr) {
                cb.s = one_stream ? c.s : m->side[side_index(br, mainbr)];
                if (!one_stream) HIPCHK(hipStreamWaitEvent(cb.s, m->ev_fork, 0));
            }
            const void* in0 = (br == 2 && l == 0) ? c.w(p.xb) : xcur;         // layer 0: bone branch starts from the bone embedding (:332-336)
            block_forward(cb, lo.blk[2 * br], lw.b[2 * br], in0, c.w(p.xl), p, p.sc[br]);
            block_forward(cb, lo.blk[2 * br + 1], lw.b[2 * br + 1], c.w(lw.b[2 * br].x_out), c.w(p.xl), p, p.sc[br]);
            if (br != mainbr && !one_stream) HIPCHK(hipEventRecord(m->ev_join[side_index(br, mainbr)], cb.s));
        }
        if (!one_stream)                                 // the joins go behind the caller's stream's OWN branch (enqueued above whatever its index)
            for (int k = 0; k < 2; ++k) HIPCHK(hipStreamWaitEvent(c.s, m->ev_join[k], 0));
        kasf_launch_gate_fwd(c.dt, c.s, c.w(lw.b[1].x_out), c.w(lw.b[3].x_out), c.w(lw.b[5].x_out), params + lo.fus_w, params + lo.fus_b, c.w(lw.gate_out),
                             (float*)c.w(lw.alpha), c.M, m->cfg.use_adaptive_fusion);
        xcur = c.w(lw.gate_out);
    }
    const TopOff& t = m->top;
    kasf_launch_linear(c.dt, c.s, xcur, 128, c.pk(t.p_fc), 128, params + t.fc_b, c.w(p.rep), 512, c.M, 512, params + t.norm_w, params + t.norm_b, nullptr, 1);
    if (flags & KASF_FLAG_RETURN_REP) kasf_launch_cast_to_f32(c.dt, c.s, c.w(p.rep), out, c.M * 512);
    else kasf_launch_head_fwd(c.dt, c.s, c.w(p.rep), params + t.head_w, params + t.head_b, out, c.M);
    HIPCHK(hipGetLastError());
    if (!g_err.empty()) return 3;
    return 0;
}

int kasf_backward(const kasf_model* m, const float* params, const void* packed, const float* dout, float* grads, void* workspace, int64_t workspace_bytes,
                  int32_t batch, int32_t flags, int32_t stage_begin, int32_t stage_end, void* stream) {
    struct ModelPath { ModelPath() { kasf_tls_model_path = 1; } ~ModelPath() { kasf_tls_model_path = 0; } } model_path;      // grid widths of the persistent launches (kernels.h)
    if (check_model(m)) return 2;
    if (!params || !packed || !dout || !grads || !workspace) return kasf_set_error(2, "null pointer argument");
    const int L = m->cfg.n_layers;
    if (stage_begin < 0 || stage_end > L + 2 || stage_begin > stage_end) return kasf_set_error(2, "bad stage range");
    Plan p;
    build_plan(m, batch, true, false, p);
    if (workspace_bytes < p.total) return kasf_set_error(5, "workspace too small (see kasf_workspace_bytes)");
    Ctx c{m, params, (const char*)packed, nullptr, grads, (char*)workspace, (hipStream_t)stream, batch, m->cfg.n_frames, m->cfg.dtype, esize(m),
          (int64_t)batch * m->cfg.n_frames * 17, true, (flags & KASF_FLAG_TRAIN) != 0, nullptr};
    g_err.clear();
    const bool one_stream = single_stream();
    const TopOff& t = m->top;
    // one sink of per-workgroup column sums per branch stream; flushed (fixed-order finish, k_reduce.hip) on the caller's stream at the end of every stage
    KasfColSink sinks[3];
    KasfColSink* sink_ptrs[3] = {&sinks[0], &sinks[1], &sinks[2]};
    for (int br = 0; br < 3; ++br) { sinks[br].scratch = (float*)c.w(p.sc[br].col); sinks[br].cap = p.col_cap; }
    c.sink = &sinks[0];
    for (int st = stage_begin; st < stage_end; ++st) {
        // the running gradient w.r.t. the current layer's output alternates between two buffers
        auto gbuf = [&](int k) { return c.w((k & 1) ? p.g_prev : p.g_layer); };
        if (st == 0) {
            HIPCHK(hipMemsetAsync(c.w(p.bstats_begin), 0, p.bstats_bytes, c.s));
            HIPCHK(hipMemsetAsync(c.w(p.g_limb), 0, c.M * 128 * c.es, c.s));
            const void* x_final = c.w(p.layers[L - 1].gate_out);
            if (flags & KASF_FLAG_RETURN_REP)      // the forward returned the tanh features: dout is [B,T,17,512], the head took no part
                kasf_launch_rep_bwd(c.dt, c.s, dout, c.w(p.rep), c.w(p.sc[0].hbuf), c.M);
            else
                kasf_launch_head_bwd(c.dt, c.s, dout, c.w(p.rep), params + t.head_w, c.w(p.sc[0].hbuf), grads + t.head_w, grads + t.head_b, c.M, c.sink);
            kasf_launch_dgrad_lnbwd(c.dt, c.s, c.w(p.sc[0].hbuf), 512, c.pk(t.p_fcT), nullptr, x_final, params + t.norm_w, nullptr, gbuf(0), 0, grads + t.norm_w,
                                    grads + t.norm_b, c.M, c.w(p.sc[0].xn_a), params + t.norm_b, c.sink);
            kasf_launch_wgrad(c.dt, c.s, c.w(p.sc[0].hbuf), 512, 512, c.w(p.sc[0].xn_a), 128, 128, nullptr, nullptr, grads + t.fc_w, 128, grads + t.fc_b, c.M,
                              (float*)c.w(p.sc[0].wg_part), WG_PARTIAL_FLOATS);
        } else if (st <= L) {
            const int l = L - st;
            const LayerOff& lo = m->layers[l];
            const LayerWs& lw = p.layers[l];
            const void* x_in = l == 0 ? c.w(p.xj) : c.w(p.layers[l - 1].gate_out);
            void* g_out = gbuf(st - 1);
            void* g_in = gbuf(st);
            // incoming gradient: the head's for the top layer, otherwise the three branch input gradients the layer above left behind (summed in-kernel)
            const bool top = l == L - 1;
            kasf_launch_gate_bwd(c.dt, c.s, top ? g_out : c.w(p.sc[0].g_in), top ? nullptr : c.w(p.sc[1].g_in), top ? nullptr : c.w(p.sc[2].g_in),
                                 c.w(lw.b[1].x_out), c.w(lw.b[3].x_out), c.w(lw.b[5].x_out), params + lo.fus_w, (const float*)c.w(lw.alpha), c.w(p.ga), c.w(p.gg),
                                 c.w(p.gb), grads + lo.fus_w, grads + lo.fus_b, c.M, m->cfg.use_adaptive_fusion, c.sink);
            const int64_t gsrc[3] = {p.ga, p.gg, p.gb};
            const int mainbr = main_branch(true);
            HIPCHK(hipEventRecord(m->ev_fork, c.s));
            for (int k3 = 0; k3 < 3; ++k3) {
                const int br = (mainbr + k3) % 3;          // the caller's stream's branch is ENQUEUED first: with a slow host (a profiler attached, a busy CPU) the stream the joins wait
                Ctx cb = c;                              // on must not be the last one to receive its work (under rocprofv3: 67 against 57 ms per step; unprofiled: no difference)
                cb.sink = &sinks[br];
                if (br != mainbr) {
                    cb.s = one_stream ? c.s : m->side[side_index(br, mainbr)];
                    if (!one_stream) HIPCHK(hipStreamWaitEvent(cb.s, m->ev_fork, 0));
                }
                const Scratch& sc = p.sc[br];
                const bool bone0 = (br == 2 && l == 0);
                const void* in0 = bone0 ? c.w(p.xb) : x_in;
                block_backward(cb, lo.blk[2 * br + 1], lw.b[2 * br + 1], c.w(lw.b[2 * br].x_out), c.w(p.xl), c.w(gsrc[br]), c.w(sc.t1), 0, p, sc);
                block_backward(cb, lo.blk[2 * br], lw.b[2 * br], in0, c.w(p.xl), c.w(sc.t1), bone0 ? c.w(p.g_bone) : c.w(sc.g_in), 0, p, sc);
                if (br != mainbr && !one_stream) HIPCHK(hipEventRecord(m->ev_join[side_index(br, mainbr)], cb.s));
            }
            if (!one_stream)
                for (int k = 0; k < 2; ++k) HIPCHK(hipStreamWaitEvent(c.s, m->ev_join[k], 0));
            // gradient w.r.t. the layer input = sum over the branches (layer 0: the bone branch fed on the bone embedding instead)
            // Only the bottom layer materialises the sum (for the embedding backward); elsewhere the next gate_bwd adds the three on the fly.
            if (l == 0) kasf_launch_add3(c.dt, c.s, g_in, c.w(p.sc[0].g_in), c.w(p.sc[1].g_in), nullptr, c.M * 128);
        } else {
            void* g_x = gbuf(L);
            const int64_t frames = (int64_t)batch * c.T;
            const KasfProOff& po = m->pro;
            const float* src[3] = {(const float*)c.w(p.x3), (const float*)c.w(p.bone3), (const float*)c.w(p.limb3)};
            const void* gs[3] = {g_x, c.w(p.g_bone), c.w(p.g_limb)};
            // the three embedding backward passes are independent: joints and bone on the side streams, limb (+ the limb-refusion MLPs that
            // consume its input gradient) on the caller's stream
            if (!one_stream) HIPCHK(hipEventRecord(m->ev_fork, c.s));
            for (int sidx = 0; sidx < 3; ++sidx) {
                hipStream_t st = (sidx == 2 || one_stream) ? c.s : m->side[sidx];
                if (st != c.s) HIPCHK(hipStreamWaitEvent(st, m->ev_fork, 0));
                kasf_launch_embed_bwd(c.dt, st, gs[sidx], src[sidx], params + po.embed_w[sidx], grads + po.embed_w[sidx], grads + po.embed_b[sidx],
                                      grads + po.pos[sidx], sidx == 2 ? (float*)c.w(p.dlimb3) : nullptr, frames, &sinks[sidx]);
                if (st != c.s) HIPCHK(hipEventRecord(m->ev_join[sidx], st));
            }
            // the 204 limb-MLP tensors occupy one contiguous range of the gradient array (build_layout): a scratch row mirrors it
            const int64_t rf_base = po.mlp[0][0], rf_end = po.mlp[50][3] + 1;
            kasf_launch_refusion_bwd(c.s, (const float*)c.w(p.x3), (const float*)c.w(p.dlimb3), params, grads, m->d_pro, frames, &sinks[2], rf_base,
                                     (int)(rf_end - rf_base));
            if (!one_stream)
                for (int sidx = 0; sidx < 2; ++sidx) HIPCHK(hipStreamWaitEvent(c.s, m->ev_join[sidx], 0));
        }
        kasf_col_flush(c.s, sink_ptrs, 3);          // every stream of the stage has joined the caller's: finish its per-channel gradients in a fixed order
    }
    HIPCHK(hipGetLastError());
    if (!g_err.empty()) return 3;
    return 0;
}

int kasf_loss3(const float* pred, const float* target, float* dpred, float* losses, int64_t losses_floats, int32_t batch, int32_t n_frames,
               float lambda_n_mpjpe, float lambda_velocity, float grad_scale, void* stream) {
    if (!pred || !target || !dpred || !losses) return kasf_set_error(2, "null pointer argument");
    if (batch < 1 || n_frames < 1) return kasf_set_error(2, "loss3: batch and n_frames must be positive");
    if (losses_floats < 4 + 4 * (int64_t)batch) return kasf_set_error(5, "loss3: `losses` must hold 4 + 4 * batch floats (the per-clip sums live behind the result)");
    kasf_launch_loss3((hipStream_t)stream, pred, target, dpred, losses, batch, n_frames, lambda_n_mpjpe, lambda_velocity, grad_scale);
    HIPCHK(hipGetLastError());
    return 0;
}

int kasf_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int32_t step_index, float grad_scale, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq) return kasf_set_error(2, "null pointer argument");
    if (n % 4 != 0 || step_index < 1) return kasf_set_error(2, "n must be a multiple of 4 and step_index >= 1");
    // torch.optim.AdamW forms the bias corrections in double (1 - 0.999^t cancels badly in fp32 at small t)
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step_index)), bc2 = (float)(1.0 - pow((double)beta2, (double)step_index));
    kasf_launch_adamw((hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
    HIPCHK(hipGetLastError());
    return 0;
}

int kasf_gather_clips(const float* x_all, const float* y_all, const int64_t* index, const uint8_t* flip, int64_t n_clips, int32_t batch,
                      int32_t n_frames, float* x_out, float* y_out, void* stream) {
    if (!x_all || !index || !x_out || (y_all && !y_out)) return kasf_set_error(2, "null pointer argument");
    if (n_clips < 1 || n_frames < 1) return kasf_set_error(2, "gather_clips: empty clip set");
    kasf_launch_gather_clips((hipStream_t)stream, x_all, y_all, index, flip, n_clips, batch, n_frames, x_out, y_out);
    HIPCHK(hipGetLastError());
    return 0;
}
int kasf_joint_flip(const float* src, float* dst, int64_t rows, void* stream) {
    if (!src || !dst || src == dst) return kasf_set_error(2, "joint_flip: null or aliased pointers");
    kasf_launch_joint_flip((hipStream_t)stream, src, dst, rows);
    HIPCHK(hipGetLastError());
    return 0;
}
int kasf_tta_merge(const float* pred, const float* pred_of_flipped, float* out, int64_t rows, void* stream) {
    if (!pred || !out) return kasf_set_error(2, "null pointer argument");
    kasf_launch_tta_merge((hipStream_t)stream, pred, pred_of_flipped, out, rows);
    HIPCHK(hipGetLastError());
    return 0;
}
int kasf_eval_metrics(const float* pred, const float* label_scaled, const float* factor, const float* res, const int32_t* action, int32_t batch,
                      int32_t n_frames, int32_t n_actions, float* mpjpe, float* p_mpjpe, float* accel, float* jpe, double* action_sums, void* stream) {
    if (!pred || !label_scaled || !factor || !res || !mpjpe || !p_mpjpe || !accel || !jpe) return kasf_set_error(2, "null pointer argument");
    if (n_frames < 3 || n_frames > 256) return kasf_set_error(2, "eval_metrics: n_frames must be in [3,256]");
    if ((action == nullptr) != (action_sums == nullptr)) return kasf_set_error(2, "eval_metrics: action and action_sums go together");
    g_err.clear();
    kasf_launch_eval_metrics((hipStream_t)stream, pred, label_scaled, factor, res, action, batch, n_frames, n_actions, mpjpe, p_mpjpe, accel, jpe, action_sums);
    HIPCHK(hipGetLastError());
    return g_err.empty() ? 0 : 3;
}

#define OP_DT_CHECK(dt) \
    if ((dt) != KASF_F32 && (dt) != KASF_BF16) return kasf_set_error(3, "bad dtype")

int kasf_op_linear(int32_t dtype, const void* a, const void* w, const float* bias, void* y, int64_t M, int32_t N, const float* ln_g, const float* ln_b,
                   void* xn_out, int32_t act, void* stream) {
    OP_DT_CHECK(dtype);
    if (N % 128 != 0) return kasf_set_error(2, "N must be a multiple of 128");
    kasf_launch_linear(dtype, (hipStream_t)stream, a, 128, w, 128, bias, y, N, M, N, ln_g, ln_b, xn_out, act);
    HIPCHK(hipGetLastError());
    return 0;
}
int kasf_op_mlp_fwd(int32_t dtype, const void* x, const float* ln_g, const float* ln_b, const void* w1, const float* b1, const void* w2, const float* b2,
                    const float* ls2, void* out, int64_t M, void* xn_out, void* stream) {
    OP_DT_CHECK(dtype);
    kasf_launch_mlp_fwd(dtype, (hipStream_t)stream, x, ln_g, ln_b, w1, b1, w2, b2, ls2, out, M, xn_out);
    HIPCHK(hipGetLastError());
    return 0;
}
int kasf_op_mlp_bwd(int32_t dtype, const void* x, const void* g, const float* ln_g, const float* ln_b, const void* w1, const float* b1,
                    const void* w2t_scaled, const void* w1t, void* hbuf, void* dzbuf, void* g_in, float* dgamma, float* dbeta, int64_t M, void* stream) {
    OP_DT_CHECK(dtype);
    kasf_launch_mlp_bwd(dtype, (hipStream_t)stream, x, g, ln_g, ln_b, w1, b1, w2t_scaled, w1t, hbuf, dzbuf, nullptr, g_in, dgamma, dbeta, M);
    HIPCHK(hipGetLastError());
    return 0;
}
int kasf_op_mlp_bwd_fused(const void* x, const void* xn, const void* g, const float* ln_g, const void* w1, const float* b1, const void* w2t_scaled,
                          const void* w1t, void* dapart, float* partial, float* dw1, float* dw2_unscaled, float* db1, float* gsum, void* g_in,
                          float* dgamma, float* dbeta, int64_t M, void* stream) {
    if (!x || !xn || !g || !dapart || !partial || !dw1 || !dw2_unscaled || !db1 || !gsum || !g_in) return kasf_set_error(2, "null pointer argument");
    kasf_launch_mlp_bwd_q((hipStream_t)stream, x, xn, g, ln_g, w1, b1, w2t_scaled, w1t, dapart, partial, dw1, dw2_unscaled, db1, gsum, g_in, dgamma,
                          dbeta, M);
    HIPCHK(hipGetLastError());
    return 0;
}
int kasf_op_wgrad(int32_t dtype, const void* g, int32_t N, const void* x, int32_t K, const float* ln_g, const float* ln_b, float* dw, float* dbias,
                  int64_t M, float* partial, int64_t partial_floats, void* stream) {
    OP_DT_CHECK(dtype);
    if (N % 128 != 0 || K % 128 != 0) return kasf_set_error(2, "N and K must be multiples of 128");
    if (ln_g != nullptr && K != 128) return kasf_set_error(2, "LayerNorm-fused wgrad needs K == 128");
    kasf_launch_wgrad(dtype, (hipStream_t)stream, g, N, N, x, K, K, ln_g, ln_b, dw, K, dbias, M, partial, partial_floats);
    HIPCHK(hipGetLastError());
    return 0;
}
int kasf_op_dgrad_lnbwd(int32_t dtype, const void* dy, int32_t Kd, const void* wt, const void* dxn_add, const void* x, const float* gamma,
                        const void* resid, void* out, int32_t accumulate, float* dgamma, float* dbeta, int64_t M, void* xn_out, const float* beta,
                        void* stream) {
    OP_DT_CHECK(dtype);
    if (Kd % 128 != 0) return kasf_set_error(2, "Kd must be a multiple of 128");
    if (xn_out != nullptr && beta == nullptr) return kasf_set_error(2, "xn_out needs beta");
    kasf_launch_dgrad_lnbwd(dtype, (hipStream_t)stream, dy, Kd, wt, dxn_add, x, gamma, resid, out, accumulate, dgamma, dbeta, M, xn_out, beta);
    HIPCHK(hipGetLastError());
    return 0;
}
int kasf_op_attention_fwd(int32_t dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int32_t batch, int32_t n_frames,
                          int32_t mode, void* stream) {
    OP_DT_CHECK(dtype);
    g_err.clear();
    kasf_launch_attn_fwd(dtype, (hipStream_t)stream, q, ldq, k, v, ldkv, o, batch, n_frames, mode);
    HIPCHK(hipGetLastError());
    return g_err.empty() ? 0 : 3;
}
int kasf_op_attention_bwd(int32_t dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq, int64_t lddq,
                          void* dk, void* dv, int64_t lddkv, int32_t batch, int32_t n_frames, int32_t mode, void* stream) {
    OP_DT_CHECK(dtype);
    g_err.clear();
    kasf_launch_attn_bwd(dtype, (hipStream_t)stream, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, batch, n_frames, mode);
    HIPCHK(hipGetLastError());
    return g_err.empty() ? 0 : 3;
}
int kasf_op_attention_fwd_heads(int32_t dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int32_t batch, int32_t n_frames,
                                int32_t mode, int32_t num_heads, void* stream) {
    OP_DT_CHECK(dtype);
    g_err.clear();
    kasf_launch_attn_fwd(dtype, (hipStream_t)stream, q, ldq, k, v, ldkv, o, batch, n_frames, mode, num_heads);
    HIPCHK(hipGetLastError());
    return g_err.empty() ? 0 : 3;
}
int kasf_op_attention_bwd_heads(int32_t dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq, int64_t lddq,
                                void* dk, void* dv, int64_t lddkv, int32_t batch, int32_t n_frames, int32_t mode, int32_t num_heads, void* stream) {
    OP_DT_CHECK(dtype);
    g_err.clear();
    kasf_launch_attn_bwd(dtype, (hipStream_t)stream, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, batch, n_frames, mode, num_heads);
    HIPCHK(hipGetLastError());
    return g_err.empty() ? 0 : 3;
}
int kasf_op_attention_bwd_fused_do(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* g_mid, const void* wproj_t_scaled, void* dq,
                                   int64_t lddq, void* dk, void* dv, int64_t lddkv, int32_t batch, int32_t n_frames, int32_t mode, int32_t form, const void* o_saved,
                                   const float* lse, void* stream) {
    g_err.clear();
    if (form < 0 || form > 1) return kasf_set_error(2, "form: 0 (persistent) or 1 (one group per workgroup)");
    if (batch < 1 || n_frames < 1 || (int64_t)batch * n_frames * 17 * 384 >= ((int64_t)1 << 31)) return kasf_set_error(2, "batch * n_frames * 17 * 384 must stay below 2^31");
    if (!kasf_launch_attn_bwd_fused_do((hipStream_t)stream, q, ldq, k, v, ldkv, g_mid, wproj_t_scaled, dq, lddq, dk, dv, lddkv, batch, n_frames, mode, form, o_saved, lse))
        return kasf_set_error(2, "fused-d_o attention backward: groups of at most 96 positions (form 1: at most 32); bf16, 8 heads");
    HIPCHK(hipGetLastError());
    return g_err.empty() ? 0 : 3;
}
int kasf_op_cast(int32_t dtype, const void* src, void* dst, int64_t n, int32_t to_f32, void* stream) {
    OP_DT_CHECK(dtype);
    if (to_f32) kasf_launch_cast_to_f32(dtype, (hipStream_t)stream, src, (float*)dst, n);
    else kasf_launch_cast_from_f32(dtype, (hipStream_t)stream, (const float*)src, dst, n);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
