// Fused attention block, BACKWARD (bf16, 8 heads, groups of at most 32 positions: every spatial block, temporal blocks up to T = 32) -- round 6.
//     forward (selfattention.py:18-57, bone_crossattention.py:19-62, KASportsFormer.py:103-110):
//         x_mid = x + ls1 * ( proj( softmax(q k^T / 4) v ) + b ),   q|k|v = LN(x) Wqkv^T        (bone: q = LN(x) Wq^T,  k|v = LN_limb(x_limb) Wkv^T)
//     this kernel, from g_mid = d(loss)/d(x_mid), x (and x_limb) ONLY:
//         d(loss)/dx  = g_mid + LNbwd( dq|dk|dv . Wqkv ),   dWqkv = (dq|dk|dv)^T LN(x),   G_proj = g_mid^T o,   d(gamma), d(beta), colsum(g_mid)
//         (bone: d(loss)/dx_limb += LNbwd_limb( dk|dv . Wkv ), dWq, dWkv and the limb LayerNorm's d(gamma), d(beta))
// Rounds 1-5 ran this as four launches that handed everything to each other through HBM: the forward SAVED q|k|v and o (1,024 B per token), k_attn_bwd_pers read
// them back and wrote dq|dk|dv (768 B), k_dgrad_r read those again with x and g_mid, k_wgrad_ring_jobs read g_mid and o a third time: 3,840 B per token and
// block in the backward pass + 1,024 B of saves in the forward, for 2.8 % of the model's FLOPs.  Here a group (the 17 joints of a frame / the T frames of a joint)
// stays on its CU from LayerNorm to the input gradient: 512 B in (x, g_mid), 256 B out per token; the forward saves nothing.
//
// Everything that has to persist across a workgroup's groups is 384 KB: the weight-gradient accumulators (dWqkv 384 x 128 + G_proj 128 x 128 in fp32: 256 KB),
// the data-gradient weights (96 KB) and the d_o weights (32 KB).  A CU's register file is 512 KB, so the kernel runs ONE wave per SIMD (4 waves, 512 registers
// each: 256 accumulator registers for the weight gradients, 128 vector registers of weights, the rest for the attention core) -- eight waves of 256 would leave
// 64 registers per wave for a core that needs ~84.  The forward projection weights (Wqkv, 96 KB) live in LDS as a swizzled [384][128] image.
//   wave w owns heads 2w, 2w + 1 (projection recompute, d_o, attention core: the 32x32x16 formulation of k_attn_mfma.hip), output channels [32w, 32w + 32) of the
//   data gradient, the 96 rows of dWqkv that belong to its two heads (their dq|dk|dv tiles are its own products) and the 32 columns of G_proj of its heads' o.
// LDS (self 153 KB / bone 160 KB): weight image 96 KB | LN(x) 8 KB (| LN_limb(x_limb) 8 KB) | g_mid 8 KB | per head q, k, v, d_o tiles [32][16] (overwritten in
// place by dq, dk, dv, o) 32 KB | one P / dS tile per wave 8 KB.  Four workgroup barriers per group.
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
constexpr int BF_THR = 256, BF_TILE = 32 * 128, BF_HT = 32 * 16;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ bf16x8 tok_frag(const bf16* s, int row, int ks, int lane) {
    const int g = lane >> 4;
    return *reinterpret_cast<const bf16x8*>(s + Tile<bf16>::chunk_off(row, 4 * ks + g));
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// position (key or query index inside its 32-tile) held by register `reg` of a 32x32 accumulator in lane half hh
__device__ __forceinline__ int pos_of(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }
__device__ __forceinline__ bf16x8 pack8(const f32x16& t, int s) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)t[8 * s + j];
    return o;
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const bf16x2_t t = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, t);
}
// lane = position, registers 0..7 of t = channels {4hh..4hh+3, 8+4hh..8+4hh+3}  ->  after the swap lane half hh holds channels 8hh .. 8hh+7 of its position
__device__ __forceinline__ u32x4_t swap_t16(const f32x16& t) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(pack2(t[0], t[1]), pack2(t[4], t[5]), false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(pack2(t[2], t[3]), pack2(t[6], t[7]), false, false);
    return u32x4_t{s0[0], s1[0], s0[1], s1[1]};
}
typedef __attribute__((address_space(3))) bf16x4 lds_v4;
// transposed fragment of k-step ks (16 positions) out of a row-major [32 positions][16 channels] head tile: element j of lane half hh = tile[kappa(ks,hh,j)][lane & 15]
// (the k order of a 32x32 accumulator used as the other operand: k_attn_mfma.hip)
__device__ __forceinline__ bf16x8 tr_frag(const bf16* s_tile, int ks, int lane) {
    const int u = lane & 15, hh = lane >> 5, q = u >> 2, p = u & 3;
    const int k0 = 16 * ks + 4 * hh;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (k0 + q) * 16 + 4 * p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (k0 + 8 + q) * 16 + 4 * p));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// [32 queries][32 keys] P / dS tile, 64-byte rows, the four 16-byte chunks of row r stored at chunk ^ ((r >> 1) & 3): the 8-byte row writes of 16 consecutive
// rows then spread over 8 bank pairs (2-way; unswizzled: 16 r mod 32 -> 8-way, 60 % of k_attn_bwd_pers' LDS cycles were conflicts: profiles/r6_attn_sq_counters_before.txt),
// and a transposed read still covers whole rows (conflict-free).
__device__ __forceinline__ int p_off(int r, int col) { return r * 32 + ((((col >> 3) ^ (r >> 1)) & 3) << 3) + (col & 7); }
__device__ __forceinline__ bf16x8 tr_frag32(const bf16* s_tile, int ks, int lane) {
    const int u = lane & 15, hh = lane >> 5, q = u >> 2, p = u & 3, c0 = 16 * ((lane >> 4) & 1);
    const int k0 = 16 * ks + 4 * hh;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + p_off(k0 + q, c0 + 4 * p)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + p_off(k0 + 8 + q, c0 + 4 * p)));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// transposed fragment for a 16x16x32 product whose reduction index is the POSITION: lane (u = lane & 15, g = lane >> 4) receives channel u of positions
// 8g + {0, 2, 4, 6, 1, 3, 5, 7} of a [32][16] head tile -- the same order frag_tr (tile_ops.h) delivers the rows of a [32][128] tile in
__device__ __forceinline__ bf16x8 tr_frag_pos(const bf16* s_tile, int lane) {
    const int u = lane & 15, g = lane >> 4, q = u >> 2, p = u & 3;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (8 * g + 2 * q) * 16 + 4 * p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (8 * g + 2 * q + 1) * 16 + 4 * p));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// 8 CONSECUTIVE rows mbase .. mbase + 7 of a swizzled [rows][128] tile, transposed: lane u of a 16-lane group receives column col0 + u (natural row order: the other
// operand of the data gradient is a plain 16-byte row piece, so the interleaved order of frag_tr does not apply; these reads are 2-way bank conflicted)
__device__ __forceinline__ bf16x8 frag_tr_nat(const bf16* s, int mbase, int col0, int lane) {
    const int u = lane & 15, q = u >> 2, p = u & 3;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s + eoff<bf16>(mbase + q, col0 + 4 * p)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s + eoff<bf16>(mbase + 4 + q, col0 + 4 * p)));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// frag_tr (tile_ops.h) with the lane id as an argument
__device__ __forceinline__ bf16x8 frag_tr_l(const bf16* s, int mbase, int col0, int lane) {
    const int u = lane & 15, q = u >> 2, p = u & 3;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s + eoff<bf16>(mbase + 2 * q, col0 + 4 * p)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s + eoff<bf16>(mbase + 2 * q + 1, col0 + 4 * p)));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // same-wave LDS write -> read ordering

struct AttnBwdFArgs {
    const bf16 *X, *XL, *G;                  // block input [M][128], limb stream (bone), g_mid [M][128]
    const float *ln_g, *ln_b, *lnl_g, *lnl_b;
    const bf16 *Wf, *Wkvf;                   // forward layout: self Wqkv [384][128]; bone Wq [128][128] + Wkv [256][128]
    const bf16 *WT, *WkvT;                   // transposed: self [128][384]; bone Wq^T [128][128] + Wkv^T [128][256]
    const bf16* Wp;                          // (ls1 . Wproj)^T [128][128]
    bf16 *OUT, *OUTL;                        // d/dx [M][128] (written); bone: d/dx_limb [M][128] (accumulated into)
    float* part;                             // [active][PLD] dgamma | dbeta (| limb dgamma | dbeta)
    bf16 *wpart, *wpart_kv, *ppart;          // [active][384 | 128][128], bone [active][256][128], [active][128][128] bf16 partial tiles
    float* pbrow;                            // [active][128] colsum(g_mid)
    int L, T, mode, groups;
};

template <bool BONE, int NR>
__global__ __launch_bounds__(BF_THR, 1) void k_attn_blk_bwd_f(const AttnBwdFArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sW = reinterpret_cast<bf16*>(smem);            // [384][128] forward projection weights (swizzled rows)
    bf16* sXn = sW + 384 * 128;                          // [32][128] LN(x); later the staging tile of the x data gradient
    bf16* sG = sXn + BF_TILE;                            // [32][128] g_mid (rows past L zero)
    bf16* sXl = sG + BF_TILE;                            // bone: [32][128] LN_limb(x_limb); later the staging tile of the limb data gradient
    bf16* sHead = sXl + (BONE ? BF_TILE : 0);            // [8 heads][q | k | v | d_o][32][16]
    bf16* sPall = sHead + 8 * 4 * BF_HT;                 // [4 waves][32][32]
    float* sLn = reinterpret_cast<float*>(sPall + 4 * 32 * 32);      // self: [2][128] gamma | beta (bone reads them from global: its LDS is full)
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Lane-derived indices are RE-DERIVED in every phase of every group from an opaque copy of the thread id (KASF_IDS): hipcc otherwise hoists the ~150 loop-invariant
    // LDS addresses of the loop body to kernel entry and spills them (first build: 280-480 spilled registers, 200 scratch loads per group); recomputing one is 2-4 VALU
    // instructions.  thread (rl, sub) owns the 16-byte chunk `sub` of rows rl and rl + 16 in the row-wise phases.
#define KASF_IDS                                                                                                                              \
    int lv_ = tid;                                                                                                                            \
    asm volatile("" : "+v"(lv_));                                                                                                             \
    const int lane = lv_ & 63, i = lane & 15, g = lane >> 4, r32 = lane & 31, hh = lane >> 5, rl = lv_ >> 4, sub = lv_ & 15;                 \
    (void)i; (void)g; (void)r32; (void)hh; (void)rl; (void)sub;
    const int lane = tid & 63, i = lane & 15, g = lane >> 4, rl = tid >> 4, sub = tid & 15;
    const int L = a.L;
    const int per = (a.groups + gridDim.x - 1) / gridDim.x;
    const int g0 = blockIdx.x * per;
    int ng = a.groups - g0;
    if (ng > per) ng = per;
    if (ng <= 0) return;                                 // workgroup-uniform
    bf16* sP = sPall + w * (32 * 32);

    // ---- one-time: the forward weight image (LDS-direct, swizzle on the source address), register-resident weights ----
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        const bf16* src = BONE ? (t < 4 ? a.Wf + t * BF_TILE : a.Wkvf + (t - 4) * BF_TILE) : a.Wf + t * BF_TILE;
        stage_tile_async<bf16, 32, BF_THR>(sW + t * BF_TILE, src, 128, 32);
    }
    bf16x8 wp[2][4];                                     // d_o weights: rows of (ls1 . Wproj)^T of this wave's two heads
#pragma unroll
    for (int hd = 0; hd < 2; ++hd)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wp[hd][ks] = *reinterpret_cast<const bf16x8*>(a.Wp + (int64_t)(16 * (2 * w + hd) + i) * 128 + 32 * ks + 8 * g);
    if (!BONE && tid < 128) { sLn[tid] = a.ln_g[tid]; sLn[128 + tid] = a.ln_b[tid]; }

    f32x4 accW[6][8];                                    // dW rows of (head 2w + hd, part p) = a = 3 hd + p, 16 rows each, x 8 column tiles
    f32x4 accP[8][2];                                    // G_proj[16 b + ..][16 (2w + hd) + ..]
    zero_acc(accW);
    zero_acc(accP);
    float dg[8], db[8], dgl[8], dbl[8], gcol[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[e] = 0.f; db[e] = 0.f; dgl[e] = 0.f; dbl[e] = 0.f; gcol[e] = 0.f; }

    const int stride = a.mode == 0 ? 1 : KASF_J;         // tokens between consecutive positions of a group
    auto base_of = [&](int G) { return a.mode == 0 ? G * KASF_J : (G / KASF_J) * a.T * KASF_J + (G % KASF_J); };
    unsigned ox[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = rl + 16 * j, rc = row < L ? row : L - 1;            // rows past L: clamped load, zeroed at the point of use
        ox[j] = (unsigned)(rc * stride) * 128u + sub * 8;
    }
    bf16x8 xN[2], gN[2], lN[2];
    auto fetch = [&](int t) {
        const unsigned b = (unsigned)base_of(g0 + t) * 128u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            xN[j] = *reinterpret_cast<const bf16x8*>(a.X + (size_t)(b + ox[j]));
            gN[j] = *reinterpret_cast<const bf16x8*>(a.G + (size_t)(b + ox[j]));
            if (BONE) lN[j] = *reinterpret_cast<const bf16x8*>(a.XL + (size_t)(b + ox[j]));
        }
    };
    // LayerNorm of one row chunk held in registers: writes LN(x) to the tile, returns the statistics the backward needs
    auto layernorm = [&](const bf16x8 raw, bf16* dst, int row, int sub, const float* gp, const float* bp, float& mean, float& rstd) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)raw[e];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        mean = reduce16(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] -= mean; q = __builtin_fmaf(v[e], v[e], q); }
        rstd = rsqrtf(reduce16(q) * (1.0f / 128.0f) + KASF_LN_EPS);
        const f32x4 g0v = *reinterpret_cast<const f32x4*>(gp + sub * 8), g1v = *reinterpret_cast<const f32x4*>(gp + sub * 8 + 4);
        const f32x4 b0v = *reinterpret_cast<const f32x4*>(bp + sub * 8), b1v = *reinterpret_cast<const f32x4*>(bp + sub * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = __builtin_fmaf(v[e], rstd * g0v[e], b0v[e]); v[4 + e] = __builtin_fmaf(v[4 + e], rstd * g1v[e], b1v[e]); }   // (the forward's arithmetic: k_attn_blk.hip)
        tile_store8(dst, row, sub * 8, v);
    };
    const float* gp = BONE ? a.ln_g : sLn;
    const float* bp = BONE ? a.ln_b : sLn + 128;

    {   // the P / dS tile starts zero: the 4-key pieces past NR are never written
        const bf16x8 z8 = {};
        *reinterpret_cast<bf16x8*>(sP + lane * 16) = z8;
        *reinterpret_cast<bf16x8*>(sP + lane * 16 + 8) = z8;
    }
    fetch(0);
    wait_async();
    __syncthreads();                                     // weight image and sLn complete
#pragma unroll
    for (int hd = 0; hd < 2; ++hd)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) touch_loaded(wp[hd][ks]);

    for (int t = 0; t < ng; ++t) {
        const int G = g0 + t;
        const bf16x8 zero = {};
        // ---------------- rows into LDS: LN(x), g_mid (| LN_limb(x_limb)); the raw x chunks and statistics stay in registers for the LayerNorm backward ----------------
        bf16x8 xr[2], lr[2];
        float mean[2], rstd[2], meanl[2], rstdl[2];
        {
        KASF_IDS
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = rl + 16 * j;
            const bool live = row < L;
            xr[j] = live ? xN[j] : zero;
            layernorm(xr[j], sXn, row, sub, gp, bp, mean[j], rstd[j]);
            *reinterpret_cast<bf16x8*>(sG + Tile<bf16>::chunk_off(row, sub)) = live ? gN[j] : zero;
            if (BONE) {
                lr[j] = live ? lN[j] : zero;
                layernorm(lr[j], sXl, row, sub, a.lnl_g, a.lnl_b, meanl[j], rstdl[j]);
            }
        }
        }
        __syncthreads();                                 // B1: the group's rows are in LDS; every thread is past the previous group's row-wise phase

        // ---------------- projections of this wave's two heads: q_h | k_h | v_h and d_o_h = g_mid (ls1 Wproj)^T -> head tiles ----------------
#pragma unroll
        for (int hd = 0; hd < 2; ++hd) {
            KASF_IDS
            const int h = 2 * w + hd;
            bf16* tq = sHead + h * 4 * BF_HT;
            f32x4 acc[3][2];
            zero_acc(acc);
            f32x4 accd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 f0 = tok_frag(sXn, i, ks, lane), f1 = tok_frag(sXn, 16 + i, ks, lane);
                const bf16x8 l0 = BONE ? tok_frag(sXl, i, ks, lane) : f0, l1 = BONE ? tok_frag(sXl, 16 + i, ks, lane) : f1;
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    const bf16x8 wa = tok_frag(sW, 128 * nt + 16 * h + i, ks, lane);
                    acc[nt][0] = mfma16(wa, nt == 0 ? f0 : l0, acc[nt][0]);
                    acc[nt][1] = mfma16(wa, nt == 0 ? f1 : l1, acc[nt][1]);
                }
                accd[0] = mfma16(wp[hd][ks], tok_frag(sG, i, ks, lane), accd[0]);
                accd[1] = mfma16(wp[hd][ks], tok_frag(sG, 16 + i, ks, lane), accd[1]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
                    store4(tq + nt * BF_HT + (16 * mt + i) * 16 + 4 * g, v);
                }
                const float lv = 16 * mt + i < L ? 1.0f : 0.0f;              // d_o rows past L are zero: those queries then give dS = 0 and add nothing to dK, dV
                float v[4] = {accd[mt][0] * lv, accd[mt][1] * lv, accd[mt][2] * lv, accd[mt][3] * lv};
                store4(tq + 3 * BF_HT + (16 * mt + i) * 16 + 4 * g, v);
            }
        }
        lds_fence();

        // ---------------- attention core of each head (k_attn_bwd_pers' arithmetic), o recomputed; dq | dk | dv | o overwrite q | k | v | d_o ----------------
#pragma unroll 1
        for (int hd = 0; hd < 2; ++hd) {
            KASF_IDS
            bf16* sQ = sHead + (2 * w + hd) * 4 * BF_HT;
            bf16* sK = sQ + BF_HT;
            bf16* sV = sK + BF_HT;
            bf16* sD = sV + BF_HT;
            const bf16x8 qf = *reinterpret_cast<const bf16x8*>(sQ + r32 * 16 + 8 * hh);
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + r32 * 16 + 8 * hh);
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + r32 * 16 + 8 * hh);
            const bf16x8 df = *reinterpret_cast<const bf16x8*>(sD + r32 * 16 + 8 * hh);
            // pass 1: lane = query
            f32x16 st = mfma32(kf, qf, zero16());        // S^T[key][query]
            f32x16 dp = mfma32(vf, df, zero16());        // dP^T[key][query]
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const float s = (pos_of(e, hh) < L) ? st[e] * 0.25f : -INFINITY;
                st[e] = s;
                mx = fmaxf(mx, s);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < NR; ++e) { st[e] = __expf(st[e] - mx); sum += st[e]; }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            float delta = 0.f;
#pragma unroll
            for (int e = 0; e < NR; ++e) { st[e] *= inv; delta += st[e] * dp[e]; }
#pragma unroll
            for (int e = NR; e < 16; ++e) st[e] = 0.f;
            delta += __shfl_xor(delta, 32);
            constexpr int NA4 = (NR + 3) / 4;
#pragma unroll
            for (int a4 = 0; a4 < NA4; ++a4) {           // P^T tile (the pieces past NR stay zero for the whole launch)
                float v4[4] = {st[4 * a4], st[4 * a4 + 1], st[4 * a4 + 2], st[4 * a4 + 3]};
                store4(sP + p_off(r32, 8 * a4 + 4 * hh), v4);
            }
            // o^T[d][query] = V^T . P^T (the forward's product, recomputed: the forward saves no attention output)
            f32x16 ot = mfma32(tr_frag(sV, 0, lane), pack8(st, 0), zero16());
            ot = mfma32(tr_frag(sV, 1, lane), pack8(st, 1), ot);
            const u32x4_t po = swap_t16(ot);
#pragma unroll
            for (int e = 0; e < NR; ++e) st[e] = st[e] * (dp[e] - delta) * 0.25f;      // dS^T (scale folded)
            f32x16 dq = mfma32(tr_frag(sK, 0, lane), pack8(st, 0), zero16());                  // dQ^T[d][query] = K^T . dS^T
            dq = mfma32(tr_frag(sK, 1, lane), pack8(st, 1), dq);
            // pass 2: lane = key.  dV^T = dO^T . P, dK^T = Q^T . dS with P / dS read back transposed from ONE tile (P first)
            lds_fence();
            f32x16 dv = mfma32(tr_frag(sD, 0, lane), tr_frag32(sP, 0, lane), zero16());
            dv = mfma32(tr_frag(sD, 1, lane), tr_frag32(sP, 1, lane), dv);
            lds_fence();                                 // the P fragments are in registers before dS overwrites the tile
#pragma unroll
            for (int a4 = 0; a4 < NA4; ++a4) {
                float v4[4] = {st[4 * a4], st[4 * a4 + 1], st[4 * a4 + 2], st[4 * a4 + 3]};
                store4(sP + p_off(r32, 8 * a4 + 4 * hh), v4);
            }
            lds_fence();
            f32x16 dk = mfma32(tr_frag(sQ, 0, lane), tr_frag32(sP, 0, lane), zero16());
            dk = mfma32(tr_frag(sQ, 1, lane), tr_frag32(sP, 1, lane), dk);
            const u32x4_t pq = swap_t16(dq), pv = swap_t16(dv), pk = swap_t16(dk);
            lds_fence();                                 // every read of the four tiles has returned
            *reinterpret_cast<u32x4_t*>(sQ + r32 * 16 + 8 * hh) = pq;
            *reinterpret_cast<u32x4_t*>(sK + r32 * 16 + 8 * hh) = pk;
            *reinterpret_cast<u32x4_t*>(sV + r32 * 16 + 8 * hh) = pv;
            *reinterpret_cast<u32x4_t*>(sD + r32 * 16 + 8 * hh) = po;
        }
        __syncthreads();                                 // B2: dq | dk | dv | o of all eight heads

        if (t + 1 < ng) fetch(t + 1);                    // the next group's rows travel under the GEMM phase
        // ---------------- data gradient: 32 output channels x 32 positions per wave, reduction over the 24 head tiles ----------------
        f32x4 accx[2][2], accl[2][2];
        zero_acc(accx);
        zero_acc(accl);
        {
            KASF_IDS
            // reduction index = (head tile, channel): k-step ks, lane group g -> tile tau = 2 ks + (g >> 1) (head tau & 7, part tau >> 3 = ks >> 2), channels 8 (g & 1) .. + 7:
            // the token operand is a 16-byte row piece of that head tile, the weight operand the same 8 rows of the LDS image read TRANSPOSED (column = output channel)
            const bf16* hb = sHead + (g >> 1) * 4 * BF_HT + 8 * (g & 1);
            const int nb = 16 * (g >> 1) + 8 * (g & 1);
#pragma unroll
            for (int ks = 0; ks < 12; ++ks) {
                const bf16* tp = hb + ((2 * ks) & 7) * 4 * BF_HT + (ks >> 2) * BF_HT;
                const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(tp + i * 16), f1 = *reinterpret_cast<const bf16x8*>(tp + (16 + i) * 16);
                const int n0 = 128 * (ks >> 2) + 16 * ((2 * ks) & 7) + nb;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const bf16x8 wa = frag_tr_nat(sW, n0, 32 * w + 16 * ct, lane);
                    if (BONE && ks >= 4) {
                        accl[ct][0] = mfma16(wa, f0, accl[ct][0]);
                        accl[ct][1] = mfma16(wa, f1, accl[ct][1]);
                    } else {
                        accx[ct][0] = mfma16(wa, f0, accx[ct][0]);
                        accx[ct][1] = mfma16(wa, f1, accx[ct][1]);
                    }
                }
            }
        }
        // ---------------- weight gradients (reduction over the group's positions: one k-step) ----------------
        {
            KASF_IDS
            bf16x8 ra[6], ro[2];
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                const bf16* tq = sHead + (2 * w + hd) * 4 * BF_HT;
#pragma unroll
                for (int p = 0; p < 3; ++p) ra[3 * hd + p] = tr_frag_pos(tq + p * BF_HT, lane);
                ro[hd] = tr_frag_pos(tq + 3 * BF_HT, lane);
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bf16x8 cx = frag_tr_l(sXn, 8 * g, 16 * b, lane);
                const bf16x8 cl = BONE ? frag_tr_l(sXl, 8 * g, 16 * b, lane) : cx;
                const bf16x8 cg = frag_tr_l(sG, 8 * g, 16 * b, lane);
#pragma unroll
                for (int aa = 0; aa < 6; ++aa) accW[aa][b] = mfma16(ra[aa], (aa % 3) == 0 ? cx : cl, accW[aa][b]);
                accP[b][0] = mfma16(cg, ro[0], accP[b][0]);
                accP[b][1] = mfma16(cg, ro[1], accP[b][1]);
            }
        }
        __syncthreads();                                 // B3: every wave is done with LN(x) (and the head tiles): the LN tiles become staging tiles
        {
        KASF_IDS
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float v[4] = {accx[ct][mt][0], accx[ct][mt][1], accx[ct][mt][2], accx[ct][mt][3]};
                store4(sXn + Tile<bf16>::off4(16 * mt + i, 32 * w + 16 * ct + 4 * g), v);
                if (BONE) {
                    float u[4] = {accl[ct][mt][0], accl[ct][mt][1], accl[ct][mt][2], accl[ct][mt][3]};
                    store4(sXl + Tile<bf16>::off4(16 * mt + i, 32 * w + 16 * ct + 4 * g), u);
                }
            }
        }
        __syncthreads();                                 // B4: staging tiles complete
        // ---------------- LayerNorm backward + residual, one 16-lane group per row; the thread's next-group writes go to the chunks it reads here ----------------
        {
            KASF_IDS
            float gm[8], gml[8];
            {
                const f32x4 g0v = *reinterpret_cast<const f32x4*>(gp + sub * 8), g1v = *reinterpret_cast<const f32x4*>(gp + sub * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { gm[e] = g0v[e]; gm[4 + e] = g1v[e]; }
                if (BONE) {
                    const f32x4 h0v = *reinterpret_cast<const f32x4*>(a.lnl_g + sub * 8), h1v = *reinterpret_cast<const f32x4*>(a.lnl_g + sub * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gml[e] = h0v[e]; gml[4 + e] = h1v[e]; }
                }
            }
            const unsigned b = (unsigned)base_of(G) * 128u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = rl + 16 * j;
                const bool live = row < L;
                float d[8], x[8], res[8], o[8];
                tile_load8(sXn, row, sub * 8, d);
                tile_load8(sG, row, sub * 8, res);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = ((float)xr[j][e] - mean[j]) * rstd[j];            // xhat
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (live) { dg[e] += d[e] * x[e]; db[e] += d[e]; gcol[e] += res[e]; }
                    d[e] *= gm[e];
                    s1 += d[e];
                    s2 += d[e] * x[e];
                }
                s1 = reduce16(s1) * (1.0f / 128.0f);
                s2 = reduce16(s2) * (1.0f / 128.0f);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd[j] * (d[e] - s1 - x[e] * s2) + res[e];
                if (live) store8(a.OUT + (size_t)(b + ox[j]), o);
                if (BONE) {
                    float dl[8], xl[8], ol[8], old[8];
                    tile_load8(sXl, row, sub * 8, dl);
                    if (live) load8(a.OUTL + (size_t)(b + ox[j]), old);
#pragma unroll
                    for (int e = 0; e < 8; ++e) xl[e] = ((float)lr[j][e] - meanl[j]) * rstdl[j];
                    float t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (live) { dgl[e] += dl[e] * xl[e]; dbl[e] += dl[e]; }
                        dl[e] *= gml[e];
                        t1 += dl[e];
                        t2 += dl[e] * xl[e];
                    }
                    t1 = reduce16(t1) * (1.0f / 128.0f);
                    t2 = reduce16(t2) * (1.0f / 128.0f);
                    if (live) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ol[e] = old[e] + rstdl[j] * (dl[e] - t1 - xl[e] * t2);
                        store8(a.OUTL + (size_t)(b + ox[j]), ol);
                    }
                }
            }
        }
    }

    // ================= end of the range: per-channel rows and the bf16 partial weight-gradient tiles of this workgroup =================
    __syncthreads();
    {
        float* sRed = reinterpret_cast<float*>(sW);      // [5][16][128] floats (40 KB of the dead weight image)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sRed[(0 * 16 + rl) * 128 + sub * 8 + e] = dg[e];
            sRed[(1 * 16 + rl) * 128 + sub * 8 + e] = db[e];
            sRed[(2 * 16 + rl) * 128 + sub * 8 + e] = gcol[e];
            if (BONE) { sRed[(3 * 16 + rl) * 128 + sub * 8 + e] = dgl[e]; sRed[(4 * 16 + rl) * 128 + sub * 8 + e] = dbl[e]; }
        }
        __syncthreads();
        constexpr int PLD = BONE ? 512 : 256;
        for (int idx = tid; idx < (BONE ? 5 : 3) * 128; idx += BF_THR) {
            const int which = idx >> 7, c = idx & 127;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += sRed[(which * 16 + k) * 128 + c];
            if (which == 2) a.pbrow[(int64_t)blockIdx.x * 128 + c] = s;
            else a.part[(int64_t)blockIdx.x * PLD + (which < 2 ? which : which - 1) * 128 + c] = s;
        }
        __syncthreads();
    }
    {
        bf16* sWst = sW;                                 // [384][128] dW image, row n = 128 part + 16 head + ..
        bf16* sPst = sHead;                              // [128][128] G_proj image
#pragma unroll
        for (int aa = 0; aa < 6; ++aa) {
            const int n0 = 128 * (aa % 3) + 16 * (2 * w + aa / 3);
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) sWst[(n0 + 4 * g + r) * 128 + 16 * b + i] = (bf16)accW[aa][b][r];
        }
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int hd = 0; hd < 2; ++hd)
#pragma unroll
                for (int r = 0; r < 4; ++r) sPst[(16 * b + 4 * g + r) * 128 + 16 * (2 * w + hd) + i] = (bf16)accP[b][hd][r];
        __syncthreads();
        if (BONE) {
            bf16* dq_ = a.wpart + (int64_t)blockIdx.x * (128 * 128);
            bf16* dkv_ = a.wpart_kv + (int64_t)blockIdx.x * (256 * 128);
            for (int c = tid; c < 128 * 16; c += BF_THR) *reinterpret_cast<f32x4*>(dq_ + c * 8) = *reinterpret_cast<const f32x4*>(sWst + c * 8);
            for (int c = tid; c < 256 * 16; c += BF_THR) *reinterpret_cast<f32x4*>(dkv_ + c * 8) = *reinterpret_cast<const f32x4*>(sWst + 128 * 128 + c * 8);
        } else {
            bf16* dst = a.wpart + (int64_t)blockIdx.x * (384 * 128);
            for (int c = tid; c < 384 * 16; c += BF_THR) *reinterpret_cast<f32x4*>(dst + c * 8) = *reinterpret_cast<const f32x4*>(sWst + c * 8);
        }
        bf16* dstp = a.ppart + (int64_t)blockIdx.x * (128 * 128);
        for (int c = tid; c < 128 * 16; c += BF_THR) *reinterpret_cast<f32x4*>(dstp + c * 8) = *reinterpret_cast<const f32x4*>(sPst + c * 8);
    }
}

template <bool BONE, int NR> int launch_bwd_f(hipStream_t s, AttnBwdFArgs a, KasfColSink* sink, float* dgamma, float* dbeta, float* dgamma_l, float* dbeta_l, int grid_cap) {
    constexpr int PLD = BONE ? 512 : 256;
    const size_t sh = (size_t)(384 * 128 + (BONE ? 3 : 2) * BF_TILE + 8 * 4 * BF_HT + 4 * 32 * 32) * 2 + (BONE ? 0 : 2 * 128 * 4);
    auto kern = k_attn_blk_bwd_f<BONE, NR>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
        kasf_set_error(3, "fused attention-block backward: cannot reserve its LDS");
        return 0;
    }
    const int grid = a.groups < grid_cap ? a.groups : grid_cap;
    const int per = (a.groups + grid - 1) / grid;
    const int active = (a.groups + per - 1) / per;       // workgroups that own at least one group (the others return at once)
    a.part = sink->take(active, PLD);
    if (a.part == nullptr) return 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BF_THR), sh, s, a);
    sink->add(a.part, PLD, active, 128, dgamma);
    sink->add(a.part + 128, PLD, active, 128, dbeta);
    if (BONE) {
        sink->add(a.part + 256, PLD, active, 128, dgamma_l);
        sink->add(a.part + 384, PLD, active, 128, dbeta_l);
    }
    return active;
}

}  // namespace

// Returns the number of partial tiles written (= active workgroups), 0 when the shape is not covered or the scratch is short (nothing launched, nothing registered).
// wpart: room for 256 x [384][128] (self) / 256 x [128][128] (bone q) bf16; wpart_kv: bone, 256 x [256][128]; ppart: 256 x [128][128]; pbrow: 256 x 128 floats.
int kasf_launch_attn_block_bwd(hipStream_t s, int bone, const void* x, const void* x_limb, const void* g_mid, const float* ln_g, const float* ln_b, const float* lnl_g,
                               const float* lnl_b, const void* Wf, const void* Wkvf, const void* WT, const void* WkvT, const void* WprojTs, void* out, void* out_limb,
                               float* dgamma, float* dbeta, float* dgamma_l, float* dbeta_l, KasfColSink* sink, void* wpart, void* wpart_kv, void* ppart, float* pbrow,
                               int B, int T, int mode) {
    const int L = mode == 0 ? KASF_J : T;
    if (L > 32 || sink == nullptr || wpart == nullptr || ppart == nullptr || pbrow == nullptr || (bone && (wpart_kv == nullptr || out_limb == nullptr))) return 0;
    AttnBwdFArgs a;
    a.X = (const bf16*)x; a.XL = (const bf16*)x_limb; a.G = (const bf16*)g_mid;
    a.ln_g = ln_g; a.ln_b = ln_b; a.lnl_g = lnl_g; a.lnl_b = lnl_b;
    a.Wf = (const bf16*)Wf; a.Wkvf = (const bf16*)Wkvf; a.WT = (const bf16*)WT; a.WkvT = (const bf16*)WkvT; a.Wp = (const bf16*)WprojTs;
    a.OUT = (bf16*)out; a.OUTL = (bf16*)out_limb; a.part = nullptr;
    a.wpart = (bf16*)wpart; a.wpart_kv = (bf16*)wpart_kv; a.ppart = (bf16*)ppart; a.pbrow = pbrow;
    a.L = L; a.T = T; a.mode = mode; a.groups = mode == 0 ? B * T : B * KASF_J;
    if (a.groups <= 0) return 0;
    const int cap = kasf_narrow_grid(KASF_NG_ATTN_BWD, 256, (int64_t)a.groups * L);
    if (bone) return L <= 17 ? launch_bwd_f<true, 9>(s, a, sink, dgamma, dbeta, dgamma_l, dbeta_l, cap) : launch_bwd_f<true, 16>(s, a, sink, dgamma, dbeta, dgamma_l, dbeta_l, cap);
    return L <= 17 ? launch_bwd_f<false, 9>(s, a, sink, dgamma, dbeta, dgamma_l, dbeta_l, cap) : launch_bwd_f<false, 16>(s, a, sink, dgamma, dbeta, dgamma_l, dbeta_l, cap);
}
