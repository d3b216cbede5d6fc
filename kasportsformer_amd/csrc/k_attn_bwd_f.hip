// Fused attention block, BACKWARD (bf16, 8 heads, groups of at most 32 positions: every spatial block, temporal blocks up to T = 32) -- round 6.
//     forward (selfattention.py:18-57, bone_crossattention.py:19-62, KASportsFormer.py:103-110):
//         x_mid = x + ls1 * ( proj( softmax(q k^T / 4) v ) + b ),   q|k|v = LN(x) Wqkv^T        (bone: q = LN(x) Wq^T,  k|v = LN_limb(x_limb) Wkv^T)
//     this kernel, from g_mid = d(loss)/d(x_mid), x (and x_limb) ONLY:
//         d(loss)/dx = g_mid + LNbwd( dq|dk|dv . Wqkv )  (bone: d(loss)/dx_limb += LNbwd_limb( dk|dv . Wkv )),  G_proj = g_mid^T o,  colsum(g_mid),  the LayerNorms' d(gamma), d(beta),
//         and -- for the streaming weight-gradient launch that follows -- dq|dk|dv and LN(x) (LN_limb(x_limb)) once each.
// Rounds 1-5 ran the block's backward as launches that handed everything to each other through HBM: the forward SAVED q|k|v and o (1,024 B per token), k_attn_bwd_pers read
// q|k|v and g_mid and wrote dq|dk|dv, k_dgrad_r read those again with x and g_mid, the weight-gradient launch read g_mid and o a third time: 3,840 B per token and block in the
// backward + 1,024 B of saves in the forward.  Here a group (the 17 joints of a frame / the T frames of a joint) stays on its CU from LayerNorm to the input gradient: the
// forward saves NOTHING, q|k|v and o are re-formed from x (98 kFLOP per token on an idle matrix pipe), dq|dk|dv reach the data gradient through LDS, and the proj weight gradient is
// accumulated here (32 registers).  HBM: 512 B in, 256 B out + 1,024 B (dq|dk|dv, LN(x)) for the one weight gradient that stays outside: dWqkv = (dq|dk|dv)^T LN(x) needs a
// 384 x 128 fp32 accumulator per workgroup -- 192 KB of a CU's 512 KB register file.  The first form of this kernel (4 waves x 512 registers, that accumulator inside; commit
// 9191739, profiles/r6_fused4w_single_stream_kernel_stats.txt) was parity-green and 2.5-4.5 x SLOWER than the four launches: one wave per SIMD issues a vector instruction every
// ~5 cycles and has nothing to cover its LDS round trips with.  This form keeps two waves per SIMD (8 waves, wave h = head h) and 224 free registers per wave.
//   wave h: projections q_h | k_h | v_h and d_o_h of the group (weights: LDS image, read as rows), the attention core of head h (the 32x32x16 formulation of k_attn_mfma.hip),
//   output channels [16h, 16h + 16) of the data gradient (weights: the SAME LDS image read transposed), columns [16h, 16h + 16) of G_proj.
// LDS (self 160 KB / bone 136 KB): Wqkv image 96 KB (bone: Wkv 64 KB; Wq lives in registers) | LN(x) 8 KB (| LN_limb(x_limb) 8 KB) | g_mid 8 KB | per head q, k, v, d_o tiles
// [32][16] (overwritten in place by dq, dk, dv, o) 32 KB | one P / dS tile per wave 16 KB.  Four workgroup barriers per group.
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
constexpr int BF_THR = 512, BF_TILE = 32 * 128, BF_HT = 32 * 16;

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

struct AttnBwdHArgs {
    const bf16 *X, *XL, *G;                  // block input [M][128], limb stream (bone), g_mid [M][128]
    const float *ln_g, *ln_b, *lnl_g, *lnl_b;
    const bf16 *Wf, *Wkvf;                   // forward layout: self Wqkv [384][128]; bone Wq [128][128] + Wkv [256][128]
    const bf16* WT;                          // bone: Wq^T [128][128] (the self form reads its image transposed instead)
    const bf16* Wp;                          // (ls1 . Wproj)^T [128][128]
    bf16 *OUT, *OUTL;                        // d/dx [M][128] (written); bone: d/dx_limb [M][128] (accumulated into)
    bf16 *DQ, *DKV;                          // dq|dk|dv [M][384] (bone: dq [M][128], dk|dv [M][256]) for the streaming weight gradient
    bf16 *XNA, *XNB;                         // LN(x) [M][128] (bone: and LN_limb(x_limb)) for the same
    float* part;                             // [active][PLD] dgamma | dbeta (| limb dgamma | dbeta)
    bf16* ppart;                             // [active][128][128] bf16 partial tiles of G_proj
    float* pbrow;                            // [active][128] colsum(g_mid)
    int L, T, mode, groups;
};

template <bool BONE, int NR>
__global__ __launch_bounds__(BF_THR, 2) void k_attn_blk_bwd_h(const AttnBwdHArgs a) {
    constexpr int WROWS = BONE ? 256 : 384;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sW = reinterpret_cast<bf16*>(smem);            // [WROWS][128] forward projection weights (swizzled rows)
    bf16* sXn = sW + WROWS * 128;                        // [32][128] LN(x); later the staging tile of the x data gradient
    bf16* sG = sXn + BF_TILE;                            // [32][128] g_mid (rows past L zero)
    bf16* sXl = sG + BF_TILE;                            // bone: [32][128] LN_limb(x_limb); later the staging tile of the limb data gradient
    bf16* sHead = sXl + (BONE ? BF_TILE : 0);            // [8 heads][q | k | v | d_o][32][16]
    bf16* sPall = sHead + 8 * 4 * BF_HT;                 // [8 waves][32][32]
    const int tid = threadIdx.x, h = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & 15, g = lane >> 4, r32 = lane & 31, hh = lane >> 5, q = i >> 2, p = i & 3;
    const int rl = tid >> 4, sub = tid & 15;             // thread (rl, sub) owns the 16-byte chunk `sub` of row rl in the row-wise phases
    const int L = a.L;
    const int per = (a.groups + gridDim.x - 1) / gridDim.x;
    const int g0 = blockIdx.x * per;
    int ng = a.groups - g0;
    if (ng > per) ng = per;
    if (ng <= 0) return;                                 // workgroup-uniform
    bf16* sP = sPall + h * (32 * 32);
    bf16* tQ = sHead + h * 4 * BF_HT;                    // this head's q | k | v | d_o tiles
    bf16* tK = tQ + BF_HT;
    bf16* tV = tK + BF_HT;
    bf16* tD = tV + BF_HT;

    // ---- one-time: the forward weight image (LDS-direct, swizzle on the source address), register-resident weights ----
#pragma unroll
    for (int t = 0; t < WROWS / 32; ++t) stage_tile_async<bf16, 32, BF_THR>(sW + t * BF_TILE, (BONE ? a.Wkvf : a.Wf) + t * BF_TILE, 128, 32);
    bf16x8 wp[4], wq[4], wtq[4];                         // d_o weights: rows of (ls1 . Wproj)^T of this head; bone: Wq rows of this head, Wq^T rows of this wave's output channels
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        wp[ks] = *reinterpret_cast<const bf16x8*>(a.Wp + (int64_t)(16 * h + i) * 128 + 32 * ks + 8 * g);
        if (BONE) {
            wq[ks] = *reinterpret_cast<const bf16x8*>(a.Wf + (int64_t)(16 * h + i) * 128 + 32 * ks + 8 * g);
            wtq[ks] = *reinterpret_cast<const bf16x8*>(a.WT + (int64_t)(16 * h + i) * 128 + 16 * (2 * ks + (g >> 1)) + 8 * (g & 1));
        }
    }
    float gm[8], bt[8], gml[8], btl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gm[e] = a.ln_g[sub * 8 + e];
        bt[e] = a.ln_b[sub * 8 + e];
        gml[e] = BONE ? a.lnl_g[sub * 8 + e] : 0.f;
        btl[e] = BONE ? a.lnl_b[sub * 8 + e] : 0.f;
    }
    f32x4 accP[8];                                       // G_proj[16 b + 4 g + r][16 h + i]
#pragma unroll
    for (int b = 0; b < 8; ++b) accP[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dg[8], db[8], dgl[8], dbl[8], gcol[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[e] = 0.f; db[e] = 0.f; dgl[e] = 0.f; dbl[e] = 0.f; gcol[e] = 0.f; }

    // ---- lane-constant LDS offsets (elements), written so that everything else is an immediate or one XOR away (the swizzles are XORs of 16-byte chunk indices) ----
    int tf[4];                                           // row i, chunk 4 ks + g of a swizzled [..][128] tile; row 16 + i: + 2048; image row R + i (R a multiple of 16): + 128 R
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) tf[ks] = i * 128 + (((4 * ks + g) ^ i) << 3);
    const int hrow = r32 * 16 + 8 * hh;                  // head tile: 16-byte row piece of position r32
    const int trb = (4 * hh + q) * 16 + 4 * p;           // head tile, transposed, k-step ks (32x32 products): lo = trb + 256 ks, hi = + 128
    const int p32b = (4 * hh + q) * 32 + ((((2 * (g & 1) + (p >> 1)) ^ (2 * hh + (q >> 1))) & 3) << 3) + 4 * (p & 1);      // P tile, transposed: lo = + 512 ks, hi = + 256
    const int pwb = r32 * 32 + 4 * hh + (((r32 >> 1) & 3) << 3);                                                       // P tile, row write of piece a4: pwb ^ (a4 << 3)
    const int hb = (g >> 1) * 4 * BF_HT + 8 * (g & 1) + i * 16;                                                          // data gradient, token operand
    const int nb = 16 * (g >> 1) + 8 * (g & 1);
    const int awl = (nb + q) * 128 + ((((2 * h + (p >> 1)) ^ ((nb + q) & 15)) & 15) << 3) + 4 * (p & 1);               // data gradient, weight operand (image, transposed): rows n0 + q
    const int awh = (awl ^ 32) + 512;                                                                                    //                                                rows n0 + 4 + q
    const int gl0 = (8 * g + 2 * q) * 128 + ((((p >> 1) ^ (8 * (g & 1) + 2 * q)) & 15) << 3) + 4 * (p & 1);            // frag_tr of a [32][128] tile, column tile b: lo = gl0 ^ (b << 4), hi = (lo ^ 8) + 128
    const int tpos = (8 * g + 2 * q) * 16 + 4 * p;       // head tile, transposed with the position as the reduction index: lo, hi = + 16
    const int st0 = i * 128 + (((2 * h + (g >> 1)) ^ i) << 3) + 4 * (g & 1);                                            // staging store of (position i, channels 16 h + 4 g ..); position 16 + i: + 2048
    const int rc = rl * 128 + ((sub ^ (rl & 15)) << 3);  // row-wise phases: chunk `sub` of row rl
    const int sa = i * 16 + 4 * g;                       // projection results into the head tiles: position i, channels 4 g ..; position 16 + i: + 256

    const int stride = a.mode == 0 ? 1 : KASF_J;         // tokens between consecutive positions of a group
    auto base_of = [&](int G) { return a.mode == 0 ? G * KASF_J : (G / KASF_J) * a.T * KASF_J + (G % KASF_J); };
    const int rlc = rl < L ? rl : L - 1;                 // rows past L: clamped load, zeroed at the point of use
    const unsigned ox = (unsigned)(rlc * stride) * 128u + sub * 8;
    const unsigned oq = (unsigned)((r32 < L ? r32 : L - 1) * stride);      // token offset of this lane's position in the core
    bf16x8 xN, gN, lN;
    auto fetch = [&](int t) {
        const unsigned b = (unsigned)base_of(g0 + t) * 128u + ox;
        xN = *reinterpret_cast<const bf16x8*>(a.X + (size_t)b);
        gN = *reinterpret_cast<const bf16x8*>(a.G + (size_t)b);
        if (BONE) lN = *reinterpret_cast<const bf16x8*>(a.XL + (size_t)b);
    };
    // LayerNorm of the thread's row chunk: LN(x) as bf16 (LDS tile + the copy the weight-gradient launch reads), statistics for the backward
    auto layernorm = [&](const bf16x8 raw, const float (&gmv)[8], const float (&btv)[8], float& mean, float& rstd) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)raw[e];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        mean = reduce16(s) * (1.0f / 128.0f);
        float qq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] -= mean; qq = __builtin_fmaf(v[e], v[e], qq); }
        rstd = rsqrtf(reduce16(qq) * (1.0f / 128.0f) + KASF_LN_EPS);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)__builtin_fmaf(v[e], rstd * gmv[e], btv[e]);       // (the forward's arithmetic: k_attn_blk.hip)
        return o;
    };

    {   // the P / dS tile starts zero: the 4-key pieces past NR are never written
        const bf16x8 z8 = {};
        *reinterpret_cast<bf16x8*>(sP + lane * 16) = z8;
        *reinterpret_cast<bf16x8*>(sP + lane * 16 + 8) = z8;
    }
    fetch(0);
    wait_async();
    __syncthreads();                                     // weight image complete
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { touch_loaded(wp[ks]); if (BONE) { touch_loaded(wq[ks]); touch_loaded(wtq[ks]); } }

#ifdef BWDH_PROF
    long long acc_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define HQ(k) do { const long long n_ = clock64(); acc_t[k] += n_ - t_; t_ = n_; } while (0)
#else
#define HQ(k) do {} while (0)
#endif
    for (int t = 0; t < ng; ++t) {
#ifdef BWDH_PROF
        long long t_ = clock64();
#endif
        const unsigned gb = (unsigned)base_of(g0 + t);   // first token of the group (wave-uniform)
        const bf16x8 zero = {};
        const bool live = rl < L;
        // ---------------- rows into LDS: LN(x), g_mid (| LN_limb(x_limb)); raw x and the statistics stay in registers for the LayerNorm backward ----------------
        float mean, rstd, meanl = 0.f, rstdl = 0.f;
        const bf16x8 xr = live ? xN : zero;
        const bf16x8 lr = BONE ? (live ? lN : zero) : zero;
        {
            const bf16x8 xn = layernorm(xr, gm, bt, mean, rstd);
            *reinterpret_cast<bf16x8*>(sXn + rc) = xn;
            *reinterpret_cast<bf16x8*>(sG + rc) = live ? gN : zero;
            if (live) *reinterpret_cast<bf16x8*>(a.XNA + (size_t)(gb * 128u + ox)) = xn;
            if (BONE) {
                const bf16x8 xl = layernorm(lr, gml, btl, meanl, rstdl);
                *reinterpret_cast<bf16x8*>(sXl + rc) = xl;
                if (live) *reinterpret_cast<bf16x8*>(a.XNB + (size_t)(gb * 128u + ox)) = xl;
            }
        }
        HQ(0);
        __syncthreads();                                 // B1: the group's rows are in LDS; every thread is past the previous group's row-wise phase
        HQ(1);

        // ---------------- projections of head h: q_h | k_h | v_h and d_o_h = g_mid (ls1 Wproj)^T -> head tiles ----------------
        {
            f32x4 acc[3][2];
            zero_acc(acc);
            f32x4 accd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const bf16* wrow = sW + 16 * h * 128;        // image rows of this head (bone image: k rows 16 h .., v rows 128 + 16 h ..)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(sXn + tf[ks]), f1 = *reinterpret_cast<const bf16x8*>(sXn + tf[ks] + 2048);
                const bf16x8 l0 = BONE ? *reinterpret_cast<const bf16x8*>(sXl + tf[ks]) : f0, l1 = BONE ? *reinterpret_cast<const bf16x8*>(sXl + tf[ks] + 2048) : f1;
                const bf16x8 g0v = *reinterpret_cast<const bf16x8*>(sG + tf[ks]), g1v = *reinterpret_cast<const bf16x8*>(sG + tf[ks] + 2048);
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    const bf16x8 wa = (BONE && nt == 0) ? wq[ks] : *reinterpret_cast<const bf16x8*>(wrow + (BONE ? nt - 1 : nt) * 128 * 128 + tf[ks]);
                    acc[nt][0] = mfma16(wa, nt == 0 ? f0 : l0, acc[nt][0]);
                    acc[nt][1] = mfma16(wa, nt == 0 ? f1 : l1, acc[nt][1]);
                }
                accd[0] = mfma16(wp[ks], g0v, accd[0]);
                accd[1] = mfma16(wp[ks], g1v, accd[1]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
                    store4(tQ + nt * BF_HT + sa + 256 * mt, v);
                }
                const float lv = 16 * mt + i < L ? 1.0f : 0.0f;              // d_o rows past L are zero: those queries then give dS = 0 and add nothing to dK, dV
                float v[4] = {accd[mt][0] * lv, accd[mt][1] * lv, accd[mt][2] * lv, accd[mt][3] * lv};
                store4(tD + sa + 256 * mt, v);
            }
        }
        lds_fence();
        HQ(2);

        // ---------------- attention core of head h (k_attn_bwd_pers' arithmetic), o recomputed; dq | dk | dv | o overwrite q | k | v | d_o ----------------
        {
            auto trh = [&](const bf16* tile, int ks) {   // transposed fragment of k-step ks of a head tile
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(tile + trb + 256 * ks));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(tile + trb + 256 * ks + 128));
                return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            auto trp = [&](int ks) {                     // transposed fragment of k-step ks of the P / dS tile
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(sP + p32b + 512 * ks));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(sP + p32b + 512 * ks + 256));
                return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            const bf16x8 qf = *reinterpret_cast<const bf16x8*>(tQ + hrow);
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(tK + hrow);
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(tV + hrow);
            const bf16x8 df = *reinterpret_cast<const bf16x8*>(tD + hrow);
            // pass 1: lane = query
            f32x16 st = mfma32(kf, qf, zero16());        // S^T[key][query]
            f32x16 dp = mfma32(vf, df, zero16());        // dP^T[key][query]
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const float sv = (pos_of(e, hh) < L) ? st[e] * 0.25f : -INFINITY;
                st[e] = sv;
                mx = fmaxf(mx, sv);
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
                store4(sP + (pwb ^ (a4 << 3)), v4);
            }
            // o^T[d][query] = V^T . P^T (the forward's product, recomputed: the forward saves no attention output)
            f32x16 ot = mfma32(trh(tV, 0), pack8(st, 0), zero16());
            ot = mfma32(trh(tV, 1), pack8(st, 1), ot);
            const u32x4_t po = swap_t16(ot);
#pragma unroll
            for (int e = 0; e < NR; ++e) st[e] = st[e] * (dp[e] - delta) * 0.25f;      // dS^T (scale folded)
            f32x16 dq = mfma32(trh(tK, 0), pack8(st, 0), zero16());                      // dQ^T[d][query] = K^T . dS^T
            dq = mfma32(trh(tK, 1), pack8(st, 1), dq);
            // pass 2: lane = key.  dV^T = dO^T . P, dK^T = Q^T . dS with P / dS read back transposed from ONE tile (P first)
            lds_fence();
            f32x16 dv = mfma32(trh(tD, 0), trp(0), zero16());
            dv = mfma32(trh(tD, 1), trp(1), dv);
            lds_fence();                                 // the P fragments are in registers before dS overwrites the tile
#pragma unroll
            for (int a4 = 0; a4 < NA4; ++a4) {
                float v4[4] = {st[4 * a4], st[4 * a4 + 1], st[4 * a4 + 2], st[4 * a4 + 3]};
                store4(sP + (pwb ^ (a4 << 3)), v4);
            }
            lds_fence();
            f32x16 dk = mfma32(trh(tQ, 0), trp(0), zero16());
            dk = mfma32(trh(tQ, 1), trp(1), dk);
            const u32x4_t pq = swap_t16(dq), pv = swap_t16(dv), pk = swap_t16(dk);
            lds_fence();                                 // every read of the four tiles has returned
            *reinterpret_cast<u32x4_t*>(tQ + hrow) = pq;
            *reinterpret_cast<u32x4_t*>(tK + hrow) = pk;
            *reinterpret_cast<u32x4_t*>(tV + hrow) = pv;
            *reinterpret_cast<u32x4_t*>(tD + hrow) = po;
            if (r32 < L) {                               // dq | dk | dv for the streaming weight gradient: 16 bytes per lane, 32 contiguous bytes per position and head
                const unsigned tok = gb + oq;
                if (BONE) {
                    *reinterpret_cast<u32x4_t*>(a.DQ + (size_t)(tok * 128u + 16 * h + 8 * hh)) = pq;
                    *reinterpret_cast<u32x4_t*>(a.DKV + (size_t)(tok * 256u + 16 * h + 8 * hh)) = pk;
                    *reinterpret_cast<u32x4_t*>(a.DKV + (size_t)(tok * 256u + 128 + 16 * h + 8 * hh)) = pv;
                } else {
                    *reinterpret_cast<u32x4_t*>(a.DQ + (size_t)(tok * 384u + 16 * h + 8 * hh)) = pq;
                    *reinterpret_cast<u32x4_t*>(a.DQ + (size_t)(tok * 384u + 128 + 16 * h + 8 * hh)) = pk;
                    *reinterpret_cast<u32x4_t*>(a.DQ + (size_t)(tok * 384u + 256 + 16 * h + 8 * hh)) = pv;
                }
            }
        }
        HQ(3);
        __syncthreads();                                 // B2: dq | dk | dv | o of all eight heads
        HQ(4);

        if (t + 1 < ng) fetch(t + 1);                    // the next group's rows travel under the GEMM phase
        // ---------------- data gradient: output channels [16 h, 16 h + 16) x 32 positions, reduction over the 24 head tiles ----------------
        // reduction index = (head tile, channel): k-step ks, lane group g -> tile tau = 2 ks + (g >> 1) (head tau & 7, part tau >> 3 = ks >> 2), channels 8 (g & 1) .. + 7:
        // the token operand is a 16-byte row piece of that head tile, the weight operand the same 8 rows of the LDS image read TRANSPOSED (column = output channel)
        f32x4 accx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, accl[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            const bf16* tp = sHead + hb + ((2 * ks) & 7) * 4 * BF_HT + (ks >> 2) * BF_HT;
            const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(tp), f1 = *reinterpret_cast<const bf16x8*>(tp + 256);
            bf16x8 wa;
            if (BONE && ks < 4) {
                wa = wtq[ks];
            } else {
                const int ro = (128 * ((ks >> 2) - (BONE ? 1 : 0)) + 16 * ((2 * ks) & 7)) * 128;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(sW + awl + ro));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(sW + awh + ro));
                wa = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            if (BONE && ks >= 4) {
                accl[0] = mfma16(wa, f0, accl[0]);
                accl[1] = mfma16(wa, f1, accl[1]);
            } else {
                accx[0] = mfma16(wa, f0, accx[0]);
                accx[1] = mfma16(wa, f1, accx[1]);
            }
        }
        // ---------------- proj weight gradient: G_proj[.][16 h + i] += g_mid^T o_h (reduction over the group's positions: one k-step) ----------------
        {
            const bf16x4 olo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(tD + tpos));
            const bf16x4 ohi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(tD + tpos + 16));
            const bf16x8 ro = bf16x8{olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int lo_ = gl0 ^ (b << 4);
                const bf16x4 glo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(sG + lo_));
                const bf16x4 ghi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(sG + (lo_ ^ 8) + 128));
                accP[b] = mfma16(bf16x8{glo[0], glo[1], glo[2], glo[3], ghi[0], ghi[1], ghi[2], ghi[3]}, ro, accP[b]);
            }
        }
        HQ(5);
        __syncthreads();                                 // B3: every wave is done with LN(x) (and the head tiles): the LN tiles become staging tiles
        HQ(6);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float v[4] = {accx[mt][0], accx[mt][1], accx[mt][2], accx[mt][3]};
            store4(sXn + st0 + 2048 * mt, v);
            if (BONE) {
                float u[4] = {accl[mt][0], accl[mt][1], accl[mt][2], accl[mt][3]};
                store4(sXl + st0 + 2048 * mt, u);
            }
        }
        HQ(7);
        __syncthreads();                                 // B4: staging tiles complete
        HQ(8);
        // ---------------- LayerNorm backward + residual, one 16-lane group per row; the thread's next-group writes go to the chunks it reads here ----------------
        {
            float d[8], x[8], res[8], o[8];
            load8(sXn + rc, d);
            load8(sG + rc, res);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = ((float)xr[e] - mean) * rstd;            // xhat
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
            for (int e = 0; e < 8; ++e) o[e] = rstd * (d[e] - s1 - x[e] * s2) + res[e];
            if (live) store8(a.OUT + (size_t)(gb * 128u + ox), o);
            if (BONE) {
                float dl[8], xl[8], ol[8], old[8];
                load8(sXl + rc, dl);
                if (live) load8(a.OUTL + (size_t)(gb * 128u + ox), old);
#pragma unroll
                for (int e = 0; e < 8; ++e) xl[e] = ((float)lr[e] - meanl) * rstdl;
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
                    for (int e = 0; e < 8; ++e) ol[e] = old[e] + rstdl * (dl[e] - t1 - xl[e] * t2);
                    store8(a.OUTL + (size_t)(gb * 128u + ox), ol);
                }
            }
        }
        HQ(9);
    }
#ifdef BWDH_PROF
    if (blockIdx.x == 77 && (tid == 0 || tid == 320)) printf("bwd_h prof bone %d NR %d wave %d groups %d: rows %lld B1 %lld project %lld core %lld B2 %lld gemm %lld B3 %lld staging %lld B4 %lld epilogue %lld\n", (int)BONE, NR, h, ng, acc_t[0], acc_t[1], acc_t[2], acc_t[3], acc_t[4], acc_t[5], acc_t[6], acc_t[7], acc_t[8], acc_t[9]);
#endif
    // ================= end of the range: per-channel rows and the bf16 partial tile of G_proj of this workgroup =================
    __syncthreads();
    {
        float* sRed = reinterpret_cast<float*>(sW);      // [5][8 waves][128] floats (20 KB of the dead weight image)
        auto put = [&](int which, const float (&v)[8]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s = v[e];                          // the wave's four rows (lanes sub, 16 + sub, 32 + sub, 48 + sub)
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                if (lane < 16) sRed[(which * 8 + h) * 128 + sub * 8 + e] = s;
            }
        };
        put(0, dg); put(1, db); put(2, gcol);
        if (BONE) { put(3, dgl); put(4, dbl); }
        __syncthreads();
        constexpr int PLD = BONE ? 512 : 256;
        for (int idx = tid; idx < (BONE ? 5 : 3) * 128; idx += BF_THR) {
            const int which = idx >> 7, c = idx & 127;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += sRed[(which * 8 + k) * 128 + c];
            if (which == 2) a.pbrow[(int64_t)blockIdx.x * 128 + c] = s;
            else a.part[(int64_t)blockIdx.x * PLD + (which < 2 ? which : which - 1) * 128 + c] = s;
        }
    }
    {
        bf16* sPst = sHead;                              // [128][128] G_proj image
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) sPst[(16 * b + 4 * g + r) * 128 + 16 * h + i] = (bf16)accP[b][r];
        __syncthreads();
        bf16* dstp = a.ppart + (int64_t)blockIdx.x * (128 * 128);
        for (int c = tid; c < 128 * 16; c += BF_THR) *reinterpret_cast<f32x4*>(dstp + c * 8) = *reinterpret_cast<const f32x4*>(sPst + c * 8);
    }
}

template <bool BONE, int NR> int launch_bwd_h(hipStream_t s, AttnBwdHArgs a, KasfColSink* sink, float* dgamma, float* dbeta, float* dgamma_l, float* dbeta_l, int grid_cap) {
    constexpr int PLD = BONE ? 512 : 256;
    const size_t sh = (size_t)((BONE ? 256 : 384) * 128 + (BONE ? 3 : 2) * BF_TILE + 8 * 4 * BF_HT + 8 * 32 * 32) * 2;
    auto kern = k_attn_blk_bwd_h<BONE, NR>;
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

// Returns the number of G_proj partial tiles written (= active workgroups), 0 when the shape is not covered or the scratch is short (nothing launched, nothing registered).
// dq: [M][384] (self) / [M][128] (bone), dkv: bone [M][256]; xn_a / xn_b: [M][128]; ppart: 256 x [128][128] bf16; pbrow: 256 x 128 floats.
int kasf_launch_attn_block_bwd(hipStream_t s, int bone, const void* x, const void* x_limb, const void* g_mid, const float* ln_g, const float* ln_b, const float* lnl_g,
                               const float* lnl_b, const void* Wf, const void* Wkvf, const void* WT, const void* WprojTs, void* out, void* out_limb, void* dq, void* dkv,
                               void* xn_a, void* xn_b, float* dgamma, float* dbeta, float* dgamma_l, float* dbeta_l, KasfColSink* sink, void* ppart, float* pbrow,
                               int B, int T, int mode) {
    const int L = mode == 0 ? KASF_J : T;
    if (L > 32 || sink == nullptr || ppart == nullptr || pbrow == nullptr || dq == nullptr || xn_a == nullptr) return 0;
    if (bone && (out_limb == nullptr || dkv == nullptr || xn_b == nullptr || WT == nullptr || Wkvf == nullptr)) return 0;
    AttnBwdHArgs a;
    a.X = (const bf16*)x; a.XL = (const bf16*)x_limb; a.G = (const bf16*)g_mid;
    a.ln_g = ln_g; a.ln_b = ln_b; a.lnl_g = lnl_g; a.lnl_b = lnl_b;
    a.Wf = (const bf16*)Wf; a.Wkvf = (const bf16*)Wkvf; a.WT = (const bf16*)WT; a.Wp = (const bf16*)WprojTs;
    a.OUT = (bf16*)out; a.OUTL = (bf16*)out_limb; a.DQ = (bf16*)dq; a.DKV = (bf16*)dkv; a.XNA = (bf16*)xn_a; a.XNB = (bf16*)xn_b;
    a.part = nullptr; a.ppart = (bf16*)ppart; a.pbrow = pbrow;
    a.L = L; a.T = T; a.mode = mode; a.groups = mode == 0 ? B * T : B * KASF_J;
    if (a.groups <= 0) return 0;
    const int cap = kasf_narrow_grid(KASF_NG_ATTN_BWD, 256, (int64_t)a.groups * L);
    if (bone) return L <= 17 ? launch_bwd_h<true, 9>(s, a, sink, dgamma, dbeta, dgamma_l, dbeta_l, cap) : launch_bwd_h<true, 16>(s, a, sink, dgamma, dbeta, dgamma_l, dbeta_l, cap);
    return L <= 17 ? launch_bwd_h<false, 9>(s, a, sink, dgamma, dbeta, dgamma_l, dbeta_l, cap) : launch_bwd_h<false, 16>(s, a, sink, dgamma, dbeta, dgamma_l, dbeta_l, cap);
}
