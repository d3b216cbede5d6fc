// Fused attention block, forward (bf16, groups of at most 32 positions: every spatial block, temporal blocks up to T = 32):
//     x_mid = x + ls1 * ( proj( softmax(q k^T / 4) v ) + b ),   q|k|v = LN(x) Wqkv^T            (selfattention.py:18-41)
//     bone:   q = LN(x) Wq^T,  k|v = LN_limb(x_limb) Wkv^T                                       (bone_crossattention.py:19-41)
// One persistent workgroup (8 waves) walks whole groups -- the 17 joints of a frame, or the T frames of one joint -- so the
// LayerNorm, the three projections, the 8 heads and the output projection of a group never leave the CU:
//   * wave w IS head w: in the QKV GEMM it owns exactly the 3 x 16 output features (q_h, k_h, v_h) its own attention core consumes,
//     so the accumulators reach the core through a wave-private LDS tile without any workgroup barrier;
//   * the core is the 32x32x16 MFMA formulation of k_attn_mfma.hip (scores transposed, softmax lane-local);
//   * the output projection + layer-scale + residual reads the 8 heads' outputs from one shared tile;
//   * weights live in registers for the whole launch (64 VGPRs), groups arrive by LDS-direct loads two groups ahead.
// HBM traffic per token: 256 B in (512 B for the bone form), 256 B out, plus -- in training only -- q|k|v and the attention output,
// which the backward pass needs (written once).  The unfused path moved 2.8 KB per token.
#include <type_traits>
#include <cstdio>
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int AB_THR = 512, AB_TILE = 32 * 128;
constexpr int AB_HPAD = 16;                        // bf16 elements between the heads' q | k | v tile images (32 bytes: the 8 heads' images 32 bytes apart modulo the 256 bytes
                                                   // of an LDS read cycle -- the row copy-out reads 16 lanes = 8 heads x 2 halves)
constexpr int RP3_PAD = 16;                        // bf16 elements between the heads' q | k | v tile images of k_attn_blk_fwd_rp3

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ bf16x8 tok_frag(const bf16* s, int row, int ks) {
    const int g = (threadIdx.x & 63) >> 4;
    return *reinterpret_cast<const bf16x8*>(s + Tile<bf16>::chunk_off(row, 4 * ks + g));
}
// position held by register `reg` of a 32x32 accumulator in lane half hh (see k_attn_mfma.hip)
__device__ __forceinline__ int pos_of(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }
__device__ __forceinline__ bf16x8 pack8(const f32x16& t, int s) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)t[8 * s + j];
    return o;
}
// transposed fragment of k-step ks out of a row-major [32][16] tile
__device__ __forceinline__ bf16x8 tr_frag(const bf16* s_tile, int ks) {
    const int lane = threadIdx.x & 63, u = lane & 15, hh = lane >> 5, q = u >> 2, p = u & 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    const int k0 = 16 * ks + 4 * hh;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (k0 + q) * 16 + 4 * p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (k0 + 8 + q) * 16 + 4 * p));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ int tok_of(int G, int i, int T, int mode) {       // token index (< 2^31: 32-bit keeps address registers short-lived)
    return mode == 0 ? G * KASF_J + i : (G / KASF_J) * T * KASF_J + i * KASF_J + (G % KASF_J);
}
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // same-wave LDS write -> read ordering

struct AttnBlkArgs {
    const bf16* X;            // [M][128] block input (query stream)
    const bf16* XL;           // bone: [M][128] limb stream
    const float *ln_g, *ln_b, *lnl_g, *lnl_b;
    const bf16* Wq;           // self: Wqkv [384][128];  bone: Wq [128][128]
    const bf16* Wkv;          // bone: [256][128]
    const bf16* Wproj;        // [128][128]
    const float *bproj, *ls1;
    bf16* Qs;                 // training: self qkv [M][384] / bone q [M][128]; nullptr in evaluation
    bf16* KVs;                // training, bone: [M][256]
    bf16* Os;                 // training: attention output [M][128]
    bf16* OUT;                // x_mid [M][128]
    float* LSE;               // training, groups of 33..96 positions: log-sum-exp of the scaled scores per (token, head) [M][8] for k_attn_bwd_kt; or nullptr
    int L, T, mode, groups;
};

// ---------------------------------------------------------------------------------------------------------------
// One-tile groups (<= 32 positions).  The inputs are prefetched through REGISTERS: thread (row tid >> 4, chunk tid & 15) loads the 16 bytes of x
// (and of x_limb) it is about to normalise -- a whole 256-byte row per 16 lanes -- one group ahead, keeps them in 4 (8) VGPRs while the current
// group computes, and LayerNorm runs straight out of those registers.  43 KB (self) / 67 KB (bone) of LDS; the q | k | v rows leave as each lane's
// 16-byte operand.  hipcc counts the prefetch loads itself: no hand-counted wait in this kernel.  (Round 1's LDS-direct ring form measured
// 56.5 / 81.2 us against 55.5 / 79.2 us and was removed.)  The self form fits 128 VGPRs, so TWO workgroups share a CU and one group's barrier /
// LDS round-trip latencies are covered by the other's work (a group is only 17-32 positions: its dependency chain, not bandwidth, bounds a lone
// workgroup).
// ---------------------------------------------------------------------------------------------------------------
// The bone form needs ~140 VGPRs (64 of them weights): at the 128 of two workgroups per CU hipcc spills four weight fragments and seven address
// registers into the loop and the launch takes 104 us instead of 79 (measured; moving the look-ahead loads behind the core did not free enough).
// So it runs ONE workgroup per CU; the self form fits 128 and runs two.
#ifndef KASF_RP_BONE_WAVES
#define KASF_RP_BONE_WAVES 2
#endif
template <bool BONE>
__global__ __launch_bounds__(AB_THR, BONE ? KASF_RP_BONE_WAVES : 4) void k_attn_blk_fwd_rp(const AttnBlkArgs a) {
    constexpr int NS = BONE ? 2 : 1, WS = 3 * 512 + AB_HPAD;       // WS: one wave's q | k | v tiles + 32 bytes (see AB_HPAD)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sX = reinterpret_cast<bf16*>(smem);           // [32][128] raw x (the residual's operand)
    bf16* sA = sX + AB_TILE;                            // [NS][32][128] LN(x) (| LN_limb(x_limb))
    bf16* sO = sA + NS * AB_TILE;                       // [32][128] attention output of the 8 heads
    bf16* sOut = sO + AB_TILE;                          // [32][128] x_mid
    bf16* sHead = sOut + AB_TILE;                       // [8 waves][q | k | v][32][16] wave-private operand tiles
    float* sLn = reinterpret_cast<float*>(sHead + 8 * WS);           // [6][128] gamma, beta, limb gamma, beta, proj bias, ls1
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, sub = lane & 15, rl = threadIdx.x >> 4;
    const int r32 = lane & 31, hh = lane >> 5;
    const int L = a.L;
    const int per = (a.groups + gridDim.x - 1) / gridDim.x;
    const int g0 = blockIdx.x * per;
    int ng = a.groups - g0;
    if (ng > per) ng = per;
    if (ng <= 0) return;
    bf16* sQh = sHead + w * WS;
    bf16* sKh = sQh + 512;
    bf16* sVh = sKh + 512;
    bf16x8 wq[3][4], wp[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (BONE) {
            wq[0][ks] = *reinterpret_cast<const bf16x8*>(a.Wq + (int64_t)(16 * w + i) * 128 + 32 * ks + 8 * g);
            wq[1][ks] = *reinterpret_cast<const bf16x8*>(a.Wkv + (int64_t)(16 * w + i) * 128 + 32 * ks + 8 * g);
            wq[2][ks] = *reinterpret_cast<const bf16x8*>(a.Wkv + (int64_t)(128 + 16 * w + i) * 128 + 32 * ks + 8 * g);
        } else {
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) wq[nt][ks] = *reinterpret_cast<const bf16x8*>(a.Wq + (int64_t)(128 * nt + 16 * w + i) * 128 + 32 * ks + 8 * g);
        }
        wp[ks] = *reinterpret_cast<const bf16x8*>(a.Wproj + (int64_t)(16 * w + i) * 128 + 32 * ks + 8 * g);
    }
    if (threadIdx.x < 128) {
        sLn[threadIdx.x] = a.ln_g[threadIdx.x];
        sLn[128 + threadIdx.x] = a.ln_b[threadIdx.x];
        if (BONE) { sLn[256 + threadIdx.x] = a.lnl_g[threadIdx.x]; sLn[384 + threadIdx.x] = a.lnl_b[threadIdx.x]; }
        sLn[512 + threadIdx.x] = a.bproj[threadIdx.x];
        sLn[640 + threadIdx.x] = a.ls1[threadIdx.x];
    }
    const int stride = a.mode == 0 ? 1 : KASF_J;                    // tokens between consecutive positions of a group
    auto base_of = [&](int G) { return a.mode == 0 ? G * KASF_J : (G / KASF_J) * a.T * KASF_J + (G % KASF_J); };
    const int rlc = rl < L ? rl : L - 1;                            // rows past L: clamped load, zeroed at the point of use
    const unsigned ox = (unsigned)(rlc * stride) * 128u + sub * 8;
    bf16x8 xN, lN;
    auto fetch = [&](int t) {
        const unsigned b = (unsigned)base_of(g0 + t) * 128u + ox;
        xN = *reinterpret_cast<const bf16x8*>(a.X + (size_t)b);
        if (BONE) lN = *reinterpret_cast<const bf16x8*>(a.XL + (size_t)b);
    };
    auto layernorm = [&](const bf16x8 raw, bf16* dst, const float* gp, const float* bp) {      // row rl from registers; gp/bp in LDS
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)raw[e];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        const float mean = reduce16(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] -= mean; q = __builtin_fmaf(v[e], v[e], q); }        // (explicit fma: the library is built with -ffp-contract=off)
        const float rstd = rsqrtf(reduce16(q) * (1.0f / 128.0f) + KASF_LN_EPS);
        const f32x4 g0v = *reinterpret_cast<const f32x4*>(gp + sub * 8), g1v = *reinterpret_cast<const f32x4*>(gp + sub * 8 + 4);
        const f32x4 b0v = *reinterpret_cast<const f32x4*>(bp + sub * 8), b1v = *reinterpret_cast<const f32x4*>(bp + sub * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = __builtin_fmaf(v[e], rstd * g0v[e], b0v[e]); v[4 + e] = __builtin_fmaf(v[4 + e], rstd * g1v[e], b1v[e]); }
        tile_store8(dst, rl, sub * 8, v);
    };
    fetch(0);
    __syncthreads();                                     // sLn
#ifdef RP_PROF
    long long acc_t[8] = {0,0,0,0,0,0,0,0};
#define RQ(k) do { const long long n_ = clock64(); acc_t[k] += n_ - t_; t_ = n_; } while (0)
#else
#define RQ(k) do {} while (0)
#endif
    for (int t = 0; t < ng; ++t) {
#ifdef RP_PROF
        long long t_ = clock64();
#endif
        const int G = g0 + t;
        const bf16x8 zero = {};
        const bf16x8 xc = rl < L ? xN : zero;
        *reinterpret_cast<bf16x8*>(sX + Tile<bf16>::chunk_off(rl, sub)) = xc;
        layernorm(xc, sA, sLn, sLn + 128);
        if (BONE) layernorm(rl < L ? lN : zero, sA + AB_TILE, sLn + 256, sLn + 384);
        if (t + 1 < ng) fetch(t + 1);
        RQ(0);
        __syncthreads();                                 // B1: raw and LN tiles complete; every wave finished the copy-out of the previous group
        RQ(1);
        if (BONE) {   // ---- q_h from LN(x), then k_h, v_h from LN_limb(x_limb): two phases keep the live accumulators + operand fragments under the 128-VGPR cap ----
            {
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    acc[0] = mfma16(wq[0][ks], tok_frag(sA, i, ks), acc[0]);
                    acc[1] = mfma16(wq[0][ks], tok_frag(sA, 16 + i, ks), acc[1]);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float v[4] = {acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]};
                    store4(sQh + (16 * mt + i) * 16 + 4 * g, v);
                }
            }
            f32x4 acc[2][2];
            zero_acc(acc);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 l0 = tok_frag(sA + AB_TILE, i, ks), l1 = tok_frag(sA + AB_TILE, 16 + i, ks);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[nt][0] = mfma16(wq[1 + nt][ks], l0, acc[nt][0]);
                    acc[nt][1] = mfma16(wq[1 + nt][ks], l1, acc[nt][1]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
                    store4(sQh + (1 + nt) * 512 + (16 * mt + i) * 16 + 4 * g, v);
                }
        } else {   // ---- q_h, k_h, v_h of the 32 positions: 3 feature tiles x 2 position tiles ----
            f32x4 acc[3][2];
            zero_acc(acc);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 f0 = tok_frag(sA, i, ks), f1 = tok_frag(sA, 16 + i, ks);
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    acc[nt][0] = mfma16(wq[nt][ks], f0, acc[nt][0]);
                    acc[nt][1] = mfma16(wq[nt][ks], f1, acc[nt][1]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
                    store4(sQh + nt * 512 + (16 * mt + i) * 16 + 4 * g, v);                       // wave-private [pos][16]
                }
        }
        lds_fence();
        RQ(2);
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sKh + r32 * 16 + 8 * hh);
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(sQh + r32 * 16 + 8 * hh);
        RQ(3);
        {   // ---- attention core of head w (k_attn_mfma.hip, one 32x32 score tile) ----
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
            f32x16 st = mfma32(kf, qf, z);               // S^T[key][query]
            // softmax with the fewest vector instructions per score (as in k_attn_blk_fwd_rp3): max over the raw scores, scale and log2(e) in one fma,
            // 1 / sum applied to the 8 outputs of the lane instead of its 16 probabilities
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                st[e] = pos_of(e, hh) < L ? st[e] : -INFINITY;
                mx = fmaxf(mx, st[e]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            constexpr float C2 = 0.25f * 1.4426950408889634f;     // scale . log2(e)
            const float nm = -mx * C2;
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[e], C2, nm)); sum += st[e]; }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            f32x16 ot = mfma32(tr_frag(sVh, 0), pack8(st, 0), z);
            ot = mfma32(tr_frag(sVh, 1), pack8(st, 1), ot);
#pragma unroll
            for (int e = 0; e < 8; ++e) ot[e] *= inv;
            float o0[4] = {ot[0], ot[1], ot[2], ot[3]}, o1[4] = {ot[4], ot[5], ot[6], ot[7]};
            store4(sO + Tile<bf16>::off4(r32, 16 * w + 4 * hh), o0);
            store4(sO + Tile<bf16>::off4(r32, 16 * w + 8 + 4 * hh), o1);
        }
        RQ(4);
        __syncthreads();                                 // B2: all heads in sO
        RQ(5);
        {   // ---- output projection + layer-scale + residual: 16 channels x 32 positions per wave ----
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0] = mfma16(wp[ks], tok_frag(sO, i, ks), acc[0]);
                acc[1] = mfma16(wp[ks], tok_frag(sO, 16 + i, ks), acc[1]);
            }
            const f32x4 bpv = *reinterpret_cast<const f32x4*>(sLn + 512 + 16 * w + 4 * g), lsv = *reinterpret_cast<const f32x4*>(sLn + 640 + 16 * w + 4 * g);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float x[4], v[4];
                load4(sX + Tile<bf16>::off4(16 * mt + i, 16 * w + 4 * g), x);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = x[r] + lsv[r] * (acc[mt][r] + bpv[r]);
                store4(sOut + Tile<bf16>::off4(16 * mt + i, 16 * w + 4 * g), v);
            }
        }
        RQ(6);
        __syncthreads();                                 // B3: x_mid tile complete
        if (rl < L) {   // ---- full-row stores: x_mid always; o only when the backward pass will need it ----
            const unsigned tok = (unsigned)(base_of(G) + rl * stride);
            const int co = Tile<bf16>::chunk_off(rl, sub);
            *reinterpret_cast<f32x4*>(a.OUT + (size_t)(tok * 128u + sub * 8)) = *reinterpret_cast<const f32x4*>(sOut + co);
            if (a.Qs != nullptr) *reinterpret_cast<f32x4*>(a.Os + (size_t)(tok * 128u + sub * 8)) = *reinterpret_cast<const f32x4*>(sO + co);
            if (BONE && a.Qs != nullptr)
                *reinterpret_cast<f32x4*>(a.Qs + (size_t)(tok * 128u + sub * 8)) = *reinterpret_cast<const f32x4*>(sHead + (sub >> 1) * WS + rl * 16 + 8 * (sub & 1));
        }
        if (a.Qs != nullptr) {   // training: the backward pass reads q | k | v.  Round 6: whole rows, thread = (position, 16-byte chunk), out of the heads' tiles (complete since
            // B2, untouched until the next group's projections behind B1) -- rounds 2-5: each lane stored the 16 bytes it was about to use as an operand, 32-byte slices per head.
            if (BONE) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int idx = (int)threadIdx.x + AB_THR * k, row = idx >> 5, c = idx & 31, part = 1 + (c >> 4), ch = c & 15;
                    if (row < L) {
                        const unsigned tok = (unsigned)(base_of(G) + row * stride);
                        *reinterpret_cast<f32x4*>(a.KVs + (size_t)(tok * 256u + c * 8)) = *reinterpret_cast<const f32x4*>(sHead + (ch >> 1) * WS + part * 512 + row * 16 + 8 * (ch & 1));
                    }
                }
            } else {
                int tx = (int)threadIdx.x;
                asm volatile("" : "+v"(tx));             // the chunk addresses are formed here, not kept in registers across the loop (128-VGPR cap of two workgroups per CU)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int idx = tx + AB_THR * k, row = idx / 48, c = idx - row * 48, part = c >> 4, ch = c & 15;
                    if (row < L) {
                        const unsigned tok = (unsigned)(base_of(G) + row * stride);
                        *reinterpret_cast<f32x4*>(a.Qs + (size_t)(tok * 384u + c * 8)) = *reinterpret_cast<const f32x4*>(sHead + (ch >> 1) * WS + part * 512 + row * 16 + 8 * (ch & 1));
                    }
                }
            }
        }
        RQ(7);
    }
#ifdef RP_PROF
    if (blockIdx.x == 77 && (threadIdx.x == 0 || threadIdx.x == 320)) printf("rp prof bone %d mode %d wave %d groups %d: LN %lld B1 %lld project %lld qkvstore %lld core %lld B2 %lld proj %lld B3+copyout %lld\n", (int)BONE, a.mode, w, ng, acc_t[0], acc_t[1], acc_t[2], acc_t[3], acc_t[4], acc_t[5], acc_t[6], acc_t[7]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Spatial blocks (a group = the 17 joints of one frame) in FLAT 32-token tiles (round 4).  The one-tile form above spends a 32-row tile on 17 rows:
// 47 % of its LayerNorm, projection, epilogue and copy-out work -- 80 % of the kernel by its phase timers, the 32 x 32 attention core is the
// other 20 % -- is padding.  The frames of a launch are consecutive in memory (token = 17 frame + joint), so a workgroup's frame range is one dense
// token range: it is walked in 32-token tiles regardless of frame boundaries, and only the core keeps the frame structure:
//   * the wave-private q | k | v tiles are rolling buffers of 48 positions (tile t's 32 rows + the <= 16 rows of a frame that began in tile t - 1);
//     after tile t's projection a wave runs the core of every frame that ends inside the tile (one or two), reading its 17 rows at a rolling offset;
//   * head outputs and raw x (the residual's operand) are kept per tile parity; the output projection of tile t - 1 runs one iteration late, when
//     the frame that straddles the tile border has been through the core;
//   * keys past 16 are dead: the softmax touches 9 of the 16 score registers (as k_attn_bwd_pers<9>).
// Same three barriers per iteration, 17 / 32 as many iterations.  LDS 79 KB (self: two workgroups per CU) / 87 KB (bone).
// ---------------------------------------------------------------------------------------------------------------
template <bool BONE>
__global__ __launch_bounds__(AB_THR, BONE ? KASF_RP_BONE_WAVES : 4) void k_attn_blk_fwd_flat(const AttnBlkArgs a) {
    constexpr int NS = BONE ? 2 : 1, J = KASF_J, HB = 48, HT = HB * 16, WS = 3 * HT + AB_HPAD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sX = reinterpret_cast<bf16*>(smem);           // [2][32][128] raw x, by tile parity
    bf16* sA = sX + 2 * AB_TILE;                        // [NS][32][128] LN(x) (| LN_limb(x_limb)); sA[0] doubles as the x_mid staging tile
    bf16* sO = sA + NS * AB_TILE;                       // [2][32][128] attention output of the 8 heads, by tile parity (row p at p & 63)
    bf16* sHead = sO + 2 * AB_TILE;                     // [8 waves][q | k | v][48][16] wave-private rolling operand tiles
    float* sLn = reinterpret_cast<float*>(sHead + 8 * WS);           // [6][128] gamma, beta, limb gamma, beta, proj bias, ls1
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, sub = lane & 15, rl = threadIdx.x >> 4;
    const int r32 = lane & 31, hh = lane >> 5;
    const int per = (a.groups + gridDim.x - 1) / gridDim.x;
    const int f0 = blockIdx.x * per;
    int nf = a.groups - f0;
    if (nf > per) nf = per;
    if (nf <= 0) return;
    const int nrows = J * nf, ntiles = (nrows + 31) >> 5;
    const unsigned tok0 = (unsigned)f0 * J;
    bf16* sQh = sHead + w * WS;
    bf16* sKh = sQh + HT;
    bf16* sVh = sKh + HT;
    bf16x8 wq[3][4], wp[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (BONE) {
            wq[0][ks] = *reinterpret_cast<const bf16x8*>(a.Wq + (int64_t)(16 * w + i) * 128 + 32 * ks + 8 * g);
            wq[1][ks] = *reinterpret_cast<const bf16x8*>(a.Wkv + (int64_t)(16 * w + i) * 128 + 32 * ks + 8 * g);
            wq[2][ks] = *reinterpret_cast<const bf16x8*>(a.Wkv + (int64_t)(128 + 16 * w + i) * 128 + 32 * ks + 8 * g);
        } else {
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) wq[nt][ks] = *reinterpret_cast<const bf16x8*>(a.Wq + (int64_t)(128 * nt + 16 * w + i) * 128 + 32 * ks + 8 * g);
        }
        wp[ks] = *reinterpret_cast<const bf16x8*>(a.Wproj + (int64_t)(16 * w + i) * 128 + 32 * ks + 8 * g);
    }
    if (threadIdx.x < 128) {
        sLn[threadIdx.x] = a.ln_g[threadIdx.x];
        sLn[128 + threadIdx.x] = a.ln_b[threadIdx.x];
        if (BONE) { sLn[256 + threadIdx.x] = a.lnl_g[threadIdx.x]; sLn[384 + threadIdx.x] = a.lnl_b[threadIdx.x]; }
        sLn[512 + threadIdx.x] = a.bproj[threadIdx.x];
        sLn[640 + threadIdx.x] = a.ls1[threadIdx.x];
    }
    {   // the rolling tiles start finite: a frame's core reads 32 rows of them, 15 of which belong to the neighbouring frames (their scores are masked by
        // selects, their V rows are multiplied by zero probabilities)
        const bf16x8 zero = {};
        for (int c = lane; c < 3 * HT / 8; c += 64) *reinterpret_cast<bf16x8*>(sQh + c * 8) = zero;
    }
    bf16x8 xN, lN;
    auto fetch = [&](int t) {
        int p = 32 * t + rl;
        p = p < nrows ? p : nrows - 1;                   // rows past the range: clamped load, zeroed at the point of use
        const unsigned b = (tok0 + (unsigned)p) * 128u + sub * 8;
        xN = *reinterpret_cast<const bf16x8*>(a.X + (size_t)b);
        if (BONE) lN = *reinterpret_cast<const bf16x8*>(a.XL + (size_t)b);
    };
    auto layernorm = [&](const bf16x8 raw, bf16* dst, const float* gp, const float* bp) {      // row rl from registers; gp/bp in LDS
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)raw[e];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        const float mean = reduce16(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] -= mean; q = __builtin_fmaf(v[e], v[e], q); }
        const float rstd = rsqrtf(reduce16(q) * (1.0f / 128.0f) + KASF_LN_EPS);
        const f32x4 g0v = *reinterpret_cast<const f32x4*>(gp + sub * 8), g1v = *reinterpret_cast<const f32x4*>(gp + sub * 8 + 4);
        const f32x4 b0v = *reinterpret_cast<const f32x4*>(bp + sub * 8), b1v = *reinterpret_cast<const f32x4*>(bp + sub * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = __builtin_fmaf(v[e], rstd * g0v[e], b0v[e]); v[4 + e] = __builtin_fmaf(v[4 + e], rstd * g1v[e], b1v[e]); }
        tile_store8(dst, rl, sub * 8, v);
    };
    auto wrap = [](int r) { return r >= HB ? r - HB : r; };
    auto core = [&](int j, int sm, int lv) {            // frame j of the range: its rows sit at rolling offset sm = (17 j) mod 48 of the wave's tiles
        const int r32 = lv & 31, hh = lv >> 5;           // (lv: the lane id behind an optimisation barrier -- the addresses below are recomputed per tile instead of
                                                         //  living in registers across the loop: the self form has none to spare at two workgroups per CU)
        const int row = wrap(sm + r32);
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sKh + row * 16 + 8 * hh);
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(sQh + row * 16 + 8 * hh);
        f32x16 z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.f;
        f32x16 st = mfma32(kf, qf, z);                   // S^T[key][query]; register e holds key pos_of(e, hh): keys 0..15 in registers 0..7, key 16 in register 8 of half 0
        st[8] = hh == 0 ? st[8] : -INFINITY;
        float mx = st[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, st[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        constexpr float C2 = 0.25f * 1.4426950408889634f;     // scale . log2(e)
        const float nm = -mx * C2;
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 9; ++e) { st[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[e], C2, nm)); sum += st[e]; }
#pragma unroll
        for (int e = 9; e < 16; ++e) st[e] = 0.f;
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        auto vfrag = [&](int ks) {                       // tr_frag at the rolling offset
            const int u = lv & 15, q = u >> 2, p = u & 3, k0 = 16 * ks + 4 * hh;
            typedef __attribute__((address_space(3))) bf16x4 lds_v4;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(sVh + wrap(sm + k0 + q) * 16 + 4 * p));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(sVh + wrap(sm + k0 + 8 + q) * 16 + 4 * p));
            return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        f32x16 ot = mfma32(vfrag(0), pack8(st, 0), z);
        ot = mfma32(vfrag(1), pack8(st, 1), ot);
        if (r32 < J) {                                   // queries past 16 belong to the next frame
#pragma unroll
            for (int e = 0; e < 8; ++e) ot[e] *= inv;
            float o0[4] = {ot[0], ot[1], ot[2], ot[3]}, o1[4] = {ot[4], ot[5], ot[6], ot[7]};
            const int prow = (J * j + r32) & 63;
            store4(sO + Tile<bf16>::off4(prow, 16 * w + 4 * hh), o0);
            store4(sO + Tile<bf16>::off4(prow, 16 * w + 8 + 4 * hh), o1);
        }
    };
    fetch(0);
    __syncthreads();                                     // sLn
    int nfr = 0, sm = 0;                                 // next frame for the core, its rolling offset
    for (int t = 0; t <= ntiles; ++t) {
        const int par = t & 1;
        int lv = lane;
        asm volatile("" : "+v"(lv));
        if (t < ntiles) {
            const bf16x8 zero = {};
            const bool live = 32 * t + rl < nrows;
            const bf16x8 xc = live ? xN : zero;
            *reinterpret_cast<bf16x8*>(sX + par * AB_TILE + Tile<bf16>::chunk_off(rl, sub)) = xc;
            layernorm(xc, sA, sLn, sLn + 128);
            if (BONE) layernorm(live ? lN : zero, sA + AB_TILE, sLn + 256, sLn + 384);
        }
        __syncthreads();                                 // B1: raw and LN tiles complete; every wave finished the copy-out of the previous tile
        if (t < ntiles) {
            const int hb = (2 * t) % 3;                  // tile t's rows start at rolling row 16 hb = (32 t) mod 48; a 16-row block never straddles the wrap
            auto blk = [&](int mt) { const int b = hb + mt; return (b >= 3 ? b - 3 : b) * 16; };
            if (BONE) {   // ---- q_h from LN(x), then k_h, v_h from LN_limb(x_limb): two phases keep the live accumulators + operand fragments under the register cap ----
                {
                    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        acc[0] = mfma16(wq[0][ks], tok_frag(sA, i, ks), acc[0]);
                        acc[1] = mfma16(wq[0][ks], tok_frag(sA, 16 + i, ks), acc[1]);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        float v[4] = {acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]};
                        store4(sQh + (blk(mt) + i) * 16 + 4 * g, v);
                    }
                }
                f32x4 acc[2][2];
                zero_acc(acc);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 l0 = tok_frag(sA + AB_TILE, i, ks), l1 = tok_frag(sA + AB_TILE, 16 + i, ks);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc[nt][0] = mfma16(wq[1 + nt][ks], l0, acc[nt][0]);
                        acc[nt][1] = mfma16(wq[1 + nt][ks], l1, acc[nt][1]);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
                        store4(sQh + (1 + nt) * HT + (blk(mt) + i) * 16 + 4 * g, v);
                    }
            } else {   // ---- q_h, k_h, v_h of the tile's 32 tokens: 3 feature tiles x 2 token tiles ----
                f32x4 acc[3][2];
                zero_acc(acc);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 f0v = tok_frag(sA, i, ks), f1v = tok_frag(sA, 16 + i, ks);
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) {
                        acc[nt][0] = mfma16(wq[nt][ks], f0v, acc[nt][0]);
                        acc[nt][1] = mfma16(wq[nt][ks], f1v, acc[nt][1]);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
                        store4(sQh + nt * HT + (blk(mt) + i) * 16 + 4 * g, v);                    // wave-private rolling [pos][16]
                    }
            }
            lds_fence();
            while (nfr < nf && J * (nfr + 1) <= 32 * (t + 1)) {      // every frame that ends inside this tile (uniform: one or two)
                core(nfr, sm, lv);
                ++nfr;
                sm = wrap(sm + J);
            }
        }
        __syncthreads();                                 // B2: the heads of every frame that ends in tile t are in sO: tile t - 1 is complete
        if (t + 1 < ntiles) fetch(t + 1);                // (behind the core: its 4 (8) registers do not fit beside the core's under the 128-VGPR cap of two workgroups per CU)
        if (t >= 1) {   // ---- output projection + layer-scale + residual of tile t - 1: 16 channels x 32 tokens per wave ----
            const int i = lv & 15, g = lv >> 4;
            const bf16* cO = sO + (par ^ 1) * AB_TILE;
            const bf16* cX = sX + (par ^ 1) * AB_TILE;
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0] = mfma16(wp[ks], tok_frag(cO, i, ks), acc[0]);
                acc[1] = mfma16(wp[ks], tok_frag(cO, 16 + i, ks), acc[1]);
            }
            const f32x4 bpv = *reinterpret_cast<const f32x4*>(sLn + 512 + 16 * w + 4 * g), lsv = *reinterpret_cast<const f32x4*>(sLn + 640 + 16 * w + 4 * g);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float x[4], v[4];
                load4(cX + Tile<bf16>::off4(16 * mt + i, 16 * w + 4 * g), x);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = x[r] + lsv[r] * (acc[mt][r] + bpv[r]);
                store4(sA + Tile<bf16>::off4(16 * mt + i, 16 * w + 4 * g), v);       // x_mid staging: LN(x_t) has no readers left; thread (rl, sub) reads the chunk back below
            }
        }
        __syncthreads();                                 // B3: x_mid tile complete
        const int rl = 4 * w + (lv >> 4), sub = lv & 15;
        if (t >= 1 && 32 * (t - 1) + rl < nrows) {   // ---- full-row stores: x_mid always; o only when the backward pass will need it ----
            const unsigned tok = tok0 + (unsigned)(32 * (t - 1) + rl);
            const int co = Tile<bf16>::chunk_off(rl, sub);
            *reinterpret_cast<f32x4*>(a.OUT + (size_t)(tok * 128u + sub * 8)) = *reinterpret_cast<const f32x4*>(sA + co);
            if (a.Qs != nullptr) *reinterpret_cast<f32x4*>(a.Os + (size_t)(tok * 128u + sub * 8)) = *reinterpret_cast<const f32x4*>(sO + (par ^ 1) * AB_TILE + co);
        }
        if (a.Qs != nullptr && t < ntiles) {   // training: the backward pass reads q | k | v.  Round 6: the rows of tile t leave as whole rows, thread = (token, 16-byte chunk), out of
            // the heads' rolling tiles (complete since B2; tile t + 1's projections, behind the next B1, overwrite tile t's first 16-row block) -- rounds 4-5: 32-byte slices per head.
            const int hb = (2 * t) % 3;
            auto rrow = [&](int r) { const int b = hb + (r >> 4); return (b >= 3 ? b - 3 : b) * 16 + (r & 15); };      // rolling row of the tile's row r
            if (BONE) {
                if (32 * t + rl < nrows) {
                    const unsigned tok = tok0 + (unsigned)(32 * t + rl);
                    *reinterpret_cast<f32x4*>(a.Qs + (size_t)(tok * 128u + sub * 8)) = *reinterpret_cast<const f32x4*>(sHead + (sub >> 1) * WS + rrow(rl) * 16 + 8 * (sub & 1));
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int idx = 64 * w + lv + AB_THR * k, row = idx >> 5, c = idx & 31, part = 1 + (c >> 4), ch = c & 15;
                    if (32 * t + row < nrows) {
                        const unsigned tok = tok0 + (unsigned)(32 * t + row);
                        *reinterpret_cast<f32x4*>(a.KVs + (size_t)(tok * 256u + c * 8)) = *reinterpret_cast<const f32x4*>(sHead + (ch >> 1) * WS + part * HT + rrow(row) * 16 + 8 * (ch & 1));
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int idx = 64 * w + lv + AB_THR * k, row = idx / 48, c = idx - row * 48, part = c >> 4, ch = c & 15;
                    if (32 * t + row < nrows) {
                        const unsigned tok = tok0 + (unsigned)(32 * t + row);
                        *reinterpret_cast<f32x4*>(a.Qs + (size_t)(tok * 384u + c * 8)) = *reinterpret_cast<const f32x4*>(sHead + (ch >> 1) * WS + part * HT + rrow(row) * 16 + 8 * (ch & 1));
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same block for groups of 33..96 positions (temporal attention at T = 81): three 32-position tiles per group, one workgroup per CU
// (147 KB of LDS: raw x, ONE LayerNorm tile, head outputs, 72 KB of wave-private q | k | v tiles).  What differs from the one-tile form:
//   * the LayerNorm tile is used twice per group in the bone form (LN(x) for q, then LN_limb(x_limb) for k | v: two more barriers, negligible
//     against a group this long) and a third time as the x_mid staging tile of the copy-out;
//   * the core is k_attn_fwd_mfma<3>'s: per 32-query tile three score tiles, lane-local softmax over 96 keys, six PV products.
// Replaces three launches (LN + QKV linear, attention core, proj + residual linear; four in the bone form) and the round trips of q | k | v
// and the head outputs through HBM between them.
// ---------------------------------------------------------------------------------------------------------------
template <bool BONE>
__global__ __launch_bounds__(AB_THR, 2) void k_attn_blk_fwd_rp3(const AttnBlkArgs a) {
    constexpr int NKT = 3, R = 32 * NKT, RT = R * 128, HT = R * 16, WS = 3 * HT + RP3_PAD;      // WS: one wave's q | k | v tiles + 32 bytes (the 8 heads' images 32 bytes apart
                                                                                                   // modulo the 256 bytes of a read cycle: the row copy-out reads 16 lanes = 8 heads x 2 halves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sX = reinterpret_cast<bf16*>(smem);           // [96][128] raw x (the residual's operand)
    bf16* sA = sX + RT;                                 // [96][128] LN(x), then (bone) LN_limb(x_limb), then x_mid
    bf16* sO = sA + RT;                                 // [96][128] attention output of the 8 heads
    bf16* sHead = sO + RT;                              // [8 waves][q | k | v][96][16] wave-private operand tiles
    float* sLn = reinterpret_cast<float*>(sHead + 8 * WS);           // [6][128]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, sub = lane & 15, rl = threadIdx.x >> 4;
    const int r32 = lane & 31, hh = lane >> 5;
    const int L = a.L;
    const int per = (a.groups + gridDim.x - 1) / gridDim.x;
    const int g0 = blockIdx.x * per;
    int ng = a.groups - g0;
    if (ng > per) ng = per;
    if (ng <= 0) return;
    bf16* sQh = sHead + w * WS;
    bf16* sKh = sQh + HT;
    bf16* sVh = sKh + HT;
    bf16x8 wq[3][4], wp[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (BONE) {
            wq[0][ks] = *reinterpret_cast<const bf16x8*>(a.Wq + (int64_t)(16 * w + i) * 128 + 32 * ks + 8 * g);
            wq[1][ks] = *reinterpret_cast<const bf16x8*>(a.Wkv + (int64_t)(16 * w + i) * 128 + 32 * ks + 8 * g);
            wq[2][ks] = *reinterpret_cast<const bf16x8*>(a.Wkv + (int64_t)(128 + 16 * w + i) * 128 + 32 * ks + 8 * g);
        } else {
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) wq[nt][ks] = *reinterpret_cast<const bf16x8*>(a.Wq + (int64_t)(128 * nt + 16 * w + i) * 128 + 32 * ks + 8 * g);
        }
        wp[ks] = *reinterpret_cast<const bf16x8*>(a.Wproj + (int64_t)(16 * w + i) * 128 + 32 * ks + 8 * g);
    }
    if (threadIdx.x < 128) {
        sLn[threadIdx.x] = a.ln_g[threadIdx.x];
        sLn[128 + threadIdx.x] = a.ln_b[threadIdx.x];
        if (BONE) { sLn[256 + threadIdx.x] = a.lnl_g[threadIdx.x]; sLn[384 + threadIdx.x] = a.lnl_b[threadIdx.x]; }
        sLn[512 + threadIdx.x] = a.bproj[threadIdx.x];
        sLn[640 + threadIdx.x] = a.ls1[threadIdx.x];
    }
    const int stride = a.mode == 0 ? 1 : KASF_J;
    auto base_of = [&](int G) { return a.mode == 0 ? G * KASF_J : (G / KASF_J) * a.T * KASF_J + (G % KASF_J); };
    unsigned ox[NKT];
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
        const int row = rl + 32 * j, rc = row < L ? row : L - 1;
        ox[j] = (unsigned)(rc * stride) * 128u + sub * 8;
    }
    bf16x8 xN[NKT], lN[NKT];
    auto fetch = [&](int t) {
        const unsigned b = (unsigned)base_of(g0 + t) * 128u;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            xN[j] = *reinterpret_cast<const bf16x8*>(a.X + (size_t)(b + ox[j]));
            if (BONE) lN[j] = *reinterpret_cast<const bf16x8*>(a.XL + (size_t)(b + ox[j]));
        }
    };
    auto layernorm = [&](const bf16x8 raw, int row, const float* gp, const float* bp) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)raw[e];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        const float mean = reduce16(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] -= mean; q = __builtin_fmaf(v[e], v[e], q); }        // (explicit fma: the library is built with -ffp-contract=off)
        const float rstd = rsqrtf(reduce16(q) * (1.0f / 128.0f) + KASF_LN_EPS);
        const f32x4 g0v = *reinterpret_cast<const f32x4*>(gp + sub * 8), g1v = *reinterpret_cast<const f32x4*>(gp + sub * 8 + 4);
        const f32x4 b0v = *reinterpret_cast<const f32x4*>(bp + sub * 8), b1v = *reinterpret_cast<const f32x4*>(bp + sub * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = __builtin_fmaf(v[e], rstd * g0v[e], b0v[e]); v[4 + e] = __builtin_fmaf(v[4 + e], rstd * g1v[e], b1v[e]); }
        tile_store8(sA, row, sub * 8, v);
    };
    // feature tiles [nt0, nt0 + NN) of this head out of the rows in sA -> the wave-private tiles
    auto project = [&](auto NT0, auto NNc) {
        constexpr int nt0 = decltype(NT0)::value, NN = decltype(NNc)::value;
#pragma unroll
        for (int jb = 0; jb < NKT; ++jb) {
            f32x4 acc[NN][2];
            zero_acc(acc);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 f0 = tok_frag(sA, 32 * jb + i, ks), f1 = tok_frag(sA, 32 * jb + 16 + i, ks);
#pragma unroll
                for (int nt = 0; nt < NN; ++nt) {
                    acc[nt][0] = mfma16(wq[nt0 + nt][ks], f0, acc[nt][0]);
                    acc[nt][1] = mfma16(wq[nt0 + nt][ks], f1, acc[nt][1]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NN; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
                    store4(sQh + (nt0 + nt) * HT + (32 * jb + 16 * mt + i) * 16 + 4 * g, v);
                }
        }
    };
    fetch(0);
    __syncthreads();                                     // sLn
#ifdef RP3_PROF
    long long acc_t[8] = {0,0,0,0,0,0,0,0};
#define RT(k) do { const long long n_ = clock64(); acc_t[k] += n_ - t_; t_ = n_; } while (0)
#else
#define RT(k) do {} while (0)
#endif
    for (int t = 0; t < ng; ++t) {
#ifdef RP3_PROF
        long long t_ = clock64();
#endif
        const int G = g0 + t;
        const bf16x8 zero = {};
        bf16x8 lc[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            const int row = rl + 32 * j;
            const bf16x8 xc = row < L ? xN[j] : zero;
            lc[j] = row < L ? lN[j] : zero;
            *reinterpret_cast<bf16x8*>(sX + Tile<bf16>::chunk_off(row, sub)) = xc;
            layernorm(xc, row, sLn, sLn + 128);
        }
        if (t + 1 < ng) fetch(t + 1);
        RT(0);
        __syncthreads();                                 // B1: raw and LN(x) tiles complete; every wave finished the copy-out of the previous group
        RT(1);
        if (BONE) {
            project(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});          // q_h from LN(x)
            __syncthreads();                             // every wave is done with LN(x)
#pragma unroll
            for (int j = 0; j < NKT; ++j) layernorm(lc[j], rl + 32 * j, sLn + 256, sLn + 384);
            __syncthreads();                             // LN_limb(x_limb) complete
            project(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});          // k_h, v_h
        } else {
            project(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
        }
        lds_fence();
        RT(2);
        RT(3);
        {   // ---- attention core of head w (k_attn_fwd_mfma<3>) ----
            bf16x8 kf[NKT];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) kf[kt] = *reinterpret_cast<const bf16x8*>(sKh + (32 * kt + r32) * 16 + 8 * hh);
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
            for (int qt = 0; qt < NKT; ++qt) {
                if (32 * qt >= L) break;
                const bf16x8 qf = *reinterpret_cast<const bf16x8*>(sQh + (32 * qt + r32) * 16 + 8 * hh);
                // Softmax with the fewest vector instructions per score (the core is bound by their issue: in-kernel stamps, 41 % of the launch): the maximum is taken
                // over the RAW scores (a positive scale commutes with max), the scale and the log2(e) of the exponential are one fma, only the tile that can hold
                // keys past L is masked, and the 1 / sum is applied to the 16 outputs instead of the 48 probabilities (P <= 1 either way: same bf16 range).
                f32x16 st[NKT];
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    st[kt] = mfma32(kf[kt], qf, z);
                    if (kt > 0) {                             // (keys 0..31 are live for every group this kernel takes: L > 32)
#pragma unroll
                        for (int e = 0; e < 16; ++e) st[kt][e] = (32 * kt + pos_of(e, hh) < L) ? st[kt][e] : -INFINITY;
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[kt][e]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                constexpr float C2 = 0.25f * 1.4426950408889634f;     // scale . log2(e)
                const float nm = -mx * C2;
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { st[kt][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][e], C2, nm)); sum += st[kt][e]; }
                sum += __shfl_xor(sum, 32);
                const float inv = 1.0f / sum;
                if (a.LSE != nullptr && hh == 0 && 32 * qt + r32 < L)      // the backward pass rebuilds P = exp(s - lse) tile by tile without a statistics pass
                    a.LSE[(size_t)(unsigned)(base_of(G) + (32 * qt + r32) * stride) * 8u + w] = mx * 0.25f + __logf(sum);
                f32x16 ot = z;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    ot = mfma32(tr_frag(sVh, 2 * kt), pack8(st[kt], 0), ot);
                    ot = mfma32(tr_frag(sVh, 2 * kt + 1), pack8(st[kt], 1), ot);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) ot[e] *= inv;
                float o0[4] = {ot[0], ot[1], ot[2], ot[3]}, o1[4] = {ot[4], ot[5], ot[6], ot[7]};
                store4(sO + Tile<bf16>::off4(32 * qt + r32, 16 * w + 4 * hh), o0);
                store4(sO + Tile<bf16>::off4(32 * qt + r32, 16 * w + 8 + 4 * hh), o1);
            }
        }
        RT(4);
        __syncthreads();                                 // B2: all heads in sO; sA has no readers left
        RT(5);
        {   // ---- output projection + layer-scale + residual: 16 channels x 96 positions per wave, x_mid staged in sA ----
            const f32x4 bpv = *reinterpret_cast<const f32x4*>(sLn + 512 + 16 * w + 4 * g), lsv = *reinterpret_cast<const f32x4*>(sLn + 640 + 16 * w + 4 * g);
#pragma unroll
            for (int jb = 0; jb < NKT; ++jb) {
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    acc[0] = mfma16(wp[ks], tok_frag(sO, 32 * jb + i, ks), acc[0]);
                    acc[1] = mfma16(wp[ks], tok_frag(sO, 32 * jb + 16 + i, ks), acc[1]);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float x[4], v[4];
                    load4(sX + Tile<bf16>::off4(32 * jb + 16 * mt + i, 16 * w + 4 * g), x);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = x[r] + lsv[r] * (acc[mt][r] + bpv[r]);
                    store4(sA + Tile<bf16>::off4(32 * jb + 16 * mt + i, 16 * w + 4 * g), v);
                }
            }
        }
        RT(6);
        __syncthreads();                                 // B3: x_mid tile complete
#pragma unroll
        for (int j = 0; j < NKT; ++j) {   // ---- full-row stores: x_mid always; o only when the backward pass will need it ----
            const int row = rl + 32 * j;
            if (row < L) {
                const unsigned tok = (unsigned)(base_of(G) + row * stride);
                const int co = Tile<bf16>::chunk_off(row, sub);
                *reinterpret_cast<f32x4*>(a.OUT + (size_t)(tok * 128u + sub * 8)) = *reinterpret_cast<const f32x4*>(sA + co);
                if (a.Qs != nullptr) *reinterpret_cast<f32x4*>(a.Os + (size_t)(tok * 128u + sub * 8)) = *reinterpret_cast<const f32x4*>(sO + co);
            }
        }
        if (a.Qs != nullptr) {   // training: the backward pass reads q | k | v.  Round 6: whole rows, thread = (position, 16-byte chunk), out of the heads' tiles (complete
            // since B2, untouched until the next group's projections) -- rounds 2-5 had every wave store its head's 32-byte slice of each row right after the projections:
            // 32 separate requests per store instruction, 24 partial writes per 768-byte row.
            if (BONE) {          // q rows (256 bytes) and k | v rows (512 bytes) live in two arrays
#pragma unroll
                for (int j = 0; j < NKT; ++j) {
                    const int row = rl + 32 * j;
                    if (row < L) {
                        const unsigned tok = (unsigned)(base_of(G) + row * stride);
                        *reinterpret_cast<f32x4*>(a.Qs + (size_t)(tok * 128u + sub * 8)) = *reinterpret_cast<const f32x4*>(sHead + (sub >> 1) * WS + row * 16 + 8 * (sub & 1));
                    }
                }
#pragma unroll
                for (int k = 0; k < 2 * NKT; ++k) {
                    const int idx = (int)threadIdx.x + AB_THR * k, row = idx >> 5, c = idx & 31, part = 1 + (c >> 4), ch = c & 15;
                    if (row < L) {
                        const unsigned tok = (unsigned)(base_of(G) + row * stride);
                        *reinterpret_cast<f32x4*>(a.KVs + (size_t)(tok * 256u + c * 8)) = *reinterpret_cast<const f32x4*>(sHead + (ch >> 1) * WS + part * HT + row * 16 + 8 * (ch & 1));
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < R * 48 / AB_THR; ++k) {
                    const int idx = (int)threadIdx.x + AB_THR * k, row = idx / 48, c = idx - row * 48, part = c >> 4, ch = c & 15;
                    if (row < L) {
                        const unsigned tok = (unsigned)(base_of(G) + row * stride);
                        *reinterpret_cast<f32x4*>(a.Qs + (size_t)(tok * 384u + c * 8)) = *reinterpret_cast<const f32x4*>(sHead + (ch >> 1) * WS + part * HT + row * 16 + 8 * (ch & 1));
                    }
                }
            }
        }
        __syncthreads();                                 // B4: the next group's LayerNorm overwrites sA
        RT(7);
    }
#ifdef RP3_PROF
    if (blockIdx.x == 77 && (threadIdx.x == 0 || threadIdx.x == 320)) printf("rp3 prof bone %d wave %d groups %d: LN %lld B1 %lld project %lld qkvstore %lld core %lld B2 %lld proj %lld copyout+B3+B4 %lld\n", (int)BONE, w, ng, acc_t[0], acc_t[1], acc_t[2], acc_t[3], acc_t[4], acc_t[5], acc_t[6], acc_t[7]);
#endif
}

}  // namespace

// Returns false when the shape is outside the fused kernels' range (groups longer than 96 positions): the caller runs the unfused sequence.
bool kasf_launch_attn_block_fwd(hipStream_t s, int bone, const void* x, const void* x_limb, const float* ln_g, const float* ln_b, const float* lnl_g,
                                const float* lnl_b, const void* Wq, const void* Wkv, const void* Wproj, const float* bproj, const float* ls1, void* q_save,
                                void* kv_save, void* o_save, void* out, int B, int T, int mode, float* lse_save) {
    const int L = mode == 0 ? KASF_J : T;
    if (L > 96) return false;
    AttnBlkArgs a;
    a.X = (const bf16*)x; a.XL = (const bf16*)x_limb; a.ln_g = ln_g; a.ln_b = ln_b; a.lnl_g = lnl_g; a.lnl_b = lnl_b;
    a.Wq = (const bf16*)Wq; a.Wkv = (const bf16*)Wkv; a.Wproj = (const bf16*)Wproj; a.bproj = bproj; a.ls1 = ls1;
    a.Qs = (bf16*)q_save; a.KVs = (bf16*)kv_save; a.Os = (bf16*)o_save; a.OUT = (bf16*)out; a.LSE = lse_save;
    a.L = L; a.T = T; a.mode = mode; a.groups = mode == 0 ? B * T : B * KASF_J;
    if (a.groups <= 0) return true;
    if (L > 32) {                                       // three-tile groups: one workgroup per CU
        const unsigned grid = (unsigned)(a.groups < 256 ? a.groups : 256);
        const size_t sh = (size_t)3 * 96 * 128 * 2 + (size_t)8 * (3 * 96 * 16 + RP3_PAD) * 2 + 6 * 128 * 4;
        if (bone) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_blk_fwd_rp3<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            hipLaunchKernelGGL(k_attn_blk_fwd_rp3<true>, dim3(grid), dim3(AB_THR), sh, s, a);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_blk_fwd_rp3<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            hipLaunchKernelGGL(k_attn_blk_fwd_rp3<false>, dim3(grid), dim3(AB_THR), sh, s, a);
        }
        return true;
    }
    const int ns = bone ? 2 : 1;
    const int cap = kasf_narrow_grid(KASF_NG_ATTN_FWD, bone ? 256 : 512, (int64_t)a.groups * L);
#ifndef KASF_NO_FLAT_SPATIAL
    if (mode == 0) {                                    // the 17 joints of a frame: flat 32-token tiles over consecutive frames
        const int want = (a.groups + 1) / 2;            // >= 2 frames per workgroup (a lone frame would pay a whole tile + the lagging iteration)
        const unsigned grid = (unsigned)(want < cap ? (want < 1 ? 1 : want) : cap);
        const size_t sh = (size_t)(2 + ns + 2) * AB_TILE * 2 + (size_t)8 * (3 * 48 * 16 + AB_HPAD) * 2 + 6 * 128 * 4;
        if (bone) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_blk_fwd_flat<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            hipLaunchKernelGGL(k_attn_blk_fwd_flat<true>, dim3(grid), dim3(AB_THR), sh, s, a);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_blk_fwd_flat<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            hipLaunchKernelGGL(k_attn_blk_fwd_flat<false>, dim3(grid), dim3(AB_THR), sh, s, a);
        }
        return true;
    }
#endif
    const unsigned grid = (unsigned)(a.groups < cap ? a.groups : cap);
    const size_t sh = (size_t)(1 + ns + 1 + 1) * AB_TILE * 2 + 8 * (3 * 512 + AB_HPAD) * 2 + 6 * 128 * 4;
    if (bone) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_blk_fwd_rp<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipLaunchKernelGGL(k_attn_blk_fwd_rp<true>, dim3(grid), dim3(AB_THR), sh, s, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_blk_fwd_rp<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipLaunchKernelGGL(k_attn_blk_fwd_rp<false>, dim3(grid), dim3(AB_THR), sh, s, a);
    }
    return true;
}
