// MFMA attention cores for bf16 mode (reference: modules/selfattention.py:18-41, bone_crossattention.py:19-41).
// One wave owns one (group, head): group = the 17 joints of a frame (spatial) or the T frames of one joint
// (temporal); head dim 16 is exactly the K of v_mfma_f32_32x32x16_bf16, so one MFMA yields a 32x32 score tile.
//
// Orientation is chosen so that nothing ever moves between lanes:
//   S^T[key][query] = K . Q^T   (A = K rows, B = Q rows: both are plain 16-byte row loads)
//   -> D layout: lane = query, registers = keys  => softmax statistics are lane-local (+1 xor-32 exchange)
//   O^T[d][query]   = V^T . P^T (A = V^T fragments by ds_read_b64_tr_b16 from the row-major V tile in LDS,
//                                B = P^T straight from this lane's score registers, normalised and cast to bf16)
//   -> D layout: lane = query, registers = d  => two 8-byte stores per lane, normalisation is lane-local.
// The accumulator->operand reuse relies on the 32x32 C/D map (row = (reg&3) + 8(reg>>2) + 4(lane>>5)): registers
// 8s..8s+7 of a tile are k-step s of the next product with key  kappa = 16s + 8(j>>2) + 4(lane>>5) + (j&3),
// and the transposed LDS reads fetch the other operand in exactly that key order.
// Backward runs two passes per (group, head): pass 1 with lane = query (softmax statistics, delta, dQ), pass 2
// with lane = key (S and dP recomputed un-transposed; dK, dV), exchanging only 3 floats per query through LDS.
#include <cstdlib>
#include <cstdio>
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ int64_t tok_of(int G, int i, int T, int mode) {
    return mode == 0 ? (int64_t)G * KASF_J + i : (int64_t)(G / KASF_J) * T * KASF_J + (int64_t)i * KASF_J + (G % KASF_J);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
    return z;
}
// position (key or query index inside its 32-tile) held by register `reg` of a 32x32 accumulator in lane half hh
__device__ __forceinline__ int pos_of(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// natural row fragment: 8 consecutive head channels of position `pos` (zero beyond L); optionally mirrored into the LDS tile
__device__ __forceinline__ bf16x8 row_frag(const bf16* base, int64_t ld, int G, int pos, int L, int Tn, int mode, int h, int hh, bf16* s_tile) {
    bf16x8 v = zero8();
    if (pos < L) v = *reinterpret_cast<const bf16x8*>(base + tok_of(G, pos, Tn, mode) * ld + h * 16 + 8 * hh);
    if (s_tile != nullptr) *reinterpret_cast<bf16x8*>(s_tile + pos * 16 + 8 * hh) = v;
    return v;
}
// transposed fragment for k-step ks out of a row-major [positions][16] tile: element j of lane half hh = tile[kappa(ks,hh,j)][lane & 15]
__device__ __forceinline__ bf16x8 tr_frag(const bf16* s_tile, int ks) {
    const int lane = threadIdx.x & 63, u = lane & 15, hh = lane >> 5, q = u >> 2, p = u & 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    const int k0 = 16 * ks + 4 * hh;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (k0 + q) * 16 + 4 * p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (k0 + 8 + q) * 16 + 4 * p));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// same, out of a row-major [32 rows][32 columns] tile: element j of lane (column = lane & 31, half hh) = tile[kappa(ks,hh,j)][lane & 31].
// Turns a tile written with lane = row into the operand of a product whose lanes are the columns (the 32 x 32 transpose of P / dS).
__device__ __forceinline__ bf16x8 tr_frag32(const bf16* s_tile, int ks) {
    const int lane = threadIdx.x & 63, u = lane & 15, hh = lane >> 5, q = u >> 2, p = u & 3, c0 = 16 * ((lane >> 4) & 1);
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    const int k0 = 16 * ks + 4 * hh;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (k0 + q) * 32 + c0 + 4 * p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (k0 + 8 + q) * 32 + c0 + 4 * p));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// registers 8s..8s+7 of an accumulator tile -> bf16 operand fragment of k-step s
__device__ __forceinline__ bf16x8 pack8(const f32x16& t, int s) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)t[8 * s + j];
    return o;
}
// lane = position (query or key), registers 0..7 = channels {4hh..4hh+3, 8+4hh..8+4hh+3} of the head
__device__ __forceinline__ void store_t(bf16* dst, const f32x16& t, int hh) {
    float a[4] = {t[0], t[1], t[2], t[3]}, b[4] = {t[4], t[5], t[6], t[7]};
    store4(dst + 4 * hh, a);
    store4(dst + 8 + 4 * hh, b);
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const bf16x2_t t = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, t);
}
// lane = position, registers 0..7 of t = channels {4hh..4hh+3, 8+4hh..8+4hh+3}  ->  after the swap lane hh holds channels 8hh .. 8hh+7
__device__ __forceinline__ u32x4_t swap_t16(const f32x16& t) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(pack2(t[0], t[1]), pack2(t[4], t[5]), false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(pack2(t[2], t[3]), pack2(t[6], t[7]), false, false);
    return u32x4_t{s0[0], s1[0], s0[1], s1[1]};
}
// the same tile as ONE 16-byte store per lane (32 contiguous bytes per position and head; store_t: two 8-byte pieces per lane, 64 requests per wave store)
__device__ __forceinline__ void store_t16(bf16* dst, const f32x16& t, int hh) { *reinterpret_cast<u32x4_t*>(dst + 8 * hh) = swap_t16(t); }

template <int NKT>
__global__ __launch_bounds__(256) void k_attn_fwd_mfma(const bf16* __restrict__ Q, int64_t ldq, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                       int64_t ldkv, bf16* __restrict__ O, int L, int Tn, int mode, int units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int unit = blockIdx.x * 4 + wave;
    if (unit >= units) return;                                   // wave-uniform; no workgroup barrier below
    const int G = unit >> 3, h = unit & 7;
    bf16* sV = reinterpret_cast<bf16*>(smem) + wave * (NKT * 32 * 16);
    bf16x8 kf[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        kf[kt] = row_frag(K, ldkv, G, 32 * kt + r, L, Tn, mode, h, hh, nullptr);
        row_frag(V, ldkv, G, 32 * kt + r, L, Tn, mode, h, hh, sV);
    }
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt) {
        if (32 * qt >= L) break;
        const int i = 32 * qt + r;
        const bf16x8 qf = row_frag(Q, ldq, G, i, L, Tn, mode, h, hh, nullptr);
        // softmax with the fewest vector instructions per score (the cores are bound by their issue): max over the RAW scores (a positive scale commutes with
        // max), scale and log2(e) in one fma, and 1 / sum applied to the lane's 8 outputs instead of its 16 NKT probabilities
        f32x16 st[NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            st[kt] = mfma32(kf[kt], qf, zero16());
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                st[kt][g] = (32 * kt + pos_of(g, hh) < L) ? st[kt][g] : -INFINITY;
                mx = fmaxf(mx, st[kt][g]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        constexpr float C2 = 0.25f * 1.4426950408889634f;             // scale . log2(e)
        const float nm = -mx * C2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int g = 0; g < 16; ++g) { st[kt][g] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][g], C2, nm)); sum += st[kt][g]; }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        f32x16 ot = zero16();
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            ot = mfma32(tr_frag(sV, 2 * kt), pack8(st[kt], 0), ot);
            ot = mfma32(tr_frag(sV, 2 * kt + 1), pack8(st[kt], 1), ot);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) ot[g] *= inv;
        if (i < L) store_t16(O + tok_of(G, i, Tn, mode) * 128 + h * 16, ot, hh);
    }
}

// FDO (one-tile groups only): the gradient of the attention output is not read from memory but formed here, d_o = g_mid . (ls1 . Wproj)
// restricted to this head's 16 channels: the workgroup is the 8 heads of ONE group, g_mid's rows are staged once (LDS-direct) and each wave
// runs 8 MFMAs against its 16 rows of the packed weight.  Saves the d_o round trip (60 MB) and a launch per attention block.
template <int NKT, bool FDO>
__global__ __launch_bounds__(FDO ? 512 : 256) void k_attn_bwd_mfma(const bf16* __restrict__ Q, int64_t ldq, const bf16* __restrict__ K,
                                                                   const bf16* __restrict__ V, int64_t ldkv, const bf16* __restrict__ dO,
                                                                   bf16* __restrict__ dQ, int64_t lddq, bf16* __restrict__ dK, bf16* __restrict__ dV,
                                                                   int64_t lddkv, int L, int Tn, int mode, int units, const bf16* __restrict__ Gmid,
                                                                   const bf16* __restrict__ Wp) {
    static_assert(NKT == 1, "one-tile groups (<= 32 positions); longer groups: k_attn_bwd_long");
    constexpr int NWAVE = FDO ? 8 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = NKT * 32 * 16;                           // bf16 elements of one [positions][16] tile
    constexpr int WAVE_BYTES = 3 * TILE * 2 + NKT * 32 * 16 + (NKT == 1 ? 2 * 32 * 32 * 2 : 0);   // K, Q, dO tiles + [positions][4] fp32 statistics
                                                                                               // (+ P and dS tiles, one-tile groups)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int unit = blockIdx.x * NWAVE + wave;
    if (!FDO && unit >= units) return;                                  // (FDO: the grid is exactly one workgroup per group)
    const int G = unit >> 3, h = unit & 7;
    bf16* sK = reinterpret_cast<bf16*>(smem + wave * WAVE_BYTES);
    bf16* sQ = sK + TILE;
    bf16* sD = sQ + TILE;
    f32x4* sStat = reinterpret_cast<f32x4*>(sD + TILE);
    bf16* sP = reinterpret_cast<bf16*>(sStat + NKT * 32);             // NKT == 1: P[query][key] and dS[query][key] of pass 1, bf16
    bf16* sdS = sP + 32 * 32;
    bf16x8 kf[NKT], vf[NKT], qf[NKT], df[NKT];
    if (FDO) {
        bf16* sG = reinterpret_cast<bf16*>(smem + NWAVE * WAVE_BYTES);           // [32][128] g_mid rows of the group (swizzled tile)
        stage_tile_async<bf16, 32, 512>(sG, Gmid + tok_of(G, 0, Tn, mode) * 128, mode == 0 ? 128 : (int64_t)KASF_J * 128, L);
    }
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        const int pos = 32 * t + r;
        kf[t] = row_frag(K, ldkv, G, pos, L, Tn, mode, h, hh, sK);
        vf[t] = row_frag(V, ldkv, G, pos, L, Tn, mode, h, hh, nullptr);
        qf[t] = row_frag(Q, ldq, G, pos, L, Tn, mode, h, hh, sQ);
        if (!FDO) df[t] = row_frag(dO, 128, G, pos, L, Tn, mode, h, hh, sD);
    }
    if (FDO) {
        const bf16* sG = reinterpret_cast<const bf16*>(smem + NWAVE * WAVE_BYTES);
        const int li = lane & 15, lg = lane >> 4;
        bf16x8 wp[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wp[ks] = *reinterpret_cast<const bf16x8*>(Wp + (int64_t)(16 * h + li) * 128 + 32 * ks + 8 * lg);
        wait_async();
        __syncthreads();
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks], *reinterpret_cast<const bf16x8*>(sG + Tile<bf16>::chunk_off(li, 4 * ks + lg)), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks], *reinterpret_cast<const bf16x8*>(sG + Tile<bf16>::chunk_off(16 + li, 4 * ks + lg)), acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {             // d_o[pos = 16 mt + li][channel 4 lg .. 4 lg + 3] of this head; rows past L are zero like row_frag's
            const float live = 16 * mt + li < L ? 1.0f : 0.0f;
            float v[4] = {acc[mt][0] * live, acc[mt][1] * live, acc[mt][2] * live, acc[mt][3] * live};
            store4(sD + (16 * mt + li) * 16 + 4 * lg, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        df[0] = *reinterpret_cast<const bf16x8*>(sD + r * 16 + 8 * hh);
    }
    // ---------------- pass 1: lane = query ----------------
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt) {
        if (32 * qt >= L) break;
        const int i = 32 * qt + r;
        f32x16 st[NKT], dp[NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            st[kt] = mfma32(kf[kt], qf[qt], zero16());            // S^T[key][query]
            dp[kt] = mfma32(vf[kt], df[qt], zero16());            // dP^T[key][query] = sum_d V[key][d] dO[query][d]
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float s = (32 * kt + pos_of(g, hh) < L) ? st[kt][g] * 0.25f : -INFINITY;
                st[kt][g] = s;
                mx = fmaxf(mx, s);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int g = 0; g < 16; ++g) { st[kt][g] = __expf(st[kt][g] - mx); sum += st[kt][g]; }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        float delta = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int g = 0; g < 16; ++g) { st[kt][g] *= inv; delta += st[kt][g] * dp[kt][g]; }
        delta += __shfl_xor(delta, 32);
        f32x16 dq = zero16();
        if (NKT == 1) {   // one-tile groups: hand P to pass 2 through LDS instead of recomputing the softmax there (lane = query -> row of the tile;
                          // registers 4a..4a+3 are the consecutive keys 8a + 4hh ..)
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                float v4[4] = {st[0][4 * a4], st[0][4 * a4 + 1], st[0][4 * a4 + 2], st[0][4 * a4 + 3]};
                store4(sP + r * 32 + 8 * a4 + 4 * hh, v4);
            }
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int g = 0; g < 16; ++g) st[kt][g] = st[kt][g] * (dp[kt][g] - delta) * 0.25f;     // dS^T (scale folded)
            dq = mfma32(tr_frag(sK, 2 * kt), pack8(st[kt], 0), dq);                                // dQ^T[d][query] += K^T . dS^T
            dq = mfma32(tr_frag(sK, 2 * kt + 1), pack8(st[kt], 1), dq);
        }
        if (i < L) store_t16(dQ + tok_of(G, i, Tn, mode) * lddq + h * 16, dq, hh);
        if (NKT == 1) {
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                float v4[4] = {st[0][4 * a4], st[0][4 * a4 + 1], st[0][4 * a4 + 2], st[0][4 * a4 + 3]};
                store4(sdS + r * 32 + 8 * a4 + 4 * hh, v4);
            }
        }
    }
    if (NKT == 1) {   // ---------------- pass 2, one-tile groups: dV^T = dO^T . P, dK^T = Q^T . dS with P / dS read back transposed ----------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the tiles were written by other lanes of this wave (LDS is in order per wave)
        const int j = r;
        f32x16 dv = zero16(), dk = zero16();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            dv = mfma32(tr_frag(sD, ks), tr_frag32(sP, ks), dv);
            dk = mfma32(tr_frag(sQ, ks), tr_frag32(sdS, ks), dk);
        }
        if (j < L) {
            const int64_t tok = tok_of(G, j, Tn, mode);
            store_t16(dV + tok * lddkv + h * 16, dv, hh);
            store_t16(dK + tok * lddkv + h * 16, dk, hh);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Backward for groups of 33..96 positions (temporal attention at T = 81), one wave per (group, head), built to fit TWO waves per SIMD
// (<= 256 VGPRs, 16 KB of LDS per wave): the two-pass kernel above needs 296 VGPRs, runs one wave per SIMD and spends most of its time
// waiting on MFMA -> VALU -> LDS dependencies it cannot hide.  Here every query tile is visited once: its softmax statistics, delta and dQ
// are formed with lane = query; each 32 x 32 block of P and dS then goes through a 2 KB LDS tile and is read back transposed
// (ds_read_b64_tr_b16) as the operand of the dV / dK products (lane = key), which accumulate in registers across the query tiles.
// K and V fragments are re-read from LDS instead of living in registers.
// ---------------------------------------------------------------------------------------------------------------
// FDO (as in the one-tile kernels): d_o = g_mid . (ls1 . Wproj)^T is formed here instead of being read -- the workgroup is then the 8 heads of ONE group
// (8 waves, 152 KB of LDS, the same two waves per SIMD), the group's g_mid rows are staged once and each wave runs 24 MFMAs against its 16 rows of the
// packed weight.  Saves the d_o linear (a launch, 45 MB written and read back) at T = 81.
template <int NKT, bool FDO>
__global__ __launch_bounds__(FDO ? 512 : 256, 2) void k_attn_bwd_long(const bf16* __restrict__ Q, int64_t ldq, const bf16* __restrict__ K,
                                                                        const bf16* __restrict__ V, int64_t ldkv, const bf16* __restrict__ dO,
                                                                        bf16* __restrict__ dQ, int64_t lddq, bf16* __restrict__ dK, bf16* __restrict__ dV,
                                                                        int64_t lddkv, int L, int Tn, int mode, int units, const bf16* __restrict__ Gmid,
                                                                        const bf16* __restrict__ Wp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = NKT * 32 * 16;                           // [positions][16] operand tile
    constexpr int WAVE_BYTES = (4 * TILE + 2 * 32 * 32) * 2;
    constexpr int NWAVE = FDO ? 8 : 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int unit = blockIdx.x * NWAVE + wave;
    if (!FDO && unit >= units) return;                            // wave-uniform; no workgroup barrier below (FDO: the grid is one workgroup per group)
    const int G = unit >> 3, h = unit & 7;
    bf16* sK = reinterpret_cast<bf16*>(smem + wave * WAVE_BYTES);
    bf16* sV = sK + TILE;
    bf16* sQ = sV + TILE;
    bf16* sD = sQ + TILE;
    bf16* sP = sD + TILE;                                         // [32 queries][32 keys] block of P
    bf16* sdS = sP + 32 * 32;                                     // ... and of dS
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        const int pos = 32 * t + r;
        row_frag(K, ldkv, G, pos, L, Tn, mode, h, hh, sK);
        row_frag(V, ldkv, G, pos, L, Tn, mode, h, hh, sV);
        row_frag(Q, ldq, G, pos, L, Tn, mode, h, hh, sQ);
        if (!FDO) row_frag(dO, 128, G, pos, L, Tn, mode, h, hh, sD);
    }
    if (FDO) {
        bf16* sG = reinterpret_cast<bf16*>(smem + NWAVE * WAVE_BYTES);          // [32 NKT][128] g_mid rows of the group (swizzled tile; rows past L zero)
        for (int c = threadIdx.x; c < NKT * 32 * 16; c += 512) {
            const int row = c >> 4, ch = c & 15;
            bf16x8 v = zero8();
            if (row < L) v = *reinterpret_cast<const bf16x8*>(Gmid + tok_of(G, row, Tn, mode) * 128 + ch * 8);
            *reinterpret_cast<bf16x8*>(sG + Tile<bf16>::chunk_off(row, ch)) = v;
        }
        const int li = lane & 15, lg = lane >> 4;
        bf16x8 wp[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wp[ks] = *reinterpret_cast<const bf16x8*>(Wp + (int64_t)(16 * h + li) * 128 + 32 * ks + 8 * lg);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2 * NKT; ++mt) {          // d_o[pos = 16 mt + li][channel 4 lg .. 4 lg + 3] of this head
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks], *reinterpret_cast<const bf16x8*>(sG + Tile<bf16>::chunk_off(16 * mt + li, 4 * ks + lg)), acc, 0, 0, 0);
            float v[4] = {acc[0], acc[1], acc[2], acc[3]};
            store4(sD + (16 * mt + li) * 16 + 4 * lg, v);
        }
    }
    f32x16 dv[NKT], dk[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) { dv[t] = zero16(); dk[t] = zero16(); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    auto rowf = [&](const bf16* tile, int t) { return *reinterpret_cast<const bf16x8*>(tile + (32 * t + r) * 16 + 8 * hh); };
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt) {
        if (32 * qt >= L) break;
        const int i = 32 * qt + r;
        const bf16x8 qf = rowf(sQ, qt), dfq = rowf(sD, qt);
        f32x16 st[NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            st[kt] = mfma32(rowf(sK, kt), qf, zero16());          // S^T[key][query]
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float sv = (32 * kt + pos_of(g, hh) < L) ? st[kt][g] * 0.25f : -INFINITY;
                st[kt][g] = sv;
                mx = fmaxf(mx, sv);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int g = 0; g < 16; ++g) { st[kt][g] = __expf(st[kt][g] - mx); sum += st[kt][g]; }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        float delta = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const f32x16 dp = mfma32(rowf(sV, kt), dfq, zero16());       // dP^T[key][query]
#pragma unroll
            for (int g = 0; g < 16; ++g) { st[kt][g] *= inv; delta += st[kt][g] * dp[g]; }
        }
        delta += __shfl_xor(delta, 32);
        f32x16 dq = zero16();
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (32 * kt >= L) break;
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {                      // P block: registers 4a..4a+3 = keys 8a + 4hh .. of this key tile
                float v4[4] = {st[kt][4 * a4], st[kt][4 * a4 + 1], st[kt][4 * a4 + 2], st[kt][4 * a4 + 3]};
                store4(sP + r * 32 + 8 * a4 + 4 * hh, v4);
            }
            const f32x16 dp = mfma32(rowf(sV, kt), dfq, zero16());       // one more MFMA instead of 16 live registers per key tile
#pragma unroll
            for (int g = 0; g < 16; ++g) st[kt][g] = st[kt][g] * (dp[g] - delta) * 0.25f;          // dS^T (scale folded)
            dq = mfma32(tr_frag(sK, 2 * kt), pack8(st[kt], 0), dq);
            dq = mfma32(tr_frag(sK, 2 * kt + 1), pack8(st[kt], 1), dq);
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                float v4[4] = {st[kt][4 * a4], st[kt][4 * a4 + 1], st[kt][4 * a4 + 2], st[kt][4 * a4 + 3]};
                store4(sdS + r * 32 + 8 * a4 + 4 * hh, v4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the blocks were written by other lanes of this wave
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {                      // lane = key: dV^T += dO^T . P,  dK^T += Q^T . dS over the 32 queries of this tile
                dv[kt] = mfma32(tr_frag(sD, 2 * qt + ks), tr_frag32(sP, ks), dv[kt]);
                dk[kt] = mfma32(tr_frag(sQ, 2 * qt + ks), tr_frag32(sdS, ks), dk[kt]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next key tile overwrites the blocks
        }
        if (i < L) store_t16(dQ + tok_of(G, i, Tn, mode) * lddq + h * 16, dq, hh);
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int j = 32 * kt + r;
        if (j < L) {
            const int64_t tok = tok_of(G, j, Tn, mode);
            store_t16(dV + tok * lddkv + h * 16, dv[kt], hh);
            store_t16(dK + tok * lddkv + h * 16, dk[kt], hh);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward for groups of 33..96 positions with d_o formed in-kernel, KEY-TILE-OUTER (round 3; replaces k_attn_bwd_long<3, true>: 242 us per launch
// at T = 81, B = 128 = 1.3 TB/s, the kernel furthest below any roof).  What changed:
//   * no softmax-statistics pass and no recomputed dP: the forward (k_attn_blk_fwd_rp3) leaves lse = max + log(sum) per (token, head), so every
//     32 x 32 block of P is rebuilt on its own as exp(s / 4 - lse);  delta = rowsum(d_o . o) comes from the saved attention output, not from P . dP;
//   * scores are formed UN-transposed, S[query][key] = Q . K^T (lane = key, registers = queries): P and dS are then already in the operand layout
//     of dV^T = d_o^T . P and dK^T = Q^T . dS (the accumulator -> operand reuse of the forward core), so only ONE 2 KB block per tile pair -- dS, for
//     dQ^T = K^T . dS^T -- goes through LDS and comes back transposed, where the query-tile-outer kernel sent P and dS both ways;
//   * one key tile's dK / dV (32 registers) and the three query tiles' dQ (48) are live instead of 96 + 16, per tile pair 8 MFMAs instead of 9 + the
//     statistics pass;
//   * the group's rows (q | k | v, o, g_mid, lse) come in cooperatively, whole rows per wave load, and reach the heads' tiles through LDS; dq / dk / dv
//     leave as 16-byte stores (see the prologue's comment: the first form, one wave per head loading its 32-byte slice of each row, was bound by the
//     number of memory requests, not by anything the CUs did: 167 -> 142 us).
// One wave per (group, head); rounds 3-5: the 8 heads of a group per workgroup (147 KB of LDS: one workgroup per CU); round 6: 4 heads per workgroup, two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------
// tr_frag32 for a block whose rows are LD elements apart (LD = 36: the 72-byte row stride spreads the 32 row stores of a block over all banks;
// with 64-byte rows every fourth lane hit the same bank)
template <int LD>
__device__ __forceinline__ bf16x8 tr_frag32s(const bf16* s_tile, int ks) {
    const int lane = threadIdx.x & 63, u = lane & 15, hh = lane >> 5, q = u >> 2, p = u & 3, c0 = 16 * ((lane >> 4) & 1);
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    const int k0 = 16 * ks + 4 * hh;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (k0 + q) * LD + c0 + 4 * p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + (k0 + 8 + q) * LD + c0 + 4 * p));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

constexpr int KT_LD = 36;                                         // row stride of a dS block
constexpr int KT_HPW = 4;                                         // heads per workgroup
#ifndef KT_PAD
#define KT_PAD 32
#endif
// NR2: score registers of the LAST query tile that can hold a live query (register g holds queries (g & 3) + 8 (g >> 2) + 4 hh of the tile: 9 when
// the tile has at most 17 positions -- T = 81 = 32 + 32 + 17 --, else 16); the vector work and block stores of the dead registers are skipped.
template <int NKT, int NR2>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_kt(const bf16* __restrict__ Q, int64_t ldq, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                        int64_t ldkv, const bf16* __restrict__ O, const float* __restrict__ LSE, bf16* __restrict__ dQ,
                                                        int64_t lddq, bf16* __restrict__ dK, bf16* __restrict__ dV, int64_t lddkv, int L, int Tn, int mode,
                                                        const bf16* __restrict__ Gmid, const bf16* __restrict__ Wp, int groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = NKT * 32 * 16;                           // [positions][16] operand tile (bf16 elements)
    constexpr int BLK = 32 * KT_LD;                               // one dS block
    constexpr int WAVE_BYTES = 3 * TILE * 2 + BLK * 2 + 2 * NKT * 32 * 4 + KT_PAD;         // K, Q, d_o tiles; the dS block; lse, delta; 32 bytes that put the four heads'
    constexpr int VTILE = TILE + KT_PAD / 2;                      //   images 32 bytes apart modulo the 128 bytes of a write cycle: the 8 lanes of one have 8 places to go
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    // Round 6: FOUR heads of a group per workgroup (rounds 3-5: all eight; 147 KB of LDS = one workgroup per CU, nothing ran under the group's row loads and gradient
    // stores: 123 + 74 KB per group = ~15 k of its ~38 k cycles at the CU's share of HBM).  73.7 KB per workgroup admits two per CU at different phases of their
    // groups; blocks b and b + 8 -- the same XCD and neighbours in dispatch order -- are the two head halves of one group, so the g_mid rows both need (all 128
    // columns: d_o of a head is a product over the whole row) meet in that XCD's L2.
    const int G = ((int)blockIdx.x >> 4) * 8 + ((int)blockIdx.x & 7), hb = KT_HPW * (((int)blockIdx.x >> 3) & 1), h = hb + wave;
    if (G >= groups) return;                                      // (workgroup-uniform: groups is rounded up to a multiple of 8 by the launcher)
#ifdef KT_PROF
    long long kt_t[8]; int kt_n = 0;
#define KTQ() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); kt_t[kt_n++] = clock64(); } while (0)
#define KTQ0() do { kt_t[kt_n++] = clock64(); } while (0)
#else
#define KTQ() do {} while (0)
#define KTQ0() do {} while (0)
#endif
    KTQ0();
    bf16* sK = reinterpret_cast<bf16*>(smem + wave * WAVE_BYTES);
    bf16* sQ = sK + TILE;
    bf16* sD = sQ + TILE;
    bf16* sdS = sD + TILE;                                        // [32 keys][KT_LD]
    float* sLse = reinterpret_cast<float*>(sdS + BLK);            // [32 NKT]  -lse
    float* sDel = sLse + NKT * 32;                                //           -delta
    bf16* sG = reinterpret_cast<bf16*>(smem + KT_HPW * WAVE_BYTES);     // [32 NKT][128] g_mid rows of the group (swizzled tile; rows past L zero); once the d_o products
    bf16* sV = sG + wave * VTILE;                                 //   have read them, the heads' V tiles (4 x 3 KB of its 24)
    auto wave_tile = [&](int hd, int which) { return reinterpret_cast<bf16*>(smem + hd * WAVE_BYTES) + which * TILE; };          // 0 K, 1 Q, 2 d_o
    auto wave_stat = [&](int hd, int which) { return reinterpret_cast<float*>(smem + hd * WAVE_BYTES + 3 * TILE * 2 + BLK * 2) + which * NKT * 32; };
    // ---- the group's rows come in COOPERATIVELY: thread = (position, 16-byte chunk of the row), so a wave load covers whole 64-byte (this half's heads of q, k, v and
    // o) or 256-byte (g_mid) row pieces.  One wave per head loading its own 32-byte slice of each of 32 rows made every load and store 32-64 separate requests, and the
    // launch was bound by exactly that: 55 us of its 167 for loads + stores alone, the same total whatever the occupancy, the phase of the co-resident workgroup or the
    // number of vector instructions (in-kernel stamps: 20-30 k of a workgroup's 35-45 k cycles went by before its first loads had landed).
    // All loads are issued before anything waits; they are distributed to the heads' tiles through LDS.
    constexpr int ROWS = NKT * 32, NQKV = ROWS * 24 / 256, NRO = ROWS * 8 / 256, NRG = ROWS * 16 / 256;                 // 9, 3 and 6 chunks per thread
    bf16x8 cq[NQKV], co[NRO], cg[NRG];
    float cl[2];
    // Issue order = use order (the vector-memory counter retires in order): g_mid first -- the d_o products run while q | k | v, lse and o are still on their way.
#pragma unroll
    for (int k = 0; k < NRG; ++k) {
        const int idx = threadIdx.x + 256 * k, row = idx >> 4, ch = idx & 15;
        cg[k] = zero8();
        if (row < L) cg[k] = *reinterpret_cast<const bf16x8*>(Gmid + tok_of(G, row, Tn, mode) * 128 + ch * 8);
    }
    const int li = lane & 15, lg = lane >> 4;
    bf16x8 wp[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wp[ks] = *reinterpret_cast<const bf16x8*>(Wp + (int64_t)(16 * h + li) * 128 + 32 * ks + 8 * lg);
#pragma unroll
    for (int k = 0; k < NQKV; ++k) {
        const int idx = threadIdx.x + 256 * k, row = idx / 24, c = idx - row * 24, part = c >> 3, ch = c & 7;
        cq[k] = zero8();
        if (row < L) {
            const int64_t tk = tok_of(G, row, Tn, mode);
            cq[k] = *reinterpret_cast<const bf16x8*>(part == 0 ? Q + tk * ldq + hb * 16 + ch * 8 : (part == 1 ? K : V) + tk * ldkv + hb * 16 + ch * 8);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = threadIdx.x + 256 * k, row = idx >> 2, hd = idx & 3;
        cl[k] = INFINITY;                                         // rows past L: exp(s - inf) = 0 keeps them out of every product
        if (idx < ROWS * KT_HPW && row < L) cl[k] = LSE[tok_of(G, row, Tn, mode) * 8 + hb + hd];
    }
#pragma unroll
    for (int k = 0; k < NRO; ++k) {
        const int idx = threadIdx.x + 256 * k, row = idx >> 3, ch = idx & 7;
        co[k] = zero8();
        if (row < L) co[k] = *reinterpret_cast<const bf16x8*>(O + tok_of(G, row, Tn, mode) * 128 + hb * 16 + ch * 8);
    }
#pragma unroll
    for (int k = 0; k < NRG; ++k) {
        const int idx = threadIdx.x + 256 * k, row = idx >> 4, ch = idx & 15;
        *reinterpret_cast<bf16x8*>(sG + Tile<bf16>::chunk_off(row, ch)) = cg[k];
    }
    KTQ0();                                                       // 1: g_mid rows landed and written
    __syncthreads();
    {   // d_o = g_mid . (ls1 . Wproj)^T restricted to this head: 24 MFMAs against the wave's 16 rows of the packed weight
#pragma unroll
        for (int mt = 0; mt < 2 * NKT; ++mt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks], *reinterpret_cast<const bf16x8*>(sG + Tile<bf16>::chunk_off(16 * mt + li, 4 * ks + lg)), acc, 0, 0, 0);
            float v[4] = {acc[0], acc[1], acc[2], acc[3]};
            store4(sD + (16 * mt + li) * 16 + 4 * lg, v);
        }
    }
    KTQ0();                                                       // 2: d_o
#pragma unroll
    for (int k = 0; k < NQKV; ++k) {
        const int idx = threadIdx.x + 256 * k, row = idx / 24, c = idx - row * 24, part = c >> 3, ch = c & 7;
        if (part == 2) continue;                                  // V waits in its registers for the g_mid rows' place
        bf16x8 v = cq[k];
        if (part == 0) {                                          // Q / 4 (a power of two: exact in bf16)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] * 0.25f);
        }
        *reinterpret_cast<bf16x8*>(wave_tile(ch >> 1, part == 0 ? 1 : 0) + row * 16 + 8 * (ch & 1)) = v;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = threadIdx.x + 256 * k, row = idx >> 2, hd = idx & 3;
        if (idx < ROWS * KT_HPW) wave_stat(hd, 0)[row] = -cl[k];  // negated: lse and delta ride into the score products as initial accumulators
    }
    __syncthreads();                                              // every head's d_o tile is complete, the g_mid rows have been read
#pragma unroll
    for (int k = 0; k < NQKV; ++k) {
        const int idx = threadIdx.x + 256 * k, row = idx / 24, c = idx - row * 24, part = c >> 3, ch = c & 7;
        if (part == 2) *reinterpret_cast<bf16x8*>(sG + (ch >> 1) * VTILE + row * 16 + 8 * (ch & 1)) = cq[k];
    }
#pragma unroll
    for (int k = 0; k < NRO; ++k) {   // delta = sum_d d_o . o per (position, head): the thread that loaded 8 channels of o meets the same 8 of d_o; the head's two halves are neighbours
        const int idx = threadIdx.x + 256 * k, row = idx >> 3, ch = idx & 7;
        const bf16x8 d8 = *reinterpret_cast<const bf16x8*>(wave_tile(ch >> 1, 2) + row * 16 + 8 * (ch & 1));
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) part += (float)co[k][e] * (float)d8[e];
        part += __shfl_xor(part, 1);
        if ((ch & 1) == 0) wave_stat(ch >> 1, 1)[row] = -part;
    }
    __syncthreads();
    KTQ0();                                                       // 3: d_o, V, delta + two barriers
    auto rowf = [&](const bf16* tile, int t) { return *reinterpret_cast<const bf16x8*>(tile + (32 * t + r) * 16 + 8 * hh); };
    const int nt = (L + 31) >> 5;                                 // live 32-position tiles (2 or 3)
    bf16x8 kf[NKT], vf[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) { kf[t] = rowf(sK, t); vf[t] = rowf(sV, t); }
    f32x16 dq[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) dq[t] = zero16();
    // The tile pairs (kt, qt) are walked kt-major.  Program order inside a pair, chosen so that no LDS latency sits between dependent steps (two waves
    // per SIMD hide little): every LDS read the pair needs -- lse / delta of its queries, the transposed d_o / Q fragments, the row fragments of the
    // NEXT pair -- is issued first, then the next pair's two score products, then the vector work, then the dS block stores (all reads of the wave's
    // tiles are ahead of them: the compiler cannot move a read across a store it cannot disambiguate), the dV / dK products and last the block's
    // transposed read-back for dQ.  sQ holds Q / 4 (exact in bf16): S and dK come out scaled, dS is stored unscaled and dQ is scaled once at the end.
    // -lse and -delta of a pair's 16 queries (registers 4a .. 4a+3 = queries 32 qt + 8a + 4hh + {0..3}) are the INITIAL ACCUMULATORS of its two score
    // products: S / 4 - lse and dP - delta come out of the matrix pipe, and the vector work per element is mul (log2 e), exp2, mul -- it was
    // sub, mul, exp2, sub, mul (1,692 -> 1,350 vector instructions per wave).
    auto stat16 = [&](const float* st, int qt) {
        f32x16 c;
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(st + 32 * qt + 8 * a4 + 4 * hh);
            c[4 * a4] = v[0]; c[4 * a4 + 1] = v[1]; c[4 * a4 + 2] = v[2]; c[4 * a4 + 3] = v[3];
        }
        return c;
    };
    f32x16 s_nxt = mfma32(rowf(sQ, 0), kf[0], stat16(sLse, 0));   // S[query][key] / 4 - lse of the first pair: lane = key, registers = queries
    f32x16 p_nxt = mfma32(rowf(sD, 0), vf[0], stat16(sDel, 0));   // dP[query][key] - delta,  dP = sum_d d_o[query][d] V[key][d]
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt >= nt) break;
        const bool keyok = 32 * kt + r < L;                       // lane = key (false only in the last tile)
        const bf16x8 kt0 = tr_frag(sK, 2 * kt), kt1 = tr_frag(sK, 2 * kt + 1);
        f32x16 dv = zero16(), dk = zero16();
#pragma unroll
        for (int qt = 0; qt < NKT; ++qt) {
            if (qt >= nt) break;
            bf16* blk = sdS;                                      // (one block: same-wave LDS requests are served in issue order, the next pair's stores land behind this pair's reads)
            const bf16x8 dt0 = tr_frag(sD, 2 * qt), dt1 = tr_frag(sD, 2 * qt + 1), qt0 = tr_frag(sQ, 2 * qt), qt1 = tr_frag(sQ, 2 * qt + 1);
            f32x16 p = s_nxt, ds = p_nxt;
            {   // the two score products of the NEXT pair (kt-major order; past the end: pair (0, 0) again, unused) run under this pair's vector work
                const int qn = qt + 1 < nt ? qt + 1 : 0;
                const bool same_kt = qt + 1 < nt;
                const bf16x8 kn = same_kt ? kf[kt] : kf[kt + 1 < NKT ? kt + 1 : 0], vn = same_kt ? vf[kt] : vf[kt + 1 < NKT ? kt + 1 : 0];
                s_nxt = mfma32(rowf(sQ, qn), kn, stat16(sLse, qn));
                p_nxt = mfma32(rowf(sD, qn), vn, stat16(sDel, qn));
            }
            const int NRQ = (qt == NKT - 1) ? NR2 : 16;           // (a constant after unrolling; NR2 < 16 is only chosen when all NKT tiles are live)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (g < NRQ) {
                    const float pv = __builtin_amdgcn_exp2f(p[g] * 1.4426950408889634f);
                    p[g] = pv;
                    ds[g] = pv * ds[g];                           // dS x 4
                } else { p[g] = 0.f; ds[g] = 0.f; }
            }
            bf16x8 dd0 = pack8(ds, 0), dd1 = pack8(ds, 1);        // operands of dK^T and, the same bits, the block dQ^T reads back transposed
            if (kt == NKT - 1 && !keyok) { dd0 = zero8(); dd1 = zero8(); }     // rows of keys past L are zero (they exist in the last tile only)
            *reinterpret_cast<bf16x4*>(blk + r * KT_LD + 4 * hh) = bf16x4{dd0[0], dd0[1], dd0[2], dd0[3]};      // dS block [key][query]: registers 4a .. 4a+3 =
            *reinterpret_cast<bf16x4*>(blk + r * KT_LD + 8 + 4 * hh) = bf16x4{dd0[4], dd0[5], dd0[6], dd0[7]};  // queries 8a + 4hh + {0..3}
            *reinterpret_cast<bf16x4*>(blk + r * KT_LD + 16 + 4 * hh) = bf16x4{dd1[0], dd1[1], dd1[2], dd1[3]};
            *reinterpret_cast<bf16x4*>(blk + r * KT_LD + 24 + 4 * hh) = bf16x4{dd1[4], dd1[5], dd1[6], dd1[7]};
            dv = mfma32(dt0, pack8(p, 0), dv);                    // dV^T[d][key] += d_o^T . P      (contraction over the tile's 32 queries, operands
            dk = mfma32(qt0, dd0, dk);                            // dK^T[d][key] += (Q / 4)^T . 4 dS     straight from the registers)
            dv = mfma32(dt1, pack8(p, 1), dv);
            dk = mfma32(qt1, dd1, dk);
            // (same-wave LDS requests are served in issue order and the transposed reads below are memory reads of `blk` to the compiler: no explicit wait)
            dq[qt] = mfma32(kt0, tr_frag32s<KT_LD>(blk, 0), dq[qt]);                         // 4 dQ^T[d][query] += K^T . (4 dS)^T
            dq[qt] = mfma32(kt1, tr_frag32s<KT_LD>(blk, 1), dq[qt]);
        }
        const int j = 32 * kt + r;
        if (j < L) {
            const int64_t tok = tok_of(G, j, Tn, mode);
            store_t16(dV + tok * lddkv + h * 16, dv, hh);
            store_t16(dK + tok * lddkv + h * 16, dk, hh);
        }
    }
    KTQ0();                                                       // 4: the tile pairs (dK / dV stores issued)
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt) {
        const int i = 32 * qt + r;
        if (i < L) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dq[qt][e] *= 0.25f;
            store_t16(dQ + tok_of(G, i, Tn, mode) * lddq + h * 16, dq[qt], hh);
        }
    }
#ifdef KT_PROF
    KTQ();                                                        // 5: every store acknowledged
    if ((blockIdx.x == 777 || blockIdx.x == 2000) && lane == 0 && (wave == 0 || wave == 3))
        printf("kt prof block %d wave %d: g_mid landed + written %lld  B1 + d_o %lld  rest landed, tiles, V, delta, B2, B3 %lld  tile pairs %lld  dq stores+acks %lld  total %lld\n", (int)blockIdx.x, wave,
               kt_t[1] - kt_t[0], kt_t[2] - kt_t[1], kt_t[3] - kt_t[2], kt_t[4] - kt_t[3], kt_t[5] - kt_t[4], kt_t[5] - kt_t[0]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Backward for groups of 97..256 positions (temporal attention of the long-clip configurations, T = 243), plain operator interface (d_o is read, nothing
// comes from the forward), one wave per (group, head).  Neither loop order of the kernels above fits: query-tile-outer keeps 32 NKT accumulator registers
// of dK / dV, key-tile-outer 16 NKT of dQ next to the score tiles.  Two passes over the head's tiles instead, each in the orientation whose products
// need no transposed P or dS:
//   pass A, lane = query (per query tile):  S^T = K . (Q/4)^T for every key tile (16 NKT registers), softmax statistics, delta = sum_keys P . dP with
//           dP^T = V . d_o^T, then dS^T = P (dP - delta) tile by tile (dP formed again: one MFMA instead of 16 NKT more live registers) and
//           dQ^T += K^T . dS^T with the operand straight out of the registers;  lse and delta of the query are left in LDS;
//   pass B, lane = key (key-tile-outer):    S = (Q/4) . K^T and dP = d_o . V^T un-transposed, P = exp(S - lse), dS = P (dP - delta), and
//           dV^T += d_o^T . P, dK^T += (Q/4)^T . dS again out of the registers: 32 accumulator registers, nothing written to LDS.
// 11 MFMAs and two exponentials per 32 x 32 tile pair against the 8 + 1 of k_attn_bwd_kt (which needs the forward's lse and o and whole groups in LDS),
// no block of P or dS ever goes through LDS, and the register count does not grow with the group in pass B and by 16 per tile in pass A.
// LDS per wave: 2 x NKT KB of operand tiles + lse / delta -- only what is read TRANSPOSED lives there (K in pass A; Q/4 and d_o in pass B, Q/4 taking
// K's place between the passes); V stays in registers through pass A, and the row fragments each pass needs once per tile come straight from memory
// (prefetched a tile ahead).  At 256 positions that is 18 KB per wave: two workgroups per CU, two waves per SIMD (all four tiles resident: one, and one
// wave per SIMD issues a vector instruction every ~5 cycles at best -- the launch was 478 us at T = 243, B = 32).
// ---------------------------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_2p(const bf16* __restrict__ Q, int64_t ldq, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                        int64_t ldkv, const bf16* __restrict__ dO, bf16* __restrict__ dQ, int64_t lddq, bf16* __restrict__ dK,
                                                        bf16* __restrict__ dV, int64_t lddkv, int L, int Tn, int mode, int units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = NKT * 32 * 16;                           // [positions][16] operand tile (bf16 elements)
    constexpr int WAVE_BYTES = 2 * TILE * 2 + 2 * NKT * 32 * 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int unit = blockIdx.x * 4 + wave;
    if (unit >= units) return;                                    // wave-uniform; no workgroup barrier below
    const int G = unit >> 3, h = unit & 7;
    bf16* sK = reinterpret_cast<bf16*>(smem + wave * WAVE_BYTES); // pass A: K;  pass B: Q / 4 (a power of two: exact in bf16)
    bf16* sQ = sK;
    bf16* sD = sK + TILE;
    float* sLse = reinterpret_cast<float*>(sD + TILE);            // [32 NKT]
    float* sDel = sLse + NKT * 32;
    const int nt = (L + 31) >> 5;                                 // live 32-position tiles
    auto q4 = [&](int t) {                                        // row fragment of Q / 4 from memory
        bf16x8 qv = row_frag(Q, ldq, G, 32 * t + r, L, Tn, mode, h, hh, nullptr);
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = (bf16)((float)qv[e] * 0.25f);
        return qv;
    };
    bf16x8 vf[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        if (t >= nt) break;
        const int pos = 32 * t + r;
        row_frag(K, ldkv, G, pos, L, Tn, mode, h, hh, sK);
        row_frag(dO, 128, G, pos, L, Tn, mode, h, hh, sD);
        vf[t] = row_frag(V, ldkv, G, pos, L, Tn, mode, h, hh, nullptr);
    }
    bf16x8 qf_n = q4(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the tiles were written by other lanes of this wave
    auto rowf = [&](const bf16* tile, int t) { return *reinterpret_cast<const bf16x8*>(tile + (32 * t + r) * 16 + 8 * hh); };
    // ---------------- pass A: lane = query ----------------
    for (int qt = 0; qt < nt; ++qt) {
        const int i = 32 * qt + r;
        const bf16x8 qf = qf_n, dfq = rowf(sD, qt);
        if (qt + 1 < nt) qf_n = q4(qt + 1);
        // (vector instructions per score kept low, as in the forward cores: exp2 of one fma, the probabilities stay UN-normalised -- e = exp(s - max) --
        //  and 1 / sum enters once per query: delta = (sum e . dP) / sum, dS = e (dP - delta) . (scale / sum))
        f32x16 st[NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt >= nt) break;
            st[kt] = mfma32(rowf(sK, kt), qf, zero16());          // S^T[key][query] / 4
            if (kt == nt - 1) {
#pragma unroll
                for (int g = 0; g < 16; ++g) st[kt][g] = (32 * kt + pos_of(g, hh) < L) ? st[kt][g] : -INFINITY;
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) mx = fmaxf(mx, st[kt][g]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float nm = -mx * 1.4426950408889634f;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt >= nt) break;
#pragma unroll
            for (int g = 0; g < 16; ++g) { st[kt][g] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][g], 1.4426950408889634f, nm)); sum += st[kt][g]; }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        float delta = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt >= nt) break;
            const f32x16 dp = mfma32(vf[kt], dfq, zero16());      // dP^T[key][query] = sum_d V[key][d] d_o[query][d]
#pragma unroll
            for (int g = 0; g < 16; ++g) delta = __builtin_fmaf(st[kt][g], dp[g], delta);
        }
        delta = (delta + __shfl_xor(delta, 32)) * inv;
        if (hh == 0) { sLse[i] = i < L ? -(mx + __logf(sum)) : -INFINITY; sDel[i] = -delta; }   // negated: they are pass B's initial accumulators; rows past L: exp(-inf) = 0
        const float ks = 0.25f * inv;
        f32x16 dq = zero16();
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt >= nt) break;
            const f32x16 dp = mfma32(vf[kt], dfq, zero16());
#pragma unroll
            for (int g = 0; g < 16; ++g) st[kt][g] = st[kt][g] * (dp[g] - delta) * ks;             // dS^T (scale and 1 / sum folded)
            dq = mfma32(tr_frag(sK, 2 * kt), pack8(st[kt], 0), dq);                                // dQ^T[d][query] += K^T . dS^T
            dq = mfma32(tr_frag(sK, 2 * kt + 1), pack8(st[kt], 1), dq);
        }
        if (i < L) store_t16(dQ + tok_of(G, i, Tn, mode) * lddq + h * 16, dq, hh);
    }
    // between the passes Q / 4 takes K's place (same-wave LDS requests are served in issue order: the stores land behind pass A's last reads)
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        if (t >= nt) break;
        *reinterpret_cast<bf16x8*>(sQ + (32 * t + r) * 16 + 8 * hh) = q4(t);
    }
    bf16x8 kf_n = row_frag(K, ldkv, G, r, L, Tn, mode, h, hh, nullptr), vf_n = vf[0];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // Q / 4, lse and delta were written by other lanes of this wave
    // ---------------- pass B: lane = key, key-tile-outer ----------------
    for (int kt = 0; kt < nt; ++kt) {
        const bf16x8 kfk = kf_n, vfk = vf_n;
        if (kt + 1 < nt) {
            kf_n = row_frag(K, ldkv, G, 32 * (kt + 1) + r, L, Tn, mode, h, hh, nullptr);
            vf_n = row_frag(V, ldkv, G, 32 * (kt + 1) + r, L, Tn, mode, h, hh, nullptr);
        }
        f32x16 dv = zero16(), dk = zero16();
        for (int qt = 0; qt < nt; ++qt) {
            f32x16 cl, cd;                                        // -lse and -delta of the tile's queries (registers 4a .. 4a+3 = queries 32 qt + 8a + 4hh + {0..3})
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + 32 * qt + 8 * a4 + 4 * hh), d4 = *reinterpret_cast<const f32x4*>(sDel + 32 * qt + 8 * a4 + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) { cl[4 * a4 + e] = l4[e]; cd[4 * a4 + e] = d4[e]; }
            }
            const bf16x8 dt0 = tr_frag(sD, 2 * qt), dt1 = tr_frag(sD, 2 * qt + 1), qt0 = tr_frag(sQ, 2 * qt), qt1 = tr_frag(sQ, 2 * qt + 1);
            f32x16 p = mfma32(rowf(sQ, qt), kfk, cl);             // S[query][key] / 4 - lse: the statistics ride in as the products' initial accumulators
            f32x16 ds = mfma32(rowf(sD, qt), vfk, cd);            // dP[query][key] - delta
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float pv = __builtin_amdgcn_exp2f(p[g] * 1.4426950408889634f);
                p[g] = pv;
                ds[g] = pv * ds[g];
            }
            dv = mfma32(dt0, pack8(p, 0), dv);                    // dV^T[d][key] += d_o^T . P
            dk = mfma32(qt0, pack8(ds, 0), dk);                   // dK^T[d][key] += (Q / 4)^T . dS
            dv = mfma32(dt1, pack8(p, 1), dv);
            dk = mfma32(qt1, pack8(ds, 1), dk);
        }
        const int j = 32 * kt + r;
        if (j < L) {
            const int64_t tok = tok_of(G, j, Tn, mode);
            store_t16(dV + tok * lddkv + h * 16, dv, hh);
            store_t16(dK + tok * lddkv + h * 16, dk, hh);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form of k_attn_bwd_mfma<1, true> (groups of <= 32 positions, d_o formed in-kernel): the same arithmetic, bit for bit, but
//   * a workgroup walks a contiguous range of groups: the 16 rows of the packed projection weight a wave needs stay in registers for the
//     whole launch (the one-group-per-workgroup form re-read 32 KB of weights per group: 221 MB of L2 traffic per launch, more than the
//     kernel's HBM bytes), and the next group's q | k | v fragments and g_mid rows are in flight while this group computes;
//   * g_mid rows travel through registers (one coalesced 16-byte load per thread, a full 256-byte row per 16 lanes) into a double-buffered
//     LDS tile: hipcc counts these loads itself, there is no hand-counted wait in this kernel; ONE workgroup barrier per group;
//   * dq / dk / dv leave as 16-byte stores: the two lane halves exchange a dword pair (v_permlane32_swap) so that each lane holds 8
//     consecutive head channels of its position instead of two 4-channel pieces.
// Two workgroups per CU (<= 128 VGPRs, 72 KB of LDS): one group's barrier and LDS round trips are covered by the other's work.
// ---------------------------------------------------------------------------------------------------------------

// NR: score registers that can hold a live key (register g holds keys (g & 3) + 8 (g >> 2) + 4 hh: 9 for groups of <= 17 positions, the spatial
// blocks): the softmax / dS arithmetic and the P / dS tile stores skip the registers past NR, which are identically zero.
#ifndef KASF_PERS_FLUSH_AT          // where group t-1's dq | dk | dv leave: 0 at the top of group t, right in front of the look-ahead loads (rounds 2-4); 1 behind the barrier; 2 behind d_o
#define KASF_PERS_FLUSH_AT 2        // (shipped, round 5); 3 behind pass 1.  Stores and loads issued back to back by one wave cost both: in step <9> 66.8 / 58.3 / 56.1 / 61.1 us, <16> 60.1 / 53.2 / 54.0 / 57.4
#endif
// [32 queries][32 keys] P / dS tiles of the persistent kernel (round 6): the four 16-byte chunks of row r are stored at chunk ^ ((r >> 1) & 3).  Unswizzled, the 8-byte row
// writes of 16 consecutive rows hit banks 16 r mod 32 -- 8-way conflicts on every write: SQ_LDS_BANK_CONFLICT was 60 % of SQ_LDS_IDX_ACTIVE in this kernel and the LDS
// busy 63 % of its time (profiles/r6_attn_sq_counters_before.txt); swizzled they are 2-way, and the transposed reads still cover whole 64-byte rows (conflict-free).
__device__ __forceinline__ int ps_off(int r, int col) { return r * 32 + ((((col >> 3) ^ (r >> 1)) & 3) << 3) + (col & 7); }
__device__ __forceinline__ bf16x8 tr_frag32z(const bf16* s_tile, int ks) {
    const int lane = threadIdx.x & 63, u = lane & 15, hh = lane >> 5, q = u >> 2, p = u & 3, c0 = 16 * ((lane >> 4) & 1);
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    const int k0 = 16 * ks + 4 * hh;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + ps_off(k0 + q, c0 + 4 * p)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s_tile + ps_off(k0 + 8 + q, c0 + 4 * p)));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
#ifndef KASF_PERS_PSWZ
#define KASF_PERS_PSWZ 1
#endif
#if KASF_PERS_PSWZ
#define PERS_POFF(r, col) ps_off(r, col)
#define PERS_TR32(t, ks) tr_frag32z(t, ks)
#else
#define PERS_POFF(r, col) ((r) * 32 + (col))
#define PERS_TR32(t, ks) tr_frag32(t, ks)
#endif
template <int NR>
__global__ __launch_bounds__(512, 4) void k_attn_bwd_pers(const bf16* __restrict__ Q, int64_t ldq, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                          int64_t ldkv, bf16* __restrict__ dQ, int64_t lddq, bf16* __restrict__ dK, bf16* __restrict__ dV,
                                                          int64_t lddkv, int L, int Tn, int mode, int groups, const bf16* __restrict__ Gmid,
                                                          const bf16* __restrict__ Wp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 32 * 16;
    constexpr int WAVE_BYTES = 3 * TILE * 2 + 2 * 32 * 32 * 2;      // K, Q, d_o tiles + P and dS tiles
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5, h = wave;
    const int per = (groups + gridDim.x - 1) / gridDim.x;
    const int g0 = blockIdx.x * per;
    int ng = groups - g0;
    if (ng > per) ng = per;
    if (ng <= 0) return;                                             // workgroup-uniform
    bf16* sK = reinterpret_cast<bf16*>(smem + wave * WAVE_BYTES);
    bf16* sQ = sK + TILE;
    bf16* sD = sQ + TILE;
    bf16* sP = sD + TILE;
    bf16* sdS = sP + 32 * 32;
    bf16* sG = reinterpret_cast<bf16*>(smem + 8 * WAVE_BYTES);      // [2][32][128] g_mid rows (swizzled tiles)
    const int li = lane & 15, lg = lane >> 4;
    const int grow = threadIdx.x >> 4, gch = threadIdx.x & 15;
    bf16x8 wp[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wp[ks] = *reinterpret_cast<const bf16x8*>(Wp + (int64_t)(16 * h + li) * 128 + 32 * ks + 8 * lg);
    bf16x8 kfN, vfN, qfN, gcN;
    const int stride = mode == 0 ? 1 : KASF_J;                      // tokens between consecutive positions of a group (32-bit indices: M * 384 < 2^31)
    auto base_of = [&](int G) { return mode == 0 ? G * KASF_J : (G / KASF_J) * Tn * KASF_J + (G % KASF_J); };
    const int rc = r < L ? r : L - 1, growc = grow < L ? grow : L - 1;
    const unsigned okv = (unsigned)(rc * stride) * (unsigned)ldkv + h * 16 + 8 * hh, oq = (unsigned)(rc * stride) * (unsigned)ldq + h * 16 + 8 * hh;
    const unsigned og = (unsigned)(growc * stride) * 128u + gch * 8;
    auto fetch = [&](int t) {
        const unsigned b = (unsigned)base_of(g0 + t);              // wave-uniform
        kfN = *reinterpret_cast<const bf16x8*>(K + (size_t)(b * (unsigned)ldkv + okv));
        vfN = *reinterpret_cast<const bf16x8*>(V + (size_t)(b * (unsigned)ldkv + okv));
        qfN = *reinterpret_cast<const bf16x8*>(Q + (size_t)(b * (unsigned)ldq + oq));
        gcN = *reinterpret_cast<const bf16x8*>(Gmid + (size_t)(b * 128u + og));
    };                                                               // (rows past L are clamped here and zeroed at the point of use: a select
                                                                     //  right behind the load would make hipcc wait for it before the group's work)
    u32x4_t pq, pv, pk;
    const unsigned sq = (unsigned)(rc * stride) * (unsigned)lddq + h * 16 + 8 * hh, skv = (unsigned)(rc * stride) * (unsigned)lddkv + h * 16 + 8 * hh;
    auto flush = [&](int G) {                                        // dq | dk | dv of group G: 16 bytes per lane, 32 contiguous bytes per position and head
        const unsigned b = (unsigned)base_of(G);
        if (r < L) {
            *reinterpret_cast<u32x4_t*>(dQ + (size_t)(b * (unsigned)lddq + sq)) = pq;
            *reinterpret_cast<u32x4_t*>(dV + (size_t)(b * (unsigned)lddkv + skv)) = pv;
            *reinterpret_cast<u32x4_t*>(dK + (size_t)(b * (unsigned)lddkv + skv)) = pk;
        }
    };
    constexpr int NA4 = (NR + 3) / 4;                                // 4-register (8-byte) pieces of a P / dS row that are ever non-zero
    if (NA4 < 4) {                                                   // the rest of both tiles stays zero for the whole launch
#pragma unroll
        for (int a4 = NA4; a4 < 4; ++a4) {
            float z4[4] = {0.f, 0.f, 0.f, 0.f};
            store4(sP + PERS_POFF(r, 8 * a4 + 4 * hh), z4);
            store4(sdS + PERS_POFF(r, 8 * a4 + 4 * hh), z4);
        }
    }
#ifdef PERS_PROF
    long long acc_t[6] = {0,0,0,0,0,0};
#define PQ(k) do { const long long n_ = clock64(); acc_t[k] += n_ - t_; t_ = n_; } while (0)
#else
#define PQ(k) do {} while (0)
#endif
    fetch(0);
    for (int t = 0; t < ng; ++t) {
#ifdef PERS_PROF
        long long t_ = clock64();
#endif
        const bf16x8 kf = r < L ? kfN : zero8(), vf = r < L ? vfN : zero8(), qf = r < L ? qfN : zero8();
        bf16* sGt = sG + (t & 1) * (32 * 128);
        *reinterpret_cast<bf16x8*>(sGt + Tile<bf16>::chunk_off(grow, gch)) = grow < L ? gcN : zero8();
        *reinterpret_cast<bf16x8*>(sK + r * 16 + 8 * hh) = kf;
        *reinterpret_cast<bf16x8*>(sQ + r * 16 + 8 * hh) = qf;
#if KASF_PERS_FLUSH_AT == 0
        if (t > 0) flush(g0 + t - 1);
#endif
        if (t + 1 < ng) fetch(t + 1);
        PQ(0);
        __syncthreads();          // g_mid rows of group t visible; every wave is past its reads of the other buffer (group t-1)
        PQ(1);
#if KASF_PERS_FLUSH_AT == 1
        if (t > 0) flush(g0 + t - 1);
#endif
        bf16x8 df;
        {   // d_o of this head: [32 positions][16 channels] = g_mid . (ls1 . Wproj)^T rows 16h .. 16h+15
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks], *reinterpret_cast<const bf16x8*>(sGt + Tile<bf16>::chunk_off(li, 4 * ks + lg)), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[ks], *reinterpret_cast<const bf16x8*>(sGt + Tile<bf16>::chunk_off(16 + li, 4 * ks + lg)), acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float live = 16 * mt + li < L ? 1.0f : 0.0f;
                float v[4] = {acc[mt][0] * live, acc[mt][1] * live, acc[mt][2] * live, acc[mt][3] * live};
                store4(sD + (16 * mt + li) * 16 + 4 * lg, v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            df = *reinterpret_cast<const bf16x8*>(sD + r * 16 + 8 * hh);
        }
        PQ(2);
#if KASF_PERS_FLUSH_AT == 2
        if (t > 0) flush(g0 + t - 1);
#endif
        // ---------------- pass 1: lane = query ----------------
        f32x16 st = mfma32(kf, qf, zero16());                    // S^T[key][query]
        f32x16 dp = mfma32(vf, df, zero16());                    // dP^T[key][query]
        float mx = -INFINITY;
#pragma unroll
        for (int g = 0; g < NR; ++g) {
            const float s = (pos_of(g, hh) < L) ? st[g] * 0.25f : -INFINITY;
            st[g] = s;
            mx = fmaxf(mx, s);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < NR; ++g) { st[g] = __expf(st[g] - mx); sum += st[g]; }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        float delta = 0.f;
#pragma unroll
        for (int g = 0; g < NR; ++g) { st[g] *= inv; delta += st[g] * dp[g]; }
#pragma unroll
        for (int g = NR; g < 16; ++g) st[g] = 0.f;
        delta += __shfl_xor(delta, 32);
#pragma unroll
        for (int a4 = 0; a4 < NA4; ++a4) {
            float v4[4] = {st[4 * a4], st[4 * a4 + 1], st[4 * a4 + 2], st[4 * a4 + 3]};
            store4(sP + PERS_POFF(r, 8 * a4 + 4 * hh), v4);
        }
#pragma unroll
        for (int g = 0; g < NR; ++g) st[g] = st[g] * (dp[g] - delta) * 0.25f;       // dS^T (scale folded)
        f32x16 dq = mfma32(tr_frag(sK, 0), pack8(st, 0), zero16());                  // dQ^T[d][query] = K^T . dS^T
        dq = mfma32(tr_frag(sK, 1), pack8(st, 1), dq);
#pragma unroll
        for (int a4 = 0; a4 < NA4; ++a4) {
            float v4[4] = {st[4 * a4], st[4 * a4 + 1], st[4 * a4 + 2], st[4 * a4 + 3]};
            store4(sdS + PERS_POFF(r, 8 * a4 + 4 * hh), v4);
        }
        PQ(3);
#if KASF_PERS_FLUSH_AT == 3
        if (t > 0) flush(g0 + t - 1);
#endif
        // ---------------- pass 2: dV^T = dO^T . P, dK^T = Q^T . dS with P / dS read back transposed (lane = key) ----------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tiles were written by other lanes of this wave (LDS is in order per wave)
        f32x16 dv = zero16(), dk = zero16();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            dv = mfma32(tr_frag(sD, ks), PERS_TR32(sP, ks), dv);
            dk = mfma32(tr_frag(sQ, ks), PERS_TR32(sdS, ks), dk);
        }
        pq = swap_t16(dq);       // stored during the NEXT group (KASF_PERS_FLUSH_AT): behind that group's look-ahead loads in the in-order vmcnt queue, so the wait
        pv = swap_t16(dv);       // for those loads never has stores ahead of it -- and not back to back with them either (round 5: -16 % / -11 %)
        pk = swap_t16(dk);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS reads are done before the next group's tile writes
        PQ(4);
    }
    flush(g0 + ng - 1);
#ifdef PERS_PROF
    if (blockIdx.x == 77 && (threadIdx.x == 0 || threadIdx.x == 320)) printf("pers prof NR %d mode %d wave %d groups %d: top(wait loads, LDS writes, flush, fetch) %lld barrier %lld d_o %lld pass1 %lld pass2 %lld\n", NR, mode, wave, ng, acc_t[0], acc_t[1], acc_t[2], acc_t[3], acc_t[4]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Head dimension 32 (num_heads = 4: the reference constructor's default, KASportsFormer.py:293).  Same orientation rules as above; what changes:
// two k-steps per score product (32 channels), the [positions][32] operand tiles are read transposed with tr_frag32 and fill all 32 rows of the
// d-indexed products (the 16-wide heads use half of them), the scale 32^-0.5 is not a power of two and stays an fp32 multiply, and a lane's 16
// output registers are 4 x 4 channels that leave as two 16-byte stores after the half swap.  One wave per (group, head), the 4 heads of a group per
// workgroup.  Forward: k_attn_fwd_mfma's program.  Backward: k_attn_bwd_2p's two passes for every group length up to 256 (V in registers through
// pass A: 8 VGPRs per key tile, so the long instantiations run one wave per SIMD).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 row_frag32(const bf16* base, int64_t ld, int G, int pos, int L, int Tn, int mode, int h, int s, int hh, bf16* s_tile) {
    bf16x8 v = zero8();
    if (pos < L) v = *reinterpret_cast<const bf16x8*>(base + tok_of(G, pos, Tn, mode) * ld + h * 32 + 16 * s + 8 * hh);
    if (s_tile != nullptr) *reinterpret_cast<bf16x8*>(s_tile + pos * 32 + 16 * s + 8 * hh) = v;
    return v;
}
// lane = position, registers 4a .. 4a+3 = channels 8a + 4hh + {0..3} (a = 0..3)  ->  lane half hh stores channels 8hh .. 8hh+7 and 16 + 8hh .. 16 + 8hh+7
__device__ __forceinline__ void store_t32(bf16* dst, const f32x16& t, int hh) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(pack2(t[0], t[1]), pack2(t[4], t[5]), false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(pack2(t[2], t[3]), pack2(t[6], t[7]), false, false);
    const auto s2 = __builtin_amdgcn_permlane32_swap(pack2(t[8], t[9]), pack2(t[12], t[13]), false, false);
    const auto s3 = __builtin_amdgcn_permlane32_swap(pack2(t[10], t[11]), pack2(t[14], t[15]), false, false);
    *reinterpret_cast<u32x4_t*>(dst + 8 * hh) = u32x4_t{s0[0], s1[0], s0[1], s1[1]};
    *reinterpret_cast<u32x4_t*>(dst + 16 + 8 * hh) = u32x4_t{s2[0], s3[0], s2[1], s3[1]};
}
constexpr float SCALE32 = 0.17677669529663687f;                  // 32 ** -0.5 (selfattention.py:12)

template <int NKT>
__global__ __launch_bounds__(256) void k_attn_fwd_mfma32(const bf16* __restrict__ Q, int64_t ldq, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                         int64_t ldkv, bf16* __restrict__ O, int L, int Tn, int mode, int units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int unit = blockIdx.x * 4 + wave;
    if (unit >= units) return;                                   // wave-uniform; no workgroup barrier below
    const int G = unit >> 2, h = unit & 3;
    bf16* sV = reinterpret_cast<bf16*>(smem) + wave * (NKT * 32 * 32);
    bf16x8 kf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kf[kt][s] = row_frag32(K, ldkv, G, 32 * kt + r, L, Tn, mode, h, s, hh, nullptr);
            row_frag32(V, ldkv, G, 32 * kt + r, L, Tn, mode, h, s, hh, sV);
        }
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt) {
        if (32 * qt >= L) break;
        const int i = 32 * qt + r;
        const bf16x8 qf0 = row_frag32(Q, ldq, G, i, L, Tn, mode, h, 0, hh, nullptr), qf1 = row_frag32(Q, ldq, G, i, L, Tn, mode, h, 1, hh, nullptr);
        f32x16 st[NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            st[kt] = mfma32(kf[kt][1], qf1, mfma32(kf[kt][0], qf0, zero16()));       // S^T[key][query]
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                st[kt][g] = (32 * kt + pos_of(g, hh) < L) ? st[kt][g] : -INFINITY;
                mx = fmaxf(mx, st[kt][g]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        constexpr float C2 = SCALE32 * 1.4426950408889634f;           // scale . log2(e): the softmax of k_attn_fwd_mfma
        const float nm = -mx * C2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int g = 0; g < 16; ++g) { st[kt][g] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][g], C2, nm)); sum += st[kt][g]; }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        f32x16 ot = zero16();
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            ot = mfma32(tr_frag32(sV + kt * 1024, 0), pack8(st[kt], 0), ot);        // O^T[d][query] += V^T . P^T, all 32 rows live
            ot = mfma32(tr_frag32(sV + kt * 1024, 1), pack8(st[kt], 1), ot);
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) ot[g] *= inv;
        if (i < L) store_t32(O + tok_of(G, i, Tn, mode) * 128 + h * 32, ot, hh);
    }
}

template <int NKT>
__global__ __launch_bounds__(256) void k_attn_bwd_2p32(const bf16* __restrict__ Q, int64_t ldq, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                       int64_t ldkv, const bf16* __restrict__ dO, bf16* __restrict__ dQ, int64_t lddq, bf16* __restrict__ dK,
                                                       bf16* __restrict__ dV, int64_t lddkv, int L, int Tn, int mode, int units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = NKT * 32 * 32;                           // [positions][32] operand tile (bf16 elements)
    constexpr int WAVE_BYTES = 2 * TILE * 2 + 2 * NKT * 32 * 4;
    constexpr float C2 = SCALE32 * 1.4426950408889634f;           // scale . log2(e)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int unit = blockIdx.x * 4 + wave;
    if (unit >= units) return;                                    // wave-uniform; no workgroup barrier below
    const int G = unit >> 2, h = unit & 3;
    bf16* sK = reinterpret_cast<bf16*>(smem + wave * WAVE_BYTES); // pass A: K;  pass B: Q
    bf16* sQ = sK;
    bf16* sD = sK + TILE;
    float* sLse = reinterpret_cast<float*>(sD + TILE);            // [32 NKT]  -lse . log2(e)
    float* sDel = sLse + NKT * 32;                                //           -delta
    const int nt = (L + 31) >> 5;
    bf16x8 vf[NKT][2];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        if (t >= nt) break;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            row_frag32(K, ldkv, G, 32 * t + r, L, Tn, mode, h, s, hh, sK);
            row_frag32(dO, 128, G, 32 * t + r, L, Tn, mode, h, s, hh, sD);
            vf[t][s] = row_frag32(V, ldkv, G, 32 * t + r, L, Tn, mode, h, s, hh, nullptr);
        }
    }
    bf16x8 qn0 = row_frag32(Q, ldq, G, r, L, Tn, mode, h, 0, hh, nullptr), qn1 = row_frag32(Q, ldq, G, r, L, Tn, mode, h, 1, hh, nullptr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the tiles were written by other lanes of this wave
    auto rowf = [&](const bf16* tile, int t, int s) { return *reinterpret_cast<const bf16x8*>(tile + (32 * t + r) * 32 + 16 * s + 8 * hh); };
    // ---------------- pass A: lane = query ----------------
    for (int qt = 0; qt < nt; ++qt) {
        const int i = 32 * qt + r;
        const bf16x8 qf0 = qn0, qf1 = qn1, df0 = rowf(sD, qt, 0), df1 = rowf(sD, qt, 1);
        if (qt + 1 < nt) {
            qn0 = row_frag32(Q, ldq, G, i + 32, L, Tn, mode, h, 0, hh, nullptr);
            qn1 = row_frag32(Q, ldq, G, i + 32, L, Tn, mode, h, 1, hh, nullptr);
        }
        f32x16 st[NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt >= nt) break;
            st[kt] = mfma32(rowf(sK, kt, 1), qf1, mfma32(rowf(sK, kt, 0), qf0, zero16()));       // S^T[key][query]
            if (kt == nt - 1) {
#pragma unroll
                for (int g = 0; g < 16; ++g) st[kt][g] = (32 * kt + pos_of(g, hh) < L) ? st[kt][g] : -INFINITY;
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) mx = fmaxf(mx, st[kt][g]);                                // (max over the RAW scores; softmax as in k_attn_bwd_2p)
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float nm = -mx * C2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt >= nt) break;
#pragma unroll
            for (int g = 0; g < 16; ++g) { st[kt][g] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][g], C2, nm)); sum += st[kt][g]; }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        float delta = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt >= nt) break;
            const f32x16 dp = mfma32(vf[kt][1], df1, mfma32(vf[kt][0], df0, zero16()));          // dP^T[key][query]
#pragma unroll
            for (int g = 0; g < 16; ++g) delta = __builtin_fmaf(st[kt][g], dp[g], delta);
        }
        delta = (delta + __shfl_xor(delta, 32)) * inv;
        if (hh == 0) { sLse[i] = i < L ? -(mx * C2 + __logf(sum) * 1.4426950408889634f) : -INFINITY; sDel[i] = -delta; }   // rows past L: exp2(-inf) = 0 in pass B
        const float ks = SCALE32 * inv;
        f32x16 dq = zero16();
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt >= nt) break;
            const f32x16 dp = mfma32(vf[kt][1], df1, mfma32(vf[kt][0], df0, zero16()));
#pragma unroll
            for (int g = 0; g < 16; ++g) st[kt][g] = st[kt][g] * (dp[g] - delta) * ks;             // dS^T (scale and 1 / sum folded)
            dq = mfma32(tr_frag32(sK + kt * 1024, 0), pack8(st[kt], 0), dq);                       // dQ^T[d][query] += K^T . dS^T
            dq = mfma32(tr_frag32(sK + kt * 1024, 1), pack8(st[kt], 1), dq);
        }
        if (i < L) store_t32(dQ + tok_of(G, i, Tn, mode) * lddq + h * 32, dq, hh);
    }
    // between the passes Q takes K's place (same-wave LDS requests are served in issue order: the stores land behind pass A's last reads)
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        if (t >= nt) break;
        row_frag32(Q, ldq, G, 32 * t + r, L, Tn, mode, h, 0, hh, sQ);
        row_frag32(Q, ldq, G, 32 * t + r, L, Tn, mode, h, 1, hh, sQ);
    }
    bf16x8 kn0 = row_frag32(K, ldkv, G, r, L, Tn, mode, h, 0, hh, nullptr), kn1 = row_frag32(K, ldkv, G, r, L, Tn, mode, h, 1, hh, nullptr);
    bf16x8 vn0 = vf[0][0], vn1 = vf[0][1];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // Q, lse and delta were written by other lanes of this wave
    auto stat16 = [&](const float* stp, int qt) {
        f32x16 c;
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(stp + 32 * qt + 8 * a4 + 4 * hh);
            c[4 * a4] = v[0]; c[4 * a4 + 1] = v[1]; c[4 * a4 + 2] = v[2]; c[4 * a4 + 3] = v[3];
        }
        return c;
    };
    // ---------------- pass B: lane = key, key-tile-outer ----------------
    for (int kt = 0; kt < nt; ++kt) {
        const bf16x8 kf0 = kn0, kf1 = kn1, vk0 = vn0, vk1 = vn1;
        if (kt + 1 < nt) {
            kn0 = row_frag32(K, ldkv, G, 32 * (kt + 1) + r, L, Tn, mode, h, 0, hh, nullptr);
            kn1 = row_frag32(K, ldkv, G, 32 * (kt + 1) + r, L, Tn, mode, h, 1, hh, nullptr);
            vn0 = row_frag32(V, ldkv, G, 32 * (kt + 1) + r, L, Tn, mode, h, 0, hh, nullptr);
            vn1 = row_frag32(V, ldkv, G, 32 * (kt + 1) + r, L, Tn, mode, h, 1, hh, nullptr);
        }
        f32x16 dv = zero16(), dk = zero16();
        for (int qt = 0; qt < nt; ++qt) {
            const f32x16 nl = stat16(sLse, qt);
            f32x16 p = mfma32(rowf(sQ, qt, 1), kf1, mfma32(rowf(sQ, qt, 0), kf0, zero16()));          // S[query][key]: lane = key, registers = queries
            f32x16 ds = mfma32(rowf(sD, qt, 1), vk1, mfma32(rowf(sD, qt, 0), vk0, stat16(sDel, qt)));  // dP[query][key] - delta (delta rides in as the initial accumulator)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(p[g], C2, nl[g]));
                p[g] = pv;
                ds[g] = pv * ds[g] * SCALE32;
            }
            dv = mfma32(tr_frag32(sD + qt * 1024, 0), pack8(p, 0), dv);                // dV^T[d][key] += d_o^T . P
            dk = mfma32(tr_frag32(sQ + qt * 1024, 0), pack8(ds, 0), dk);               // dK^T[d][key] += Q^T . dS
            dv = mfma32(tr_frag32(sD + qt * 1024, 1), pack8(p, 1), dv);
            dk = mfma32(tr_frag32(sQ + qt * 1024, 1), pack8(ds, 1), dk);
        }
        const int j = 32 * kt + r;
        if (j < L) {
            const int64_t tok = tok_of(G, j, Tn, mode);
            store_t32(dV + tok * lddkv + h * 32, dv, hh);
            store_t32(dK + tok * lddkv + h * 32, dk, hh);
        }
    }
}

template <typename K> bool set_smem(K k, size_t bytes) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) kasf_set_error(1000 + (int)e, "attention (MFMA): cannot reserve the group's LDS tiles");
    return e == hipSuccess;
}

}  // namespace

// returns false when the shape is outside the MFMA kernels' range (caller falls back to the VALU kernels)
bool kasf_launch_attn_fwd_mfma(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int B, int Tn, int mode) {
    const int L = mode == 0 ? KASF_J : Tn, groups = mode == 0 ? B * Tn : B * KASF_J, units = groups * 8;
    if (L > 256) return false;
    const dim3 grid((units + 3) / 4);
    auto go = [&](auto NK) {                            // all score tiles of a query tile stay in registers (16 per key tile): 8 tiles = 128 VGPRs
        constexpr int NKT = decltype(NK)::value;
        hipLaunchKernelGGL(k_attn_fwd_mfma<NKT>, grid, dim3(256), 4 * NKT * 32 * 16 * 2, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (bf16*)o, L,
                           Tn, mode, units);
    };
    if (L <= 32) go(std::integral_constant<int, 1>{});
    else if (L <= 96) go(std::integral_constant<int, 3>{});
    else if (L <= 128) go(std::integral_constant<int, 4>{});
    else if (L <= 192) go(std::integral_constant<int, 6>{});
    else go(std::integral_constant<int, 8>{});
    return true;
}
bool kasf_launch_attn_bwd_mfma(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq,
                               int64_t lddq, void* dk, void* dv, int64_t lddkv, int B, int Tn, int mode) {
    const int L = mode == 0 ? KASF_J : Tn, groups = mode == 0 ? B * Tn : B * KASF_J, units = groups * 8;
    if (L > 256) return false;
    const dim3 grid((units + 3) / 4);
    if (L > 96) {                                       // two-pass kernel: 2 NKT KB of operand tiles + lse / delta per wave
        auto go = [&](auto NK) {
            constexpr int NKT = decltype(NK)::value;
            const size_t sh2 = 4 * (size_t)(2 * NKT * 32 * 16 * 2 + 2 * NKT * 32 * 4);
            if (!set_smem(k_attn_bwd_2p<NKT>, sh2)) return;
            hipLaunchKernelGGL(k_attn_bwd_2p<NKT>, grid, dim3(256), sh2, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (const bf16*)d_o, (bf16*)dq,
                               lddq, (bf16*)dk, (bf16*)dv, lddkv, L, Tn, mode, units);
        };
        if (L <= 128) go(std::integral_constant<int, 4>{});
        else if (L <= 192) go(std::integral_constant<int, 6>{});
        else go(std::integral_constant<int, 8>{});
        return true;
    }
    if (L <= 32) {
        const size_t sh = 4 * (3 * 32 * 16 * 2 + 32 * 16 + 2 * 32 * 32 * 2);
        hipLaunchKernelGGL((k_attn_bwd_mfma<1, false>), grid, dim3(256), sh, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (const bf16*)d_o,
                           (bf16*)dq, lddq, (bf16*)dk, (bf16*)dv, lddkv, L, Tn, mode, units, (const bf16*)nullptr, (const bf16*)nullptr);
    } else {
        const size_t shl = 4 * (size_t)(4 * 96 * 16 + 2 * 32 * 32) * 2;
        if (!set_smem(k_attn_bwd_long<3, false>, shl)) return true;
        hipLaunchKernelGGL((k_attn_bwd_long<3, false>), grid, dim3(256), shl, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (const bf16*)d_o,
                           (bf16*)dq, lddq, (bf16*)dk, (bf16*)dv, lddkv, L, Tn, mode, units, (const bf16*)nullptr, (const bf16*)nullptr);
    }
    return true;
}

// Attention backward with the projection's data gradient fused in (bf16, groups of <= 32 positions): d_o = g_mid . WprojTs^T is formed per head
// inside the kernel.  false: shape not covered (the caller computes d_o with a linear and calls kasf_launch_attn_bwd).
bool kasf_launch_attn_bwd_fused_do(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* g_mid,
                                   const void* WprojTs, void* dq, int64_t lddq, void* dk, void* dv, int64_t lddkv, int B, int Tn, int mode, int form,
                                   const void* o_saved, const float* lse) {
    const int L = mode == 0 ? KASF_J : Tn, groups = mode == 0 ? B * Tn : B * KASF_J;
    if (L > 96) return false;
    if (groups <= 0) return true;
    if (L > 32) {                                       // three-tile groups (temporal attention at T = 81): one workgroup = the 8 heads of one group
        if (form == 1) return false;                    // (the one-group-per-workgroup comparison form exists for one-tile groups only)
        if (o_saved != nullptr && lse != nullptr) {     // key-tile-outer kernel: statistics and delta from what the forward left behind
            const size_t shk = KT_HPW * (size_t)(3 * 96 * 16 * 2 + 32 * KT_LD * 2 + 2 * 96 * 4 + KT_PAD) + (size_t)96 * 128 * 2;      // 73,856 bytes: two workgroups per CU
            const unsigned ktgrid = 16u * (unsigned)((groups + 7) / 8);          // two workgroups (head halves) per group, blocks b and b + 8
            if (L > 64 && L <= 81) {                    // last tile of at most 17 positions: 7 of its 16 query registers are dead
                if (!set_smem(k_attn_bwd_kt<3, 9>, shk)) return true;
                hipLaunchKernelGGL((k_attn_bwd_kt<3, 9>), dim3(ktgrid), dim3(256), shk, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (const bf16*)o_saved, lse,
                                   (bf16*)dq, lddq, (bf16*)dk, (bf16*)dv, lddkv, L, Tn, mode, (const bf16*)g_mid, (const bf16*)WprojTs, groups);
            } else {
                if (!set_smem(k_attn_bwd_kt<3, 16>, shk)) return true;
                hipLaunchKernelGGL((k_attn_bwd_kt<3, 16>), dim3(ktgrid), dim3(256), shk, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (const bf16*)o_saved, lse,
                                   (bf16*)dq, lddq, (bf16*)dk, (bf16*)dv, lddkv, L, Tn, mode, (const bf16*)g_mid, (const bf16*)WprojTs, groups);
            }
            return true;
        }
        const size_t shl = 8 * (size_t)(4 * 96 * 16 + 2 * 32 * 32) * 2 + (size_t)96 * 128 * 2;
        if (!set_smem(k_attn_bwd_long<3, true>, shl)) return true;
        hipLaunchKernelGGL((k_attn_bwd_long<3, true>), dim3(groups), dim3(512), shl, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (const bf16*)nullptr,
                           (bf16*)dq, lddq, (bf16*)dk, (bf16*)dv, lddkv, L, Tn, mode, groups * 8, (const bf16*)g_mid, (const bf16*)WprojTs);
        return true;
    }
    if (form != 1) {                                    // form 1: the one-group-per-workgroup kernel the persistent one is compared with bit for bit (tests)
        const size_t shp = 8 * (3 * 32 * 16 * 2 + 2 * 32 * 32 * 2) + 2 * 32 * 128 * 2;
        const int pcap = kasf_narrow_grid(KASF_NG_ATTN_BWD, 512, (int64_t)groups * L);
        const dim3 grid(groups < pcap ? groups : pcap);   // two workgroups per CU
        if (L <= 17) {
            if (!set_smem(k_attn_bwd_pers<9>, shp)) return true;
            hipLaunchKernelGGL(k_attn_bwd_pers<9>, grid, dim3(512), shp, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (bf16*)dq, lddq, (bf16*)dk,
                               (bf16*)dv, lddkv, L, Tn, mode, groups, (const bf16*)g_mid, (const bf16*)WprojTs);
        } else {
            if (!set_smem(k_attn_bwd_pers<16>, shp)) return true;
            hipLaunchKernelGGL(k_attn_bwd_pers<16>, grid, dim3(512), shp, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (bf16*)dq, lddq, (bf16*)dk,
                               (bf16*)dv, lddkv, L, Tn, mode, groups, (const bf16*)g_mid, (const bf16*)WprojTs);
        }
        return true;
    }
    const size_t sh = 8 * (3 * 32 * 16 * 2 + 32 * 16 + 2 * 32 * 32 * 2) + 32 * 128 * 2;
    if (!set_smem(k_attn_bwd_mfma<1, true>, sh)) return true;
    hipLaunchKernelGGL((k_attn_bwd_mfma<1, true>), dim3(groups), dim3(512), sh, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv,
                       (const bf16*)nullptr, (bf16*)dq, lddq, (bf16*)dk, (bf16*)dv, lddkv, L, Tn, mode, groups * 8, (const bf16*)g_mid, (const bf16*)WprojTs);
    return true;
}

// num_heads = 4 (head dimension 32): groups of up to 256 positions.  false: shape not covered (the caller runs the LDS-resident fp32 cores)
bool kasf_launch_attn_fwd_mfma32(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int B, int Tn, int mode) {
    const int L = mode == 0 ? KASF_J : Tn, groups = mode == 0 ? B * Tn : B * KASF_J, units = groups * 4;
    if (L > 256) return false;
    auto go = [&](auto NK) {
        constexpr int NKT = decltype(NK)::value;
        const size_t sh = 4 * (size_t)NKT * 32 * 32 * 2;
        if (!set_smem(k_attn_fwd_mfma32<NKT>, sh)) return;
        hipLaunchKernelGGL(k_attn_fwd_mfma32<NKT>, dim3(groups), dim3(256), sh, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (bf16*)o, L, Tn, mode, units);
    };
    if (L <= 32) go(std::integral_constant<int, 1>{});
    else if (L <= 96) go(std::integral_constant<int, 3>{});
    else if (L <= 128) go(std::integral_constant<int, 4>{});
    else if (L <= 192) go(std::integral_constant<int, 6>{});
    else go(std::integral_constant<int, 8>{});
    return true;
}
bool kasf_launch_attn_bwd_mfma32(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq,
                                 int64_t lddq, void* dk, void* dv, int64_t lddkv, int B, int Tn, int mode) {
    const int L = mode == 0 ? KASF_J : Tn, groups = mode == 0 ? B * Tn : B * KASF_J, units = groups * 4;
    if (L > 256) return false;
    auto go = [&](auto NK) {
        constexpr int NKT = decltype(NK)::value;
        const size_t sh = 4 * (size_t)(2 * NKT * 32 * 32 * 2 + 2 * NKT * 32 * 4);
        if (!set_smem(k_attn_bwd_2p32<NKT>, sh)) return;
        hipLaunchKernelGGL(k_attn_bwd_2p32<NKT>, dim3(groups), dim3(256), sh, s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, (const bf16*)d_o, (bf16*)dq, lddq,
                           (bf16*)dk, (bf16*)dv, lddkv, L, Tn, mode, units);
    };
    if (L <= 32) go(std::integral_constant<int, 1>{});
    else if (L <= 96) go(std::integral_constant<int, 3>{});
    else if (L <= 128) go(std::integral_constant<int, 4>{});
    else if (L <= 192) go(std::integral_constant<int, 6>{});
    else go(std::integral_constant<int, 8>{});
    return true;
}
