// MLP backward, second generation (bf16 mode): hidden-quarter ownership with register-resident weights.
//
// Problem with the first generation (k_mlp_bwd + two k_wgrad launches): the 512-wide H and dZ had to round-trip
// through HBM (480 MB per block at B=256) and every 128x128 weight block was re-streamed from L2 per 64 tokens,
// with a workgroup barrier per block: the kernel spent >50 % of its wave-cycles waiting.
//
// Here a workgroup (8 waves) owns ONE QUARTER of the hidden units (128 of 512) and a long range of tokens:
//   * its slices of W1, (ls2.W2)^T and W1^T live in VGPRs for the whole kernel (3 x 16 registers per wave):
//     wave w owns hidden units 16w..16w+15 of the quarter for Z / dH and channels 16w..16w+15 for dA;
//   * per 64-token tile:  Z_q = W1_q LN(x)^T,  dH_q = (ls2.W2)_q^T g^T,  H_q = GELU(Z_q),  dZ_q = dH_q GELU'(Z_q)
//     go registers -> LDS (bf16) once;  dA_q = W1_q^T dZ_q is stored as a bf16 partial (summed over the four
//     quarters by k_lnbwd_sum4, which also does the LayerNorm backward + residual);
//   * the weight gradients  dW1_q += dZ_q^T LN(x)  and  dW2_q += g^T H_q  accumulate in registers over the whole
//     token range (2 x 32 registers per lane) from transposed LDS fragments (ds_read_b64_tr_b16), and leave the
//     kernel once as per-range partial tiles for the deterministic k_wgrad_reduce.
// HBM traffic per token drops from ~5.5 KB (dgrad + two wgrads) to ~3.3 KB, nothing streams from L2 in the loop,
// and there are two barriers per 64 tokens.
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int Q_BM = 32;          // tokens per tile (64 overflows the 256-register budget of a 2-waves-per-SIMD wave)
constexpr int Q_MT = Q_BM / 16;   // 16-token tiles per tile
constexpr int Q_NW = 8;           // waves per workgroup of k_mlp_bwd_q (see the kernel comment)

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// B-operand fragment (rows = tokens) of k-step ks from a swizzled [rows][128] tile
__device__ __forceinline__ bf16x8 tok_frag(const bf16* s, int row, int ks) {
    const int g = (threadIdx.x & 63) >> 4;
    return *reinterpret_cast<const bf16x8*>(s + Tile<bf16>::chunk_off(row, 4 * ks + g));
}

// NW waves per workgroup: 8 (2 per SIMD, 256 registers each) or 4 "fat" waves (1 per SIMD, 512 registers: room for the
// compiler to batch LDS operand reads ahead of the MFMAs).  A wave owns HW = 128/NW hidden units (Z, dH), HW channels
// (dA) and a 32 x (1024/NW) block of each 128 x 128 weight-gradient quarter.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_mlp_bwd_q(const bf16* __restrict__ XN, const bf16* __restrict__ G, const bf16* __restrict__ W1,
                                                       const float* __restrict__ b1,
                                                       const bf16* __restrict__ W2ts, const bf16* __restrict__ W1t, bf16* __restrict__ dApart,
                                                       float* __restrict__ dW1part, float* __restrict__ dW2part, float* __restrict__ db1,
                                                       int64_t M, int tiles_per_range) {
    constexpr int NTHR = NW * 64, NTW = 8 / NW;           // 16-row tiles of hidden units / channels per wave
    constexpr int CT = 32 / NW;                           // 16-column tiles of the weight-gradient block per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sA = reinterpret_cast<bf16*>(smem);            // [3][BM][128] LN(x) as the forward pass stored it: in use / landed / in flight
    bf16* sG = sA + 3 * Q_BM * 128;                      // [3][BM][128] upstream gradient, same ring
    bf16* sH = sG + 3 * Q_BM * 128;                      // [BM][128]    H of this quarter
    bf16* sD = sH + Q_BM * 128;                          // [BM][128]    dZ of this quarter
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), so the four hidden quarters of one token range
    // sit at blockIdx b, b+8, b+16, b+24: same XCD, same L2 -> x and g cross the fabric once, not four times.
    int q, range;
    {
        const int used = gridDim.x >> 2, full = used & ~7, b = blockIdx.x;
        if (b < 4 * full) { q = (b >> 3) & 3; range = (b & 7) + 8 * (b >> 5); }
        else { q = (b - 4 * full) & 3; range = full + ((b - 4 * full) >> 2); }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int64_t tile0 = (int64_t)range * tiles_per_range;
    const int64_t ntiles_total = (M + Q_BM - 1) / Q_BM;
    int64_t ntiles = ntiles_total - tile0;
    if (ntiles > tiles_per_range) ntiles = tiles_per_range;
    const int h0 = 16 * NTW * w;                          // first hidden unit / channel of this wave inside the quarter

    // ---- register-resident weight slices ----
    bf16x8 w1f[NTW][4], w2f[NTW][4], wtf[NTW][4];
    f32x4 bias4[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            w1f[nt][ks] = *reinterpret_cast<const bf16x8*>(W1 + (int64_t)(q * 128 + h0 + 16 * nt + i) * 128 + 32 * ks + 8 * g);
            w2f[nt][ks] = *reinterpret_cast<const bf16x8*>(W2ts + (int64_t)(q * 128 + h0 + 16 * nt + i) * 128 + 32 * ks + 8 * g);
            wtf[nt][ks] = *reinterpret_cast<const bf16x8*>(W1t + (int64_t)(h0 + 16 * nt + i) * 512 + q * 128 + 32 * ks + 8 * g);
        }
        bias4[nt] = *reinterpret_cast<const f32x4*>(b1 + q * 128 + h0 + 16 * nt + 4 * g);
    }
    const int tr0 = NW == 4 ? 32 * w : 32 * (w >> 1), tc0 = NW == 4 ? 0 : 64 * (w & 1);
    f32x4 accW1[2][CT], accW2[2][CT];
    zero_acc(accW1);
    zero_acc(accW2);
    f32x4 db1acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) db1acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sub = threadIdx.x & 15;
    // Tile t's g and LN(x) rows are requested TWO tiles ahead by LDS-direct loads (HBM latency is longer than one tile's work) and are
    // consumed straight from the landing tiles: no LayerNorm is recomputed here.  Every request is exactly LPI loads per wave, so
    // "all but the youngest LPI" == "everything requested before the last issue".  Rows past M are clamped copies; their H / dZ are
    // zeroed below, so they reach neither weight gradient.
    constexpr int LPI = 2 * (Q_BM / 4 / NW);
    auto nx3 = [](int sl) { return sl == 2 ? 0 : sl + 1; };     // ring slots roll (a 64-bit "% 3" is a dozen scalar instructions)
    auto stage_issue = [&](int64_t t, int sl) {
        if (t < ntiles) {
            const int64_t row0 = (tile0 + t) * Q_BM;
            const int nvalid = (int)((M - row0) < Q_BM ? (M - row0) : Q_BM);
            stage_tile_async<bf16, Q_BM, NTHR>(sG + sl * Q_BM * 128, G + row0 * 128, 128, nvalid);
            stage_tile_async<bf16, Q_BM, NTHR>(sA + sl * Q_BM * 128, XN + row0 * 128, 128, nvalid);
        } else {                                         // keep the per-issue load count constant (harmless re-read of the first row)
            stage_tile_async<bf16, Q_BM, NTHR>(sG + sl * Q_BM * 128, G, 128, 1);
            stage_tile_async<bf16, Q_BM, NTHR>(sA + sl * Q_BM * 128, XN, 128, 1);
        }
    };
    stage_issue(0, 0);
    stage_issue(1, 1);
    int sl = 0;
    for (int64_t t = 0; t < ntiles; ++t, sl = nx3(sl)) {
        const bf16* cA = sA + sl * Q_BM * 128;
        const bf16* cG = sG + sl * Q_BM * 128;
        const int64_t row0 = (tile0 + t) * Q_BM;
        const int nvalid = (int)((M - row0) < Q_BM ? (M - row0) : Q_BM);
        wait_async_le<LPI>();                            // g(t) (requested two tiles ago) is complete; the youngest issue stays in flight
        barrier_keep_async();                            // B1: tile t staged; every wave is past tile t-1
        // ---- Z_q and dH_q for this wave's hidden units x 64 tokens ----
        {
            f32x4 accZ[NTW][Q_MT], accH[NTW][Q_MT];
            zero_acc(accZ);
            zero_acc(accH);
            // fragments of k-step ks+1 are requested before the MFMAs of k-step ks; the scheduling barriers keep hipcc from
            // hoisting every LDS read of the phase to the top (which spills) or sinking them next to their use (which stalls)
            bf16x8 fa[2][Q_MT], fg[2][Q_MT];
#pragma unroll
            for (int mt = 0; mt < Q_MT; ++mt) { fa[0][mt] = tok_frag(cA, mt * 16 + i, 0); fg[0][mt] = tok_frag(cG, mt * 16 + i, 0); }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int mt = 0; mt < Q_MT; ++mt) { fa[(ks + 1) & 1][mt] = tok_frag(cA, mt * 16 + i, ks + 1); fg[(ks + 1) & 1][mt] = tok_frag(cG, mt * 16 + i, ks + 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < Q_MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        accZ[nt][mt] = mfma16(w1f[nt][ks], fa[ks & 1][mt], accZ[nt][mt]);
                        accH[nt][mt] = mfma16(w2f[nt][ks], fg[ks & 1][mt], accH[nt][mt]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // rows past M (only the last tile of the last range has any) must not leak GELU(b1) into anything: masked variant there only
            auto epilogue = [&](auto MASKED) {
                constexpr bool masked = decltype(MASKED)::value;
                constexpr int NP = NTW * Q_MT * 2;           // all pairs of the wave advance together (dependent packed FMAs back to back cost wait states)
                f32x2 z[NP], dg[NP];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < Q_MT; ++mt)
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp)
                            z[(nt * Q_MT + mt) * 2 + hp] = f32x2{accZ[nt][mt][2 * hp] + bias4[nt][2 * hp], accZ[nt][mt][2 * hp + 1] + bias4[nt][2 * hp + 1]};
                gelu_grad_pairs_fast(z, dg);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < Q_MT; ++mt) {
                        float h[4], dz[4];
                        const float live = (!masked || mt * 16 + i < nvalid) ? 1.0f : 0.0f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            h[r] = z[(nt * Q_MT + mt) * 2 + (r >> 1)][r & 1];
                            dz[r] = accH[nt][mt][r] * dg[(nt * Q_MT + mt) * 2 + (r >> 1)][r & 1];
                            if (masked) { h[r] *= live; dz[r] *= live; }
                            db1acc[nt][r] += dz[r];
                        }
                        store4(sH + Tile<bf16>::off4(mt * 16 + i, h0 + 16 * nt + 4 * g), h);
                        store4(sD + Tile<bf16>::off4(mt * 16 + i, h0 + 16 * nt + 4 * g), dz);
                    }
            };
            if (nvalid == Q_BM) epilogue(std::false_type{}); else epilogue(std::true_type{});
        }
        barrier_keep_async();                            // B2: H_q / dZ_q of all 128 hidden units are in LDS
        // ---- dA_q partial: this wave's channels x 64 tokens ----
        {
            f32x4 accA[NTW][Q_MT];
            zero_acc(accA);
            bf16x8 fd[2][Q_MT];
#pragma unroll
            for (int mt = 0; mt < Q_MT; ++mt) fd[0][mt] = tok_frag(sD, mt * 16 + i, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int mt = 0; mt < Q_MT; ++mt) fd[(ks + 1) & 1][mt] = tok_frag(sD, mt * 16 + i, ks + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < Q_MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) accA[nt][mt] = mfma16(wtf[nt][ks], fd[ks & 1][mt], accA[nt][mt]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = 0; mt < Q_MT; ++mt) {
                const int64_t row = row0 + mt * 16 + i;
                if (row < M) {
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        float v[4] = {accA[nt][mt][0], accA[nt][mt][1], accA[nt][mt][2], accA[nt][mt][3]};
                        store4(dApart + ((int64_t)q * M + row) * 128 + h0 + 16 * nt + 4 * g, v);
                    }
                }
            }
        }
        stage_issue(t + 2, nx3(nx3(sl)));                // after the dA stores: older than this issue == safe to count on
        // ---- weight gradients: reduction over the tokens of the tile (32-token k-steps), pipelined stages ----
        {
            bf16x8 ra[2][2], cb[2][CT];
            auto load_stage = [&](int st, int slot) {        // st = 2*s + which (0: dW1 operands dZ^T, LN(x); 1: dW2 operands g^T, H)
                const int mb = 32 * (st >> 1) + 8 * g;
                const bf16* rowsrc = (st & 1) ? cG : sD;
                const bf16* colsrc = (st & 1) ? sH : cA;
#pragma unroll
                for (int a = 0; a < 2; ++a) ra[slot][a] = frag_tr(rowsrc, mb, tr0 + 16 * a);
#pragma unroll
                for (int b = 0; b < CT; ++b) cb[slot][b] = frag_tr(colsrc, mb, tc0 + 16 * b);
            };
            load_stage(0, 0);
#pragma unroll
            for (int st = 0; st < 2 * (Q_BM / 32); ++st) {
                if (st + 1 < 2 * (Q_BM / 32)) load_stage(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < CT; ++b) {
                        if (st & 1) accW2[a][b] = mfma16(ra[st & 1][a], cb[st & 1][b], accW2[a][b]);     // dW2[c][hq] += g^T H
                        else accW1[a][b] = mfma16(ra[st & 1][a], cb[st & 1][b], accW1[a][b]);            // dW1[hq][k] += dZ^T LN(x)
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    wait_async();                                        // drain the look-ahead requests before the wave retires
    // ---- leave: per-range partial tiles (summed by k_wgrad_reduce), bias partials by atomics ----
    {
        float* p1 = dW1part + (int64_t)range * 512 * 128;        // [512][128]
        float* p2 = dW2part + (int64_t)range * 128 * 512;        // [128][512]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < CT; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = tr0 + 16 * a + 4 * g + r, cc = tc0 + 16 * b + i;
                    p1[(int64_t)(q * 128 + rr) * 128 + cc] = accW1[a][b][r];
                    p2[(int64_t)rr * 512 + q * 128 + cc] = accW2[a][b][r];
                }
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = db1acc[nt][r];                     // sum over the 16 token lanes that share (g, r)
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (i == 0) atomicAdd(db1 + q * 128 + h0 + 16 * nt + 4 * g + r, v);
        }
}

// g_in = g + LNbwd( sum_q dApart[q] ; x, gamma );  dgamma / dbeta block partials -> atomics.  16 lanes per token row.
// The kernel is pure streaming (6 x 256 B in, 256 B out per token) and used to run at what 8 waves per CU with 96 B in flight per lane can
// pull (5.5 TB/s): RPT rows per thread are now loaded back to back before any of them is consumed, which doubles the bytes in flight at the
// same number of workgroups (the grid stays capped: every workgroup ends with 384 same-address atomics).
template <int RPT>
__device__ __forceinline__ void lnbwd_sum4_body(const bf16* __restrict__ dApart, const bf16* __restrict__ X, const bf16* __restrict__ G,
                                                const float* __restrict__ gamma, bf16* __restrict__ g_in, float* __restrict__ dgamma,
                                                float* __restrict__ dbeta, float* __restrict__ gsum, int64_t M, int bid, int nblk) {
    __shared__ float red[3][16][128];
    const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    float gm[8], dg[8], db[8], gsv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[e] = gamma[sub * 8 + e]; dg[e] = 0.f; db[e] = 0.f; gsv[e] = 0.f; }
    for (int64_t row0 = ((int64_t)bid * RPT) * 16 + rl; row0 < M; row0 += (int64_t)nblk * 16 * RPT) {
        bf16x8 pq[RPT][4], xr[RPT], gr[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int64_t row = row0 + 16 * u;
            if (row < M) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) pq[u][qq] = *reinterpret_cast<const bf16x8*>(dApart + ((int64_t)qq * M + row) * 128 + sub * 8);
                xr[u] = *reinterpret_cast<const bf16x8*>(X + row * 128 + sub * 8);
                gr[u] = *reinterpret_cast<const bf16x8*>(G + row * 128 + sub * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int64_t row = row0 + 16 * u;
            if (row >= M) break;                        // uniform over the 16 lanes of a row
            float d[8], x[8], t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                d[e] = ((float)pq[u][0][e] + (float)pq[u][1][e]) + ((float)pq[u][2][e] + (float)pq[u][3][e]);
                x[e] = (float)xr[u][e];
                t[e] = (float)gr[u][e];
            }
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[e];
            const float mean = reduce16(s) * (1.0f / 128.0f);
            float qv = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[e] -= mean; qv += x[e] * x[e]; }
            const float rstd = rsqrtf(reduce16(qv) * (1.0f / 128.0f) + KASF_LN_EPS);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[e] *= rstd;
                dg[e] += d[e] * x[e];
                db[e] += d[e];
                d[e] *= gm[e];
                s1 += d[e];
                s2 += d[e] * x[e];
            }
            s1 = reduce16(s1) * (1.0f / 128.0f);
            s2 = reduce16(s2) * (1.0f / 128.0f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { gsv[e] += t[e]; t[e] += rstd * (d[e] - s1 - x[e] * s2); }      // gsv: colsum(g) for the fc2 bias / layer-scale gradients
            store8(g_in + row * 128 + sub * 8, t);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][rl][sub * 8 + e] = dg[e]; red[1][rl][sub * 8 + e] = db[e]; red[2][rl][sub * 8 + e] = gsv[e]; }
    __syncthreads();
    for (int item = threadIdx.x; item < 3 * 128; item += 256) {
        const int c = item & 127, which = item >> 7;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[which][k][c];
        atomicAdd((which == 0 ? dgamma : (which == 1 ? dbeta : gsum)) + c, s);
    }
}
template <int RPT>
__global__ __launch_bounds__(256) void k_lnbwd_sum4(const bf16* __restrict__ dApart, const bf16* __restrict__ X, const bf16* __restrict__ G,
                                                    const float* __restrict__ gamma, bf16* __restrict__ g_in, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, float* __restrict__ gsum, int64_t M) {
    lnbwd_sum4_body<RPT>(dApart, X, G, gamma, g_in, dgamma, dbeta, gsum, M, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------
// MLP forward, second generation (bf16): persistent workgroups, ALL weights resident in registers.
//   wave w keeps W1 rows [64w, 64w+64) (GEMM1: its 64 hidden units) and W2 rows [16w, 16w+16) (GEMM2: its 16 output
//   channels over all 512 hidden units): 2 x 64 VGPRs.  Per 32-token tile: LN(x) -> sA, GEMM1 + GELU -> sH (bf16,
//   [32][512] as four swizzled [32][128] tiles), barrier, GEMM2 over the whole hidden axis, epilogue x + ls2 (y + b2)
//   with x taken from the raw tile that is still in LDS.  Raw x tiles arrive by LDS-direct loads two tiles ahead.
//   Nothing is streamed from L2 inside the loop and there are two barriers per tile.
// ---------------------------------------------------------------------------------------------------------------
constexpr int F_BM = 32, F_NW = 8, F_THR = F_NW * 64;

__global__ __launch_bounds__(F_THR) void k_mlp_fwd_r(const bf16* __restrict__ X, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                     const bf16* __restrict__ W1, const float* __restrict__ b1, const bf16* __restrict__ W2,
                                                     const float* __restrict__ b2, const float* __restrict__ ls2, bf16* __restrict__ out, int64_t M,
                                                     bf16* __restrict__ xn_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sA = reinterpret_cast<bf16*>(smem);            // [2][32][128]  LN(x)
    bf16* sH = sA + 2 * F_BM * 128;                      // [4][32][128]  GELU output, hidden chunk major
    bf16* sXr = sH + 4 * F_BM * 128;                     // [3][32][128]  raw x: in use / landed / in flight
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, sub = lane & 15;
    const int64_t ntiles_total = (M + F_BM - 1) / F_BM;
    const int64_t per = (ntiles_total + gridDim.x - 1) / gridDim.x;
    const int64_t tile0 = (int64_t)blockIdx.x * per;
    int64_t ntiles = ntiles_total - tile0;
    if (ntiles > per) ntiles = per;
    if (ntiles <= 0) return;

    bf16x8 w1f[4][4], w2f[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w1f[nt][ks] = *reinterpret_cast<const bf16x8*>(W1 + (int64_t)(64 * w + 16 * nt + i) * 128 + 32 * ks + 8 * g);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w2f[ks] = *reinterpret_cast<const bf16x8*>(W2 + (int64_t)(16 * w + i) * 512 + 32 * ks + 8 * g);
    f32x4 b1v[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b1v[nt] = *reinterpret_cast<const f32x4*>(b1 + 64 * w + 16 * nt + 4 * g);
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(b2 + 16 * w + 4 * g), lsv = *reinterpret_cast<const f32x4*>(ls2 + 16 * w + 4 * g);

    auto nx3 = [](int sl) { return sl == 2 ? 0 : sl + 1; };
    auto issue = [&](int64_t t, int sl) {                // exactly one LDS-direct load per wave per call
        const int64_t tt = t < ntiles ? t : ntiles - 1;
        const int64_t row0 = (tile0 + tt) * F_BM;
        const int nvalid = (int)((M - row0) < F_BM ? (M - row0) : F_BM);
        stage_tile_async<bf16, F_BM, F_THR>(sXr + sl * F_BM * 128, X + row0 * 128, 128, nvalid);
    };
    auto layernorm = [&](int64_t t, int sl) {            // wave w normalises the 4 rows its own load delivered
        float gmv[8], btv[8];
        {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(ln_g + sub * 8), g1 = *reinterpret_cast<const f32x4*>(ln_g + sub * 8 + 4);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(ln_b + sub * 8), c1 = *reinterpret_cast<const f32x4*>(ln_b + sub * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gmv[e] = g0[e]; gmv[4 + e] = g1[e]; btv[e] = c0[e]; btv[4 + e] = c1[e]; }
        }
        const int r = 4 * w + (lane >> 4);
        float v[8];
        tile_load8(sXr + sl * F_BM * 128, r, sub * 8, v);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        const float mean = reduce16(s) * (1.0f / 128.0f);
        float qv = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] -= mean; qv += v[e] * v[e]; }
        const float rstd = rsqrtf(reduce16(qv) * (1.0f / 128.0f) + KASF_LN_EPS);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * rstd * gmv[e] + btv[e];
        tile_store8(sA + (int)(t & 1) * F_BM * 128, r, sub * 8, v);
        if (xn_out != nullptr) {                         // training: the backward pass streams LN(x) instead of recomputing it
            const int64_t row = (tile0 + t) * F_BM + r;
            if (row < M) store8(xn_out + row * 128 + sub * 8, v);
        }
    };
    issue(0, 0);
    issue(1, 1);
    wait_async_le<1>();
    layernorm(0, 0);
    int sl = 0;
    for (int64_t t = 0; t < ntiles; ++t, sl = nx3(sl)) {
        const bf16* cA = sA + (int)(t & 1) * F_BM * 128;
        const bf16* cX = sXr + sl * F_BM * 128;
        const int64_t row0 = (tile0 + t) * F_BM;
        barrier_keep_async();                            // B1: LN(x_t) complete; everyone is past tile t-1
        {   // ---- GEMM1: this wave's 64 hidden units x 32 tokens, then GELU -> sH ----
            f32x4 acc1[4][2];
            zero_acc(acc1);
            bf16x8 fa[2][2];
            fa[0][0] = tok_frag(cA, i, 0);
            fa[0][1] = tok_frag(cA, 16 + i, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) { fa[(ks + 1) & 1][0] = tok_frag(cA, i, ks + 1); fa[(ks + 1) & 1][1] = tok_frag(cA, 16 + i, ks + 1); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc1[nt][mt] = mfma16(w1f[nt][ks], fa[ks & 1][mt], acc1[nt][mt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            bf16* hT = sH + (w >> 1) * F_BM * 128;       // hidden chunk of this wave
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float h[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = gelu_f<bf16>(acc1[nt][mt][r] + b1v[nt][r]);
                    store4(hT + Tile<bf16>::off4(mt * 16 + i, 64 * (w & 1) + 16 * nt + 4 * g), h);
                }
        }
        barrier_keep_async();                            // B2: the whole [32][512] hidden tile is in LDS
        issue(t + 2, nx3(nx3(sl)));
        {   // ---- GEMM2: this wave's 16 output channels x 32 tokens over all 512 hidden units ----
            f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            bf16x8 fh[2][2];
            fh[0][0] = tok_frag(sH, i, 0);
            fh[0][1] = tok_frag(sH, 16 + i, 0);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks + 1 < 16) {
                    const bf16* hT = sH + ((ks + 1) >> 2) * F_BM * 128;
                    fh[(ks + 1) & 1][0] = tok_frag(hT, i, (ks + 1) & 3);
                    fh[(ks + 1) & 1][1] = tok_frag(hT, 16 + i, (ks + 1) & 3);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc2[0] = mfma16(w2f[ks], fh[ks & 1][0], acc2[0]);
                acc2[1] = mfma16(w2f[ks], fh[ks & 1][1], acc2[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // x(t+1) has landed when only the youngest request (t+2) is still outstanding; this tile's output stores are
            // issued AFTER this wait so that they never count as "youngest" (they get the whole next tile to drain)
            wait_async_le<1>();
            if (t + 1 < ntiles) layernorm(t + 1, nx3(sl));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int64_t row = row0 + mt * 16 + i;
                if (row < M) {
                    float x[4], y[4];
                    load4(cX + Tile<bf16>::off4(mt * 16 + i, 16 * w + 4 * g), x);      // residual from the raw tile still in LDS
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = x[r] + lsv[r] * (acc2[mt][r] + b2v[r]);
                    store4(out + row * 128 + 16 * w + 4 * g, y);
                }
            }
        }
    }
    wait_async();
}


// ---------------------------------------------------------------------------------------------------------------
// End of the MLP backward in ONE launch: sums the per-range partial tiles of dW1 [512][128] and dW2 [128][512] (fixed order)
// and, for fc2, applies the layer-scale algebra that k_finalize_ls would: with G = g^T H (unscaled),
//   dls[c] += sum_k W2[c][k] G[c][k] + b2[c] gsum[c];   dW2[c][:] += ls[c] G[c][:];   db2[c] = ls[c] gsum[c].
// Workgroups 0..127 own 4 rows of dW1 each, workgroups 128..255 one row of dW2 each (512 floats per workgroup); the two
// halves of a workgroup take the even / odd splits.
// ---------------------------------------------------------------------------------------------------------------
// DEFER: the column sums of g are not complete yet (the launch shares k_lnbwd_sum4's grid): leave their terms to that kernel's last workgroup
template <bool DEFER>
__device__ __forceinline__ void mlp_wfinish_body(const float* __restrict__ p1, const float* __restrict__ p2, float* __restrict__ dW1,
                                                 float* __restrict__ dW2, int splits, const float* __restrict__ W2, const float* __restrict__ b2,
                                                 const float* __restrict__ ls, float* __restrict__ gsum_db2, float* __restrict__ dls, int bid) {
    __shared__ f32x4 sHalf[128];
    __shared__ float sDot[2];
    const int lane = threadIdx.x & 127, half = threadIdx.x >> 7;
    const bool second = bid >= 128;
    const int blk = second ? bid - 128 : bid;
    const float* part = second ? p2 : p1;
    const int64_t e = (int64_t)blk * 512 + lane * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    int z = half;
    for (; z + 6 < splits; z += 8) {                    // four partial tiles in flight per thread; the order of the additions is the two-at-a-time loop's
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(part + (int64_t)z * 65536 + e);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(part + (int64_t)(z + 2) * 65536 + e);
        const f32x4 u1 = *reinterpret_cast<const f32x4*>(part + (int64_t)(z + 4) * 65536 + e);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(part + (int64_t)(z + 6) * 65536 + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] += u0[q]; b[q] += v0[q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] += u1[q]; b[q] += v1[q]; }
    }
    for (; z + 2 < splits; z += 4) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(part + (int64_t)z * 65536 + e);
        const f32x4 v = *reinterpret_cast<const f32x4*>(part + (int64_t)(z + 2) * 65536 + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] += u[q]; b[q] += v[q]; }
    }
    if (z < splits) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(part + (int64_t)z * 65536 + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += u[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] += b[q];
    if (half == 1) sHalf[lane] = a;
    __syncthreads();
    if (half == 0) {
        const f32x4 o = sHalf[lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += o[q];                    // a = the summed gradient (G for fc2)
    }
    const bool fin = second && W2 != nullptr;
    float dot = 0.f;
    if (half == 0) {
        float* dst = (second ? dW2 : dW1) + e;
        f32x4 cur = *reinterpret_cast<f32x4*>(dst);
        const float l = fin ? ls[blk] : 1.0f;
        if (fin) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(W2 + e);
#pragma unroll
            for (int q = 0; q < 4; ++q) dot += w[q] * a[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] += l * a[q];
        *reinterpret_cast<f32x4*>(dst) = cur;
    }
    if (fin) {
        dot = reduce64(dot);
        if (half == 0 && (threadIdx.x & 63) == 0) sDot[threadIdx.x >> 6] = dot;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (DEFER) atomicAdd(dls + blk, sDot[0] + sDot[1]);       // (the b2 . gsum term arrives by a second atomic add)
            else {
                const float gs = gsum_db2[blk];
                dls[blk] += sDot[0] + sDot[1] + b2[blk] * gs;
                gsum_db2[blk] = gs * ls[blk];
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_mlp_wfinish(const float* __restrict__ p1, const float* __restrict__ p2, float* __restrict__ dW1,
                                                     float* __restrict__ dW2, int splits, const float* __restrict__ W2, const float* __restrict__ b2,
                                                     const float* __restrict__ ls, float* __restrict__ gsum_db2, float* __restrict__ dls,
                                                     unsigned* __restrict__ zero_words, int n_zero) {
    if (blockIdx.x == 255)         // the hand-off flags of k_mlp_bwd_s<true> (complete by stream order) are cleared for the next launch on this scratch
        for (int k = threadIdx.x; k < n_zero; k += 256) zero_words[k] = 0u;
    mlp_wfinish_body<false>(p1, p2, dW1, dW2, splits, W2, b2, ls, gsum_db2, dls, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// k_lnbwd_sum4 and k_mlp_wfinish in ONE launch: workgroups 0..255 reduce the weight-gradient partial tiles (they only read what k_mlp_bwd_s left),
// the others stream the dA partials; the two terms of the finish that need colsum(g) -- complete only when the LAST streaming workgroup has
// added its share -- are applied by that workgroup (ticket counter; every workgroup waits for the acknowledgement of its atomic adds before it
// takes its ticket, the reader takes the sums with agent-scope loads).  One launch and ~8 us of dependent latency less per MLP block, 156 times per step.
// ---------------------------------------------------------------------------------------------------------------
struct MlpFinArgs {
    const float *p1, *p2;
    float *dW1, *dW2;
    int splits;
    const float *W2, *b2, *ls;
    float *gsum_db2, *dls;
    unsigned* ticket;              // zero before the launch; the last workgroup leaves it zero again
};
template <int RPT>
__global__ __launch_bounds__(256) void k_lnbwd_sum4_fin(const bf16* __restrict__ dApart, const bf16* __restrict__ X, const bf16* __restrict__ G,
                                                        const float* __restrict__ gamma, bf16* __restrict__ g_in, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, float* __restrict__ gsum, int64_t M, const MlpFinArgs fa) {
    if (blockIdx.x < 256) {
        mlp_wfinish_body<true>(fa.p1, fa.p2, fa.dW1, fa.dW2, fa.splits, fa.W2, fa.b2, fa.ls, fa.gsum_db2, fa.dls, (int)blockIdx.x);
        return;
    }
    const int nblk = (int)gridDim.x - 256;
    lnbwd_sum4_body<RPT>(dApart, X, G, gamma, g_in, dgamma, dbeta, gsum, M, (int)blockIdx.x - 256, nblk);
    __shared__ int sLast;
    // this workgroup's atomic adds (device scope, performed at the memory side) have been acknowledged before its ticket is taken.  NOT __threadfence():
    // that also writes back the XCD's dirty L2 lines -- the g_in rows just stored -- from every thread of every workgroup (measured: +14 ms per step)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sLast = atomicAdd(fa.ticket, 1u) == (unsigned)(nblk - 1);
    __syncthreads();
    if (sLast) {
        if (fa.W2 != nullptr && threadIdx.x < 128) {
            const int c = threadIdx.x;
            const float gs = __hip_atomic_load(fa.gsum_db2 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(fa.dls + c, fa.b2[c] * gs);
            fa.gsum_db2[c] = gs * fa.ls[c];
        }
        if (threadIdx.x == 0) atomicExch(fa.ticket, 0u);
    }
}

}  // namespace

bool kasf_mlp_bwd_xchg_enabled() {
    static const bool on = getenv("KASF_MLP_BWD_XCHG") != nullptr;
    return on;
}
// scratch needs: dApart = 4*M*128 bf16;  partial >= 2 * ranges * 65536 floats (returned through *ranges_out)
int kasf_mlp_bwd_q_ranges(int64_t M) {
    const int64_t tiles = (M + Q_BM - 1) / Q_BM;
    static const int cap = getenv("KASF_MLP_BWD_RANGES") ? atoi(getenv("KASF_MLP_BWD_RANGES")) : 64;   // measurement switch
    return (int)(tiles < cap ? tiles : cap);
}
void kasf_launch_mlp_bwd_q(hipStream_t s, const void* x, const void* xn, const void* g, const float* ln_g, const void* W1, const float* b1,
                           const void* W2ts, const void* W1t, void* dApart, float* partial, float* dW1, float* dW2, float* db1, float* gsum, void* g_in,
                           float* dgamma, float* dbeta, int64_t M, const float* W2, const float* b2, const float* ls2, float* dls2, unsigned* err) {
    const int ranges = kasf_mlp_bwd_q_ranges(M);
    const int64_t tiles = (M + Q_BM - 1) / Q_BM;
    const int tpr = (int)((tiles + ranges - 1) / ranges);
    const int used = (int)((tiles + tpr - 1) / tpr);             // ranges that own at least one tile
    float* p1 = partial;
    float* p2 = partial + (int64_t)used * 512 * 128;
    static const bool lockstep = getenv("KASF_MLP_BWD_LOCKSTEP") != nullptr;                          // measurement switch: the symmetric kernel
    // The in-kernel reduction of the four dA partials (k_mlp_bwd_s<true>) is correct and placement-independent but SLOWER than the
    // two-kernel chain on the hardware (173 vs 159 us per launch): it stays an opt-in experiment.
    const bool sum4 = !kasf_mlp_bwd_xchg_enabled();
    unsigned* flags = reinterpret_cast<unsigned*>(partial + KASF_MLP_PARTIAL_FLOATS);                 // [64 ranges][16] + the timeout word
    if (!lockstep && !sum4 && M * 1024 < (int64_t(1) << 32)) {        // (the hand-off addresses its [4][M][128] buffer with 32-bit byte offsets)
        // in-kernel reduction of the four dA partials: no second pass over them.  The flags are zero here: cleared by the engine at the start
        // of a backward pass / by the op entry point, and by every k_mlp_wfinish for the launch that follows it.
        kasf_launch_mlp_bwd_x(s, x, xn, g, ln_g, W1, b1, W2ts, W1t, dApart, p1, p2, db1, gsum, g_in, dgamma, dbeta, flags, err ? err : flags + KASF_MLP_ERR_WORD, M,
                              tpr, used);
        hipLaunchKernelGGL(k_mlp_wfinish, dim3(256), dim3(256), 0, s, p1, p2, dW1, dW2, used, W2, b2, ls2, gsum, dls2, flags, 16 * used);
        return;
    }
    if (!lockstep) {
        kasf_launch_mlp_bwd_s(s, xn, g, W1, b1, W2ts, W1t, dApart, p1, p2, db1, M, tpr, used);
    } else {
        const size_t sh = (size_t)(8 * Q_BM * 128) * sizeof(bf16);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_bwd_q<Q_NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipLaunchKernelGGL(k_mlp_bwd_q<Q_NW>, dim3(4 * used), dim3(Q_NW * 64), sh, s, (const bf16*)xn, (const bf16*)g, (const bf16*)W1, b1, (const bf16*)W2ts,
                           (const bf16*)W1t, (bf16*)dApart, p1, p2, db1, M, tpr);
    }
    static const int rpt = getenv("KASF_SUM4_RPT") ? atoi(getenv("KASF_SUM4_RPT")) : 2;               // measurement switches
    static const int cap = getenv("KASF_SUM4_BLOCKS") ? atoi(getenv("KASF_SUM4_BLOCKS")) : 512;
    static const bool two_launches = getenv("KASF_MLP_FINISH_SEPARATE") != nullptr;                   // k_lnbwd_sum4, then k_mlp_wfinish (round-1 form)
    int64_t blocks = (M + 16 * rpt - 1) / (16 * rpt);
    if (blocks > cap) blocks = cap;                     // few blocks: every block ends with 384 same-address atomics (contended atomics serialise)
    if (!two_launches && rpt == 2) {
        // gsum doubles as the fc2 bias gradient slot; the ticket word lives behind the hand-off flags of the scratch (zeroed by the engine at the start
        // of a backward pass / by the op entry point, left zero by every launch)
        const MlpFinArgs fa{p1, p2, dW1, dW2, used, W2, b2, ls2, gsum, dls2, flags + KASF_MLP_TICKET_WORD};
        hipLaunchKernelGGL(k_lnbwd_sum4_fin<2>, dim3((unsigned)blocks + 256), dim3(256), 0, s, (const bf16*)dApart, (const bf16*)x, (const bf16*)g, ln_g,
                           (bf16*)g_in, dgamma, dbeta, gsum, M, fa);
        return;
    }
    if (rpt == 1) hipLaunchKernelGGL(k_lnbwd_sum4<1>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16*)dApart, (const bf16*)x, (const bf16*)g, ln_g, (bf16*)g_in,
                                     dgamma, dbeta, gsum, M);
    else if (rpt == 2) hipLaunchKernelGGL(k_lnbwd_sum4<2>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16*)dApart, (const bf16*)x, (const bf16*)g, ln_g,
                                          (bf16*)g_in, dgamma, dbeta, gsum, M);
    else hipLaunchKernelGGL(k_lnbwd_sum4<4>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16*)dApart, (const bf16*)x, (const bf16*)g, ln_g, (bf16*)g_in,
                            dgamma, dbeta, gsum, M);
    // after k_lnbwd_sum4 (gsum complete): both partial reductions + the fc2 layer-scale algebra (W2 == nullptr: dW2 stays unscaled)
    hipLaunchKernelGGL(k_mlp_wfinish, dim3(256), dim3(256), 0, s, p1, p2, dW1, dW2, used, W2, b2, ls2, gsum, dls2, (unsigned*)nullptr, 0);
}

void kasf_launch_mlp_fwd_r(hipStream_t s, const void* x, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2,
                           const float* b2, const float* ls2, void* out, int64_t M, void* xn_out) {
    const int64_t tiles = (M + F_BM - 1) / F_BM;
    static const int cap = getenv("KASF_MLP_FWD_GRID") ? atoi(getenv("KASF_MLP_FWD_GRID")) : 256;     // measurement switch
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    static const bool lockstep = getenv("KASF_MLP_FWD_LOCKSTEP") != nullptr;                          // measurement switch: the symmetric kernel
    if (!lockstep) {
        kasf_launch_mlp_fwd_s(s, x, ln_g, ln_b, W1, b1, W2, b2, ls2, out, M, xn_out, grid);
        return;
    }
    const size_t sh = (size_t)(9 * F_BM * 128) * sizeof(bf16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_fwd_r), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(k_mlp_fwd_r, dim3(grid), dim3(F_THR), sh, s, (const bf16*)x, ln_g, ln_b, (const bf16*)W1, b1, (const bf16*)W2, b2, ls2, (bf16*)out, M, (bf16*)xn_out);
}
