// MLP forward, third generation (bf16): the waves of a workgroup are SPECIALISED.
//
// The second generation gave every wave the same program: GEMM1 (32 MFMAs), GELU (~370 vector instructions), barrier, GEMM2 (32 MFMAs),
// epilogue.  The two waves that share a SIMD belong to the same workgroup and move in lockstep between the barriers, so the matrix pipe idles
// during GELU and the vector ALU idles during the GEMMs: measured 18.6 % MFMA busy, 42 % VALU busy, and a tile time equal to the SUM of the
// phases plus the stalls (SQ counters in profiles/r1_mlp_sq_counters.json).  Software-pipelining the symmetric program did not help.
//
// Here waves 0-3 (one per SIMD) are PRODUCERS: wave p keeps W1 rows [128p, 128p+128) in 128 VGPRs and does GEMM1 + GELU for its quarter of the
// hidden units.  Waves 4-7 (again one per SIMD) are CONSUMERS: wave c keeps W2 rows [32c, 32c+32) in 128 VGPRs and does GEMM2 of the PREVIOUS
// tile, the residual epilogue, the LDS-direct loads and the LayerNorm of the NEXT tile.  A SIMD therefore always holds one vector-heavy and one
// matrix-heavy wave with no common phase, B fragments are shared by twice as many A fragments (LDS reads per MFMA halve), and there is ONE
// barrier per tile:
//     iteration t:   producers  GEMM1(t) + GELU(t) -> sH[t & 1]        consumers  GEMM2(t-1) <- sH[(t-1) & 1], epilogue(t-1), LN(t+1) -> sA[(t+1) & 1]
// Packed fp32 instructions (v_pk_fma_f32 ...) do not overlap with MFMAs of another wave on gfx950 (tools/valu_probe.hip: an MFMA wave and a
// v_pk_fma_f32 wave on one SIMD take the sum of their times, an MFMA wave and a v_fma_f32 wave the maximum), but one wave issues at most one
// vector instruction per ~5.3 cycles whatever it is, so the producers' GELU is written in packed form anyway (half the instructions; building this
// file with and without -fno-slp-vectorize measured the same, the explicit packed GELU 6 % faster than either).
// LDS: sA 2 x 8 KB, sH 2 x 32 KB, raw x ring 4 x 8 KB (tiles t-1, t, t+1 in use, t+2 in flight), parameters 4 KB.
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"
#include <type_traits>

#ifdef KASF_LSTAMP
constexpr unsigned LSTAMP_CAP = 16384;
__device__ long long g_lstamp[LSTAMP_CAP];
__device__ unsigned g_lstamp_n;
#endif
#ifdef KASF_PROBE_TIMERS
__device__ long long g_prof[32];
#define TMARK(k) do { const long long _n = clock64(); if (lane == 0 && blockIdx.x == 7 && (w & 3) == 0) g_prof[k] += _n - _t; _t = _n; } while (0)   // wave 0 (producer) and wave 4 (consumer) of one workgroup only: no lost updates
#define TSTART() long long _t = clock64()
#else
#define TMARK(k) do {} while (0)
#define TSTART() do {} while (0)
#endif

#ifndef KASF_FWD_SGB            // vector instructions scheduled behind each MFMA of the next slice
#define KASF_FWD_SGB 9
#endif

namespace {

constexpr int S_BM = 32, S_THR = 512, TL = S_BM * 128;
#ifndef KASF_BWD_DASTORE_EARLY
#define KASF_BWD_DASTORE_EARLY 0
#endif
#ifndef KASF_BWD_RING
#define KASF_BWD_RING 5
#endif
#ifndef KASF_BWD_ISSUE_AT          // where in the producers' iteration the look-ahead loads are issued: 0 at the loop top, 1 behind the first MFMA phase (shipped), 2 in front of the
#define KASF_BWD_ISSUE_AT 1        // last GELU slice.  At the loop top they follow the consumers' dA stores of the previous iteration through the same address path: k_mlp_bwd_s
#endif                             // 156.7 us in step (0), 146.5 (1), 151.3 (2); the dA stores in front of the weight-gradient MFMAs instead (KASF_BWD_DASTORE_EARLY): 147.1, both: 148.9
constexpr int BW_RING = KASF_BWD_RING, BW_AHEAD = BW_RING - 2;      // backward: LN(x) / g ring slots; tiles t-1 .. t+AHEAD live (t+2 .. t+AHEAD in flight)

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// GEMM2 of the forward pass: H and W2 are FP16 bit patterns in bf16-typed registers / tiles (same size, same layouts)
__device__ __forceinline__ f32x4 mfma16h(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 tok_frag(const bf16* s, int row, int ks) {
    const int g = (threadIdx.x & 63) >> 4;
    return *reinterpret_cast<const bf16x8*>(s + Tile<bf16>::chunk_off(row, 4 * ks + g));
}

__global__ __launch_bounds__(S_THR) void k_mlp_fwd_s(const bf16* __restrict__ X, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                     const bf16* __restrict__ W1, const float* __restrict__ b1, const bf16* __restrict__ W2,
                                                     const float* __restrict__ b2, const float* __restrict__ ls2, bf16* __restrict__ out, int64_t M,
                                                     bf16* __restrict__ xn_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sA = reinterpret_cast<bf16*>(smem);            // [2][32][128]     LN(x)
    bf16* sH = sA + 2 * TL;                              // [2][4][32][128]  GELU output, hidden quarter major
    bf16* sXr = sH + 8 * TL;                             // [4][32][128]     raw x ring
    float* sPar = reinterpret_cast<float*>(sXr + 4 * TL);   // b1[512] | b2[128] | ls2[128] | ln_g[128] | ln_b[128]
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int64_t ntiles_total = (M + S_BM - 1) / S_BM;
    const int64_t per = (ntiles_total + gridDim.x - 1) / gridDim.x;
    const int64_t tile0 = (int64_t)blockIdx.x * per;
    int64_t ntiles = ntiles_total - tile0;
    if (ntiles > per) ntiles = per;
    if (ntiles <= 0) return;

    sPar[threadIdx.x] = b1[threadIdx.x];
    if (threadIdx.x < 128) {
        sPar[512 + threadIdx.x] = b2[threadIdx.x];
        sPar[640 + threadIdx.x] = ls2[threadIdx.x];
        sPar[768 + threadIdx.x] = ln_g[threadIdx.x];
        sPar[896 + threadIdx.x] = ln_b[threadIdx.x];
    }
    __syncthreads();

    if (w < 4) {
        // ------------------------------------------------ producer: GEMM1 + GELU of hidden units [128w, 128w + 128) ------------------------------------------------
        bf16x8 w1f[8][4];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) w1f[nt][ks] = *reinterpret_cast<const bf16x8*>(W1 + (int64_t)(128 * w + 16 * nt + i) * 128 + 32 * ks + 8 * g);
        f32x4 b1v[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) b1v[nt] = *reinterpret_cast<const f32x4*>(sPar + 128 * w + 16 * nt + 4 * g);
        barrier_keep_async();                            // LN(x_0) is in sA[0]
        TSTART();
        for (int64_t t = 0; t <= ntiles; ++t) {
            if (t < ntiles) {
                const bf16* cA = sA + (int)(t & 1) * TL;
                bf16* hT = sH + (int)(t & 1) * 4 * TL + w * TL;
                bf16x8 fa[4][2];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { fa[ks][0] = tok_frag(cA, i, ks); fa[ks][1] = tok_frag(cA, 16 + i, ks); }
                f32x4 acc[2][2];
                auto gemm1 = [&](f32x4 (&a)[2], int nt) {      // accumulators start from the bias (the lane's 4 rows = 4 hidden units): no add afterwards
                    a[0] = b1v[nt];
                    a[1] = b1v[nt];
#pragma unroll
#ifdef KASF_PROBE_NO_MFMA1
                    for (int ks = 0; ks < 1; ++ks) {
#else
                    for (int ks = 0; ks < 4; ++ks) {
#endif
                        a[0] = mfma16(w1f[nt][ks], fa[ks][0], a[0]);
                        a[1] = mfma16(w1f[nt][ks], fa[ks][1], a[1]);
                    }
                };
                gemm1(acc[0], 0);
                __builtin_amdgcn_sched_barrier(0);
                TMARK(0);
                auto slice = [&](auto NT) {              // MFMAs of slice nt+1 spread over the GELU of slice nt
                    constexpr int nt = decltype(NT)::value;
                    if (nt < 7) gemm1(acc[(nt + 1) & 1], nt + 1);
                    f32x2 y[4];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        y[2 * mt] = f32x2{acc[nt & 1][mt][0], acc[nt & 1][mt][1]};
                        y[2 * mt + 1] = f32x2{acc[nt & 1][mt][2], acc[nt & 1][mt][3]};
                    }
                    f16x2 hh[4];
#ifdef KASF_KO_FGELU
#pragma unroll
                    for (int k = 0; k < 4; ++k) hh[k] = __builtin_convertvector(y[k], f16x2);
#else
                    gelu_pairs_h(y, hh);                 // packed fp16, H stays fp16 (same 2-byte tile layout)
#endif
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const f16x4 h4 = {hh[2 * mt][0], hh[2 * mt][1], hh[2 * mt + 1][0], hh[2 * mt + 1][1]};
                        *reinterpret_cast<f16x4*>(hT + Tile<bf16>::off4(mt * 16 + i, 16 * nt + 4 * g)) = h4;
                    }
                    if (nt < 7) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, KASF_FWD_SGB, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                slice(std::integral_constant<int, 0>{});
                slice(std::integral_constant<int, 1>{});
                slice(std::integral_constant<int, 2>{});
                slice(std::integral_constant<int, 3>{});
                slice(std::integral_constant<int, 4>{});
                slice(std::integral_constant<int, 5>{});
                slice(std::integral_constant<int, 6>{});
                slice(std::integral_constant<int, 7>{});
                TMARK(1);
            }
            barrier_keep_async();
            TMARK(2);
        }
    } else {
        // ------------------------------------------------ consumer: GEMM2 for output channels [32c, 32c + 32), epilogue, loads, LayerNorm ------------------------------------------------
        const int c = w - 4, sub = lane & 15;
#ifndef KASF_PROBE_NO_PRIO
        __builtin_amdgcn_s_setprio(3);                   // short latency-bound sections: win the issue arbitration against the producer on the same SIMD
#endif
        bf16x8 w2f[2][16];

        // two LDS-direct loads per consumer wave: rows [8c, 8c + 8) of tile t, the rows it will normalise.  Tile-invariant parts once (see k_mlp_bwd_s).
        const int nt32 = (int)ntiles;
        const int64_t rows_here = M - tile0 * S_BM;
        const int rows_in_range = (int)(rows_here < (int64_t)nt32 * S_BM ? rows_here : (int64_t)nt32 * S_BM);
        const void* uxb = uniform_ptr(X + tile0 * S_BM * 128);
        const unsigned ldsX = __builtin_amdgcn_readfirstlane(lds_addr(sXr));
        auto issue = [&](int t) {
            const int tt = t < nt32 ? t : nt32 - 1;
            int nvalid = rows_in_range - tt * S_BM;
            nvalid = nvalid < S_BM ? nvalid : S_BM;
            const void* ub = (const char*)uxb + (unsigned)tt * (S_BM * 256u);
            const unsigned base = ldsX + (unsigned)(t & 3) * (TL * 2u);
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int inst = 2 * c + k, row = inst * 4 + (ln >> 4), ch = (ln & 15) ^ (row & 15);
                const int srow = row < nvalid ? row : nvalid - 1;
                glds16_s(ub, (unsigned)srow * 256u + (unsigned)ch * 16u, base + (unsigned)inst * 1024u);
            }
        };
        auto layernorm = [&](int64_t t) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sPar + 768 + sub * 8), g1 = *reinterpret_cast<const f32x4*>(sPar + 772 + sub * 8);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(sPar + 896 + sub * 8), c1 = *reinterpret_cast<const f32x4*>(sPar + 900 + sub * 8);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = 8 * c + 4 * k + (lane >> 4);
                float v[8];
                tile_load8(sXr + (int)(t & 3) * TL, r, sub * 8, v);
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[e];
                const float mean = reduce16(s) * (1.0f / 128.0f);
                float qv = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] -= mean; qv = __builtin_fmaf(v[e], v[e], qv); }
                const float rstd = rsqrtf(reduce16(qv) * (1.0f / 128.0f) + KASF_LN_EPS);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = __builtin_fmaf(v[e], rstd * g0[e], c0[e]); v[4 + e] = __builtin_fmaf(v[4 + e], rstd * g1[e], c1[e]); }
                tile_store8(sA + (int)(t & 1) * TL, r, sub * 8, v);
                if (xn_out != nullptr) {                 // training: the backward pass streams LN(x) instead of recomputing it
                    const int64_t row = (tile0 + t) * S_BM + r;
                    if (row < M) store8(xn_out + row * 128 + sub * 8, v);
                }
            }
        };
        issue(0);                                        // the first tiles are in flight while the weights arrive from L2
        issue(1);
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) w2f[n2][ks] = *reinterpret_cast<const bf16x8*>(W2 + (int64_t)(32 * c + 16 * n2 + i) * 512 + 32 * ks + 8 * g);
        wait_async();                                    // tiles 0, 1 and the weights
        layernorm(0);
        barrier_keep_async();
        TSTART();
        for (int64_t t = 0; t <= ntiles; ++t) {
            issue((int)t + 2);
            TMARK(8);                                // into the slot of x(t-2)
            f32x4 acc2[2][2];
            if (t >= 1) {                                // GEMM2 of tile t-1 over all 512 hidden units
                const bf16* cH = sH + (int)((t - 1) & 1) * 4 * TL;
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) acc2[n2][0] = acc2[n2][1] = *reinterpret_cast<const f32x4*>(sPar + 512 + 32 * c + 16 * n2 + 4 * g);   // starts from b2
                bf16x8 fh[3][2];                          // fragment ring, two k-steps ahead: LDS latency under load exceeds the 64 cycles of one step's MFMAs
                auto frag = [&](int ks, int slot) {
                    const bf16* hT = cH + (ks >> 2) * TL;
                    fh[slot][0] = tok_frag(hT, i, ks & 3);
                    fh[slot][1] = tok_frag(hT, 16 + i, ks & 3);
                };
                frag(0, 0);
                frag(1, 1);
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    if (ks + 2 < 16) frag(ks + 2, (ks + 2) % 3);
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2) {
                        acc2[n2][0] = mfma16h(w2f[n2][ks], fh[ks % 3][0], acc2[n2][0]);
                        acc2[n2][1] = mfma16h(w2f[n2][ks], fh[ks % 3][1], acc2[n2][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            TMARK(9);
            wait_async_le<2>();                          // everything older than the two loads just issued: x(t+1) has landed, earlier stores have drained
            TMARK(10);
            if (t + 1 < ntiles) layernorm(t + 1);
            TMARK(11);
            if (t >= 1) {
                const int64_t row0 = (tile0 + t - 1) * S_BM;
                const bf16* cX = sXr + (int)((t - 1) & 3) * TL;
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const int col = 32 * c + 16 * n2 + 4 * g;
                    const f32x4 lsv = *reinterpret_cast<const f32x4*>(sPar + 640 + col);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int64_t row = row0 + mt * 16 + i;
                        if (row < M) {
                            float x[4], y[4];
                            load4(cX + Tile<bf16>::off4(mt * 16 + i, col), x);      // residual from the raw tile still in LDS
#pragma unroll
                            for (int r = 0; r < 4; ++r) y[r] = __builtin_fmaf(lsv[r], acc2[n2][mt][r], x[r]);
                            store4(out + row * 128 + col, y);
                        }
                    }
                }
            }
            TMARK(12);
            barrier_keep_async();
            TMARK(13);
        }
        wait_async();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Fused MLP backward (bf16): hidden-quarter ownership with register-resident weights and specialised waves.
// A workgroup owns ONE QUARTER of the hidden units (128 of 512) and a long range of tokens: its slices of W1, (ls2.W2)^T and W1^T live in VGPRs for
// the whole kernel; per 32-token tile  Z_q = W1_q LN(x)^T,  dH_q = (ls2.W2)_q^T g^T,  H_q = GELU(Z_q),  dZ_q = dH_q GELU'(Z_q)  go registers -> LDS
// (bf16) once;  dA_q = W1_q^T dZ_q is stored as a bf16 partial (summed over the four quarters by k_lnbwd_sum4_fin, which also does the LayerNorm
// backward + residual); the weight gradients  dW1_q += dZ_q^T LN(x)  and  dW2_q += g^T H_q  accumulate in registers over the whole token range from
// transposed LDS fragments (ds_read_b64_tr_b16) and leave the kernel once as per-range partial tiles.  The 512-wide H / dZ never reach HBM.
// XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), so the four hidden quarters of one token range sit at
// blockIdx b, b+8, b+16, b+24: same XCD, same L2 -> LN(x) and g cross the fabric once, not four times.
// Waves 0-3 (one per SIMD) are producers and waves 4-7 consumers:
//     iteration t:   producers  Z_q(t), dH_q(t), GELU / GELU' -> sH[t & 1], sD[t & 1]                       (wave p: hidden units [32p, 32p + 32) of the quarter)
//                    consumers  loads of tile t+2;  dA_q(t-1) = W1_q^T dZ_q  (wave c: channels [32c, 32c + 32)),
//                               dW1_q += dZ_q^T LN(x),  dW2_q += g^T H_q  of tile t-1  (wave c: a 64 x 64 block of each 128 x 128 quarter)
// One barrier per 32-token tile.  Against the symmetric kernel every B fragment feeds twice as many MFMAs (LDS reads per tile: 176 KB, was
// 288 KB), the per-wave loop overheads exist once per role instead of eight times, and a SIMD holds a vector-heavy and a matrix-heavy
// wave.  LDS: LN(x) and g rings 5 x (8 + 8) KB (tile t-1 still feeds the weight gradients while t is in use and t+1 .. t+3 land), H / dZ
// double-buffered 2 x (8 + 8) KB.
// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// DZOUT (round 6): the consumers do NOT form the quarter's dA partial; they copy the quarter's dZ tile out ([M][512] bf16, the same 1,024 B per token as the four dA
// partials) and the launch that follows -- k_dgrad_r<4, ..., MLPFIN> (k_gemm2.hip) -- forms dA = dZ W1 on ITS idle matrix pipe in front of the LayerNorm backward:
// 16 of this kernel's 80 MFMAs per tile, their operand reads, the W1^T registers and the 8-byte scattered partial stores leave the issue-bound kernel
// (knock-outs, profiles/r5_mlp_ab.md: -9 us for the products, -9 us for the stores, of ~110).
template <bool DZOUT>
__global__ __launch_bounds__(S_THR) void k_mlp_bwd_s(const bf16* __restrict__ XN, const bf16* __restrict__ G, const bf16* __restrict__ W1,
                                                     const float* __restrict__ b1, const bf16* __restrict__ W2ts, const bf16* __restrict__ W1t,
                                                     bf16* __restrict__ dApart, bf16* __restrict__ dW1part, bf16* __restrict__ dW2part,
                                                     float* __restrict__ db1, float* __restrict__ db1_rows, int64_t M, int tiles_per_range) {
#ifdef KASF_LSTAMP                                       // un-profiled timeline of the MLP-backward launches (tools/mlp_launch_stamps.py): first and last workgroup, start / end, constant 100 MHz clock
    unsigned ls_slot = 0;
    const bool ls_on = (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && threadIdx.x == 0;
    if (ls_on) {
        ls_slot = atomicAdd(&g_lstamp_n, 4u) & (LSTAMP_CAP - 1);
        g_lstamp[ls_slot] = (long long)(size_t)W1;
        g_lstamp[ls_slot + 1] = (long long)blockIdx.x;
        g_lstamp[ls_slot + 2] = wall_clock64();
    }
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sA = reinterpret_cast<bf16*>(smem);            // [RING][32][128] LN(x) ring
    bf16* sG = sA + BW_RING * TL;                        // [RING][32][128] upstream gradient ring
    bf16* sH = sG + BW_RING * TL;                        // [2][32][128] H of this quarter
    bf16* sD = sH + 2 * TL;                              // [2][32][128] dZ of this quarter
    int q, range;
    {   // the four hidden quarters of one token range sit on one XCD
        const int used = gridDim.x >> 2, full = used & ~7, b = blockIdx.x;
        if (b < 4 * full) { q = (b >> 3) & 3; range = (b & 7) + 8 * (b >> 5); }
        else { q = (b - 4 * full) & 3; range = full + ((b - 4 * full) >> 2); }
    }
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int64_t tile0 = (int64_t)range * tiles_per_range;
    const int64_t ntiles_total = (M + S_BM - 1) / S_BM;
    int64_t ntiles = ntiles_total - tile0;
    if (ntiles > tiles_per_range) ntiles = tiles_per_range;

    if (w < 4) {
        // ------------------------------------------------ producer: hidden units [32w, 32w + 32) of the quarter ------------------------------------------------
        const int h0 = 32 * w;
        // The LDS-direct look-ahead loads are the PRODUCERS' job since round 5: wave w brings rows [8w, 8w + 8) of g and LN(x) of tile t + 3.  Segment timers of the
        // round-4 form (consumers loading) after the producers' GELU went to packed fp16: producer busy 3.6 k cycles per tile, consumer 5.4 k -- 0.9 k of it issuing
        // these four loads and 0.7 k waiting for them -- and the producers 2.5 k at the barrier.  The producers have no other vector-memory traffic, so their
        // counted wait sees loads only; the consumers' queue holds their dA-partial stores only and is never waited on.
        // Everything that does not change from tile to tile is computed once: the two wave-uniform bases of this range (SGPR pairs), the per-lane byte offset
        // inside a tile and the LDS offset of the lane group; per tile a 32-bit tile offset is added (a range is far below 4 GB).
        const int nt32 = (int)ntiles;
        const int64_t rows_here = M - tile0 * S_BM;
        const int rows_in_range = (int)(rows_here < (int64_t)nt32 * S_BM ? rows_here : (int64_t)nt32 * S_BM);
        const void* ugb = uniform_ptr(G + tile0 * S_BM * 128);
        const void* uab = uniform_ptr(XN + tile0 * S_BM * 128);
        const unsigned ldsG = __builtin_amdgcn_readfirstlane(lds_addr(sG)), ldsA = __builtin_amdgcn_readfirstlane(lds_addr(sA));
        auto issue = [&](int t, int slot) {
            const int tt = t < nt32 ? t : 0;             // past the range: harmless re-read that keeps the per-issue load count constant
            int nvalid = rows_in_range - tt * S_BM;
            nvalid = t < nt32 ? (nvalid < S_BM ? nvalid : S_BM) : 1;
            const unsigned tile_off = (unsigned)tt * (S_BM * 256u), slot_off = (unsigned)slot * (TL * 2u);
            const void* ug = (const char*)ugb + tile_off;             // two scalar adds each
            const void* ua = (const char*)uab + tile_off;
            int ln = lane;
            asm volatile("" : "+v"(ln));                 // re-derive the lane's row / chunk here (six vector instructions) instead of holding four more registers across the loop
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int row = (2 * w + k) * 4 + (ln >> 4);
                const int srow = row < nvalid ? row : nvalid - 1;
                const unsigned off = (unsigned)srow * 256u + (unsigned)(((ln & 15) ^ (row & 15)) * 16);
                const unsigned lo = slot_off + (unsigned)(2 * w + k) * 1024u;
                glds16_s(ug, off, ldsG + lo);
                glds16_s(ua, off, ldsA + lo);
            }
        };
#pragma unroll
        for (int k = 0; k < BW_AHEAD; ++k) issue(k, k);  // the first tiles are in flight while the weights arrive from L2
        bf16x8 w1f[2][4], w2f[2][4];
        f32x4 bias4[2], db1acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                w1f[nt][ks] = *reinterpret_cast<const bf16x8*>(W1 + (int64_t)(q * 128 + h0 + 16 * nt + i) * 128 + 32 * ks + 8 * g);
                w2f[nt][ks] = *reinterpret_cast<const bf16x8*>(W2ts + (int64_t)(q * 128 + h0 + 16 * nt + i) * 128 + 32 * ks + 8 * g);
            }
            bias4[nt] = *reinterpret_cast<const f32x4*>(b1 + q * 128 + h0 + 16 * nt + 4 * g);
            db1acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        wait_async();                                    // tiles 0..2 and the weights.  NOT a counted wait: hipcc is free to sink the (const, restrict) weight
                                                         // loads below this point, and a count that assumes them would then let tile 0 through unfinished
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {                 // ... and a use of every one of them HERE, so that hipcc's own wait for them is not placed inside the loop
            touch_loaded(bias4[nt]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { touch_loaded(w1f[nt][ks]); touch_loaded(w2f[nt][ks]); }
        }
        barrier_keep_async();                            // every producer's rows of tile 0 have landed
        TSTART();
        int sp = 0, si = BW_AHEAD;                       // ring slots of tile t and of tile t+AHEAD (mod RING, rolling)
        for (int64_t t = 0; t <= ntiles; ++t, sp = sp == BW_RING - 1 ? 0 : sp + 1, si = si == BW_RING - 1 ? 0 : si + 1) {
#if KASF_BWD_ISSUE_AT == 0
            issue((int)t + BW_AHEAD, si);                // into the slot of tile t-2 (free since the barrier that ended iteration t-1): AHEAD tiles of HBM latency cover
#endif
            TMARK(19);
            if (t < ntiles) {
                const bf16* cA = sA + sp * TL;
                const bf16* cG = sG + sp * TL;
                bf16* cH = sH + (int)(t & 1) * TL;
                bf16* cD = sD + (int)(t & 1) * TL;
                const int64_t row0 = (tile0 + t) * S_BM;
                const int nvalid = (int)((M - row0) < S_BM ? (M - row0) : S_BM);
                // all 16 operand fragments first (64 VGPRs), then the MFMAs of hidden slice 1 are issued BEFORE the GELU of slice 0 so that the matrix pipe
                // works through them while the vector ALU does the GELU
                bf16x8 fa[4][2], fg[4][2];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) { fa[ks][mt] = tok_frag(cA, mt * 16 + i, ks); fg[ks][mt] = tok_frag(cG, mt * 16 + i, ks); }
                f32x4 accZ[2][2], accH[2][2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) { accZ[nt][0] = bias4[nt]; accZ[nt][1] = bias4[nt]; }      // Z accumulates on top of the bias
                zero_acc(accH);
                auto gemm = [&](int nt) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            accZ[nt][mt] = mfma16(w1f[nt][ks], fa[ks][mt], accZ[nt][mt]);
                            accH[nt][mt] = mfma16(w2f[nt][ks], fg[ks][mt], accH[nt][mt]);
                        }
                };
                const bool ragged = nvalid != S_BM;      // rows past M (only the last tile of the last range has any) must not leak GELU(b1) into anything
                auto act = [&](int nt) {                 // Phi and GELU' in packed fp16, the products in fp32 (v_fma_mix_f32), H / dZ rounded to bf16 once
                    f32x2 z[4];
                    f16x2 ph[4], dgh[4];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp)
                            z[2 * mt + hp] = f32x2{accZ[nt][mt][2 * hp], accZ[nt][mt][2 * hp + 1]};
#ifdef KASF_KO_GELU
#pragma unroll
                    for (int k = 0; k < 4; ++k) { ph[k] = __builtin_convertvector(z[k], f16x2); dgh[k] = ph[k]; }
#else
                    gelu_grad_pairs_h(z, ph, dgh);
#endif
                    if (ragged) {                        // rows past M: H = dZ = 0 (wave-uniform branch; only the last tile of the last range is ragged)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            const f16 lv = (mt * 16 + i < nvalid) ? (f16)1.0f : (f16)0.0f;
#pragma unroll
                            for (int hp = 0; hp < 2; ++hp) { ph[2 * mt + hp] *= f16x2{lv, lv}; dgh[2 * mt + hp] *= f16x2{lv, lv}; }
                        }
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        float h[4], dz[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const f16 p1 = ph[2 * mt + (r >> 1)][r & 1], d1 = dgh[2 * mt + (r >> 1)][r & 1];
                            h[r] = __builtin_fmaf((float)p1, accZ[nt][mt][r], 0.0f);
                            dz[r] = __builtin_fmaf((float)d1, accH[nt][mt][r], 0.0f);
                            db1acc[nt][r] = __builtin_fmaf((float)d1, accH[nt][mt][r], db1acc[nt][r]);
                        }
                        store4(cH + Tile<bf16>::off4(mt * 16 + i, h0 + 16 * nt + 4 * g), h);
                        store4(cD + Tile<bf16>::off4(mt * 16 + i, h0 + 16 * nt + 4 * g), dz);
                    }
                };
                gemm(0);
                __builtin_amdgcn_sched_barrier(0);
#if KASF_BWD_ISSUE_AT == 1
                issue((int)t + BW_AHEAD, si);
                __builtin_amdgcn_sched_barrier(0);
#endif
                TMARK(16);
                gemm(1);
                act(0);
#pragma unroll
                for (int k = 0; k < 16; ++k) {           // one MFMA of slice 1 to five vector instructions of slice 0
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#if KASF_BWD_ISSUE_AT == 2
                issue((int)t + BW_AHEAD, si);
                __builtin_amdgcn_sched_barrier(0);
#endif
                act(1);
                TMARK(17);
            }
#if KASF_BWD_ISSUE_AT == 1 || KASF_BWD_ISSUE_AT == 2
            else issue((int)t + BW_AHEAD, si);           // (the iteration past the last tile keeps the per-iteration load count)
#endif
            wait_async_le<4 * (BW_AHEAD - 1)>();         // tile t+1 has landed: only the requests of the BW_AHEAD - 1 tiles behind it (4 loads each, this wave's only vector-memory traffic) may be in flight
            TMARK(21);
            barrier_keep_async();
            TMARK(18);
        }
        wait_async();                                    // drain the look-ahead requests before the wave goes on to its epilogue
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = db1acc[nt][r];                 // sum over the 16 token lanes that share (g, r)
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                if (i == 0) {                            // one row of db1 per token range (added in a fixed order by k_col_finish), or an atomic without scratch
                    if (db1_rows != nullptr) db1_rows[(int64_t)range * 512 + q * 128 + h0 + 16 * nt + 4 * g + r] = v;
                    else atomicAdd(db1 + q * 128 + h0 + 16 * nt + 4 * g + r, v);
                }
            }
    } else {
        // ------------------------------------------------ consumer: channels [32c, 32c + 32) of dA, a 64 x 64 block of each weight-gradient quarter ------------------------------------------------
        const int c = w - 4, ch0 = 32 * c, tr0 = 64 * (c >> 1), tc0 = 64 * (c & 1);
#ifdef KASF_BWD_CONS_PRIO
        __builtin_amdgcn_s_setprio(KASF_BWD_CONS_PRIO);
#endif
        bf16x8 wtf[2][4];
        f32x4 accW1[4][4], accW2[4][4];
        zero_acc(accW1);
        zero_acc(accW2);
        if (!DZOUT) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wtf[nt][ks] = *reinterpret_cast<const bf16x8*>(W1t + (int64_t)(ch0 + 16 * nt + i) * 512 + q * 128 + 32 * ks + 8 * g);
        }
        barrier_keep_async();                            // (the producers' rows of tile 0 have landed)
        TSTART();
        int sc = BW_RING - 1;                            // ring slot of tile t-1 (consumed), rolling mod RING
        for (int64_t t = 0; t <= ntiles; ++t, sc = sc == BW_RING - 1 ? 0 : sc + 1) {
            TMARK(24);
            f32x4 accA[2][2];
            auto store_dA = [&]() {
                const int64_t row0 = (tile0 + t - 1) * S_BM;
                if (DZOUT) {   // the quarter's dZ tile of tile t - 1, whole 256-byte row pieces: consumer thread (rl, sub) copies chunk `sub` of rows rl and rl + 16
                    const bf16* cDz = sD + (int)((t - 1) & 1) * TL;
                    const int ct = c * 64 + lane, rlz = ct >> 4, subz = ct & 15;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int64_t row = row0 + rlz + 16 * k;
                        if (row < M)
                            *reinterpret_cast<f32x4*>(dApart + row * 512 + q * 128 + subz * 8) = *reinterpret_cast<const f32x4*>(cDz + Tile<bf16>::chunk_off(rlz + 16 * k, subz));
                    }
                    return;
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int64_t row = row0 + mt * 16 + i;
                    if (row < M) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const float v[4] = {accA[nt][mt][0], accA[nt][mt][1], accA[nt][mt][2], accA[nt][mt][3]};
                            store4(dApart + ((int64_t)q * M + row) * 128 + ch0 + 16 * nt + 4 * g, v);
                        }
                    }
                }
            };
            if (t >= 1) {
                const bf16* cA = sA + sc * TL;
                const bf16* cG = sG + sc * TL;
                const bf16* cH = sH + (int)((t - 1) & 1) * TL;
                const bf16* cD = sD + (int)((t - 1) & 1) * TL;
                zero_acc(accA);
#ifndef KASF_KO_DA
                if (!DZOUT) {   // ---- dA_q partial: 32 channels x 32 tokens over the 128 hidden units of the quarter ----
                    bf16x8 fd[2][2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) fd[0][mt] = tok_frag(cD, mt * 16 + i, 0);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        if (ks + 1 < 4) {
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt) fd[(ks + 1) & 1][mt] = tok_frag(cD, mt * 16 + i, ks + 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) accA[nt][mt] = mfma16(wtf[nt][ks], fd[ks & 1][mt], accA[nt][mt]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#endif
                TMARK(25);
#if KASF_BWD_DASTORE_EARLY                  // the dA partial rows leave BEFORE the weight-gradient MFMAs: their address arithmetic and the four stores issue under the matrix pipe's time
                store_dA();
                __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef KASF_KO_WGRAD
                {   // ---- weight gradients: reduction over the 32 tokens of the tile (one k-step) ----
                    bf16x8 ra[4], cb[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) ra[a] = frag_tr(cD, 8 * g, tr0 + 16 * a);
#pragma unroll
                    for (int b = 0; b < 4; ++b) cb[b] = frag_tr(cA, 8 * g, tc0 + 16 * b);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) accW1[a][b] = mfma16(ra[a], cb[b], accW1[a][b]);             // dW1[hq][k] += dZ^T LN(x)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < 4; ++a) ra[a] = frag_tr(cG, 8 * g, tr0 + 16 * a);
#pragma unroll
                    for (int b = 0; b < 4; ++b) cb[b] = frag_tr(cH, 8 * g, tc0 + 16 * b);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) accW2[a][b] = mfma16(ra[a], cb[b], accW2[a][b]);             // dW2[c][hq] += g^T H
                }
#endif
            }
            TMARK(26);
#if !KASF_BWD_DASTORE_EARLY
#ifdef KASF_KO_DASTORE
            if (t >= 1 && accA[0][0][0] == 123.456f) store_dA();
#else
            if (t >= 1) store_dA();
#endif
#endif
            TMARK(28);
            barrier_keep_async();
            TMARK(29);
        }
        // the quarter's weight-gradient tiles leave as bf16 (round 4: 64 ranges x 0.2 % rounding noise average out two orders below the bf16 operands' own;
        // half the partial bytes of the step's 156 MLP blocks): accumulators -> two [128][128] images in the (now dead) LDS rings -> whole rows out below
        __builtin_amdgcn_s_barrier();                    // (with the producers, which are past their last tile: the rings are free)
        bf16* sW1 = reinterpret_cast<bf16*>(smem);
        bf16* sW2 = sW1 + 128 * 128;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = tr0 + 16 * a + 4 * g + r, cc = tc0 + 16 * b + i;
                    sW1[rr * 128 + cc] = (bf16)accW1[a][b][r];
                    sW2[rr * 128 + cc] = (bf16)accW2[a][b][r];
                }
    }
    if (w < 4) __builtin_amdgcn_s_barrier();             // pairs with the consumers' barrier above
    __syncthreads();
    {
        const bf16* sW1 = reinterpret_cast<const bf16*>(smem);
        const bf16* sW2 = sW1 + 128 * 128;
        bf16* p1 = dW1part + (int64_t)range * 512 * 128 + (int64_t)q * 128 * 128;      // rows [128 q, +128) of [512][128]
        bf16* p2 = dW2part + (int64_t)range * 128 * 512 + q * 128;                      // columns [128 q, +128) of [128][512]
        for (int c = threadIdx.x; c < 128 * 16; c += S_THR) {
            const int rr = c >> 4, ch = c & 15;
            *reinterpret_cast<f32x4*>(p1 + rr * 128 + ch * 8) = *reinterpret_cast<const f32x4*>(sW1 + rr * 128 + ch * 8);
            *reinterpret_cast<f32x4*>(p2 + (int64_t)rr * 512 + ch * 8) = *reinterpret_cast<const f32x4*>(sW2 + rr * 128 + ch * 8);
        }
    }
#ifdef KASF_LSTAMP
    if (ls_on) g_lstamp[ls_slot + 3] = wall_clock64();
#endif
}

}  // namespace

void kasf_launch_mlp_fwd_s(hipStream_t s, const void* x, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2,
                           const float* b2, const float* ls2, void* out, int64_t M, void* xn_out, unsigned grid) {
    const size_t sh = (size_t)(14 * TL) * sizeof(bf16) + 1024 * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_fwd_s), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(k_mlp_fwd_s, dim3(grid), dim3(S_THR), sh, s, (const bf16*)x, ln_g, ln_b, (const bf16*)W1, b1, (const bf16*)W2, b2, ls2, (bf16*)out, M,
                       (bf16*)xn_out);
}

#ifdef KASF_PROBE_TIMERS
extern "C" void kasf_debug_read_prof(long long* dst, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_prof), sizeof(long long) * 32);
    if (reset) { long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
}
#endif

#ifdef KASF_LSTAMP
extern "C" int kasf_debug_read_lstamps(long long* dst, int reset) {      // dst: LSTAMP_CAP entries; returns the number of entries written since the last reset
    (void)hipDeviceSynchronize();
    unsigned n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_lstamp_n), sizeof(n));
    (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lstamp), sizeof(long long) * LSTAMP_CAP);
    if (reset) { const unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lstamp_n), &z, sizeof(z)); }
    return (int)n;
}
#endif

void kasf_launch_mlp_bwd_s(hipStream_t s, const void* xn, const void* g, const void* W1, const float* b1, const void* W2ts, const void* W1t, void* dApart,
                           void* p1, void* p2, float* db1, float* db1_rows, int64_t M, int tiles_per_range, int used, bool dz_out) {
    const size_t sh = (size_t)((2 * BW_RING + 4) * TL) * sizeof(bf16);
    if (dz_out) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_bwd_s<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipLaunchKernelGGL(k_mlp_bwd_s<true>, dim3(4 * used), dim3(S_THR), sh, s, (const bf16*)xn, (const bf16*)g, (const bf16*)W1, b1, (const bf16*)W2ts,
                           (const bf16*)W1t, (bf16*)dApart, (bf16*)p1, (bf16*)p2, db1, db1_rows, M, tiles_per_range);
        return;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_bwd_s<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(k_mlp_bwd_s<false>, dim3(4 * used), dim3(S_THR), sh, s, (const bf16*)xn, (const bf16*)g, (const bf16*)W1, b1, (const bf16*)W2ts,
                       (const bf16*)W1t, (bf16*)dApart, (bf16*)p1, (bf16*)p2, db1, db1_rows, M, tiles_per_range);
}
