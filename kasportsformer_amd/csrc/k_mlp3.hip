// MLP forward, third generation (bf16): the waves of a workgroup are SPECIALISED.
//
// k_mlp_fwd_r (k_mlp2.hip) gives every wave the same program: GEMM1 (32 MFMAs), GELU (~370 vector instructions), barrier, GEMM2 (32 MFMAs),
// epilogue.  The two waves that share a SIMD belong to the same workgroup and move in lockstep between the barriers, so the matrix pipe idles
// during GELU and the vector ALU idles during the GEMMs: measured 18.6 % MFMA busy, 42 % VALU busy, and a tile time equal to the SUM of the
// phases plus the stalls (SQ counters in profiles/r1_mlp_sq_counters.md).  Software-pipelining the symmetric program did not help.
//
// Here waves 0-3 (one per SIMD) are PRODUCERS: wave p keeps W1 rows [128p, 128p+128) in 128 VGPRs and does GEMM1 + GELU for its quarter of the
// hidden units.  Waves 4-7 (again one per SIMD) are CONSUMERS: wave c keeps W2 rows [32c, 32c+32) in 128 VGPRs and does GEMM2 of the PREVIOUS
// tile, the residual epilogue, the LDS-direct loads and the LayerNorm of the NEXT tile.  A SIMD therefore always holds one vector-heavy and one
// matrix-heavy wave with no common phase, B fragments are shared by twice as many A fragments (LDS reads per MFMA halve), and there is ONE
// barrier per tile:
//     iteration t:   producers  GEMM1(t) + GELU(t) -> sH[t & 1]        consumers  GEMM2(t-1) <- sH[(t-1) & 1], epilogue(t-1), LN(t+1) -> sA[(t+1) & 1]
// Packed fp32 instructions (v_pk_fma_f32 ...) do not overlap with MFMAs of another wave on gfx950 (tools/valu_probe.hip: an MFMA wave and a
// v_pk_fma_f32 wave on one SIMD take the sum of their times, an MFMA wave and a v_fma_f32 wave the maximum), so this file is compiled with
// -fno-slp-vectorize.
// LDS: sA 2 x 8 KB, sH 2 x 32 KB, raw x ring 4 x 8 KB (tiles t-1, t, t+1 in use, t+2 in flight), parameters 4 KB.
#include "common.h"
#include "kernels.h"
#include <type_traits>

#ifdef KASF_PROBE_TIMERS
__device__ long long g_prof[32];
#define TMARK(k) do { const long long _n = clock64(); if (lane == 0 && blockIdx.x == 7) g_prof[k] += _n - _t; _t = _n; } while (0)
#define TSTART() long long _t = clock64()
#else
#define TMARK(k) do {} while (0)
#define TSTART() do {} while (0)
#endif

namespace {

constexpr int S_BM = 32, S_THR = 512, TL = S_BM * 128;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
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
                auto gemm1 = [&](f32x4 (&a)[2], int nt) {
                    a[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    a[1] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                    const f32x4 bv = b1v[nt];
                    f32x2 y[4];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        y[2 * mt] = f32x2{acc[nt & 1][mt][0] + bv[0], acc[nt & 1][mt][1] + bv[1]};
                        y[2 * mt + 1] = f32x2{acc[nt & 1][mt][2] + bv[2], acc[nt & 1][mt][3] + bv[3]};
                    }
                    gelu_pairs_fast(y);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const float h[4] = {y[2 * mt][0], y[2 * mt][1], y[2 * mt + 1][0], y[2 * mt + 1][1]};
                        store4(hT + Tile<bf16>::off4(mt * 16 + i, 16 * nt + 4 * g), h);
                    }
                    if (nt < 7) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
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
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) w2f[n2][ks] = *reinterpret_cast<const bf16x8*>(W2 + (int64_t)(32 * c + 16 * n2 + i) * 512 + 32 * ks + 8 * g);

        auto issue = [&](int64_t t) {                    // two LDS-direct loads per consumer wave: rows [8c, 8c + 8) of tile t, the rows it will normalise
            const int64_t tt = t < ntiles ? t : ntiles - 1;
            const int64_t row0 = (tile0 + tt) * S_BM;
            const int nvalid = (int)((M - row0) < S_BM ? (M - row0) : S_BM);
            const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(sXr + (int)(t & 3) * TL));
            const void* ub = uniform_ptr(X + row0 * 128);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int inst = 2 * c + k, row = inst * 4 + (lane >> 4), pc = lane & 15, ch = pc ^ (row & 15);
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
                for (int e = 0; e < 8; ++e) { v[e] -= mean; qv += v[e] * v[e]; }
                const float rstd = rsqrtf(reduce16(qv) * (1.0f / 128.0f) + KASF_LN_EPS);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = v[e] * rstd * g0[e] + c0[e]; v[4 + e] = v[4 + e] * rstd * g1[e] + c1[e]; }
                tile_store8(sA + (int)(t & 1) * TL, r, sub * 8, v);
                if (xn_out != nullptr) {                 // training: the backward pass streams LN(x) instead of recomputing it
                    const int64_t row = (tile0 + t) * S_BM + r;
                    if (row < M) store8(xn_out + row * 128 + sub * 8, v);
                }
            }
        };
        issue(0);
        issue(1);
        wait_async_le<2>();
        layernorm(0);
        barrier_keep_async();
        TSTART();
        for (int64_t t = 0; t <= ntiles; ++t) {
            issue(t + 2);
            TMARK(8);                                // into the slot of x(t-2)
            f32x4 acc2[2][2];
            if (t >= 1) {                                // GEMM2 of tile t-1 over all 512 hidden units
                const bf16* cH = sH + (int)((t - 1) & 1) * 4 * TL;
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) { acc2[n2][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[n2][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
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
                        acc2[n2][0] = mfma16(w2f[n2][ks], fh[ks % 3][0], acc2[n2][0]);
                        acc2[n2][1] = mfma16(w2f[n2][ks], fh[ks % 3][1], acc2[n2][1]);
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
                    const f32x4 b2v = *reinterpret_cast<const f32x4*>(sPar + 512 + col), lsv = *reinterpret_cast<const f32x4*>(sPar + 640 + col);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int64_t row = row0 + mt * 16 + i;
                        if (row < M) {
                            float x[4], y[4];
                            load4(cX + Tile<bf16>::off4(mt * 16 + i, col), x);      // residual from the raw tile still in LDS
#pragma unroll
                            for (int r = 0; r < 4; ++r) y[r] = x[r] + lsv[r] * (acc2[n2][mt][r] + b2v[r]);
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
