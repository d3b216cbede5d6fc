// MLP backward, bf16 mode: the launches around k_mlp_bwd_s (k_mlp3.hip, which owns the hidden-quarter algorithm).
//
//   k_mlp_bwd_s        per (token range, hidden quarter): Z, dH, H, dZ in LDS, the quarter's dA partial (bf16) to memory, dW1 / dW2 accumulated in
//                      registers over the range and left as per-range partial tiles, db1 as one row per range
//   k_lnbwd_sum4_fin   workgroups 0..255: fixed-order sum of the weight-gradient partial tiles + the fc2 layer-scale algebra (W2 . G term);
//                      the other workgroups stream the four dA partials: sum, LayerNorm backward + residual -> g_in, and per-workgroup rows of
//                      dgamma | dbeta | colsum(g)
//   k_col_finish       (k_reduce.hip, once per backward stage) adds those rows in a fixed order and applies the colsum(g) terms of fc2
//                      (db2 = ls2 . colsum, dls2 += b2 . colsum)
// Every reduction across workgroups is "store a row, add the rows later in a fixed order": the gradients are bit-reproducible and no kernel
// synchronises with another workgroup (round 2 ended this chain with fp32 atomics and a last-workgroup ticket).
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"
#include "mlp_fin.h"

namespace {

constexpr int Q_BM = 32;          // tokens per tile of k_mlp_bwd_s

// dApart [4][M][128] (one slab per hidden quarter) -> g_in = g + LNbwd(sum of the slabs; x, gamma).  Per workgroup one row of
// dgamma | dbeta | colsum(g): stored to `part` (row `bid`, 384 floats) or, without scratch, added atomically.
template <int RPT>
__device__ __forceinline__ void lnbwd_sum4_body(const bf16* __restrict__ dApart, const bf16* __restrict__ X, const bf16* __restrict__ G,
                                                const float* __restrict__ gamma, bf16* __restrict__ g_in, float* __restrict__ dgamma,
                                                float* __restrict__ dbeta, float* __restrict__ gsum, float* __restrict__ part, int64_t M, int bid, int nblk) {
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
        if (part != nullptr) part[(int64_t)bid * 384 + item] = s;
        else atomicAdd((which == 0 ? dgamma : (which == 1 ? dbeta : gsum)) + c, s);
    }
}
template <int RPT>
__global__ __launch_bounds__(256) void k_lnbwd_sum4_fin(const bf16* __restrict__ dApart, const bf16* __restrict__ X, const bf16* __restrict__ G,
                                                        const float* __restrict__ gamma, bf16* __restrict__ g_in, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, float* __restrict__ gsum, float* __restrict__ part, int64_t M,
                                                        const MlpFinArgs fa) {
    if (blockIdx.x < 256) {
        __shared__ f32x4 sHalf[128];
        __shared__ float sDot[2];
        mlp_wfinish_body(fa, (int)blockIdx.x, (int)threadIdx.x, sHalf, sDot);
        return;
    }
    lnbwd_sum4_body<RPT>(dApart, X, G, gamma, g_in, dgamma, dbeta, gsum, part, M, (int)blockIdx.x - 256, (int)gridDim.x - 256);
}

}  // namespace

// scratch needs: dApart = 4*M*128 bf16;  partial >= 2 * ranges * 65536 floats
int kasf_mlp_bwd_q_ranges(int64_t M) {
    const int64_t tiles = (M + Q_BM - 1) / Q_BM;
    return (int)(tiles < 64 ? tiles : 64);
}
void kasf_launch_mlp_bwd_q(hipStream_t s, const void* x, const void* xn, const void* g, const float* ln_g, const void* W1, const float* b1,
                           const void* W2ts, const void* W1t, void* dApart, float* partial, float* dW1, float* dW2, float* db1, float* gsum, void* g_in,
                           float* dgamma, float* dbeta, int64_t M, const float* W2, const float* b2, const float* ls2, float* dls2, KasfColSink* sink) {
    int ranges = kasf_mlp_bwd_q_ranges(M);
    const int64_t tiles = (M + Q_BM - 1) / Q_BM;
    { const int nr = kasf_narrow_grid(KASF_NG_MLP_BWD, 64, M); if (ranges > nr) ranges = nr; }
    const int tpr = (int)((tiles + ranges - 1) / ranges);
    const int used = (int)((tiles + tpr - 1) / tpr);             // ranges that own at least one tile
    bf16* p1 = reinterpret_cast<bf16*>(partial);          // (the scratch is sized in floats for the round-3 fp32 tiles: half of it is used)
    bf16* p2 = p1 + (int64_t)used * 512 * 128;
    float* db1_rows = sink != nullptr ? sink->take(used, 512) : nullptr;          // one row of db1 per token range
    // Round 6, opt-in (KASF_MLP_BWD_DZ=1): the dZ form -- k_mlp_bwd_s<DZOUT> leaves dZ [M][512] where the four dA partials went, and the second launch
    // (k_dgrad_r<4, ..., MLPFIN>, k_gemm2.hip) forms dA = dZ W1 on its own matrix pipe in front of the LayerNorm backward.  Same bytes, one GEMM moved out of the
    // issue-bound kernel into the HBM-bound one: k_mlp_bwd_s -10 us, the second launch +9..+28 us, step +0.4 % (inside box noise) -- so the partial-sum form of
    // rounds 2-5 (k_lnbwd_sum4_fin) stays the default.  profiles/r6_mlp_bwd_dz_form_ab.md
    static const bool dz_form = [] { const char* e = getenv("KASF_MLP_BWD_DZ"); return e != nullptr && *e == '1'; }();
    kasf_launch_mlp_bwd_s(s, xn, g, W1, b1, W2ts, W1t, dApart, p1, p2, db1, db1_rows, M, tpr, used, dz_form);
    if (db1_rows != nullptr) sink->add(db1_rows, 512, used, 512, db1);
    if (dz_form) {
        const MlpFinArgs fa{p1, p2, dW1, dW2, used, W2, b2, ls2, dls2};
        kasf_launch_mlp_dgrad_fin(s, dApart, W1t, x, ln_g, g, g_in, dgamma, dbeta, gsum, M, sink, &fa, b2, ls2, dls2, W2 != nullptr);
        return;
    }
    int64_t blocks = (M + 31) / 32;                     // two rows of 16 lanes per thread
    if (blocks > 512) blocks = 512;
    float* rows = sink != nullptr ? sink->take((int)blocks, 384) : nullptr;       // dgamma | dbeta | colsum(g) per streaming workgroup
    const MlpFinArgs fa{p1, p2, dW1, dW2, used, W2, b2, ls2, dls2};
    hipLaunchKernelGGL(k_lnbwd_sum4_fin<2>, dim3((unsigned)blocks + 256), dim3(256), 0, s, (const bf16*)dApart, (const bf16*)x, (const bf16*)g, ln_g,
                       (bf16*)g_in, dgamma, dbeta, gsum, rows, M, fa);
    if (rows != nullptr) {
        sink->add(rows, 384, (int)blocks, 128, dgamma);
        sink->add(rows + 128, 384, (int)blocks, 128, dbeta);
        if (W2 != nullptr) sink->add(rows + 256, 384, (int)blocks, 128, gsum, 1, b2, ls2, dls2);      // gsum is the fc2 bias gradient slot: db2 = ls2 . colsum(g)
        else sink->add(rows + 256, 384, (int)blocks, 128, gsum);
    }
}
