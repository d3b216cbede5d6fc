// Fused MLP half of a FormerModule (reference: model/KASportsFormer.py:111, modules/mlp.py:24-30):
//     out = x + ls2 * ( GELU( LN(x) W1^T + b1 ) W2^T + b2 )
// ~69 % of the model's FLOPs.  One workgroup owns BM tokens; the [BM x 512] hidden activation never
// leaves the CU: it is produced 128 columns at a time into LDS and consumed immediately by the second
// GEMM.  W1 / W2 blocks (128 x 128) stream from L2 through one LDS buffer.
//
// Backward recomputes Z = LN(x) W1^T + b1 (no activation stash in HBM during forward), produces
// dZ and H = GELU(Z) for the two weight-gradient GEMMs, chains dA = dZ . W1 in registers and ends
// with the LayerNorm backward + residual:   g_in = g + LNbwd(dA).
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"

namespace {

// Weight blocks are consumed in a fixed order (8 per token tile forward, 12 backward).  Block b+1 is
// requested (LDS-direct, asynchronous) right after the barrier that opens block b, into the other
// tile of the ring, and lands while block b is being multiplied: one barrier per block.  With NBUF == 1
// (fp32 parity mode: tiles are twice as large) the request is issued after the block's MFMAs instead.
template <typename T, int NBUF, int NTHR, typename NextFn>
struct WeightRing {
    T* base;
    NextFn next;            // next(b) -> (pointer, ld) of weight block b
    int nblocks;
    __device__ __forceinline__ T* tile(int b) const { return base + (b % NBUF) * 128 * 128; }
    __device__ __forceinline__ void request(int b) const {
        if (b < nblocks) {
            int64_t ld;
            const T* p = next(b, ld);
            stage_tile_async<T, 128, NTHR>(tile(b), p, ld, 128);
        }
    }
    // call at the top of block b (after everything written for it is in flight); returns the tile to multiply
    __device__ __forceinline__ const T* open(int b) const {
        wait_async();
        __syncthreads();
        if (NBUF == 2) request(b + 1);
        return tile(b);
    }
    // call after the MFMAs of block b
    __device__ __forceinline__ void close(int b) const {
        if (NBUF == 1 && b + 1 < nblocks) {
            __syncthreads();
            request(b + 1);
        }
    }
};

template <typename T, int BM, int NBUF, int NTHR>
__global__ __launch_bounds__(NTHR) void k_mlp_fwd(const T* __restrict__ X, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                 const T* __restrict__ W1, const float* __restrict__ b1, const T* __restrict__ W2,
                                                 const float* __restrict__ b2, const float* __restrict__ ls2, T* __restrict__ out, int64_t M) {
    constexpr int MT = BM / (NTHR / 128) / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);        // LN(x)          [BM][128]
    T* sH = sA + BM * 128;                     // GELU chunk     [BM][128]
    T* sW = sH + BM * 128;                     // weight ring    [NBUF][128][128]
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int wn0 = wave_n0(), wm0 = wave_m0_n<BM, NTHR>();
    // block 2*hc: W1 rows [128*hc, +128) (k = channels); block 2*hc+1: W2[:, 128*hc .. +128) (k = hidden chunk)
    auto nextw = [&](int b, int64_t& ld) -> const T* {
        const int hc = b >> 1;
        if (b & 1) { ld = 512; return W2 + hc * 128; }
        ld = 128;
        return W1 + (int64_t)hc * 128 * 128;
    };
    WeightRing<T, NBUF, NTHR, decltype(nextw)> ring{sW, nextw, 8};
    ring.request(0);
    stage_rows<T, BM, true, NTHR>(sA, X, 128, row0, M, ln_g, ln_b, nullptr);
    f32x4 acc2[4][MT];
    zero_acc(acc2);
    for (int hc = 0; hc < 4; ++hc) {
        const T* w1 = ring.open(2 * hc);
        f32x4 acc1[4][MT];
        zero_acc(acc1);
        mma_k128<4, MT>(w1, wn0, sA, wm0, acc1);
        acc_to_tile<T>(sH, acc1, wn0, wm0, [&](float v, int n) { return gelu_f<T>(v + b1[hc * 128 + n]); });
        ring.close(2 * hc);
        const T* w2 = ring.open(2 * hc + 1);            // barrier: sH complete, GEMM1 done with its tile
        mma_k128<4, MT>(w2, wn0, sH, wm0, acc2);
        ring.close(2 * hc + 1);
    }
    acc_foreach<4, MT>(acc2, wn0, wm0, row0, M, [&](float (&v)[4], int64_t row, int n) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + n), l = *reinterpret_cast<const f32x4*>(ls2 + n);
        float x[4];
        load4(X + row * 128 + n, x);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x[r] + l[r] * (v[r] + b[r]);
        store4(out + row * 128 + n, v);
    });
}

template <typename T, int BM, int NBUF, int NTHR>
__global__ __launch_bounds__(NTHR) void k_mlp_bwd(const T* __restrict__ X, const T* __restrict__ G, const float* __restrict__ ln_g,
                                                 const float* __restrict__ ln_b, const T* __restrict__ W1, const float* __restrict__ b1,
                                                 const T* __restrict__ W2ts, const T* __restrict__ W1t, T* __restrict__ Hbuf,
                                                 T* __restrict__ dZbuf, T* __restrict__ xn_buf, T* __restrict__ g_in,
                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t M, float* __restrict__ part) {
    constexpr int MT = BM / (NTHR / 128) / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);        // LN(x)        [BM][128]
    T* sG = sA + BM * 128;                     // upstream g   [BM][128]
    T* sD = sG + BM * 128;                     // dZ chunk     [BM][128]
    T* sHs = sD + BM * 128;                    // H chunk (copy-out staging) [BM][128]
    T* sW = sHs + BM * 128;                    // weight ring  [NBUF][128][128]
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int wn0 = wave_n0(), wm0 = wave_m0_n<BM, NTHR>();
    const int nvalid = (int)((M - row0) < BM ? (M - row0) : BM);
    // per hidden chunk hc: W1 rows (Z), (ls2.W2)^T rows (dH), W1^T columns (dA)
    auto nextw = [&](int b, int64_t& ld) -> const T* {
        const int hc = b / 3, which = b % 3;
        if (which == 2) { ld = 512; return W1t + hc * 128; }
        ld = 128;
        return (which == 0 ? W1 : W2ts) + (int64_t)hc * 128 * 128;
    };
    WeightRing<T, NBUF, NTHR, decltype(nextw)> ring{sW, nextw, 12};
    ring.request(0);
    stage_tile_async<T, BM, NTHR>(sG, G + row0 * 128, 128, nvalid);
    stage_rows<T, BM, true, NTHR>(sA, X, 128, row0, M, ln_g, ln_b, xn_buf);      // LN(x) is also the fc1 weight-gradient operand
    f32x4 accA[4][MT];
    zero_acc(accA);
    for (int hc = 0; hc < 4; ++hc) {
        f32x4 accZ[4][MT], accH[4][MT];
        zero_acc(accZ);
        zero_acc(accH);
        const T* w = ring.open(3 * hc);
        if (hc > 0) {                                   // copy out the previous chunk's H / dZ while this chunk starts
            for (int idx = threadIdx.x; idx < BM * 16; idx += NTHR) {
                const int r = idx >> 4, sub = idx & 15;
                if (row0 + r < M) {
                    float v[8];
                    tile_load8(sHs, r, sub * 8, v);
                    store8(Hbuf + (row0 + r) * 512 + (hc - 1) * 128 + sub * 8, v);
                    tile_load8(sD, r, sub * 8, v);
                    store8(dZbuf + (row0 + r) * 512 + (hc - 1) * 128 + sub * 8, v);
                }
            }
        }
        mma_k128<4, MT>(w, wn0, sA, wm0, accZ);                         // Z^T[h][m]
        ring.close(3 * hc);
        w = ring.open(3 * hc + 1);                      // barrier: previous chunk's sHs/sD fully copied out and dA-multiplied
        mma_k128<4, MT>(w, wn0, sG, wm0, accH);                         // dH^T[h][m] = sum_c ls2[c] W2[c][h] g[m][c]
        {
            const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int n = wn0 + nt * 16 + g * 4, m = wm0 + mt * 16 + i;
                    float h[4], dz[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = accZ[nt][mt][r] + b1[hc * 128 + n + r];
                        float dg;
                        gelu_and_grad<T>(z, h[r], dg);
                        dz[r] = accH[nt][mt][r] * dg;
                    }
                    store4(sHs + Tile<T>::off4(m, n), h);
                    store4(sD + Tile<T>::off4(m, n), dz);
                }
        }
        ring.close(3 * hc + 1);
        w = ring.open(3 * hc + 2);                      // barrier: sD / sHs complete
        mma_k128<4, MT>(w, wn0, sD, wm0, accA);                         // dA^T[k][m] += sum_h W1[h][k] dZ[m][h]
        ring.close(3 * hc + 2);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < BM * 16; idx += NTHR) {           // last chunk's H / dZ
        const int r = idx >> 4, sub = idx & 15;
        if (row0 + r < M) {
            float v[8];
            tile_load8(sHs, r, sub * 8, v);
            store8(Hbuf + (row0 + r) * 512 + 3 * 128 + sub * 8, v);
            tile_load8(sD, r, sub * 8, v);
            store8(dZbuf + (row0 + r) * 512 + 3 * 128 + sub * 8, v);
        }
    }
    __syncthreads();
    acc_to_tile<T>(sD, accA, wn0, wm0, [](float v, int) { return v; });
    __syncthreads();
    lnbwd_rows<T, BM, NTHR>(sD, X, ln_g, (const T*)nullptr, G, g_in, 0, dgamma, dbeta, row0, M, reinterpret_cast<float*>(sW), (T*)nullptr, nullptr, part);
}

template <typename T> struct MlpCfg;
// bf16: 8 waves per workgroup (2 per SIMD) so one wave's VALU epilogue (GELU, LayerNorm) overlaps the other's MFMAs
template <> struct MlpCfg<bf16> { static constexpr int BM_F = 128, BM_B = 64, NBUF = 2, NTHR = 512; };
template <> struct MlpCfg<float> { static constexpr int BM_F = 64, BM_B = 32, NBUF = 1, NTHR = 256; };

template <typename K> void set_smem(K k, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T>
void mlp_fwd_T(hipStream_t s, const void* x, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2, const float* b2,
               const float* ls2, void* out, int64_t M) {
    constexpr int BM = MlpCfg<T>::BM_F, NBUF = MlpCfg<T>::NBUF, NTHR = MlpCfg<T>::NTHR;
    const size_t sh = (size_t)(2 * BM * 128 + NBUF * 128 * 128) * sizeof(T);
    set_smem(k_mlp_fwd<T, BM, NBUF, NTHR>, sh);
    hipLaunchKernelGGL((k_mlp_fwd<T, BM, NBUF, NTHR>), dim3((unsigned)((M + BM - 1) / BM)), dim3(NTHR), sh, s, (const T*)x, ln_g, ln_b, (const T*)W1, b1,
                       (const T*)W2, b2, ls2, (T*)out, M);
}
template <typename T>
void mlp_bwd_T(hipStream_t s, const void* x, const void* g, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2ts,
               const void* W1t, void* Hbuf, void* dZbuf, void* xn_buf, void* g_in, float* dgamma, float* dbeta, int64_t M, KasfColSink* sink) {
    constexpr int BM = MlpCfg<T>::BM_B, NBUF = MlpCfg<T>::NBUF, NTHR = MlpCfg<T>::NTHR;
    const size_t sh = (size_t)(4 * BM * 128 + NBUF * 128 * 128) * sizeof(T);
    set_smem(k_mlp_bwd<T, BM, NBUF, NTHR>, sh);
    const int grid = (int)((M + BM - 1) / BM);
    float* part = sink != nullptr ? sink->take(grid, 256) : nullptr;          // one [dgamma | dbeta] row per workgroup
    hipLaunchKernelGGL((k_mlp_bwd<T, BM, NBUF, NTHR>), dim3((unsigned)grid), dim3(NTHR), sh, s, (const T*)x, (const T*)g, ln_g, ln_b,
                       (const T*)W1, b1, (const T*)W2ts, (const T*)W1t, (T*)Hbuf, (T*)dZbuf, (T*)xn_buf, (T*)g_in, dgamma, dbeta, M, part);
    if (part != nullptr) { sink->add(part, 256, grid, 128, dgamma); sink->add(part + 128, 256, grid, 128, dbeta); }
}

}  // namespace

void kasf_launch_mlp_fwd(int dt, hipStream_t s, const void* x, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2,
                         const float* b2, const float* ls2, void* out, int64_t M, void* xn_out) {
    if (M <= 0) return;
    if (dt == KASF_F32) { mlp_fwd_T<float>(s, x, ln_g, ln_b, W1, b1, W2, b2, ls2, out, M); return; }
    const int64_t tiles = (M + 31) / 32;                 // bf16: persistent producer / consumer kernel (k_mlp3.hip), one workgroup per CU
    const int64_t cap = kasf_narrow_grid(KASF_NG_MLP_FWD, 256, M);
    kasf_launch_mlp_fwd_s(s, x, ln_g, ln_b, W1, b1, W2, b2, ls2, out, M, xn_out, (unsigned)(tiles < cap ? tiles : cap));
}
void kasf_launch_mlp_bwd(int dt, hipStream_t s, const void* x, const void* g, const float* ln_g, const float* ln_b, const void* W1, const float* b1,
                         const void* W2ts, const void* W1t, void* Hbuf, void* dZbuf, void* xn_buf, void* g_in, float* dgamma, float* dbeta,
                         int64_t M, KasfColSink* sink) {
    if (M <= 0) return;
    if (dt == KASF_F32) mlp_bwd_T<float>(s, x, g, ln_g, ln_b, W1, b1, W2ts, W1t, Hbuf, dZbuf, xn_buf, g_in, dgamma, dbeta, M, sink);
    else mlp_bwd_T<bf16>(s, x, g, ln_g, ln_b, W1, b1, W2ts, W1t, Hbuf, dZbuf, xn_buf, g_in, dgamma, dbeta, M, sink);
}
