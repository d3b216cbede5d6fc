// Fused MLP half of a FormerModule (reference: model/KASportsFormer.py:111, modules/mlp.py:24-30):
//     out = x + ls2 * ( GELU( LN(x) W1^T + b1 ) W2^T + b2 )
// ~69 % of the model's FLOPs.  One workgroup owns BM tokens; the [BM x 512] hidden activation never
// leaves the CU: it is produced 128 columns at a time into LDS and consumed immediately by the second
// GEMM.  W1 / W2 blocks (128 x 128) stream from L2 through one LDS buffer.
//
// Backward recomputes Z = LN(x) W1^T + b1 (no activation stash in HBM during forward), produces
// dZ and H = GELU(Z) for the two weight-gradient GEMMs, chains dA = dZ . W1 in registers and ends
// with the LayerNorm backward + residual:   g_in = g + LNbwd(dA).
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"

namespace {

template <typename T, int BM>
__global__ __launch_bounds__(256) void k_mlp_fwd(const T* __restrict__ X, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                 const T* __restrict__ W1, const float* __restrict__ b1, const T* __restrict__ W2,
                                                 const float* __restrict__ b2, const float* __restrict__ ls2, T* __restrict__ out, int64_t M) {
    constexpr int MT = BM / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);        // LN(x)          [BM][128]
    T* sH = sA + BM * 128;                     // GELU chunk     [BM][128]
    T* sB = sH + BM * 128;                     // weight block   [128][128]
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int wn0 = wave_n0(), wm0 = wave_m0_bm<BM>();
    stage_rows<T, BM, true>(sA, X, 128, row0, M, ln_g, ln_b, nullptr);
    f32x4 acc2[4][MT];
    zero_acc(acc2);
    for (int hc = 0; hc < 4; ++hc) {
        __syncthreads();
        stage_w<T>(sB, W1 + (int64_t)hc * 128 * 128, 128);              // rows = hidden units of this chunk
        __syncthreads();
        f32x4 acc1[4][MT];
        zero_acc(acc1);
        mma_k128<4, MT>(sB, wn0, sA, wm0, acc1);
        acc_to_tile<T>(sH, acc1, wn0, wm0, [&](float v, int n) { return gelu_f(v + b1[hc * 128 + n]); });
        __syncthreads();
        stage_w<T>(sB, W2 + hc * 128, 512);                             // rows = output channels, k = this hidden chunk
        __syncthreads();
        mma_k128<4, MT>(sB, wn0, sH, wm0, acc2);
    }
    __syncthreads();
    acc_to_tile<T>(sH, acc2, wn0, wm0, [&](float v, int n) { return (v + b2[n]) * ls2[n]; });
    __syncthreads();
    for (int idx = threadIdx.x; idx < BM * 16; idx += 256) {
        const int r = idx >> 4, sub = idx & 15;
        if (row0 + r < M) {
            float v[8], x[8];
            tile_load8(sH, r, sub * 8, v);
            load8(X + (row0 + r) * 128 + sub * 8, x);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += x[i];
            store8(out + (row0 + r) * 128 + sub * 8, v);
        }
    }
}

template <typename T, int BM>
__global__ __launch_bounds__(256) void k_mlp_bwd(const T* __restrict__ X, const T* __restrict__ G, const float* __restrict__ ln_g,
                                                 const float* __restrict__ ln_b, const T* __restrict__ W1, const float* __restrict__ b1,
                                                 const T* __restrict__ W2ts, const T* __restrict__ W1t, T* __restrict__ Hbuf,
                                                 T* __restrict__ dZbuf, T* __restrict__ g_in, float* __restrict__ dgamma,
                                                 float* __restrict__ dbeta, int64_t M) {
    constexpr int MT = BM / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);        // LN(x)        [BM][128]
    T* sG = sA + BM * 128;                     // upstream g   [BM][128]
    T* sD = sG + BM * 128;                     // dZ chunk     [BM][128]
    T* sB = sD + BM * 128;                     // weight block [128][128]  (also staging for the H chunk copy-out)
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int wn0 = wave_n0(), wm0 = wave_m0_bm<BM>();
    stage_rows<T, BM, true>(sA, X, 128, row0, M, ln_g, ln_b, nullptr);
    stage_rows<T, BM, false>(sG, G, 128, row0, M, nullptr, nullptr, nullptr);
    f32x4 accA[4][MT];
    zero_acc(accA);
    for (int hc = 0; hc < 4; ++hc) {
        __syncthreads();
        stage_w<T>(sB, W1 + (int64_t)hc * 128 * 128, 128);
        __syncthreads();
        f32x4 accZ[4][MT], accH[4][MT];
        zero_acc(accZ);
        zero_acc(accH);
        mma_k128<4, MT>(sB, wn0, sA, wm0, accZ);                        // Z^T[h][m]
        __syncthreads();
        stage_w<T>(sB, W2ts + (int64_t)hc * 128 * 128, 128);            // (ls2 . W2)^T : rows = hidden units, k = channels
        __syncthreads();
        mma_k128<4, MT>(sB, wn0, sG, wm0, accH);                        // dH^T[h][m]
        __syncthreads();                                                // every wave is done with sB
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
                        h[r] = gelu_f(z);
                        dz[r] = accH[nt][mt][r] * gelu_grad_f(z);
                    }
                    store4(sB + Tile<T>::off4(m, n), h);                // only rows < BM of sB are used
                    store4(sD + Tile<T>::off4(m, n), dz);
                }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < BM * 16; idx += 256) {
            const int r = idx >> 4, sub = idx & 15;
            if (row0 + r < M) {
                float v[8];
                tile_load8(sB, r, sub * 8, v);
                store8(Hbuf + (row0 + r) * 512 + hc * 128 + sub * 8, v);
                tile_load8(sD, r, sub * 8, v);
                store8(dZbuf + (row0 + r) * 512 + hc * 128 + sub * 8, v);
            }
        }
        __syncthreads();
        stage_w<T>(sB, W1t + hc * 128, 512);                            // W1^T block: rows = channels k, reduction = hidden chunk
        __syncthreads();
        mma_k128<4, MT>(sB, wn0, sD, wm0, accA);                        // dA^T[k][m] += sum_h W1[h][k] dZ[m][h]
    }
    __syncthreads();
    acc_to_tile<T>(sD, accA, wn0, wm0, [](float v, int) { return v; });
    __syncthreads();
    lnbwd_rows<T, BM>(sD, X, ln_g, (const T*)nullptr, G, g_in, 0, dgamma, dbeta, row0, M, reinterpret_cast<float*>(sB));
}

template <typename T> struct MlpCfg;
template <> struct MlpCfg<bf16> { static constexpr int BM_F = 128, BM_B = 128; };
template <> struct MlpCfg<float> { static constexpr int BM_F = 64, BM_B = 32; };

template <typename K> void set_smem(K k, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T>
void mlp_fwd_T(hipStream_t s, const void* x, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2, const float* b2,
               const float* ls2, void* out, int64_t M) {
    constexpr int BM = MlpCfg<T>::BM_F;
    const size_t sh = (2 * BM * 128 + 128 * 128) * sizeof(T);
    set_smem(k_mlp_fwd<T, BM>, sh);
    hipLaunchKernelGGL((k_mlp_fwd<T, BM>), dim3((unsigned)((M + BM - 1) / BM)), dim3(256), sh, s, (const T*)x, ln_g, ln_b, (const T*)W1, b1,
                       (const T*)W2, b2, ls2, (T*)out, M);
}
template <typename T>
void mlp_bwd_T(hipStream_t s, const void* x, const void* g, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2ts,
               const void* W1t, void* Hbuf, void* dZbuf, void* g_in, float* dgamma, float* dbeta, int64_t M) {
    constexpr int BM = MlpCfg<T>::BM_B;
    const size_t sh = (3 * BM * 128 + 128 * 128) * sizeof(T);
    set_smem(k_mlp_bwd<T, BM>, sh);
    hipLaunchKernelGGL((k_mlp_bwd<T, BM>), dim3((unsigned)((M + BM - 1) / BM)), dim3(256), sh, s, (const T*)x, (const T*)g, ln_g, ln_b,
                       (const T*)W1, b1, (const T*)W2ts, (const T*)W1t, (T*)Hbuf, (T*)dZbuf, (T*)g_in, dgamma, dbeta, M);
}

}  // namespace

void kasf_launch_mlp_fwd(int dt, hipStream_t s, const void* x, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2,
                         const float* b2, const float* ls2, void* out, int64_t M) {
    if (dt == KASF_F32) mlp_fwd_T<float>(s, x, ln_g, ln_b, W1, b1, W2, b2, ls2, out, M);
    else mlp_fwd_T<bf16>(s, x, ln_g, ln_b, W1, b1, W2, b2, ls2, out, M);
}
void kasf_launch_mlp_bwd(int dt, hipStream_t s, const void* x, const void* g, const float* ln_g, const float* ln_b, const void* W1, const float* b1,
                         const void* W2ts, const void* W1t, void* Hbuf, void* dZbuf, void* g_in, float* dgamma, float* dbeta, int64_t M) {
    if (dt == KASF_F32) mlp_bwd_T<float>(s, x, g, ln_g, ln_b, W1, b1, W2ts, W1t, Hbuf, dZbuf, g_in, dgamma, dbeta, M);
    else mlp_bwd_T<bf16>(s, x, g, ln_g, ln_b, W1, b1, W2ts, W1t, Hbuf, dZbuf, g_in, dgamma, dbeta, M);
}
