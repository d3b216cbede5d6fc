// GEMM-family kernels of the KASportsFormer path (gfx950).  All contractions have one short
// side (128..512) and a very long token side M = B*T*17, so every kernel tiles 128 tokens per
// workgroup, keeps the 128-wide operand tiles in swizzled LDS and runs 16x16 MFMA tiles
// (bf16 16x16x32 in fast mode, exact-f32 16x16x4 in parity mode).
//
//   k_linear        C = act( LN?(A) . W^T + bias )            qkv / q / kv / U|V / rep_logit.fc
//   k_linear_res    C = resid + ls * (A . W^T + bias)          attention proj (+layer-scale, +residual)
//   k_dgrad_lnbwd   g_in = [resid +] LNbwd( dY . Wt^T [+ add] ) dgrad of an LN-fused linear + LN backward
//   k_wgrad         dW[N][K] += G^T . LN?(X), db += colsum(G)   split over M, LDS-transpose reads
//   k_pack          fp32 master weights -> T (optionally transposed / row-scaled) kernel arena
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"

namespace {

// ---------------------------------------------------------------------------------------------
template <typename T, bool LN, int ACT>
__global__ __launch_bounds__(256) void k_linear(const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw,
                                                const float* __restrict__ bias, T* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                const float* __restrict__ ln_g, const float* __restrict__ ln_b, T* __restrict__ xn_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);
    T* sB = sA + 128 * 128;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    stage_rows<T, 128, LN>(sA, A, lda, row0, M, ln_g, ln_b, xn_out);
    for (int n0 = 0; n0 < N; n0 += 128) {
        __syncthreads();
        stage_w<T>(sB, W + (int64_t)n0 * ldw, ldw);
        __syncthreads();
        f32x4 acc[4][4];
        zero_acc(acc);
        mma_k128<4, 4>(sB, wave_n0(), sA, wave_m0(), acc);
        __syncthreads();
        acc_to_tile<T>(sB, acc, wave_n0(), wave_m0(), [&](float v, int n) {
            if (bias != nullptr) v += bias[n0 + n];
            if (ACT == 1) v = tanhf(v);
            return v;
        });
        __syncthreads();
        for (int idx = threadIdx.x; idx < 128 * 16; idx += 256) {
            const int r = idx >> 4, sub = idx & 15;
            if (row0 + r < M) {
                float v[8];
                tile_load8(sB, r, sub * 8, v);
                store8(C + (row0 + r) * ldc + n0 + sub * 8, v);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_linear_res(const T* __restrict__ A, const T* __restrict__ W, const float* __restrict__ bias,
                                                    const float* __restrict__ ls, const T* __restrict__ resid, T* __restrict__ C, int64_t M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);
    T* sB = sA + 128 * 128;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    stage_rows<T, 128, false>(sA, A, 128, row0, M, nullptr, nullptr, nullptr);
    stage_w<T>(sB, W, 128);
    __syncthreads();
    f32x4 acc[4][4];
    zero_acc(acc);
    mma_k128<4, 4>(sB, wave_n0(), sA, wave_m0(), acc);
    __syncthreads();
    acc_to_tile<T>(sB, acc, wave_n0(), wave_m0(), [&](float v, int n) { return (v + bias[n]) * ls[n]; });
    __syncthreads();
    for (int idx = threadIdx.x; idx < 128 * 16; idx += 256) {
        const int r = idx >> 4, sub = idx & 15;
        if (row0 + r < M) {
            float v[8], x[8];
            tile_load8(sB, r, sub * 8, v);
            load8(resid + (row0 + r) * 128 + sub * 8, x);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += x[i];
            store8(C + (row0 + r) * 128 + sub * 8, v);
        }
    }
}

// dxn = dY[M x Kd] . Wt[128 x Kd]^T (+ dxn_add);  out = (resid?) + (accumulate? out) + LNbwd(dxn; x, gamma)
// dgamma += sum_m dxn*xhat, dbeta += sum_m dxn  (block partials -> fp32 atomics)
template <typename T>
__global__ __launch_bounds__(256) void k_dgrad_lnbwd(const T* __restrict__ dY, int Kd, const T* __restrict__ Wt, const T* __restrict__ dxn_add,
                                                     const T* __restrict__ X, const float* __restrict__ gamma, const T* __restrict__ resid,
                                                     T* __restrict__ out, int accumulate, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int64_t M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);
    T* sB = sA + 128 * 128;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    f32x4 acc[4][4];
    zero_acc(acc);
    for (int k0 = 0; k0 < Kd; k0 += 128) {
        __syncthreads();
        stage_rows<T, 128, false>(sA, dY + k0, Kd, row0, M, nullptr, nullptr, nullptr);
        stage_w<T>(sB, Wt + k0, Kd);
        __syncthreads();
        mma_k128<4, 4>(sB, wave_n0(), sA, wave_m0(), acc);
    }
    __syncthreads();
    acc_to_tile<T>(sB, acc, wave_n0(), wave_m0(), [](float v, int) { return v; });
    __syncthreads();
    lnbwd_rows<T, 128>(sB, X, gamma, dxn_add, resid, out, accumulate, dgamma, dbeta, row0, M, reinterpret_cast<float*>(smem));
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: out[n][k] += sum_m G[m][n0+n] * Xp[m][k0+k] over this workgroup's slice of M,
// Xp = LN(X) when LN (then ldx == 128).  The reduction index m is the ROW index of both operands
// in memory, so bf16 fragments are fetched with ds_read_b64_tr_b16 (hardware LDS transpose) from
// row-major [m][*] tiles padded to 144 elements per row.  grid = (N/128, K/128, splits).
// ---------------------------------------------------------------------------------------------
constexpr int WG_LD = 144;
constexpr int WG_BM = 128;

__device__ __forceinline__ bf16x8 frag_tr(const bf16* s, int mbase, int col0) {
    // group of 16 lanes: lane u = 4q+p supplies row (mbase+q), columns col0+4p..; lane u receives column col0+u of 4 rows
    const int u = threadIdx.x & 15, q = u >> 2, p = u & 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s + (mbase + q) * WG_LD + col0 + 4 * p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s + (mbase + 4 + q) * WG_LD + col0 + 4 * p));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ void wgrad_mma(const bf16* sG, const bf16* sX, int wn0, int wk0, f32x4 (&acc)[4][4]) {
    const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int s = 0; s < WG_BM / 32; ++s) {
        bf16x8 a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = frag_tr(sG, 32 * s + 8 * g, wn0 + 16 * t);
            b[t] = frag_tr(sX, 32 * s + 8 * g, wk0 + 16 * t);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], b[kt], acc[nt][kt], 0, 0, 0);
    }
}
__device__ __forceinline__ void wgrad_mma(const float* sG, const float* sX, int wn0, int wk0, f32x4 (&acc)[4][4]) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll 4
    for (int s = 0; s < WG_BM / 4; ++s) {
        float a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = sG[(4 * s + g) * WG_LD + wn0 + 16 * t + i];
            b[t] = sX[(4 * s + g) * WG_LD + wk0 + 16 * t + i];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt], b[kt], acc[nt][kt], 0, 0, 0);
    }
}

template <typename T, bool LN>
__global__ __launch_bounds__(256) void k_wgrad(const T* __restrict__ G, int64_t ldg, const T* __restrict__ X, int64_t ldx,
                                               const float* __restrict__ ln_g, const float* __restrict__ ln_b, float* __restrict__ out,
                                               int64_t ldo, float* __restrict__ dbias, int64_t M, int64_t slice) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sG = reinterpret_cast<T*>(smem);
    T* sX = sG + WG_BM * WG_LD;
    const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
    const int64_t m_begin = (int64_t)blockIdx.z * slice, m_end = (m_begin + slice < M) ? m_begin + slice : M;
    const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    float gm[8], bt[8], bsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { bsum[i] = 0.f; gm[i] = LN ? ln_g[sub * 8 + i] : 0.f; bt[i] = LN ? ln_b[sub * 8 + i] : 0.f; }
    f32x4 acc[4][4];
    zero_acc(acc);
    for (int64_t m0 = m_begin; m0 < m_end; m0 += WG_BM) {
        __syncthreads();
        for (int r = rl; r < WG_BM; r += 16) {
            const int64_t row = m0 + r;
            const bool ok = row < m_end;
            float v[8], x[8];
            if (ok) { load8(G + row * ldg + n0 + sub * 8, v); load8(X + row * ldx + k0 + sub * 8, x); }
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[i] = 0.f; x[i] = 0.f; }
            }
            if (LN) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += x[i];
                const float mean = reduce16(s) * (1.0f / 128.0f);
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { x[i] -= mean; q += x[i] * x[i]; }
                const float rstd = rsqrtf(reduce16(q) * (1.0f / 128.0f) + KASF_LN_EPS);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = ok ? x[i] * rstd * gm[i] + bt[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) bsum[i] += v[i];
            store8(sG + r * WG_LD + sub * 8, v);
            store8(sX + r * WG_LD + sub * 8, x);
        }
        __syncthreads();
        wgrad_mma(sG, sX, wave_n0(), wave_m0(), acc);      // wave_m0() doubles as the k-half here
    }
    {   // out[n][k] += acc : lane owns 4 consecutive n (rows of out) for one k -> strided fp32 atomics (L2-resident tile)
        const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wave_n0() + nt * 16 + g * 4 + r, k = k0 + wave_m0() + kt * 16 + i;
                    atomicAdd(out + (int64_t)n * ldo + k, acc[nt][kt][r]);
                }
    }
    if (dbias != nullptr && blockIdx.y == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [16][128]
#pragma unroll
        for (int i = 0; i < 8; ++i) red[rl * 128 + sub * 8 + i] = bsum[i];
        __syncthreads();
        if (threadIdx.x < 128) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * 128 + threadIdx.x];
            atomicAdd(dbias + n0 + threadIdx.x, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pack: one 32x32 tile per workgroup, table driven (one launch for the whole model).
// dst[r][c] = scale[r] * src[r][c]      (transpose == 0, dst is [rows][cols])
// dst[c][r] = scale[r] * src[r][c]      (transpose == 1, dst is [cols][rows])
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_pack(const float* __restrict__ params, T* __restrict__ arena, const KasfPackDesc* __restrict__ desc,
                                              const int* __restrict__ tile_start, int ndesc) {
    __shared__ float tile[32][33];
    int lo = 0, hi = ndesc - 1;                 // last descriptor with tile_start <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const KasfPackDesc d = desc[lo];
    const int t = blockIdx.x - tile_start[lo], tiles_c = d.cols / 32;
    const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const float* src = params + d.src;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k;
        float v = src[(int64_t)r * d.cols + c0 + tx];
        if (d.scale >= 0) v *= params[d.scale + r];
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
    T* dst = arena + d.dst;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (d.transpose) dst[(int64_t)(c0 + ty + 8 * k) * d.rows + r0 + tx] = from_f<T>(tile[tx][ty + 8 * k]);
        else dst[(int64_t)(r0 + ty + 8 * k) * d.cols + c0 + tx] = from_f<T>(tile[ty + 8 * k][tx]);
    }
}

template <typename T> constexpr size_t tile_bytes() { return 2 * 128 * 128 * sizeof(T); }

}  // namespace

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
#define DT_DISPATCH(dt, CALL_F32, CALL_BF16) \
    do { if ((dt) == KASF_F32) { CALL_F32; } else { CALL_BF16; } } while (0)

template <typename K> static void set_smem(K k, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T>
static void linear_T(hipStream_t s, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc, int64_t M,
                     int N, const float* ln_g, const float* ln_b, void* xn_out, int act) {
    const dim3 grid((unsigned)((M + 127) / 128));
    const size_t sh = tile_bytes<T>();
    auto go = [&](auto kern) {
        set_smem(kern, sh);
        hipLaunchKernelGGL(kern, grid, dim3(256), sh, s, (const T*)A, lda, (const T*)W, ldw, bias, (T*)C, ldc, M, N, ln_g, ln_b, (T*)xn_out);
    };
    if (ln_g != nullptr) { if (act == 1) go(k_linear<T, true, 1>); else go(k_linear<T, true, 0>); }
    else { if (act == 1) go(k_linear<T, false, 1>); else go(k_linear<T, false, 0>); }
}
void kasf_launch_linear(int dt, hipStream_t s, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                        int64_t M, int N, const float* ln_g, const float* ln_b, void* xn_out, int act) {
    DT_DISPATCH(dt, (linear_T<float>(s, A, lda, W, ldw, bias, C, ldc, M, N, ln_g, ln_b, xn_out, act)),
                (linear_T<bf16>(s, A, lda, W, ldw, bias, C, ldc, M, N, ln_g, ln_b, xn_out, act)));
}

template <typename T>
static void linear_res_T(hipStream_t s, const void* A, const void* W, const float* bias, const float* ls, const void* resid, void* C, int64_t M) {
    set_smem(k_linear_res<T>, tile_bytes<T>());
    hipLaunchKernelGGL(k_linear_res<T>, dim3((unsigned)((M + 127) / 128)), dim3(256), tile_bytes<T>(), s, (const T*)A, (const T*)W, bias, ls,
                       (const T*)resid, (T*)C, M);
}
void kasf_launch_linear_res(int dt, hipStream_t s, const void* A, const void* W, const float* bias, const float* ls, const void* resid, void* C,
                            int64_t M) {
    DT_DISPATCH(dt, (linear_res_T<float>(s, A, W, bias, ls, resid, C, M)), (linear_res_T<bf16>(s, A, W, bias, ls, resid, C, M)));
}

template <typename T>
static void dgrad_lnbwd_T(hipStream_t s, const void* dY, int Kd, const void* Wt, const void* dxn_add, const void* X, const float* gamma,
                          const void* resid, void* out, int accumulate, float* dgamma, float* dbeta, int64_t M) {
    set_smem(k_dgrad_lnbwd<T>, tile_bytes<T>());
    hipLaunchKernelGGL(k_dgrad_lnbwd<T>, dim3((unsigned)((M + 127) / 128)), dim3(256), tile_bytes<T>(), s, (const T*)dY, Kd, (const T*)Wt,
                       (const T*)dxn_add, (const T*)X, gamma, (const T*)resid, (T*)out, accumulate, dgamma, dbeta, M);
}
void kasf_launch_dgrad_lnbwd(int dt, hipStream_t s, const void* dY, int Kd, const void* Wt, const void* dxn_add, const void* X, const float* gamma,
                             const void* resid, void* out, int accumulate, float* dgamma, float* dbeta, int64_t M) {
    DT_DISPATCH(dt, (dgrad_lnbwd_T<float>(s, dY, Kd, Wt, dxn_add, X, gamma, resid, out, accumulate, dgamma, dbeta, M)),
                (dgrad_lnbwd_T<bf16>(s, dY, Kd, Wt, dxn_add, X, gamma, resid, out, accumulate, dgamma, dbeta, M)));
}

template <typename T>
static void wgrad_T(hipStream_t s, const void* G, int64_t ldg, int N, const void* X, int64_t ldx, int K, const float* ln_g, const float* ln_b,
                    float* out, int64_t ldo, float* dbias, int64_t M) {
    const int tiles = (N / 128) * (K / 128);
    int splits = (768 + tiles - 1) / tiles;                         // ~3 workgroups per CU in total
    const int64_t max_splits = (M + WG_BM - 1) / WG_BM;
    if (splits > max_splits) splits = (int)max_splits;
    if (splits < 1) splits = 1;
    int64_t slice = (M + splits - 1) / splits;
    slice = (slice + WG_BM - 1) / WG_BM * WG_BM;
    splits = (int)((M + slice - 1) / slice);
    const size_t sh = 2 * WG_BM * WG_LD * sizeof(T);
    const dim3 grid(N / 128, K / 128, splits);
    if (ln_g != nullptr) {
        set_smem(k_wgrad<T, true>, sh);
        hipLaunchKernelGGL((k_wgrad<T, true>), grid, dim3(256), sh, s, (const T*)G, ldg, (const T*)X, ldx, ln_g, ln_b, out, ldo, dbias, M, slice);
    } else {
        set_smem(k_wgrad<T, false>, sh);
        hipLaunchKernelGGL((k_wgrad<T, false>), grid, dim3(256), sh, s, (const T*)G, ldg, (const T*)X, ldx, ln_g, ln_b, out, ldo, dbias, M, slice);
    }
}
void kasf_launch_wgrad(int dt, hipStream_t s, const void* G, int64_t ldg, int N, const void* X, int64_t ldx, int K, const float* ln_g,
                       const float* ln_b, float* out, int64_t ldo, float* dbias, int64_t M) {
    DT_DISPATCH(dt, (wgrad_T<float>(s, G, ldg, N, X, ldx, K, ln_g, ln_b, out, ldo, dbias, M)),
                (wgrad_T<bf16>(s, G, ldg, N, X, ldx, K, ln_g, ln_b, out, ldo, dbias, M)));
}

void kasf_launch_pack(int dt, hipStream_t s, const float* params, void* arena, const KasfPackDesc* desc, const int* tile_start, int ndesc,
                      int total_tiles) {
    if (total_tiles <= 0) return;
    if (dt == KASF_F32) hipLaunchKernelGGL(k_pack<float>, dim3(total_tiles), dim3(256), 0, s, params, (float*)arena, desc, tile_start, ndesc);
    else hipLaunchKernelGGL(k_pack<bf16>, dim3(total_tiles), dim3(256), 0, s, params, (bf16*)arena, desc, tile_start, ndesc);
}
