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
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"

namespace {

template <typename T> __device__ __forceinline__ void add_bias4(float (&v)[4], const float* bias) {
    if (bias != nullptr) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += b[r];
    }
}

// ---------------------------------------------------------------------------------------------
// C[M x N] = act(LN?(A)[M x 128] . W[N x 128]^T + bias).  One workgroup = BM tokens; the weight is
// streamed 128 rows at a time by LDS-direct loads into NBUF tiles (block nb+1 lands while block nb
// is multiplied); the epilogue goes straight from the accumulators to HBM.
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int NBUF, bool LN, int ACT>
__global__ __launch_bounds__(256) void k_linear(const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw,
                                                const float* __restrict__ bias, T* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                const float* __restrict__ ln_g, const float* __restrict__ ln_b, T* __restrict__ xn_out) {
    constexpr int MT = BM / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);
    T* sW = sA + BM * 128;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int wn0 = wave_n0(), wm0 = wave_m0_bm<BM>();
    stage_tile_async<T, 128>(sW, W, ldw, 128);
    stage_rows<T, BM, LN>(sA, A, lda, row0, M, ln_g, ln_b, xn_out);
    const int NB = N / 128;
    for (int nb = 0; nb < NB; ++nb) {
        const T* cur = sW + (nb % NBUF) * 128 * 128;
        wait_async();
        __syncthreads();
        if (NBUF == 2 && nb + 1 < NB) stage_tile_async<T, 128>(sW + ((nb + 1) & 1) * 128 * 128, W + (int64_t)(nb + 1) * 128 * ldw, ldw, 128);
        f32x4 acc[4][MT];
        zero_acc(acc);
        mma_k128<4, MT>(cur, wn0, sA, wm0, acc);
        const int n0 = nb * 128;
        acc_foreach<4, MT>(acc, wn0, wm0, row0, M, [&](float (&v)[4], int64_t row, int n) {
            add_bias4<T>(v, bias != nullptr ? bias + n0 + n : nullptr);
            if (ACT == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
            }
            store4(C + row * ldc + n0 + n, v);
        });
        if (NBUF == 1 && nb + 1 < NB) {
            __syncthreads();
            stage_tile_async<T, 128>(sW, W + (int64_t)(nb + 1) * 128 * ldw, ldw, 128);
        }
    }
}

template <typename T, int BM>
__global__ __launch_bounds__(256) void k_linear_res(const T* __restrict__ A, const T* __restrict__ W, const float* __restrict__ bias,
                                                    const float* __restrict__ ls, const T* __restrict__ resid, T* __restrict__ C, int64_t M) {
    constexpr int MT = BM / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);
    T* sW = sA + BM * 128;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int wn0 = wave_n0(), wm0 = wave_m0_bm<BM>();
    const int nvalid = (int)((M - row0) < BM ? (M - row0) : BM);
    stage_tile_async<T, 128>(sW, W, 128, 128);
    stage_tile_async<T, BM>(sA, A + row0 * 128, 128, nvalid);
    wait_async();
    __syncthreads();
    f32x4 acc[4][MT];
    zero_acc(acc);
    mma_k128<4, MT>(sW, wn0, sA, wm0, acc);
    acc_foreach<4, MT>(acc, wn0, wm0, row0, M, [&](float (&v)[4], int64_t row, int n) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + n), l = *reinterpret_cast<const f32x4*>(ls + n);
        float x[4];
        load4(resid + row * 128 + n, x);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x[r] + l[r] * (v[r] + b[r]);
        store4(C + row * 128 + n, v);
    });
}

// dxn = dY[M x Kd] . Wt[128 x Kd]^T (+ dxn_add);  out = (resid?) + (accumulate? out) + LNbwd(dxn; x, gamma)
// dgamma += sum_m dxn*xhat, dbeta += sum_m dxn  (one [dgamma | dbeta] row per workgroup in `part`, see lnbwd_rows)
template <typename T, int BM, int NBUF>
__global__ __launch_bounds__(256) void k_dgrad_lnbwd(const T* __restrict__ dY, int Kd, const T* __restrict__ Wt, const T* __restrict__ dxn_add,
                                                     const T* __restrict__ X, const float* __restrict__ gamma, const T* __restrict__ resid,
                                                     T* __restrict__ out, int accumulate, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int64_t M, T* __restrict__ xn_out, const float* __restrict__ beta, float* __restrict__ part) {
    constexpr int MT = BM / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);                 // [NBUF][BM][128]
    T* sW = sA + NBUF * BM * 128;                       // [NBUF][128][128]
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int wn0 = wave_n0(), wm0 = wave_m0_bm<BM>();
    const int nvalid = (int)((M - row0) < BM ? (M - row0) : BM);
    const T* a0 = dY + row0 * Kd;
    stage_tile_async<T, BM>(sA, a0, Kd, nvalid);
    stage_tile_async<T, 128>(sW, Wt, Kd, 128);
    f32x4 acc[4][MT];
    zero_acc(acc);
    const int KC = Kd / 128;
    for (int kc = 0; kc < KC; ++kc) {
        const int cur = kc % NBUF;
        wait_async();
        __syncthreads();
        if (NBUF == 2 && kc + 1 < KC) {
            stage_tile_async<T, BM>(sA + (cur ^ 1) * BM * 128, a0 + (kc + 1) * 128, Kd, nvalid);
            stage_tile_async<T, 128>(sW + (cur ^ 1) * 128 * 128, Wt + (kc + 1) * 128, Kd, 128);
        }
        mma_k128<4, MT>(sW + cur * 128 * 128, wn0, sA + cur * BM * 128, wm0, acc);
        if (NBUF == 1 && kc + 1 < KC) {
            __syncthreads();
            stage_tile_async<T, BM>(sA, a0 + (kc + 1) * 128, Kd, nvalid);
            stage_tile_async<T, 128>(sW, Wt + (kc + 1) * 128, Kd, 128);
        }
    }
    __syncthreads();
    acc_to_tile<T>(sW, acc, wn0, wm0, [](float v, int) { return v; });
    __syncthreads();
    lnbwd_rows<T, BM>(sW, X, gamma, dxn_add, resid, out, accumulate, dgamma, dbeta, row0, M, reinterpret_cast<float*>(smem), xn_out, beta, part);
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: out[n][k] += sum_m G[m][n0+n] * Xp[m][k0+k] over this workgroup's slice of M,
// Xp = LN(X) when LN (then ldx == 128).  The reduction index m is the ROW index of both operands in
// memory, so bf16 fragments come from ds_read_b64_tr_b16 (hardware LDS transpose) on row-major
// [m][128] swizzled tiles.  Full 128-row tiles arrive by LDS-direct loads, double buffered; the ragged
// tail (and the LayerNorm-ed operand) is staged through registers with zero fill.
// grid = (N/128, K/128, splits).
// ---------------------------------------------------------------------------------------------
constexpr int WG_BM = 128;

template <int ROWS = WG_BM>
__device__ __forceinline__ void wgrad_mma(const bf16* sG, const bf16* sX, int wn0, int wk0, f32x4 (&acc)[4][4]) {
    const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int s = 0; s < ROWS / 32; ++s) {
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
template <int ROWS = WG_BM>
__device__ __forceinline__ void wgrad_mma(const float* sG, const float* sX, int wn0, int wk0, f32x4 (&acc)[4][4]) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll 4
    for (int s = 0; s < ROWS / 4; ++s) {
        float a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = sG[eoff<float>(4 * s + g, wn0 + 16 * t + i)];
            b[t] = sX[eoff<float>(4 * s + g, wk0 + 16 * t + i)];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt], b[kt], acc[nt][kt], 0, 0, 0);
    }
}

// register-path staging of one [128 x 128] operand tile with zero fill past m_end (and optional LayerNorm)
template <typename T, bool LN>
__device__ __forceinline__ void wgrad_stage_sync(T* sT, const T* src, int64_t ld, int64_t m0, int64_t m_end, const float (&gm)[8], const float (&bt)[8]) {
    const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
#pragma unroll 2
    for (int r = rl; r < WG_BM; r += 16) {
        const int64_t row = m0 + r;
        const bool ok = row < m_end;
        float x[8];
        if (ok) load8(src + row * ld + sub * 8, x);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = 0.f;
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
        tile_store8(sT, r, sub * 8, x);
    }
}

template <typename T, int NBUF, bool LN>
__global__ __launch_bounds__(256) void k_wgrad(const T* __restrict__ G, int64_t ldg, const T* __restrict__ X, int64_t ldx,
                                               const float* __restrict__ ln_g, const float* __restrict__ ln_b, float* __restrict__ out,
                                               int64_t ldo, float* __restrict__ dbias, int64_t M, int64_t slice, float* __restrict__ partial, float* __restrict__ brow) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sG = reinterpret_cast<T*>(smem);                 // [NBUF][128][128]
    T* sX = sG + NBUF * 128 * 128;                      // [NBUF][128][128]
    const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
    const int64_t m_begin = (int64_t)blockIdx.z * slice, m_end = (m_begin + slice < M) ? m_begin + slice : M;
    const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const bool want_bias = dbias != nullptr && blockIdx.y == 0;
    float gm[8], bt[8], bsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { bsum[i] = 0.f; gm[i] = LN ? ln_g[sub * 8 + i] : 0.f; bt[i] = LN ? ln_b[sub * 8 + i] : 0.f; }
    f32x4 acc[4][4];
    zero_acc(acc);
    auto issue = [&](int buf, int64_t m0) {             // full tiles only
        stage_tile_async<T, 128>(sG + buf * 128 * 128, G + m0 * ldg + n0, ldg, 128);
        if (!LN) stage_tile_async<T, 128>(sX + buf * 128 * 128, X + m0 * ldx + k0, ldx, 128);
    };
    const int64_t ntiles = (m_end - m_begin + WG_BM - 1) / WG_BM;
    auto full = [&](int64_t t) { return m_begin + (t + 1) * WG_BM <= m_end; };
    if (ntiles > 0 && full(0)) issue(0, m_begin);
    for (int64_t t = 0; t < ntiles; ++t) {
        const int cur = (int)(t % NBUF);
        const int64_t m0 = m_begin + t * WG_BM;
        T* cG = sG + cur * 128 * 128;
        T* cX = sX + cur * 128 * 128;
        if (!full(t)) {                                  // ragged tail: synchronous, zero filled
            __syncthreads();
            wgrad_stage_sync<T, false>(cG, G + n0, ldg, m0, m_end, gm, bt);
            wgrad_stage_sync<T, LN>(cX, X + k0, ldx, m0, m_end, gm, bt);
        } else if (LN) {
            if (NBUF == 1) __syncthreads();
            wgrad_stage_sync<T, true>(cX, X + k0, ldx, m0, m_end, gm, bt);
        }
        wait_async();
        __syncthreads();
        if (NBUF == 2 && t + 1 < ntiles && full(t + 1)) issue(cur ^ 1, m0 + WG_BM);
        if (want_bias) {
            for (int r = rl; r < WG_BM; r += 16) {
                float v[8];
                tile_load8(cG, r, sub * 8, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) bsum[i] += v[i];
            }
        }
        wgrad_mma(cG, cX, wave_n0(), wave_m0(), acc);      // wave_m0() doubles as the k-half here
        if (NBUF == 1 && t + 1 < ntiles) {
            __syncthreads();
            if (full(t + 1)) issue(0, m0 + WG_BM);
        }
    }
    {   // lane owns 4 consecutive n (rows of out) for one k
        const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
        // partial != nullptr: plain stores of this split's tile into partial[z][N][K] (N = 128*gridDim.x, K = 128*gridDim.y),
        // summed afterwards by k_wgrad_reduce (deterministic; plain stores run ~5x the fp32-atomic rate).
        float* dst = partial != nullptr ? partial + (int64_t)blockIdx.z * (gridDim.x * 128) * (gridDim.y * 128) : out;
        const int64_t ld = partial != nullptr ? (int64_t)gridDim.y * 128 : ldo;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wave_n0() + nt * 16 + g * 4 + r, k = k0 + wave_m0() + kt * 16 + i;
                    if (partial != nullptr) dst[(int64_t)n * ld + k] = acc[nt][kt][r];
                    else atomicAdd(dst + (int64_t)n * ld + k, acc[nt][kt][r]);
                }
    }
    if (want_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [16][128]
#pragma unroll
        for (int i = 0; i < 8; ++i) red[rl * 128 + sub * 8 + i] = bsum[i];
        __syncthreads();
        if (threadIdx.x < 128) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * 128 + threadIdx.x];
            // one row of column sums per split (brow [splits][N], added in a fixed order by the finishing kernel), or an atomic without scratch
            if (brow != nullptr) brow[(int64_t)blockIdx.z * (gridDim.x * 128) + n0 + threadIdx.x] = s;
            else atomicAdd(dbias + n0 + threadIdx.x, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming variant (bf16, no LayerNorm): the reduction over tokens is an HBM stream of G and X, and a
// CU needs ~50 KB in flight to hide the ~2 us HBM latency.  64-row tiles go through a WR_ST-stage LDS ring
// filled by LDS-direct loads issued WR_ST-1 tiles ahead (2 stages measured equal to 4 and leave LDS for co-resident kernels); each wave waits only for ITS loads of the
// current tile (counted vmcnt, 8 loads per tile and wave) and the workgroup meets at a raw s_barrier,
// so two to three tiles stay in flight across every barrier.  The ragged tail goes through registers.
// ---------------------------------------------------------------------------------------------
constexpr int WR_BM = 64, WR_ST = 2, WR_IPT = 8;      // rows per tile, ring stages, LDS-direct loads per tile per wave

// One workgroup: the 128 x 128 tile (n0, k0) of split z.  N, K = full dimensions of the weight (partial tiles are laid out [z][N][K]).
// BF16P (the jobs launch of the bf16 engine, round 4): the split's partial tile leaves as bf16, staged through the (dead) ring and stored as whole 256-byte rows
// -- half the partial bytes of the step's 104 attention / bone blocks, and 8 wide stores per lane where the fp32 tile took 64 scattered 4-byte ones.
template <typename T, bool BF16P = false>
__device__ __forceinline__ void wgrad_ring_body(const T* __restrict__ G, int64_t ldg, const T* __restrict__ X, int64_t ldx, float* __restrict__ out,
                                                int64_t ldo, float* __restrict__ dbias, int64_t M, int64_t slice, float* __restrict__ partial, float* __restrict__ brow,
                                                int N, int K, int n0, int k0, int z) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* ring = reinterpret_cast<T*>(smem);               // [WR_ST][2][64][128]  (G tile, X tile)
    const int64_t m_begin = (int64_t)z * slice, m_end = (m_begin + slice < M) ? m_begin + slice : M;
    const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const bool want_bias = dbias != nullptr && k0 == 0;
    float bsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bsum[i] = 0.f;
    f32x4 acc[4][4];
    zero_acc(acc);
    const int nfull = (int)((m_end - m_begin) / WR_BM);
    auto issue = [&](int t) {
        T* st = ring + (t % WR_ST) * 2 * WR_BM * 128;
        const int64_t m0 = m_begin + (int64_t)t * WR_BM;
        stage_tile_async<T, WR_BM>(st, G + m0 * ldg + n0, ldg, WR_BM);
        stage_tile_async<T, WR_BM>(st + WR_BM * 128, X + m0 * ldx + k0, ldx, WR_BM);
    };
    auto consume = [&](const T* cG, const T* cX) {
        if (want_bias) {
            for (int r = rl; r < WR_BM; r += 16) {
                float v[8];
                tile_load8(cG, r, sub * 8, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) bsum[i] += v[i];
            }
        }
        wgrad_mma<WR_BM>(cG, cX, wave_n0(), wave_m0(), acc);
    };
    for (int t = 0; t < WR_ST - 1 && t < nfull; ++t) issue(t);
    for (int t = 0; t < nfull; ++t) {
        const int newer = nfull - 1 - t < WR_ST - 2 ? nfull - 1 - t : WR_ST - 2;     // younger tiles already requested
        if (newer >= 2) wait_async_le<2 * WR_IPT>();
        else if (newer == 1) wait_async_le<WR_IPT>();
        else wait_async_le<0>();
        barrier_keep_async();                           // tile t complete for every wave; everyone is done with tile t-1
        if (t + WR_ST - 1 < nfull) issue(t + WR_ST - 1);    // refills the stage tile t-1 used
        const T* st = ring + (t % WR_ST) * 2 * WR_BM * 128;
        consume(st, st + WR_BM * 128);
    }
    const int64_t m_tail = m_begin + (int64_t)nfull * WR_BM;
    if (m_tail < m_end) {                               // < 64 leftover rows: register path, zero filled (block-uniform branch)
        __syncthreads();
        T* st = ring;
        for (int r = rl; r < WR_BM; r += 16) {
            const int64_t row = m_tail + r;
            float v[8], x[8];
            if (row < m_end) { load8(G + row * ldg + n0 + sub * 8, v); load8(X + row * ldx + k0 + sub * 8, x); }
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[i] = 0.f; x[i] = 0.f; }
            }
            tile_store8(st, r, sub * 8, v);
            tile_store8(st + WR_BM * 128, r, sub * 8, x);
        }
        __syncthreads();
        consume(st, st + WR_BM * 128);
    }
    if constexpr (BF16P) {
        const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
        __syncthreads();                                 // every wave is done with the ring
        bf16* sT = reinterpret_cast<bf16*>(smem);        // [128][128] image of this split's tile
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) sT[(wave_n0() + nt * 16 + g * 4 + r) * 128 + wave_m0() + kt * 16 + i] = (bf16)acc[nt][kt][r];
        __syncthreads();
        bf16* dstb = reinterpret_cast<bf16*>(partial) + (int64_t)z * N * K + (int64_t)n0 * K + k0;
        for (int c = threadIdx.x; c < 128 * 16; c += 256)
            *reinterpret_cast<f32x4*>(dstb + (int64_t)(c >> 4) * K + (c & 15) * 8) = *reinterpret_cast<const f32x4*>(sT + (c >> 4) * 128 + (c & 15) * 8);
    } else {
        const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
        float* dst = partial != nullptr ? partial + (int64_t)z * N * K : out;
        const int64_t ld = partial != nullptr ? (int64_t)K : ldo;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wave_n0() + nt * 16 + g * 4 + r, k = k0 + wave_m0() + kt * 16 + i;
                    if (partial != nullptr) dst[(int64_t)n * ld + k] = acc[nt][kt][r];
                    else atomicAdd(dst + (int64_t)n * ld + k, acc[nt][kt][r]);
                }
    }
    if (want_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [16][128]
#pragma unroll
        for (int i = 0; i < 8; ++i) red[rl * 128 + sub * 8 + i] = bsum[i];
        __syncthreads();
        if (threadIdx.x < 128) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * 128 + threadIdx.x];
            // one row of column sums per split (brow [splits][N], added in a fixed order by the finishing kernel), or an atomic without scratch
            if (brow != nullptr) brow[(int64_t)z * N + n0 + threadIdx.x] = s;
            else atomicAdd(dbias + n0 + threadIdx.x, s);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_wgrad_ring(const T* __restrict__ G, int64_t ldg, const T* __restrict__ X, int64_t ldx,
                                                    float* __restrict__ out, int64_t ldo, float* __restrict__ dbias, int64_t M, int64_t slice,
                                                    float* __restrict__ partial, float* __restrict__ brow) {
    wgrad_ring_body<T>(G, ldg, X, ldx, out, ldo, dbias, M, slice, partial, brow, gridDim.x * 128, gridDim.y * 128, blockIdx.x * 128, blockIdx.y * 128, blockIdx.z);
}

// Several weight gradients of one block in ONE launch (bf16): the launch / ramp-up / partial-write overhead of these short streaming kernels
// is paid once, and one k_wgrad_finish_jobs launch sums every job's partial tiles (and applies the layer-scale algebra of the proj weight).
struct WgJob {
    const bf16 *G, *X;
    int64_t ldg, ldx;
    float *partial, *dbias, *brow;   // brow: [splits][N] column sums of G per split (the bias gradient's partial rows), or nullptr
    int N, K, first;          // first workgroup of this job; it owns (N/128)(K/128) tiles x splits workgroups
};
struct WgJobs {
    int n, splits;
    int64_t slice, M;
    WgJob j[3];
};
__global__ __launch_bounds__(256) void k_wgrad_ring_jobs(const WgJobs js) {
    int ji = 0;
#pragma unroll
    for (int q = 1; q < 3; ++q)
        if (q < js.n && (int)blockIdx.x >= js.j[q].first) ji = q;
    const WgJob& jb = js.j[ji];
    const int tiles_k = jb.K / 128, tiles = (jb.N / 128) * tiles_k, rel = blockIdx.x - jb.first;
    // XCD-aware order (round 4; speed only, the partial-tile layout and every result bit are unchanged): workgroups are dealt round-robin to the 8 XCDs, and the
    // `tiles` workgroups of one token split all stream the SAME X rows (LN(x): the three 128-column tiles of the qkv gradient, the two of kv / U|V).  With the
    // tile index fastest they sat on consecutive XCDs and each fetched its own copy through the fabric (2,048 B per token moved for 1,536 B of operands);
    // with blockIdx = first + 8 tiles grp + 8 tile + (z mod 8) they share an XCD and its L2.
    int tile, z;
    {
        const int full = (js.splits / 8) * 8 * tiles;        // workgroups of the complete groups of 8 splits
        if (rel < full) {
            const int grp = rel / (8 * tiles), in = rel - grp * 8 * tiles;
            z = grp * 8 + (in & 7);
            tile = in >> 3;
        } else {
            const int r2 = rel - full, rs = js.splits - (js.splits / 8) * 8;
            z = (js.splits / 8) * 8 + r2 % rs;
            tile = r2 / rs;
        }
    }
    wgrad_ring_body<bf16, true>(jb.G, jb.ldg, jb.X, jb.ldx, nullptr, 0, jb.dbias, js.M, js.slice, jb.partial, jb.brow, jb.N, jb.K, (tile / tiles_k) * 128,
                          (tile % tiles_k) * 128, z);
}
struct FinJob {
    const float* partial;
    float* out;               // [N][K] dense, accumulated into
    int N, K, first;          // first workgroup; N workgroups (one per output row of K = 128 columns)
    const float *W, *bias, *ls;   // W != nullptr (K == 128): out += ls[n] * G, dls[n] += <W[n], G[n]> + bias[n] * colsum[n], db[n] += ls[n] * colsum[n]
    float *db, *dls;
    const float* brow;            // [splits][N] per-split column sums of G (colsum = their fixed-order sum)
    int nsplit;                   // partial tiles of THIS job (round 6: a job whose tiles come out of a fused kernel -- one per workgroup -- rides beside streaming jobs)
};
struct FinRed {                // bf16 partial tiles of a fused data + weight gradient launch (k_dgrad_r<..., WG>): out[e] += sum_z part[z][e]
    const bf16* part;
    float* out;
    int nparts, elems, first;  // first workgroup; elems / 128 workgroups
};
struct FinJobs {
    int n, splits, nred;
    FinJob j[3];
    FinRed r[2];
};
__global__ __launch_bounds__(256) void k_wgrad_finish_jobs(const FinJobs js) {
    if (js.nred > 0 && (int)blockIdx.x >= js.r[0].first) {
        // ---- reduction role (round 4): a workgroup owns 128 consecutive elements (16 lanes x 8); its 16 lane groups each add every 16th partial tile,
        // the 16 sums meet in LDS and are added in a fixed tree: bitwise reproducible ----
        __shared__ float sR[16][128];
        const FinRed& rd = js.r[(js.nred > 1 && (int)blockIdx.x >= js.r[1].first) ? 1 : 0];
        const int l = threadIdx.x & 15, zl = threadIdx.x >> 4;
        const int64_t e = ((int64_t)((int)blockIdx.x - rd.first) * 16 + l) * 8;
        float s0[8], s1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s0[k] = 0.f; s1[k] = 0.f; }
        int z = zl;
        for (; z + 16 < rd.nparts; z += 32) {
            float a[8], b[8];
            load8(rd.part + (int64_t)z * rd.elems + e, a);
            load8(rd.part + (int64_t)(z + 16) * rd.elems + e, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s0[k] += a[k]; s1[k] += b[k]; }
        }
        if (z < rd.nparts) {
            float a[8];
            load8(rd.part + (int64_t)z * rd.elems + e, a);
#pragma unroll
            for (int k = 0; k < 8; ++k) s0[k] += a[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) sR[zl][l * 8 + k] = s0[k] + s1[k];
        __syncthreads();
        if (threadIdx.x < 128) {
            float t[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) t[q] = sR[q][threadIdx.x];
#pragma unroll
            for (int h = 8; h >= 1; h >>= 1)
#pragma unroll
                for (int q = 0; q < h; ++q) t[q] += t[q + h];
            rd.out[((int64_t)((int)blockIdx.x - rd.first)) * 128 + threadIdx.x] += t[0];
        }
        return;
    }
    int ji = 0;
#pragma unroll
    for (int q = 1; q < 3; ++q)
        if (q < js.n && (int)blockIdx.x >= js.j[q].first) ji = q;
    const FinJob& jb = js.j[ji];
    // ---- one output ROW (K = 128 columns) per workgroup (round 4: was 256 outputs per workgroup with four waves taking every 4th split -- the proj job alone has
    // 248 splits and 64 such workgroups: a 62-deep dependent load chain per wave, 14 us per launch).  16 lane groups take every 16th split, 16 lanes x 8 columns
    // each; the 16 sums meet in LDS and are added in a fixed tree; threads 128..255 add the split rows of the bias gradient meanwhile. ----
    __shared__ float sR[16][128];
    __shared__ float sDot[4];
    const int l = threadIdx.x & 15, zl = threadIdx.x >> 4, n = (int)blockIdx.x - jb.first;
    const int64_t e = (int64_t)n * 128 + l * 8, stride = (int64_t)jb.N * jb.K;
    float s0[8], s1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s0[k] = 0.f; s1[k] = 0.f; }
    int z = zl;
    for (; z + 16 < jb.nsplit; z += 32) {
        float a[8], b[8];
        load8(reinterpret_cast<const bf16*>(jb.partial) + (int64_t)z * stride + e, a);           // (the jobs launch leaves bf16 tiles)
        load8(reinterpret_cast<const bf16*>(jb.partial) + (int64_t)(z + 16) * stride + e, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s0[k] += a[k]; s1[k] += b[k]; }
    }
    if (z < jb.nsplit) {
        float a[8];
        load8(reinterpret_cast<const bf16*>(jb.partial) + (int64_t)z * stride + e, a);
#pragma unroll
        for (int k = 0; k < 8; ++k) s0[k] += a[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sR[zl][l * 8 + k] = s0[k] + s1[k];
    float gs = 0.f;                                      // colsum(g)[n] = fixed-order sum of the per-split rows (threads 128..255: two waves)
    if (jb.W != nullptr && threadIdx.x >= 128) {
        for (int z2 = threadIdx.x - 128; z2 < jb.nsplit; z2 += 128) gs += jb.brow[(int64_t)z2 * jb.N + n];
        gs = reduce64(gs);
        if ((threadIdx.x & 63) == 0) sDot[2 + ((threadIdx.x - 128) >> 6)] = gs;
    }
    __syncthreads();
    float dot = 0.f;
    if (threadIdx.x < 128) {
        float t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = sR[q][threadIdx.x];
#pragma unroll
        for (int h = 8; h >= 1; h >>= 1)
#pragma unroll
            for (int q = 0; q < h; ++q) t[q] += t[q + h];
        const float gsum = t[0];
        float* o = jb.out + (int64_t)n * 128 + threadIdx.x;
        if (jb.W != nullptr) {                           // proj: out += ls[n] G, and the row's share of dls
            dot = jb.W[(int64_t)n * 128 + threadIdx.x] * gsum;
            *o += jb.ls[n] * gsum;
        } else {
            *o += gsum;
        }
        if (jb.W != nullptr) {
            dot = reduce64(dot);
            if ((threadIdx.x & 63) == 0) sDot[threadIdx.x >> 6] = dot;
        }
    }
    if (jb.W != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const float g2 = sDot[2] + sDot[3], lsn = jb.ls[n];
            jb.dls[n] += (sDot[0] + sDot[1]) + jb.bias[n] * g2;
            jb.db[n] += g2 * lsn;
        }
    }
}

// out[n][k] += sum_z partial[z][n][k]   (N*K multiple of 256; out row stride ldo, partial row stride K).
// A workgroup owns 64 consecutive float4 outputs; its 4 waves each sum every 4th split with 4 independent
// loads in flight, then the partial sums meet in LDS (fixed order => bitwise reproducible gradients).
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ out, int64_t ldo, int N, int K, int splits,
                                                      const float* __restrict__ brow, float* __restrict__ dbias) {
    __shared__ f32x4 sPart[4][64];
    const int tile_wgs = (int)((int64_t)N * K / 256);
    if ((int)blockIdx.x >= tile_wgs) {                   // extra workgroups (64 columns each): dbias[n] += sum_z brow[z][n] in a fixed order --
        __shared__ float sB[4][64];                      // four split lanes per column, four independent chains per lane (the loads are what takes time)
        const int cl = threadIdx.x & 63, zl = threadIdx.x >> 6, n = ((int)blockIdx.x - tile_wgs) * 64 + cl;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (n < N) {
            int z = zl;
            for (; z + 12 < splits; z += 16) {
                b0 += brow[(int64_t)z * N + n];
                b1 += brow[(int64_t)(z + 4) * N + n];
                b2 += brow[(int64_t)(z + 8) * N + n];
                b3 += brow[(int64_t)(z + 12) * N + n];
            }
            for (; z < splits; z += 4) b0 += brow[(int64_t)z * N + n];
        }
        sB[zl][cl] = (b0 + b1) + (b2 + b3);
        __syncthreads();
        if (zl == 0 && n < N) dbias[n] += (sB[0][cl] + sB[1][cl]) + (sB[2][cl] + sB[3][cl]);
        return;
    }
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t e = ((int64_t)blockIdx.x * 64 + lane) * 4, stride = (int64_t)N * K;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int z = part;
    for (; z + 12 < splits; z += 16) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(partial + (int64_t)z * stride + e);
        const f32x4 b = *reinterpret_cast<const f32x4*>(partial + (int64_t)(z + 4) * stride + e);
        const f32x4 c = *reinterpret_cast<const f32x4*>(partial + (int64_t)(z + 8) * stride + e);
        const f32x4 d = *reinterpret_cast<const f32x4*>(partial + (int64_t)(z + 12) * stride + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) { s0[q] += a[q]; s1[q] += b[q]; s2[q] += c[q]; s3[q] += d[q]; }
    }
    for (; z < splits; z += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(partial + (int64_t)z * stride + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) s0[q] += a[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) s0[q] = (s0[q] + s1[q]) + (s2[q] + s3[q]);
    sPart[part][lane] = s0;
    __syncthreads();
    if (part == 0) {
        const f32x4 a = sPart[0][lane], b = sPart[1][lane], c = sPart[2][lane], d = sPart[3][lane];
        const int n = (int)(e / K), k = (int)(e % K);
        f32x4* o = reinterpret_cast<f32x4*>(out + (int64_t)n * ldo + k);
        f32x4 cur = *o;
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] += (a[q] + b[q]) + (c[q] + d[q]);
        *o = cur;
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
    auto put = [&](int64_t at, float v) {
        if (sizeof(T) == 2 && d.fp16) reinterpret_cast<f16*>(dst)[at] = (f16)v;      // master weights are O(0.1): no fp16 range concern; below 6e-5 they round on a 6e-8 grid
        else dst[at] = from_f<T>(v);
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (d.transpose) put((int64_t)(c0 + ty + 8 * k) * d.rows + r0 + tx, tile[tx][ty + 8 * k]);
        else put((int64_t)(r0 + ty + 8 * k) * d.cols + c0 + tx, tile[ty + 8 * k][tx]);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
#define DT_DISPATCH(dt, CALL_F32, CALL_BF16) \
    do { if ((dt) == KASF_F32) { CALL_F32; } else { CALL_BF16; } } while (0)

template <typename K> static void set_smem(K k, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// tile shapes per dtype: bf16 keeps two weight tiles in flight; fp32 (parity mode) has twice the bytes and single-buffers
template <typename T> struct GemmCfg;
template <> struct GemmCfg<bf16> { static constexpr int BM = 64, NBUF = 1, WG_NBUF = 2; };   // 48 KB of LDS -> 3 workgroups per CU
template <> struct GemmCfg<float> { static constexpr int BM = 64, NBUF = 1, WG_NBUF = 1; };

template <typename T>
static void linear_T(hipStream_t s, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc, int64_t M,
                     int N, const float* ln_g, const float* ln_b, void* xn_out, int act) {
    constexpr int BM = GemmCfg<T>::BM, NBUF = GemmCfg<T>::NBUF;
    const dim3 grid((unsigned)((M + BM - 1) / BM));
    const size_t sh = (size_t)(BM * 128 + NBUF * 128 * 128) * sizeof(T);
    auto go = [&](auto kern) {
        set_smem(kern, sh);
        hipLaunchKernelGGL(kern, grid, dim3(256), sh, s, (const T*)A, lda, (const T*)W, ldw, bias, (T*)C, ldc, M, N, ln_g, ln_b, (T*)xn_out);
    };
    if (ln_g != nullptr) { if (act == 1) go(k_linear<T, BM, NBUF, true, 1>); else go(k_linear<T, BM, NBUF, true, 0>); }
    else { if (act == 1) go(k_linear<T, BM, NBUF, false, 1>); else go(k_linear<T, BM, NBUF, false, 0>); }
}
void kasf_launch_linear(int dt, hipStream_t s, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                        int64_t M, int N, const float* ln_g, const float* ln_b, void* xn_out, int act) {
    if (dt == KASF_BF16 && act == 0 && lda == 128 && ldw == 128 && ldc == N && kasf_launch_linear_r(s, A, W, bias, C, M, N, ln_g, ln_b, xn_out)) return;
    DT_DISPATCH(dt, (linear_T<float>(s, A, lda, W, ldw, bias, C, ldc, M, N, ln_g, ln_b, xn_out, act)),
                (linear_T<bf16>(s, A, lda, W, ldw, bias, C, ldc, M, N, ln_g, ln_b, xn_out, act)));
}

template <typename T>
static void linear_res_T(hipStream_t s, const void* A, const void* W, const float* bias, const float* ls, const void* resid, void* C, int64_t M) {
    constexpr int BM = GemmCfg<T>::BM;
    const size_t sh = (size_t)(BM * 128 + 128 * 128) * sizeof(T);
    set_smem(k_linear_res<T, BM>, sh);
    hipLaunchKernelGGL((k_linear_res<T, BM>), dim3((unsigned)((M + BM - 1) / BM)), dim3(256), sh, s, (const T*)A, (const T*)W, bias, ls, (const T*)resid,
                       (T*)C, M);
}
void kasf_launch_linear_res(int dt, hipStream_t s, const void* A, const void* W, const float* bias, const float* ls, const void* resid, void* C,
                            int64_t M) {
    if (dt == KASF_BF16) { kasf_launch_linear_res_r(s, A, W, bias, ls, resid, C, M); return; }
    DT_DISPATCH(dt, (linear_res_T<float>(s, A, W, bias, ls, resid, C, M)), (linear_res_T<bf16>(s, A, W, bias, ls, resid, C, M)));
}

template <typename T>
static void dgrad_lnbwd_T(hipStream_t s, const void* dY, int Kd, const void* Wt, const void* dxn_add, const void* X, const float* gamma,
                          const void* resid, void* out, int accumulate, float* dgamma, float* dbeta, int64_t M, void* xn_out, const float* beta,
                          KasfColSink* sink) {
    constexpr int BM = GemmCfg<T>::BM, NBUF = GemmCfg<T>::NBUF;
    const size_t sh = (size_t)NBUF * (BM * 128 + 128 * 128) * sizeof(T);
    set_smem(k_dgrad_lnbwd<T, BM, NBUF>, sh);
    const int grid = (int)((M + BM - 1) / BM);
    float* part = sink != nullptr ? sink->take(grid, 256) : nullptr;
    hipLaunchKernelGGL((k_dgrad_lnbwd<T, BM, NBUF>), dim3((unsigned)grid), dim3(256), sh, s, (const T*)dY, Kd, (const T*)Wt,
                       (const T*)dxn_add, (const T*)X, gamma, (const T*)resid, (T*)out, accumulate, dgamma, dbeta, M, (T*)xn_out, beta, part);
    if (part != nullptr) { sink->add(part, 256, grid, 128, dgamma); sink->add(part + 128, 256, grid, 128, dbeta); }
}
void kasf_launch_dgrad_lnbwd(int dt, hipStream_t s, const void* dY, int Kd, const void* Wt, const void* dxn_add, const void* X, const float* gamma,
                             const void* resid, void* out, int accumulate, float* dgamma, float* dbeta, int64_t M, void* xn_out, const float* beta,
                             KasfColSink* sink) {
    if (M <= 0) return;
    if (dt == KASF_BF16 && kasf_launch_dgrad_r(s, dY, Kd, Wt, dxn_add, X, gamma, resid, out, accumulate, dgamma, dbeta, M, xn_out, beta, sink)) return;
    DT_DISPATCH(dt, (dgrad_lnbwd_T<float>(s, dY, Kd, Wt, dxn_add, X, gamma, resid, out, accumulate, dgamma, dbeta, M, xn_out, beta, sink)),
                (dgrad_lnbwd_T<bf16>(s, dY, Kd, Wt, dxn_add, X, gamma, resid, out, accumulate, dgamma, dbeta, M, xn_out, beta, sink)));
}

template <typename T>
static void wgrad_T(hipStream_t s, const void* G, int64_t ldg, int N, const void* X, int64_t ldx, int K, const float* ln_g, const float* ln_b,
                    float* out, int64_t ldo, float* dbias, int64_t M, float* partial, int64_t partial_floats) {
    constexpr int NBUF = GemmCfg<T>::WG_NBUF;
    const int tiles = (N / 128) * (K / 128);
    // one workgroup per CU (the 128 KB of LDS allow only one resident anyway): fewest splits that still fill the chip
    constexpr int target = 248;                          // (496 = two per CU measured slower: the partial tiles double)
    int splits = (target + tiles - 1) / tiles;
    const int64_t max_splits = (M + WG_BM - 1) / WG_BM;
    if (splits > max_splits) splits = (int)max_splits;
    if (splits < 1) splits = 1;
    int64_t slice = (M + splits - 1) / splits;
    slice = (slice + WG_BM - 1) / WG_BM * WG_BM;
    splits = (int)((M + slice - 1) / slice);
    if (partial != nullptr && (int64_t)splits * N * K + (int64_t)splits * N > partial_floats) partial = nullptr;     // fall back to atomics
    float* brow = (partial != nullptr && dbias != nullptr) ? partial + (int64_t)splits * N * K : nullptr;      // [splits][N] rows of the bias gradient
    const size_t sh = (size_t)2 * NBUF * 128 * 128 * sizeof(T);
    const dim3 grid(N / 128, K / 128, splits);
    if (ln_g == nullptr && sizeof(T) == 2) {
        const size_t shr = (size_t)WR_ST * 2 * WR_BM * 128 * sizeof(T);
        set_smem(k_wgrad_ring<T>, shr);
        hipLaunchKernelGGL(k_wgrad_ring<T>, grid, dim3(256), shr, s, (const T*)G, ldg, (const T*)X, ldx, out, ldo, dbias, M, slice, partial, brow);
    } else if (ln_g != nullptr) {
        set_smem(k_wgrad<T, NBUF, true>, sh);
        hipLaunchKernelGGL((k_wgrad<T, NBUF, true>), grid, dim3(256), sh, s, (const T*)G, ldg, (const T*)X, ldx, ln_g, ln_b, out, ldo, dbias, M, slice,
                           partial, brow);
    } else {
        set_smem(k_wgrad<T, NBUF, false>, sh);
        hipLaunchKernelGGL((k_wgrad<T, NBUF, false>), grid, dim3(256), sh, s, (const T*)G, ldg, (const T*)X, ldx, ln_g, ln_b, out, ldo, dbias, M, slice,
                           partial, brow);
    }
    if (partial != nullptr)                              // fixed-order sum of the per-split tiles (+ (N + 63) / 64 workgroups for the bias rows)
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((int64_t)N * K / 256 + (brow != nullptr ? (N + 63) / 64 : 0))), dim3(256), 0, s, partial, out, ldo, N, K,
                           splits, brow, dbias);
}
void kasf_launch_wgrad(int dt, hipStream_t s, const void* G, int64_t ldg, int N, const void* X, int64_t ldx, int K, const float* ln_g,
                       const float* ln_b, float* out, int64_t ldo, float* dbias, int64_t M, float* partial, int64_t partial_floats) {
    DT_DISPATCH(dt, (wgrad_T<float>(s, G, ldg, N, X, ldx, K, ln_g, ln_b, out, ldo, dbias, M, partial, partial_floats)),
                (wgrad_T<bf16>(s, G, ldg, N, X, ldx, K, ln_g, ln_b, out, ldo, dbias, M, partial, partial_floats)));
}

void kasf_launch_pack(int dt, hipStream_t s, const float* params, void* arena, const KasfPackDesc* desc, const int* tile_start, int ndesc,
                      int total_tiles) {
    if (total_tiles <= 0) return;
    if (dt == KASF_F32) hipLaunchKernelGGL(k_pack<float>, dim3(total_tiles), dim3(256), 0, s, params, (float*)arena, desc, tile_start, ndesc);
    else hipLaunchKernelGGL(k_pack<bf16>, dim3(total_tiles), dim3(256), 0, s, params, (bf16*)arena, desc, tile_start, ndesc);
}

// The finish launch of a block whose proj gradient came out of a fused data-gradient kernel (no streaming jobs launch): the proj job's fixed-order sum + layer-scale
// algebra over `nparts` bf16 tiles and bias rows, and the block's other bf16 partial-tile reductions, in one launch.
void kasf_launch_proj_finish(hipStream_t s, const void* proj_part, const float* proj_brow, int nparts, float* dW, const float* W, const float* bias, const float* ls,
                             float* db, float* dls, int nred, const KasfBf16Reduce* red) {
    if (nparts < 1 || nred < 0 || nred > 2) return;
    FinJobs fj;
    fj.n = 1; fj.splits = nparts; fj.nred = nred;
    fj.j[0] = FinJob{(const float*)proj_part, dW, 128, 128, 0, W, bias, ls, db, dls, proj_brow, nparts};
    int first = 128;
    for (int k = 0; k < nred; ++k) {
        fj.r[k] = FinRed{(const bf16*)red[k].part, red[k].out, red[k].nparts, red[k].elems, first};
        first += red[k].elems / 128;
    }
    hipLaunchKernelGGL(k_wgrad_finish_jobs, dim3(first), dim3(256), 0, s, fj);
}

void kasf_launch_bf16_reduce(hipStream_t s, int nred, const KasfBf16Reduce* red) {
    if (nred < 1 || nred > 2) return;
    FinJobs fj;
    fj.n = 0; fj.splits = 0; fj.nred = nred;
    int first = 0;
    for (int k = 0; k < nred; ++k) {
        fj.r[k] = FinRed{(const bf16*)red[k].part, red[k].out, red[k].nparts, red[k].elems, first};
        first += red[k].elems / 128;
    }
    hipLaunchKernelGGL(k_wgrad_finish_jobs, dim3(first), dim3(256), 0, s, fj);
}

// Up to three bf16 weight gradients  dW_j[N_j][128] += G_j^T X_j  (dense operands: ldg = N_j, ldx = 128) in one streaming launch plus one
// finishing launch.  fin_* of job j (optional): the layer-scale algebra of k_finalize_ls applied to that job (the proj weight).
// Returns false (nothing launched) when the scratch is too small.
bool kasf_launch_wgrad_jobs(hipStream_t s, int njobs, const void* const* G, const void* const* X, const int* N, float* const* dW, float* const* dbias,
                            int fin_job, const float* fin_W, const float* fin_bias, const float* fin_ls, float* fin_dls, int64_t M, float* partial,
                            int64_t partial_floats, int nred, const KasfBf16Reduce* red, const void* ext_part, const float* ext_brow, int ext_nparts,
                            float* ext_dW, float* ext_db) {
    // ext_* (round 6): the proj job when its tiles G = g_mid^T o and rows colsum(g_mid) were accumulated by the fused attention-block backward (one bf16 tile and one row per
    // workgroup): it takes no part in the streaming launch and is finished -- fin_W / fin_bias / fin_ls / fin_dls are then ITS layer-scale algebra -- by the same finish launch
    if (njobs < 1 || njobs > 3 || M <= 0 || nred < 0 || nred > 2) return false;
    if (ext_part != nullptr && (njobs > 2 || fin_job >= 0 || ext_brow == nullptr || ext_nparts < 1 || ext_dW == nullptr || ext_db == nullptr)) return false;
    int tiles = 0;
    for (int j = 0; j < njobs; ++j) tiles += N[j] / 128;
    const int target = kasf_narrow_grid(KASF_NG_WGRAD, 248, M);
    int splits = (target + tiles - 1) / tiles;
    const int64_t max_splits = (M + WG_BM - 1) / WG_BM;
    if (splits > max_splits) splits = (int)max_splits;
    int64_t slice = (M + splits - 1) / splits;
    slice = (slice + WG_BM - 1) / WG_BM * WG_BM;
    splits = (int)((M + slice - 1) / slice);
    int64_t need = 0;
    for (int j = 0; j < njobs; ++j) {
        need += (int64_t)splits * N[j] * 128 + (dbias[j] != nullptr ? (int64_t)splits * N[j] : 0);
        if (dbias[j] != nullptr && j != fin_job) return false;        // a bias gradient is finished together with the layer-scale algebra only
    }
    if (partial == nullptr || need > partial_floats) return false;
    WgJobs js;
    FinJobs fj;
    js.n = fj.n = njobs; js.splits = fj.splits = splits; js.slice = slice; js.M = M;
    fj.nred = nred;
    int first = 0, ffirst = 0;
    float* pp = partial;
    for (int j = 0; j < njobs; ++j) {
        float* brow = dbias[j] != nullptr ? pp + (int64_t)splits * N[j] * 128 : nullptr;      // per-split rows of the bias gradient, behind the job's tiles
        js.j[j] = WgJob{(const bf16*)G[j], (const bf16*)X[j], N[j], 128, pp, dbias[j], brow, N[j], 128, first};
        const bool fin = j == fin_job;
        fj.j[j] = FinJob{pp, dW[j], N[j], 128, ffirst, fin ? fin_W : nullptr, fin_bias, fin_ls, dbias[j], fin_dls, brow, splits};
        first += (N[j] / 128) * splits;
        ffirst += N[j];                                   // one finishing workgroup per output row
        pp += (int64_t)splits * N[j] * 128 + (brow != nullptr ? (int64_t)splits * N[j] : 0);
    }
    if (ext_part != nullptr) {
        fj.j[njobs] = FinJob{(const float*)ext_part, ext_dW, 128, 128, ffirst, fin_W, fin_bias, fin_ls, ext_db, fin_dls, ext_brow, ext_nparts};
        fj.n = njobs + 1;
        ffirst += 128;
    }
    const size_t shr = (size_t)WR_ST * 2 * WR_BM * 128 * sizeof(bf16);
    set_smem(k_wgrad_ring_jobs, shr);
    for (int k = 0; k < nred; ++k) {                     // the bf16 partial tiles of the block's fused data + weight gradient launches ride in the same finish
        fj.r[k] = FinRed{(const bf16*)red[k].part, red[k].out, red[k].nparts, red[k].elems, ffirst};
        ffirst += red[k].elems / 128;
    }
    hipLaunchKernelGGL(k_wgrad_ring_jobs, dim3(first), dim3(256), shr, s, js);
    hipLaunchKernelGGL(k_wgrad_finish_jobs, dim3(ffirst), dim3(256), 0, s, fj);
    return true;
}
