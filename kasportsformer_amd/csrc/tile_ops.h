// Tile-level helpers shared by the GEMM-family and fused-MLP kernels.
#pragma once
#include "common.h"

// wave w of a 256-thread workgroup owns output features [64*(w&1), +64) x tokens [64*(w>>1), +64)
__device__ __forceinline__ int wave_n0() { return ((threadIdx.x >> 6) & 1) * 64; }
__device__ __forceinline__ int wave_m0() { return (threadIdx.x >> 7) * 64; }
template <int BM> __device__ __forceinline__ int wave_m0_bm() { return (threadIdx.x >> 7) * (BM / 2); }
// NTHR/64 waves as 2 (feature halves) x NTHR/128 (token groups)
template <int BM, int NTHR> __device__ __forceinline__ int wave_m0_n() { return (threadIdx.x >> 7) * (BM / (NTHR / 128)); }

// Epilogue straight from the accumulators: each lane owns 4 consecutive output features of one token, so it
// issues one 8-byte (bf16) / 16-byte (f32) global access per tile; f(v, row, n) transforms and stores.
template <int NT, int MT, typename F>
__device__ __forceinline__ void acc_foreach(f32x4 (&acc)[NT][MT], int wn0, int wm0, int64_t row0, int64_t M, F f) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t row = row0 + wm0 + mt * 16 + i;
        if (row < M) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
                f(v, row, wn0 + nt * 16 + g * 4);
            }
        }
    }
}

template <typename T, int NT, int MT, typename F>
__device__ __forceinline__ void acc_to_tile(T* sC, f32x4 (&acc)[NT][MT], int wn0, int wm0, F f) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int n = wn0 + nt * 16 + g * 4, m = wm0 + mt * 16 + i;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = f(acc[nt][mt][r], n + r);
            store4(sC + Tile<T>::off4(m, n), v);
        }
}


// LayerNorm backward over the BM x 128 tile of d(LN output) held in sC (swizzled, type T):
//   out = [resid +] [out +] rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat)),  dxh = (sC [+ dxn_add]) * gamma
// and dgamma += sum_m d*xhat, dbeta += sum_m d: this workgroup's sums go to row blockIdx.x of `part` ([dgamma | dbeta], 256 floats, added in a
// fixed order by k_col_finish) or, without scratch (part == nullptr), into the gradients by fp32 atomics.  xhat/rstd are
// recomputed from X.  `red` is >= NTHR * 64 B of LDS that is free at this point (must not alias sC).
template <typename T, int BM, int NTHR = 256>
__device__ __forceinline__ void lnbwd_rows(const T* sC, const T* __restrict__ X, const float* __restrict__ gamma, const T* __restrict__ dxn_add,
                                           const T* __restrict__ resid, T* __restrict__ out, int accumulate, float* __restrict__ dgamma,
                                           float* __restrict__ dbeta, int64_t row0, int64_t M, float* red, T* __restrict__ xn_out = nullptr,
                                           const float* __restrict__ beta = nullptr, float* __restrict__ part = nullptr) {
    const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    float gm[8], dg[8], db[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { gm[i] = gamma[sub * 8 + i]; dg[i] = 0.f; db[i] = 0.f; }
    constexpr int RS = NTHR / 16;
    for (int r = rl; r < BM; r += RS) {
        const int64_t row = row0 + r;
        if (row >= M) break;        // uniform across the 16 lanes that share a row
        float d[8], x[8];
        tile_load8(sC, r, sub * 8, d);
        if (dxn_add != nullptr) {
            float a[8];
            load8(dxn_add + row * 128 + sub * 8, a);
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] += a[i];
        }
        load8(X + row * 128 + sub * 8, x);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[i];
        const float mean = reduce16(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[i] -= mean; q += x[i] * x[i]; }
        const float rstd = rsqrtf(reduce16(q) * (1.0f / 128.0f) + KASF_LN_EPS);
        float s1 = 0.f, s2 = 0.f;
        if (xn_out != nullptr) {                // LN(x) = xhat*gamma+beta: operand of the matching weight-gradient GEMM
            float xn[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xn[i] = x[i] * rstd * gm[i] + beta[sub * 8 + i];
            store8(xn_out + row * 128 + sub * 8, xn);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x[i] *= rstd;                       // xhat
            dg[i] += d[i] * x[i];
            db[i] += d[i];
            d[i] *= gm[i];                      // dxhat
            s1 += d[i];
            s2 += d[i] * x[i];
        }
        s1 = reduce16(s1) * (1.0f / 128.0f);
        s2 = reduce16(s2) * (1.0f / 128.0f);
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rstd * (d[i] - s1 - x[i] * s2);
        if (resid != nullptr) {
            float a[8];
            load8(resid + row * 128 + sub * 8, a);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += a[i];
        }
        if (accumulate) {
            float a[8];
            load8(out + row * 128 + sub * 8, a);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += a[i];
        }
        store8(out + row * 128 + sub * 8, o);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[rl * 128 + sub * 8 + i] = dg[i];
        red[RS * 128 + rl * 128 + sub * 8 + i] = db[i];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int c = threadIdx.x & 127, which = threadIdx.x >> 7;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < RS; ++k) s += red[which * RS * 128 + k * 128 + c];
        if (part != nullptr) part[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
        else atomicAdd((which ? dbeta : dgamma) + c, s);
    }
}

// ---- transposed fragments: the reduction index is the ROW index of a row-major swizzled [m][128] tile ----
template <typename T> __device__ __forceinline__ int eoff(int row, int col) { return Tile<T>::chunk_off(row, col / Tile<T>::EPC) + (col % Tile<T>::EPC); }

// Row order inside a group of 8 reduction rows (round 6).  The swizzle puts element (row, col) at 16-byte slot (col / 8) ^ (row & 15) of its 256-byte row, and a
// ds_read_b64_tr_b16 is serviced in two 32-lane halves whose lanes read 4 rows x 2 adjacent chunks each (c0 = col0 / 8, always even, and c0 + 1): with CONSECUTIVE
// rows {m, m+1, m+2, m+3} and {m+8, ..., m+11} slot c0 ^ r of row r is also slot (c0 + 1) ^ (r ^ 1) of row r ^ 1 -- every lane shares its two banks with one
// other lane (2-way conflict on every transposed read: SQ_LDS_BANK_CONFLICT = 33 % of SQ_LDS_IDX_ACTIVE in k_mlp_bwd_s, rounds 4-5; profiles/r6_mlp_sq_counters.txt).
// With the EVEN rows of the group in the first read and the ODD rows in the second the 32 lanes of a half touch 16 distinct slots x 2 halves = all 64 banks once.
// The reduction index inside an MFMA k-step is then a permutation (j < 4: row m + 2j, j >= 4: row m + 2(j - 4) + 1) -- the same one for BOTH operands of every
// product that uses frag_tr (all of them take both from this function), so the sums are the same sums.
#ifndef KASF_FRAG_TR_INTERLEAVE
#define KASF_FRAG_TR_INTERLEAVE 1
#endif
__device__ __forceinline__ bf16x8 frag_tr(const bf16* s, int mbase, int col0) {
    // group of 16 lanes: lane u = 4q+p supplies one row of the group, columns col0+4p..; lane u receives column col0+u of the 4 rows supplied by q = 0..3
    const int u = threadIdx.x & 15, q = u >> 2, p = u & 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
#if KASF_FRAG_TR_INTERLEAVE
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s + eoff<bf16>(mbase + 2 * q, col0 + 4 * p)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s + eoff<bf16>(mbase + 2 * q + 1, col0 + 4 * p)));
#else
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s + eoff<bf16>(mbase + q, col0 + 4 * p)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(s + eoff<bf16>(mbase + 4 + q, col0 + 4 * p)));
#endif
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

