// End of the MLP backward (bf16): fixed-order sum of the per-range partial tiles of dW1 [512][128] and dW2 [128][512] and, for fc2, the layer-scale algebra that
// k_finalize_ls would apply: with G = g^T H (unscaled),   dls[c] += sum_k W2[c][k] G[c][k];   dW2[c][:] += ls[c] G[c][:]
// (the colsum(g) terms -- b2 . gsum into dls, db2 = ls . gsum -- are applied by the stage's k_col_finish: the column sums are complete only after this launch).
// 256 "fin blocks" of 256 threads: blocks 0..127 own 4 rows of dW1 each, blocks 128..255 one row of dW2 each (512 floats per block); the two halves of a block take the
// even / odd splits.  Shared by k_lnbwd_sum4_fin (k_mlp2.hip: 256 extra 256-thread workgroups) and k_dgrad_r<..., MLPFIN> (k_gemm2.hip: 128 extra 512-thread workgroups,
// two fin blocks each).
#pragma once
#include "common.h"

struct MlpFinArgs {
    const bf16 *p1, *p2;       // bf16 partial tiles per token range: [ranges][512][128] and [ranges][128][512]
    float *dW1, *dW2;
    int splits;
    const float *W2, *b2, *ls;
    float* dls;
};

__device__ __forceinline__ f32x4 mlp_fin_ld4(const bf16* p) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
}
// tid: 0..255 inside the fin block; sHalf [128] and sDot [2]: LDS private to the fin block; every thread of the WORKGROUP must call this (it contains workgroup barriers:
// one, plus one more when the fin block finishes fc2 -- all fin blocks of a workgroup are on the same side of bid 128)
__device__ __forceinline__ void mlp_wfinish_body(const MlpFinArgs& fa, int bid, int tid, f32x4* sHalf, float* sDot) {
    const int lane = tid & 127, half = tid >> 7;
    const bool second = bid >= 128;
    const int blk = second ? bid - 128 : bid;
    const bf16* part = second ? fa.p2 : fa.p1;
    const int splits = fa.splits;
    const int64_t e = (int64_t)blk * 512 + lane * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    int z = half;
    for (; z + 6 < splits; z += 8) {                    // four partial tiles in flight per thread; the order of the additions is the two-at-a-time loop's
        const f32x4 u0 = mlp_fin_ld4(part + (int64_t)z * 65536 + e);
        const f32x4 v0 = mlp_fin_ld4(part + (int64_t)(z + 2) * 65536 + e);
        const f32x4 u1 = mlp_fin_ld4(part + (int64_t)(z + 4) * 65536 + e);
        const f32x4 v1 = mlp_fin_ld4(part + (int64_t)(z + 6) * 65536 + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] += u0[q]; b[q] += v0[q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] += u1[q]; b[q] += v1[q]; }
    }
    for (; z + 2 < splits; z += 4) {
        const f32x4 u = mlp_fin_ld4(part + (int64_t)z * 65536 + e);
        const f32x4 v = mlp_fin_ld4(part + (int64_t)(z + 2) * 65536 + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] += u[q]; b[q] += v[q]; }
    }
    if (z < splits) {
        const f32x4 u = mlp_fin_ld4(part + (int64_t)z * 65536 + e);
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
    const bool fin = second && fa.W2 != nullptr;
    float dot = 0.f;
    if (half == 0) {
        float* dst = (second ? fa.dW2 : fa.dW1) + e;
        f32x4 cur = *reinterpret_cast<f32x4*>(dst);
        const float l = fin ? fa.ls[blk] : 1.0f;
        if (fin) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(fa.W2 + e);
#pragma unroll
            for (int q = 0; q < 4; ++q) dot += w[q] * a[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] += l * a[q];
        *reinterpret_cast<f32x4*>(dst) = cur;
    }
    if (fin) {
        dot = reduce64(dot);
        if (half == 0 && (tid & 63) == 0) sDot[tid >> 6] = dot;
        __syncthreads();
        if (tid == 0) fa.dls[blk] += sDot[0] + sDot[1];      // this fin block owns row blk: a plain read-modify-write
    }
}
