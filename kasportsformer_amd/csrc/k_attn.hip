// Attention cores (reference: modules/selfattention.py:18-41, modules/bone_crossattention.py:19-41).
// The problems are tiny and numerous: per (group, head) a [L x 16] . [16 x L] score matrix with
// L = 17 (spatial: the joints of one frame) or L = T (temporal: one joint's track).  One workgroup
// owns one group; K and V of the group live in LDS as fp32, one thread owns one (query, head) row
// (exact max-subtracted softmax in two sweeps over the keys, scale 16^-0.5 folded into q).
// Token of (group G, position i):  spatial  tok = 17*G + i            (G = b*T + t)
//                                  temporal tok = (G/17)*T*17 + 17*i + G%17   (G = b*17 + j)
// Backward recomputes the probabilities (nothing but q,k,v,dO is read):
//   phase 1 (thread = query row):  P, delta = sum_j P dP, dS = P (dP - delta) * scale, dQ = dS K
//   phase 2 (thread = key row):    dK = dS^T Q, dV = P^T dO   (P, dS recomputed from per-row max/sum/delta)
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int64_t tok_of(int G, int i, int T, int mode) {
    return mode == 0 ? (int64_t)G * KASF_J + i : (int64_t)(G / KASF_J) * T * KASF_J + (int64_t)i * KASF_J + (G % KASF_J);
}

template <typename T> __device__ __forceinline__ void load16(const T* p, float (&v)[16]) {
    float a[8], b[8];
    load8(p, a);
    load8(p + 8, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = a[i]; v[8 + i] = b[i]; }
}
template <typename T> __device__ __forceinline__ void store16(T* p, const float (&v)[16]) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = v[i]; b[i] = v[8 + i]; }
    store8(p, a);
    store8(p + 8, b);
}
__device__ __forceinline__ float dot16(const float (&a)[16], const float* b) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 16; d += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(b + d);
        s += a[d] * t[0] + a[d + 1] * t[1] + a[d + 2] * t[2] + a[d + 3] * t[3];
    }
    return s;
}
__device__ __forceinline__ void axpy16(float (&acc)[16], float a, const float* b) {
#pragma unroll
    for (int d = 0; d < 16; d += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(b + d);
        acc[d] += a * t[0]; acc[d + 1] += a * t[1]; acc[d + 2] += a * t[2]; acc[d + 3] += a * t[3];
    }
}

// copy [L rows] x [HP*16 cols] (head window h0) of a token-strided tensor into LDS as fp32
template <typename T>
__device__ __forceinline__ void stage_group(float* s, const T* src, int64_t ld, int G, int L, int Tn, int mode, int h0, int HP, float scale) {
    const int CH = HP * 2;                      // 8-element chunks per row
    for (int idx = threadIdx.x; idx < L * CH; idx += blockDim.x) {
        const int i = idx / CH, c = idx % CH;
        float v[8];
        load8(src + tok_of(G, i, Tn, mode) * ld + h0 * 16 + c * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[i * (HP * 16) + c * 8 + e] = v[e] * scale;
    }
}

// No per-thread score arrays (an earlier version kept L scores in registers, spilled to scratch and
// miscomputed): every pass recomputes the 16-wide dot products from LDS.
template <typename T>
__global__ __launch_bounds__(256) void k_attn_fwd(const T* __restrict__ Q, int64_t ldq, const T* __restrict__ K, const T* __restrict__ V, int64_t ldkv,
                                                  T* __restrict__ O, int L, int Tn, int mode, int HP) {
    const int W = HP * 16;                        // HP heads at a time (8 unless the track is too long for K and V of all heads to fit LDS)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sK = reinterpret_cast<float*>(smem);
    float* sV = sK + L * W;
    const int G = blockIdx.x;
    for (int h0 = 0; h0 < 8; h0 += HP) {
        __syncthreads();
        stage_group<T>(sK, K, ldkv, G, L, Tn, mode, h0, HP, 1.0f);
        stage_group<T>(sV, V, ldkv, G, L, Tn, mode, h0, HP, 1.0f);
        __syncthreads();
        for (int item = threadIdx.x; item < L * HP; item += blockDim.x) {
            const int i = item / HP, h = item % HP;
            const int64_t tok = tok_of(G, i, Tn, mode);
            float q[16];
            load16(Q + tok * ldq + (h0 + h) * 16, q);
#pragma unroll
            for (int d = 0; d < 16; ++d) q[d] *= 0.25f;
            float mx = -INFINITY;
            for (int j = 0; j < L; ++j) mx = fmaxf(mx, dot16(q, sK + j * W + h * 16));
            float sum = 0.f, o[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) o[d] = 0.f;
            for (int j = 0; j < L; ++j) {
                const float e = __expf(dot16(q, sK + j * W + h * 16) - mx);
                sum += e;
                axpy16(o, e, sV + j * W + h * 16);
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int d = 0; d < 16; ++d) o[d] *= inv;
            store16(O + tok * 128 + (h0 + h) * 16, o);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_attn_bwd(const T* __restrict__ Q, int64_t ldq, const T* __restrict__ K, const T* __restrict__ V, int64_t ldkv,
                                                  const T* __restrict__ dO, T* __restrict__ dQ, int64_t lddq, T* __restrict__ dK, T* __restrict__ dV,
                                                  int64_t lddkv, int L, int Tn, int mode, int HP) {
    const int W = HP * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sQ = reinterpret_cast<float*>(smem);   // pre-scaled by 0.25
    float* sK = sQ + L * W;
    float* sV = sK + L * W;
    float* sD = sV + L * W;                       // dO
    float* sStat = sD + L * W;                    // [L*HP][4] : max, 1/sum, delta
    const int G = blockIdx.x;
    for (int h0 = 0; h0 < 8; h0 += HP) {
        __syncthreads();
        stage_group<T>(sQ, Q, ldq, G, L, Tn, mode, h0, HP, 0.25f);
        stage_group<T>(sK, K, ldkv, G, L, Tn, mode, h0, HP, 1.0f);
        stage_group<T>(sV, V, ldkv, G, L, Tn, mode, h0, HP, 1.0f);
        stage_group<T>(sD, dO, 128, G, L, Tn, mode, h0, HP, 1.0f);
        __syncthreads();
        // ---- phase 1: one thread per (query i, head): softmax statistics, delta, dQ ----
        for (int item = threadIdx.x; item < L * HP; item += blockDim.x) {
            const int i = item / HP, h = item % HP;
            float q[16], d_o[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) { q[d] = sQ[i * W + h * 16 + d]; d_o[d] = sD[i * W + h * 16 + d]; }
            float mx = -INFINITY;
            for (int j = 0; j < L; ++j) mx = fmaxf(mx, dot16(q, sK + j * W + h * 16));
            float sum = 0.f, acc = 0.f;
            for (int j = 0; j < L; ++j) {
                const float e = __expf(dot16(q, sK + j * W + h * 16) - mx);
                sum += e;
                acc += e * dot16(d_o, sV + j * W + h * 16);
            }
            const float inv = 1.0f / sum, delta = acc * inv;
            float dq[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) dq[d] = 0.f;
            for (int j = 0; j < L; ++j) {
                const float pj = __expf(dot16(q, sK + j * W + h * 16) - mx) * inv;
                const float dpj = dot16(d_o, sV + j * W + h * 16);
                axpy16(dq, pj * (dpj - delta) * 0.25f, sK + j * W + h * 16);
            }
            store16(dQ + tok_of(G, i, Tn, mode) * lddq + (h0 + h) * 16, dq);
            sStat[item * 4 + 0] = mx;
            sStat[item * 4 + 1] = inv;
            sStat[item * 4 + 2] = delta;
        }
        __syncthreads();
        // ---- phase 2: one thread per (key j, head): dK, dV ----
        for (int item = threadIdx.x; item < L * HP; item += blockDim.x) {
            const int j = item / HP, h = item % HP;
            float kk[16], vv[16], dk[16], dv[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) { kk[d] = sK[j * W + h * 16 + d]; vv[d] = sV[j * W + h * 16 + d]; dk[d] = 0.f; dv[d] = 0.f; }
            for (int i = 0; i < L; ++i) {
                const float* st = sStat + (i * HP + h) * 4;
                const float pij = __expf(dot16(kk, sQ + i * W + h * 16) - st[0]) * st[1];
                const float dpij = dot16(vv, sD + i * W + h * 16);
                const float ds = pij * (dpij - st[2]);            // scale is already inside sQ
                axpy16(dk, ds, sQ + i * W + h * 16);
                axpy16(dv, pij, sD + i * W + h * 16);
            }
            const int64_t tok = tok_of(G, j, Tn, mode);
            store16(dK + tok * lddkv + (h0 + h) * 16, dk);
            store16(dV + tok * lddkv + (h0 + h) * 16, dv);
        }
    }
}

template <typename K> void set_smem(K k, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T>
void fwd_T(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int B, int Tn, int mode) {
    const int L = mode == 0 ? KASF_J : Tn, groups = mode == 0 ? B * Tn : B * KASF_J;
    int HP = 8;
    auto bytes = [&](int hp) { return (size_t)2 * L * hp * 16 * sizeof(float); };
    while (HP > 1 && bytes(HP) > 128 * 1024) HP >>= 1;
    if (bytes(HP) > 160 * 1024) { kasf_set_error(3, "attention: n_frames too large for the LDS-resident kernel"); return; }
    int threads = ((L * HP + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    set_smem(k_attn_fwd<T>, bytes(HP));
    hipLaunchKernelGGL(k_attn_fwd<T>, dim3(groups), dim3(threads), bytes(HP), s, (const T*)q, ldq, (const T*)k, (const T*)v, ldkv, (T*)o, L, Tn, mode, HP);
}
template <typename T>
void bwd_T(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq, int64_t lddq, void* dk,
           void* dv, int64_t lddkv, int B, int Tn, int mode) {
    const int L = mode == 0 ? KASF_J : Tn, groups = mode == 0 ? B * Tn : B * KASF_J;
    int HP = 8;
    auto bytes = [&](int hp) { return (size_t)(4 * L * hp * 16 + L * hp * 4) * sizeof(float); };
    while (HP > 1 && bytes(HP) > 96 * 1024) HP >>= 1;
    if (bytes(HP) > 160 * 1024) { kasf_set_error(3, "attention backward: n_frames too large for the LDS-resident kernel"); return; }
    int threads = ((L * HP + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    set_smem(k_attn_bwd<T>, bytes(HP));
    hipLaunchKernelGGL(k_attn_bwd<T>, dim3(groups), dim3(threads), bytes(HP), s, (const T*)q, ldq, (const T*)k, (const T*)v, ldkv, (const T*)d_o, (T*)dq,
                       lddq, (T*)dk, (T*)dv, lddkv, L, Tn, mode, HP);
}

}  // namespace

void kasf_launch_attn_fwd(int dt, hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int B, int T,
                          int mode) {
    if (dt == KASF_F32) fwd_T<float>(s, q, ldq, k, v, ldkv, o, B, T, mode);
    else if (!kasf_launch_attn_fwd_mfma(s, q, ldq, k, v, ldkv, o, B, T, mode)) fwd_T<bf16>(s, q, ldq, k, v, ldkv, o, B, T, mode);
}
void kasf_launch_attn_bwd(int dt, hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq,
                          int64_t lddq, void* dk, void* dv, int64_t lddkv, int B, int T, int mode) {
    if (dt == KASF_F32) bwd_T<float>(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, T, mode);
    else if (!kasf_launch_attn_bwd_mfma(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, T, mode))
        bwd_T<bf16>(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, T, mode);
}
