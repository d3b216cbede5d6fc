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

// D = head dimension = 128 / num_heads (16 in every shipped yaml; the reference's constructor default num_heads=4 gives 32)
template <typename T, int D> __device__ __forceinline__ void load16(const T* p, float (&v)[D]) {
#pragma unroll
    for (int c = 0; c < D; c += 8) {
        float a[8];
        load8(p + c, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c + i] = a[i];
    }
}
template <typename T, int D> __device__ __forceinline__ void store16(T* p, const float (&v)[D]) {
#pragma unroll
    for (int c = 0; c < D; c += 8) {
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = v[c + i];
        store8(p + c, a);
    }
}
template <int D> __device__ __forceinline__ float dot16(const float (&a)[D], const float* b) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(b + d);
        s += a[d] * t[0] + a[d + 1] * t[1] + a[d + 2] * t[2] + a[d + 3] * t[3];
    }
    return s;
}
template <int D> __device__ __forceinline__ void axpy16(float (&acc)[D], float a, const float* b) {
#pragma unroll
    for (int d = 0; d < D; d += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(b + d);
        acc[d] += a * t[0]; acc[d + 1] += a * t[1]; acc[d + 2] += a * t[2]; acc[d + 3] += a * t[3];
    }
}

// copy [L rows] x [HP*16 cols] (head window h0) of a token-strided tensor into LDS as fp32
template <typename T, int D>
__device__ __forceinline__ void stage_group(float* s, const T* src, int64_t ld, int G, int L, int Tn, int mode, int h0, int HP, float scale) {
    const int CH = HP * D / 8;                  // 8-element chunks per row
    for (int idx = threadIdx.x; idx < L * CH; idx += blockDim.x) {
        const int i = idx / CH, c = idx % CH;
        float v[8];
        load8(src + tok_of(G, i, Tn, mode) * ld + h0 * D + c * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[i * (HP * D) + c * 8 + e] = v[e] * scale;
    }
}

// No per-thread score arrays (an earlier version kept L scores in registers, spilled to scratch and
// miscomputed): every pass recomputes the 16-wide dot products from LDS.
template <typename T, int D>
__global__ __launch_bounds__(256) void k_attn_fwd(const T* __restrict__ Q, int64_t ldq, const T* __restrict__ K, const T* __restrict__ V, int64_t ldkv,
                                                  T* __restrict__ O, int L, int Tn, int mode, int HP) {
    constexpr int H = 128 / D;
    const float scale = 1.0f / sqrtf((float)D);   // head_dim ** -0.5 (selfattention.py:12)
    const int W = HP * D;                         // HP heads at a time (all of them unless the track is too long for K and V of every head to fit LDS)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sK = reinterpret_cast<float*>(smem);
    float* sV = sK + L * W;
    const int G = blockIdx.x;
    for (int h0 = 0; h0 < H; h0 += HP) {
        __syncthreads();
        stage_group<T, D>(sK, K, ldkv, G, L, Tn, mode, h0, HP, 1.0f);
        stage_group<T, D>(sV, V, ldkv, G, L, Tn, mode, h0, HP, 1.0f);
        __syncthreads();
        for (int item = threadIdx.x; item < L * HP; item += blockDim.x) {
            const int i = item / HP, h = item % HP;
            const int64_t tok = tok_of(G, i, Tn, mode);
            float q[D];
            load16<T, D>(Q + tok * ldq + (h0 + h) * D, q);
#pragma unroll
            for (int d = 0; d < D; ++d) q[d] *= scale;
            float mx = -INFINITY;
            for (int j = 0; j < L; ++j) mx = fmaxf(mx, dot16<D>(q, sK + j * W + h * D));
            float sum = 0.f, o[D];
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] = 0.f;
            for (int j = 0; j < L; ++j) {
                const float e = __expf(dot16<D>(q, sK + j * W + h * D) - mx);
                sum += e;
                axpy16<D>(o, e, sV + j * W + h * D);
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] *= inv;
            store16<T, D>(O + tok * 128 + (h0 + h) * D, o);
        }
    }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void k_attn_bwd(const T* __restrict__ Q, int64_t ldq, const T* __restrict__ K, const T* __restrict__ V, int64_t ldkv,
                                                  const T* __restrict__ dO, T* __restrict__ dQ, int64_t lddq, T* __restrict__ dK, T* __restrict__ dV,
                                                  int64_t lddkv, int L, int Tn, int mode, int HP) {
    constexpr int H = 128 / D;
    const float scale = 1.0f / sqrtf((float)D);
    const int W = HP * D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sQ = reinterpret_cast<float*>(smem);   // pre-scaled by head_dim ** -0.5
    float* sK = sQ + L * W;
    float* sV = sK + L * W;
    float* sD = sV + L * W;                       // dO
    float* sStat = sD + L * W;                    // [L*HP][4] : max, 1/sum, delta
    const int G = blockIdx.x;
    for (int h0 = 0; h0 < H; h0 += HP) {
        __syncthreads();
        stage_group<T, D>(sQ, Q, ldq, G, L, Tn, mode, h0, HP, scale);
        stage_group<T, D>(sK, K, ldkv, G, L, Tn, mode, h0, HP, 1.0f);
        stage_group<T, D>(sV, V, ldkv, G, L, Tn, mode, h0, HP, 1.0f);
        stage_group<T, D>(sD, dO, 128, G, L, Tn, mode, h0, HP, 1.0f);
        __syncthreads();
        // ---- phase 1: one thread per (query i, head): softmax statistics, delta, dQ ----
        for (int item = threadIdx.x; item < L * HP; item += blockDim.x) {
            const int i = item / HP, h = item % HP;
            float q[D], d_o[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { q[d] = sQ[i * W + h * D + d]; d_o[d] = sD[i * W + h * D + d]; }
            float mx = -INFINITY;
            for (int j = 0; j < L; ++j) mx = fmaxf(mx, dot16<D>(q, sK + j * W + h * D));
            float sum = 0.f, acc = 0.f;
            for (int j = 0; j < L; ++j) {
                const float e = __expf(dot16<D>(q, sK + j * W + h * D) - mx);
                sum += e;
                acc += e * dot16<D>(d_o, sV + j * W + h * D);
            }
            const float inv = 1.0f / sum, delta = acc * inv;
            float dq[D];
#pragma unroll
            for (int d = 0; d < D; ++d) dq[d] = 0.f;
            for (int j = 0; j < L; ++j) {
                const float pj = __expf(dot16<D>(q, sK + j * W + h * D) - mx) * inv;
                const float dpj = dot16<D>(d_o, sV + j * W + h * D);
                axpy16<D>(dq, pj * (dpj - delta) * scale, sK + j * W + h * D);
            }
            store16<T, D>(dQ + tok_of(G, i, Tn, mode) * lddq + (h0 + h) * D, dq);
            sStat[item * 4 + 0] = mx;
            sStat[item * 4 + 1] = inv;
            sStat[item * 4 + 2] = delta;
        }
        __syncthreads();
        // ---- phase 2: one thread per (key j, head): dK, dV ----
        for (int item = threadIdx.x; item < L * HP; item += blockDim.x) {
            const int j = item / HP, h = item % HP;
            float kk[D], vv[D], dk[D], dv[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { kk[d] = sK[j * W + h * D + d]; vv[d] = sV[j * W + h * D + d]; dk[d] = 0.f; dv[d] = 0.f; }
            for (int i = 0; i < L; ++i) {
                const float* st = sStat + (i * HP + h) * 4;
                const float pij = __expf(dot16<D>(kk, sQ + i * W + h * D) - st[0]) * st[1];
                const float dpij = dot16<D>(vv, sD + i * W + h * D);
                const float ds = pij * (dpij - st[2]);            // scale is already inside sQ
                axpy16<D>(dk, ds, sQ + i * W + h * D);
                axpy16<D>(dv, pij, sD + i * W + h * D);
            }
            const int64_t tok = tok_of(G, j, Tn, mode);
            store16<T, D>(dK + tok * lddkv + (h0 + h) * D, dk);
            store16<T, D>(dV + tok * lddkv + (h0 + h) * D, dv);
        }
    }
}

template <typename K> bool set_smem(K k, size_t bytes) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) kasf_set_error(1000 + (int)e, "attention: cannot reserve the group's LDS tiles");
    return e == hipSuccess;
}

template <typename T, int D>
void fwd_TD(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int B, int Tn, int mode) {
    constexpr int H = 128 / D;
    const int L = mode == 0 ? KASF_J : Tn, groups = mode == 0 ? B * Tn : B * KASF_J;
    int HP = H;
    auto bytes = [&](int hp) { return (size_t)2 * L * hp * D * sizeof(float); };
    while (HP > 1 && bytes(HP) > 128 * 1024) HP >>= 1;
    if (bytes(HP) > 160 * 1024) { kasf_set_error(3, "attention: n_frames too large for the LDS-resident kernel"); return; }
    int threads = ((L * HP + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    if (!set_smem(k_attn_fwd<T, D>, bytes(HP))) return;
    hipLaunchKernelGGL((k_attn_fwd<T, D>), dim3(groups), dim3(threads), bytes(HP), s, (const T*)q, ldq, (const T*)k, (const T*)v, ldkv, (T*)o, L, Tn, mode, HP);
}
template <typename T, int D>
void bwd_TD(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq, int64_t lddq, void* dk,
            void* dv, int64_t lddkv, int B, int Tn, int mode) {
    constexpr int H = 128 / D;
    const int L = mode == 0 ? KASF_J : Tn, groups = mode == 0 ? B * Tn : B * KASF_J;
    int HP = H;
    auto bytes = [&](int hp) { return (size_t)(4 * L * hp * D + L * hp * 4) * sizeof(float); };
    while (HP > 1 && bytes(HP) > 96 * 1024) HP >>= 1;
    if (bytes(HP) > 160 * 1024) { kasf_set_error(3, "attention backward: n_frames too large for the LDS-resident kernel"); return; }
    int threads = ((L * HP + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    if (!set_smem(k_attn_bwd<T, D>, bytes(HP))) return;
    hipLaunchKernelGGL((k_attn_bwd<T, D>), dim3(groups), dim3(threads), bytes(HP), s, (const T*)q, ldq, (const T*)k, (const T*)v, ldkv, (const T*)d_o, (T*)dq,
                       lddq, (T*)dk, (T*)dv, lddkv, L, Tn, mode, HP);
}
template <typename T>
void fwd_T(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int B, int Tn, int mode, int heads) {
    if (heads == 8) fwd_TD<T, 16>(s, q, ldq, k, v, ldkv, o, B, Tn, mode);
    else if (heads == 4) fwd_TD<T, 32>(s, q, ldq, k, v, ldkv, o, B, Tn, mode);
    else if (heads == 16) fwd_TD<T, 8>(s, q, ldq, k, v, ldkv, o, B, Tn, mode);
    else if (heads == 2) fwd_TD<T, 64>(s, q, ldq, k, v, ldkv, o, B, Tn, mode);
    else kasf_set_error(3, "attention: num_heads must be 2, 4, 8 or 16");
}
template <typename T>
void bwd_T(hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq, int64_t lddq, void* dk,
           void* dv, int64_t lddkv, int B, int Tn, int mode, int heads) {
    if (heads == 8) bwd_TD<T, 16>(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, Tn, mode);
    else if (heads == 4) bwd_TD<T, 32>(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, Tn, mode);
    else if (heads == 16) bwd_TD<T, 8>(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, Tn, mode);
    else if (heads == 2) bwd_TD<T, 64>(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, Tn, mode);
    else kasf_set_error(3, "attention: num_heads must be 2, 4, 8 or 16");
}

}  // namespace

// heads = 8 and heads = 4 (head dimensions 16 and 32) have MFMA cores in bf16 mode (k_attn_mfma.hip); 2 and 16 heads, and fp32 mode, run the kernels above
void kasf_launch_attn_fwd(int dt, hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int B, int T,
                          int mode, int heads) {
    if (dt == KASF_F32) fwd_T<float>(s, q, ldq, k, v, ldkv, o, B, T, mode, heads);
    else if (heads == 4 && kasf_launch_attn_fwd_mfma32(s, q, ldq, k, v, ldkv, o, B, T, mode)) return;
    else if (heads != 8 || !kasf_launch_attn_fwd_mfma(s, q, ldq, k, v, ldkv, o, B, T, mode)) fwd_T<bf16>(s, q, ldq, k, v, ldkv, o, B, T, mode, heads);
}
void kasf_launch_attn_bwd(int dt, hipStream_t s, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* d_o, void* dq,
                          int64_t lddq, void* dk, void* dv, int64_t lddkv, int B, int T, int mode, int heads) {
    if (dt == KASF_F32) bwd_T<float>(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, T, mode, heads);
    else if (heads == 4 && kasf_launch_attn_bwd_mfma32(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, T, mode)) return;
    else if (heads != 8 || !kasf_launch_attn_bwd_mfma(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, T, mode))
        bwd_T<bf16>(s, q, ldq, k, v, ldkv, d_o, dq, lddq, dk, dv, lddkv, B, T, mode, heads);
}
