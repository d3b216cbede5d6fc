// Memory-bound pieces of the KASportsFormer path (gfx950): prologue (bone decomposition, limb refusion,
// three embeddings), gate fusion, head, losses, AdamW, casts.  One aligned group of 16 lanes owns one
// token row of 128 channels (8 channels / 16 B per lane) so every global access is a full 256/512-B line.
#include "common.h"
#include "kernels.h"

namespace {

// model/KASportsFormer.py:46-47 and modules/bone_refusion.py:34-40 -- index tables copied as data (bit-exact)
__constant__ int c_bone_child[16] = {0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15};
__constant__ int c_bone_parent[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ int c_limb_n[17] = {3, 3, 2, 2, 3, 3, 4, 4, 4, 4, 3, 4, 4, 4, 4, 2, 2};
__constant__ int c_limb_idx[17][4] = {{0, 1, 2, 0}, {3, 4, 5, 0}, {6, 7, 0, 0}, {8, 9, 0, 0}, {10, 11, 12, 0}, {13, 14, 15, 0},
                                      {6, 7, 1, 2}, {6, 7, 4, 5}, {6, 7, 11, 12}, {6, 7, 14, 15}, {6, 7, 9, 0},
                                      {14, 15, 11, 12}, {1, 2, 4, 5}, {14, 15, 4, 5}, {11, 12, 4, 5}, {10, 0, 0, 0}, {13, 3, 0, 0}};

inline unsigned ew_grid(int64_t items, int cap = 4096) {
    int64_t blocks = (items + 255) / 256;
    return (unsigned)(blocks > cap ? cap : (blocks < 1 ? 1 : blocks));
}

// ------------------------------------------------------------------------------------------------
// Prologue forward (KASportsFormer.py:42-62, 323-330; bone_refusion.py:61-70; bone_MLP.py:16-27).
// One workgroup walks frames; per frame: 51 input floats -> bone_info[17][3], limb[17][3] (51 tiny
// MLPs n->16->1, GELU) -> three Linear(3,128)+pos embeddings written as T.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_prologue_fwd(const float* __restrict__ x, const float* __restrict__ P, const KasfProOff* __restrict__ offp,
                                                      T* __restrict__ xj, T* __restrict__ xb, T* __restrict__ xl, float* __restrict__ bone3,
                                                      float* __restrict__ limb3, int64_t frames) {
    __shared__ float sIn[3][KASF_J][3];       // 0: joints (x,y,conf), 1: bone_info, 2: limb
    __shared__ KasfProOff off;
    for (int i = threadIdx.x; i < (int)(sizeof(KasfProOff) / sizeof(int64_t)); i += 256) reinterpret_cast<int64_t*>(&off)[i] = reinterpret_cast<const int64_t*>(offp)[i];
    __syncthreads();
    for (int64_t f = blockIdx.x; f < frames; f += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < 51) (&sIn[0][0][0])[threadIdx.x] = x[f * 51 + threadIdx.x];
        __syncthreads();
        if (threadIdx.x < 16) {                // bones: direction = child - parent on (x, y); zero length -> 1
            const int c = c_bone_child[threadIdx.x], p = c_bone_parent[threadIdx.x];
            const float dx = sIn[0][c][0] - sIn[0][p][0], dy = sIn[0][c][1] - sIn[0][p][1];
            float len = sqrtf(dx * dx + dy * dy);
            if (len == 0.f) len = 1.f;
            sIn[1][threadIdx.x][0] = dx / len;
            sIn[1][threadIdx.x][1] = dy / len;
            sIn[1][threadIdx.x][2] = len;
        } else if (threadIdx.x >= 64 && threadIdx.x < 64 + 51) {   // limb refusion MLPs on the RAW joints (KASportsFormer.py:324)
            const int t = threadIdx.x - 64, i = t / 3, ch = t % 3, n = c_limb_n[i];
            const int64_t* o = off.mlp[t];
            float in[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) in[k] = k < n ? sIn[0][c_limb_idx[i][k]][ch] : 0.f;
            float outv = P[o[3]];
            for (int h = 0; h < 16; ++h) {
                float z = P[o[1] + h];
                for (int k = 0; k < n; ++k) z += P[o[0] + h * n + k] * in[k];
                outv += P[o[2] + h] * gelu_f(z);
            }
            sIn[2][i][ch] = outv;
        }
        __syncthreads();
        if (threadIdx.x < 3) {                 // row 16 = mean over the 16 bones
            float s = 0.f;
            for (int b = 0; b < 16; ++b) s += sIn[1][b][threadIdx.x];
            sIn[1][16][threadIdx.x] = s * (1.0f / 16.0f);
        }
        __syncthreads();
        if (threadIdx.x < 51) {
            bone3[f * 51 + threadIdx.x] = (&sIn[1][0][0])[threadIdx.x];
            limb3[f * 51 + threadIdx.x] = (&sIn[2][0][0])[threadIdx.x];
        }
        for (int item = threadIdx.x; item < 3 * KASF_J * 16; item += 256) {
            const int st = item / (KASF_J * 16), j = (item / 16) % KASF_J, sub = item & 15;
            const float a0 = sIn[st][j][0], a1 = sIn[st][j][1], a2 = sIn[st][j][2];
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = sub * 8 + e;
                const float* w = P + off.embed_w[st] + c * 3;
                v[e] = w[0] * a0 + w[1] * a1 + w[2] * a2 + P[off.embed_b[st] + c] + P[off.pos[st] + j * 128 + c];
            }
            T* dst = st == 0 ? xj : (st == 1 ? xb : xl);
            store8(dst + (f * KASF_J + j) * 128 + sub * 8, v);
        }
    }
}

// Embedding backward for one stream: dW[128][3], db[128], dpos[17][128], optional din3[M][3] = g . W
constexpr int EMBED_ROW = 17 * 128 + 128 * 3 + 128;
template <typename T>
__global__ __launch_bounds__(256) void k_embed_bwd(const T* __restrict__ g, const float* __restrict__ in3, const float* __restrict__ W,
                                                   float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dpos, float* __restrict__ din3,
                                                   int64_t frames, float* __restrict__ rows) {
    // rows != nullptr: this workgroup's sums go to row blockIdx.x of `rows` (dpos[17][128] | dW[128][3] | db[128] = EMBED_ROW floats) and are added in a
    // fixed order by k_col_finish; otherwise fp32 atomics
    float* prow = rows != nullptr ? rows + (int64_t)blockIdx.x * EMBED_ROW : nullptr;
    __shared__ float sRed[16][128];
    const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    float w[8][3], aw[8][3], ab[8], apA[8], apB[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ab[e] = apA[e] = apB[e] = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) { w[e][d] = W[(sub * 8 + e) * 3 + d]; aw[e][d] = 0.f; }
    }
    // Few workgroups (every one ends in 2.7 K same-address atomics: at 512 workgroups those, not the 30 MB stream, were the launch's 198 us), so
    // each keeps EU frames in flight: all loads of a pass are issued before the first use.
    constexpr int EU = 4;
    auto accumulate = [&](const float (&gv)[8], float a0, float a1, float a2, int64_t tok, bool second) {
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            aw[e][0] += gv[e] * a0; aw[e][1] += gv[e] * a1; aw[e][2] += gv[e] * a2;
            ab[e] += gv[e];
            if (!second) apA[e] += gv[e]; else apB[e] += gv[e];
            d0 += gv[e] * w[e][0]; d1 += gv[e] * w[e][1]; d2 += gv[e] * w[e][2];
        }
        if (din3 != nullptr) {
            d0 = reduce16(d0); d1 = reduce16(d1); d2 = reduce16(d2);
            if (sub == 0) { din3[tok * 3] = d0; din3[tok * 3 + 1] = d1; din3[tok * 3 + 2] = d2; }
        }
    };
    for (int64_t f0 = blockIdx.x; f0 < frames; f0 += (int64_t)EU * gridDim.x) {
        float gv[EU][8], a[EU][3], gw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, aq[3] = {0.f, 0.f, 0.f};
        // joints 0..15 of EU frames: row rl of each; joint 16 of those frames: row rl < EU takes frame rl's
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int64_t f = f0 + (int64_t)u * gridDim.x;
            const int64_t tok = (f < frames ? f : f0) * KASF_J + rl;
            load8(g + tok * 128 + sub * 8, gv[u]);
            a[u][0] = in3[tok * 3]; a[u][1] = in3[tok * 3 + 1]; a[u][2] = in3[tok * 3 + 2];
        }
        const int64_t f16 = f0 + (int64_t)rl * gridDim.x;
        const bool has16 = rl < EU && f16 < frames;
        if (has16) {
            const int64_t tok = f16 * KASF_J + 16;
            load8(g + tok * 128 + sub * 8, gw);
            aq[0] = in3[tok * 3]; aq[1] = in3[tok * 3 + 1]; aq[2] = in3[tok * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int64_t f = f0 + (int64_t)u * gridDim.x;
            if (f < frames) accumulate(gv[u], a[u][0], a[u][1], a[u][2], f * KASF_J + rl, false);
        }
        if (has16) accumulate(gw, aq[0], aq[1], aq[2], f16 * KASF_J + 16, true);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (prow != nullptr) prow[rl * 128 + sub * 8 + e] = apA[e]; else atomicAdd(dpos + rl * 128 + sub * 8 + e, apA[e]);
        float b16 = apB[e];                          // joint 16: rows rl < EU hold a share each (lanes 16 rl + sub of the first waves)
        b16 += __shfl_xor(b16, 16);
        b16 += __shfl_xor(b16, 32);
        if (threadIdx.x < 16) { if (prow != nullptr) prow[16 * 128 + sub * 8 + e] = b16; else atomicAdd(dpos + 16 * 128 + sub * 8 + e, b16); }
    }
    for (int q = 0; q < 4; ++q) {               // q<3: dW[:, q], q==3: db
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) sRed[rl][sub * 8 + e] = q < 3 ? aw[e][q == 0 ? 0 : (q == 1 ? 1 : 2)] : ab[e];
        __syncthreads();
        if (threadIdx.x < 128) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += sRed[k][threadIdx.x];
            if (prow != nullptr) { if (q < 3) prow[2176 + threadIdx.x * 3 + q] = s; else prow[2560 + threadIdx.x] = s; }
            else if (q < 3) atomicAdd(dW + threadIdx.x * 3 + q, s); else atomicAdd(db + threadIdx.x, s);
        }
    }
}

// Limb-refusion backward: thread = one of the 51 (group, channel) MLPs, register accumulation over frames.
__global__ __launch_bounds__(64) void k_refusion_bwd(const float* __restrict__ x, const float* __restrict__ dlimb3, const float* __restrict__ P,
                                                     float* __restrict__ Gr, const KasfProOff* __restrict__ offp, int64_t frames,
                                                     float* __restrict__ rows, int64_t base, int row_len) {
    const int t = threadIdx.x;
    // rows != nullptr: row blockIdx.x of `rows` mirrors the gradient range [base, base + row_len) of the 204 limb-MLP tensors (alignment gaps zeroed);
    // k_col_finish adds the rows in a fixed order.  Otherwise fp32 atomics into the gradients.
    float* prow = rows != nullptr ? rows + (int64_t)blockIdx.x * row_len - base : nullptr;
    if (rows != nullptr) {
        for (int k = t; k < row_len; k += 64) rows[(int64_t)blockIdx.x * row_len + k] = 0.f;
        __syncthreads();
    }
    if (t >= 51) return;
    const int i = t / 3, ch = t % 3, n = c_limb_n[i];
    int64_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = offp->mlp[t][k];
    float w1[16][4], b1[16], w2[16];
    float dw1[16][4], db1[16], dw2[16], db2 = 0.f;
#pragma unroll
    for (int h = 0; h < 16; ++h) {
        b1[h] = P[o[1] + h]; w2[h] = P[o[2] + h]; db1[h] = 0.f; dw2[h] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { w1[h][k] = k < n ? P[o[0] + h * n + k] : 0.f; dw1[h][k] = 0.f; }
    }
    for (int64_t f = blockIdx.x; f < frames; f += gridDim.x) {
        float in[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) in[k] = k < n ? x[f * 51 + c_limb_idx[i][k] * 3 + ch] : 0.f;
        const float d = dlimb3[f * 51 + i * 3 + ch];
        db2 += d;
#pragma unroll
        for (int h = 0; h < 16; ++h) {
            const float z = b1[h] + w1[h][0] * in[0] + w1[h][1] * in[1] + w1[h][2] * in[2] + w1[h][3] * in[3];
            float gz, dgz;
            gelu_and_grad(z, gz, dgz);
            dw2[h] += d * gz;
            const float dz = d * w2[h] * dgz;
            db1[h] += dz;
#pragma unroll
            for (int k = 0; k < 4; ++k) dw1[h][k] += dz * in[k];
        }
    }
    if (prow != nullptr) {
        prow[o[3]] = db2;
#pragma unroll
        for (int h = 0; h < 16; ++h) {
            prow[o[1] + h] = db1[h];
            prow[o[2] + h] = dw2[h];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < n) prow[o[0] + h * n + k] = dw1[h][k];
        }
        return;
    }
    atomicAdd(Gr + o[3], db2);
#pragma unroll
    for (int h = 0; h < 16; ++h) {
        atomicAdd(Gr + o[1] + h, db1[h]);
        atomicAdd(Gr + o[2] + h, dw2[h]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n) atomicAdd(Gr + o[0] + h * n + k, dw1[h][k]);
    }
}

// ------------------------------------------------------------------------------------------------
// Gate fusion (KASportsFormer.py:278-284): alpha = softmax(Linear(384->3)(cat(xa,xg,xb))), out = sum alpha_k x_k
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_gate_fwd(const T* __restrict__ xa, const T* __restrict__ xg, const T* __restrict__ xb, const float* __restrict__ W,
                                                  const float* __restrict__ bias, T* __restrict__ out, float* __restrict__ alpha, int64_t M, int adaptive) {
    const int sub = threadIdx.x & 15;
    float w[3][3][8];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) w[j][s][e] = W[j * 384 + s * 128 + sub * 8 + e];
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2];
    for (int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x; item < M * 16; item += (int64_t)gridDim.x * 256) {
        const int64_t tok = item >> 4;
        float x[3][8];
        load8(xa + tok * 128 + sub * 8, x[0]);
        load8(xg + tok * 128 + sub * 8, x[1]);
        load8(xb + tok * 128 + sub * 8, x[2]);
        float a0, a1, a2;
        if (adaptive) {
            float l[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e) l[j] += x[s][e] * w[j][s][e];
            l[0] = reduce16(l[0]) + b0; l[1] = reduce16(l[1]) + b1; l[2] = reduce16(l[2]) + b2;
            const float mx = fmaxf(l[0], fmaxf(l[1], l[2]));
            a0 = __expf(l[0] - mx); a1 = __expf(l[1] - mx); a2 = __expf(l[2] - mx);
            const float inv = 1.0f / (a0 + a1 + a2);
            a0 *= inv; a1 *= inv; a2 *= inv;
        } else {
            a0 = a1 = a2 = 1.0f / 3.0f;
        }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = x[0][e] * a0 + x[1][e] * a1 + x[2][e] * a2;
        store8(out + tok * 128 + sub * 8, o);
        if (alpha != nullptr && sub == 0) { alpha[tok * 4] = a0; alpha[tok * 4 + 1] = a1; alpha[tok * 4 + 2] = a2; }
    }
}

// g = g0 (+ g1 + g2: the three branch input gradients of the layer above, summed here instead of by a separate pass).
// Weight / bias gradients leave each workgroup as ONE partial row (part != nullptr; summed in fixed order by k_gate_reduce) so that
// the grid can be large enough to hide HBM latency without contended atomics.
constexpr int GATE_PART_LD = 1160;      // 3 x 384 weight gradients + 3 bias gradients, padded
template <typename T>
__global__ __launch_bounds__(256, 3) void k_gate_bwd(const T* __restrict__ g, const T* __restrict__ g1, const T* __restrict__ g2, const T* __restrict__ xa,
                                                     const T* __restrict__ xg, const T* __restrict__ xb, const float* __restrict__ W,
                                                     const float* __restrict__ alpha, T* __restrict__ ga, T* __restrict__ gg, T* __restrict__ gb,
                                                     float* __restrict__ dW, float* __restrict__ db, float* __restrict__ part, int64_t M, int adaptive) {
    // Round 4: 128 VGPRs, four waves per SIMD (was 217 / two: 69 % of the wave-cycles were waits on the seven streams).  The 72 weight values a lane
    // multiplies by are re-read from LDS per token instead of living in registers, the three branch rows stay in their loaded (bf16) form between the
    // two passes, and each branch's gradient row is stored as soon as it is formed.
    __shared__ float sRed[16][384];
    __shared__ float sW[3][384];
    const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    for (int c = threadIdx.x; c < 3 * 384; c += 256) sW[c / 384][c % 384] = W[c];
    __syncthreads();
    float dw[3][3][8], dbl[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) dw[j][s][e] = 0.f;
    T* const outs[3] = {ga, gg, gb};
    for (int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x; item < M * 16; item += (int64_t)gridDim.x * 256) {
        const int64_t tok = item >> 4;
        Raw8<T> xr[3];
        float gv[8];
        xr[0].load(xa + tok * 128 + sub * 8);
        xr[1].load(xg + tok * 128 + sub * 8);
        xr[2].load(xb + tok * 128 + sub * 8);
        load8(g + tok * 128 + sub * 8, gv);
        if (g1 != nullptr) {
            float t[8];
            load8(g1 + tok * 128 + sub * 8, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] += t[e];
        }
        if (g2 != nullptr) {
            float t[8];
            load8(g2 + tok * 128 + sub * 8, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] += t[e];
        }
        const f32x4 av = *reinterpret_cast<const f32x4*>(alpha + tok * 4);
        const float a[3] = {av[0], av[1], av[2]};
        float dl[3] = {0.f, 0.f, 0.f};
        if (adaptive) {
            float da[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                float xs[8];
                xr[s].get(xs);
#pragma unroll
                for (int e = 0; e < 8; ++e) da[s] += gv[e] * xs[e];
                da[s] = reduce16(da[s]);
            }
            const float dot = a[0] * da[0] + a[1] * da[1] + a[2] * da[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) { dl[j] = a[j] * (da[j] - dot); if (sub == 0) dbl[j] += dl[j]; }
        }
        int wo = sub * 8;
        asm volatile("" : "+v"(wo));                     // the weight reads below stay in the loop
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float xs[8], o[8], wv[3][8];
            xr[s].get(xs);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(&sW[j][s * 128 + wo]), w1 = *reinterpret_cast<const f32x4*>(&sW[j][s * 128 + wo + 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { wv[j][e] = w0[e]; wv[j][4 + e] = w1[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = a[s] * gv[e] + dl[0] * wv[0][e] + dl[1] * wv[1][e] + dl[2] * wv[2][e];
#pragma unroll
                for (int j = 0; j < 3; ++j) dw[j][s][e] += dl[j] * xs[e];
            }
            store8(outs[s] + tok * 128 + sub * 8, o);
        }
    }
    if (!adaptive) return;
    float* prow = part != nullptr ? part + (int64_t)blockIdx.x * GATE_PART_LD : nullptr;
    for (int j = 0; j < 3; ++j) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) sRed[rl][s * 128 + sub * 8 + e] = j == 0 ? dw[0][s][e] : (j == 1 ? dw[1][s][e] : dw[2][s][e]);
        __syncthreads();
        for (int c = threadIdx.x; c < 384; c += 256) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += sRed[k][c];
            if (prow != nullptr) prow[j * 384 + c] = s;
            else atomicAdd(dW + j * 384 + c, s);
        }
    }
    __syncthreads();
    if (sub == 0) { sRed[rl][0] = dbl[0]; sRed[rl][1] = dbl[1]; sRed[rl][2] = dbl[2]; }      // the 16 groups' shares, added in a fixed order
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sRed[k][threadIdx.x];
        if (prow != nullptr) prow[1152 + threadIdx.x] = t;
        else atomicAdd(db + threadIdx.x, t);
    }
}

// ------------------------------------------------------------------------------------------------
// Head (KASportsFormer.py:339-345): rep = tanh(fc(LN(x))) comes from k_linear; here Linear(512->3) and its backward
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_head_fwd(const T* __restrict__ rep, const float* __restrict__ W, const float* __restrict__ bias,
                                                  float* __restrict__ out, int64_t M) {
    const int sub = threadIdx.x & 15;
    float w[3][32];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) w[j][c * 8 + e] = W[j * 512 + c * 128 + sub * 8 + e];
    for (int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x; item < M * 16; item += (int64_t)gridDim.x * 256) {
        const int64_t tok = item >> 4;
        float l[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float r[8];
            load8(rep + tok * 512 + c * 128 + sub * 8, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) { l[0] += r[e] * w[0][c * 8 + e]; l[1] += r[e] * w[1][c * 8 + e]; l[2] += r[e] * w[2][c * 8 + e]; }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) l[j] = reduce16(l[j]);
        if (sub == 0) { out[tok * 3] = l[0] + bias[0]; out[tok * 3 + 1] = l[1] + bias[1]; out[tok * 3 + 2] = l[2] + bias[2]; }
    }
}

constexpr int HEAD_ROW = 3 * 512 + 4;
template <typename T>
__global__ __launch_bounds__(256) void k_head_bwd(const float* __restrict__ dy, const T* __restrict__ rep, const float* __restrict__ W, T* __restrict__ dpre,
                                                  float* __restrict__ dW, float* __restrict__ db, int64_t M, float* __restrict__ rows) {
    float* prow = rows != nullptr ? rows + (int64_t)blockIdx.x * HEAD_ROW : nullptr;      // dW[3][512] | db[3] of this workgroup (fixed-order finish)
    __shared__ float sRed[16][512];
    const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    float w[3][32], dw[3][32], dbl[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) { w[j][c * 8 + e] = W[j * 512 + c * 128 + sub * 8 + e]; dw[j][c * 8 + e] = 0.f; }
    for (int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x; item < M * 16; item += (int64_t)gridDim.x * 256) {
        const int64_t tok = item >> 4;
        const float d0 = dy[tok * 3], d1 = dy[tok * 3 + 1], d2 = dy[tok * 3 + 2];
        if (sub == 0) { dbl[0] += d0; dbl[1] += d1; dbl[2] += d2; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float r[8], o[8];
            load8(rep + tok * 512 + c * 128 + sub * 8, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = c * 8 + e;
                o[e] = (d0 * w[0][k] + d1 * w[1][k] + d2 * w[2][k]) * (1.0f - r[e] * r[e]);
                dw[0][k] += d0 * r[e]; dw[1][k] += d1 * r[e]; dw[2][k] += d2 * r[e];
            }
            store8(dpre + tok * 512 + c * 128 + sub * 8, o);
        }
    }
    for (int j = 0; j < 3; ++j) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) sRed[rl][c * 128 + sub * 8 + e] = j == 0 ? dw[0][c * 8 + e] : (j == 1 ? dw[1][c * 8 + e] : dw[2][c * 8 + e]);
        __syncthreads();
        for (int c = threadIdx.x; c < 512; c += 256) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += sRed[k][c];
            if (prow != nullptr) prow[j * 512 + c] = s; else atomicAdd(dW + j * 512 + c, s);
        }
    }
    __syncthreads();
    if (sub == 0) { sRed[rl][0] = dbl[0]; sRed[rl][1] = dbl[1]; sRed[rl][2] = dbl[2]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sRed[k][threadIdx.x];
        if (prow != nullptr) prow[1536 + threadIdx.x] = t; else atomicAdd(db + threadIdx.x, t);
    }
}

// ------------------------------------------------------------------------------------------------
template <typename T> __global__ void k_cast_to_f32(const T* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = to_f(src[i]);
}
template <typename T> __global__ void k_cast_from_f32(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = from_f<T>(src[i]);
}
template <typename T> __global__ void k_add_inplace(T* __restrict__ dst, const T* __restrict__ a, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], y[8];
        load8(dst + i * 8, x);
        load8(a + i * 8, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += y[e];
        store8(dst + i * 8, x);
    }
}

template <typename T> __global__ void k_add3(T* __restrict__ dst, const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], y[8];
        load8(a + i * 8, x);
        load8(b + i * 8, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += y[e];
        if (c != nullptr) {
            load8(c + i * 8, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += y[e];
        }
        store8(dst + i * 8, x);
    }
}

// dls[n] = sum_k W[n][k] G[n][k] + bias[n] * gsum[n];  G[n][:] *= ls[n];  gsum[n] *= ls[n]   (one workgroup per row n)
__global__ __launch_bounds__(128) void k_finalize_ls(float* __restrict__ dW, const float* __restrict__ W, const float* __restrict__ bias,
                                                     const float* __restrict__ ls, float* __restrict__ db, float* __restrict__ dls, int K) {
    __shared__ float sPart[2];
    const int n = blockIdx.x;
    const float l = ls[n];
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 128) {
        const float gk = dW[(int64_t)n * K + k];
        s += W[(int64_t)n * K + k] * gk;
        dW[(int64_t)n * K + k] = gk * l;
    }
    s = reduce64(s);
    if ((threadIdx.x & 63) == 0) sPart[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float gs = db[n];
        dls[n] += sPart[0] + sPart[1] + bias[n] * gs;
        db[n] = gs * l;
    }
}

// ------------------------------------------------------------------------------------------------
// 3-term training loss and its gradient (utils/loss_calc.py:6-27; train_and_evaluate_sp.py:212-222):
//   L = mpjpe + lambda_n * n_mpjpe + lambda_v * velocity.   One workgroup per clip.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float norm3(float a, float b, float c) { return sqrtf(a * a + b * b + c * c); }

__global__ __launch_bounds__(256) void k_loss3(const float* __restrict__ P, const float* __restrict__ Y, float* __restrict__ dP, float* __restrict__ losses,
                                               int B, int T, float lam_n, float lam_v, float gscale) {
    // Bit-reproducible: every sum below is formed in an order that does not depend on scheduling (a thread's own loop, a fixed tree over waves); the
    // per-clip loss terms go to losses[4 + 4 b ..] and k_loss3_finish adds the clips in a fixed order (no atomics anywhere: dP feeds the whole backward).
    extern __shared__ float sm[];
    float* sS = sm;            // [T] scale
    float* sDen = sS + T;      // [T]
    float* sWt = sDen + T;     // [T] sum_j u_j . p_j
    __shared__ float sL[4][3];
    const int b = blockIdx.x;
    const float* p = P + (int64_t)b * T * 51;
    const float* y = Y + (int64_t)b * T * 51;
    float* dp = dP + (int64_t)b * T * 51;
    for (int t = threadIdx.x; t < T; t += 256) {
        float den = 0.f, num = 0.f;
        for (int k = 0; k < 51; ++k) { den += p[t * 51 + k] * p[t * 51 + k]; num += y[t * 51 + k] * p[t * 51 + k]; }
        sDen[t] = den;
        const float s = num / den;
        sS[t] = s;
        float wt = 0.f;                                  // sum over the frame's joints of u_j . p_j, u_j = e_j / |e_j|, e_j = s p_j - y_j
        for (int j = 0; j < KASF_J; ++j) {
            const int it = t * KASF_J + j;
            const float e0 = s * p[it * 3] - y[it * 3], e1 = s * p[it * 3 + 1] - y[it * 3 + 1], e2 = s * p[it * 3 + 2] - y[it * 3 + 2];
            const float n = norm3(e0, e1, e2);
            if (n > 0.f) wt += (e0 * p[it * 3] + e1 * p[it * 3 + 1] + e2 * p[it * 3 + 2]) / n;
        }
        sWt[t] = wt;
    }
    __syncthreads();
    const float n1 = (float)B * T * KASF_J, n3 = (float)B * (T - 1) * KASF_J;
    float l1 = 0.f, l2 = 0.f, l3 = 0.f;
    for (int it = threadIdx.x; it < T * KASF_J; it += 256) {
        const int t = it / KASF_J;
        const float s = sS[t];
        const float e0 = s * p[it * 3] - y[it * 3], e1 = s * p[it * 3 + 1] - y[it * 3 + 1], e2 = s * p[it * 3 + 2] - y[it * 3 + 2];
        l2 += norm3(e0, e1, e2);
    }
    for (int it = threadIdx.x; it < T * KASF_J; it += 256) {
        const int t = it / KASF_J;
        float gr[3] = {0.f, 0.f, 0.f};
        const float pv[3] = {p[it * 3], p[it * 3 + 1], p[it * 3 + 2]}, yv[3] = {y[it * 3], y[it * 3 + 1], y[it * 3 + 2]};
        {   // mpjpe
            const float e0 = pv[0] - yv[0], e1 = pv[1] - yv[1], e2 = pv[2] - yv[2], n = norm3(e0, e1, e2);
            l1 += n;
            if (n > 0.f) { gr[0] += e0 / n / n1; gr[1] += e1 / n / n1; gr[2] += e2 / n / n1; }
        }
        {   // n_mpjpe: e = s p - y, s = <y,p>/<p,p> per frame; d/dp_k = s u_k + (sum_j u_j.p_j) (y_k - 2 s p_k) / <p,p>
            const float s = sS[t], wt = sWt[t], den = sDen[t];
            const float e0 = s * pv[0] - yv[0], e1 = s * pv[1] - yv[1], e2 = s * pv[2] - yv[2], n = norm3(e0, e1, e2);
            const float u[3] = {n > 0.f ? e0 / n : 0.f, n > 0.f ? e1 / n : 0.f, n > 0.f ? e2 / n : 0.f};
#pragma unroll
            for (int d = 0; d < 3; ++d) gr[d] += lam_n * (s * u[d] + wt * (yv[d] - 2.f * s * pv[d]) / den) / n1;
        }
        if (T > 1) {   // velocity
            if (t < T - 1) {
                const int nx = it + KASF_J;
                const float v0 = (p[nx * 3] - pv[0]) - (y[nx * 3] - yv[0]), v1 = (p[nx * 3 + 1] - pv[1]) - (y[nx * 3 + 1] - yv[1]),
                            v2 = (p[nx * 3 + 2] - pv[2]) - (y[nx * 3 + 2] - yv[2]);
                const float n = norm3(v0, v1, v2);
                l3 += n;
                if (n > 0.f) { gr[0] -= lam_v * v0 / n / n3; gr[1] -= lam_v * v1 / n / n3; gr[2] -= lam_v * v2 / n / n3; }
            }
            if (t > 0) {
                const int pr = it - KASF_J;
                const float v0 = (pv[0] - p[pr * 3]) - (yv[0] - y[pr * 3]), v1 = (pv[1] - p[pr * 3 + 1]) - (yv[1] - y[pr * 3 + 1]),
                            v2 = (pv[2] - p[pr * 3 + 2]) - (yv[2] - y[pr * 3 + 2]);
                const float n = norm3(v0, v1, v2);
                if (n > 0.f) { gr[0] += lam_v * v0 / n / n3; gr[1] += lam_v * v1 / n / n3; gr[2] += lam_v * v2 / n / n3; }
            }
        }
        dp[it * 3] = gr[0] * gscale; dp[it * 3 + 1] = gr[1] * gscale; dp[it * 3 + 2] = gr[2] * gscale;
    }
    l1 = reduce64(l1); l2 = reduce64(l2); l3 = reduce64(l3);
    if ((threadIdx.x & 63) == 0) { sL[threadIdx.x >> 6][0] = l1; sL[threadIdx.x >> 6][1] = l2; sL[threadIdx.x >> 6][2] = l3; }
    __syncthreads();
    if (threadIdx.x < 3) losses[4 + 4 * b + threadIdx.x] = (sL[0][threadIdx.x] + sL[1][threadIdx.x]) + (sL[2][threadIdx.x] + sL[3][threadIdx.x]);
}
// losses[0..3] = {total, mpjpe, n_mpjpe, velocity} from the per-clip sums at losses[4 + 4 b + {0, 1, 2}]: one workgroup, fixed order
__global__ __launch_bounds__(256) void k_loss3_finish(float* __restrict__ losses, int B, int T, float lam_n, float lam_v) {
    __shared__ float sP[256][3];
    float a[3] = {0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < B; b += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] += losses[4 + 4 * b + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) sP[threadIdx.x][k] = a[k];
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if (threadIdx.x < h)
#pragma unroll
            for (int k = 0; k < 3; ++k) sP[threadIdx.x][k] += sP[threadIdx.x + h][k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n1 = (float)B * T * KASF_J, n3 = (float)B * (T - 1) * KASF_J;
        const float m1 = sP[0][0] / n1, m2 = sP[0][1] / n1, m3 = T > 1 ? sP[0][2] / n3 : 0.f;
        losses[0] = m1 + lam_n * m2 + lam_v * m3;
        losses[1] = m1; losses[2] = m2; losses[3] = m3;
    }
}

// torch.optim.AdamW semantics (decoupled decay first, bias-corrected moments); grad_scale folds the 1/world_size of DP averaging
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                               float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float gs) {
    const float step = lr / bc1, isq = 1.0f / sqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<const f32x4*>(g)[i], mm = reinterpret_cast<f32x4*>(m)[i],
              vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * gs;
            pp[e] *= 1.0f - lr * wd;
            mm[e] = b1 * mm[e] + (1.0f - b1) * gr;
            vv[e] = b2 * vv[e] + (1.0f - b2) * gr * gr;
            pp[e] -= step * mm[e] / (sqrtf(vv[e]) * isq + eps);
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
}

}  // namespace

void kasf_launch_prologue_fwd(int dt, hipStream_t s, const float* x, const float* params, const KasfProOff* off, void* xj, void* xb, void* xl,
                              float* bone3, float* limb3, int64_t frames) {
    const unsigned grid = (unsigned)(frames < 2048 ? frames : 2048);
    if (dt == KASF_F32) hipLaunchKernelGGL(k_prologue_fwd<float>, dim3(grid), dim3(256), 0, s, x, params, off, (float*)xj, (float*)xb, (float*)xl, bone3, limb3, frames);
    else hipLaunchKernelGGL(k_prologue_fwd<bf16>, dim3(grid), dim3(256), 0, s, x, params, off, (bf16*)xj, (bf16*)xb, (bf16*)xl, bone3, limb3, frames);
}
void kasf_launch_embed_bwd(int dt, hipStream_t s, const void* g, const float* in3, const float* W, float* dW, float* db, float* dpos, float* din3,
                           int64_t frames, KasfColSink* sink) {
    const unsigned grid = (unsigned)(frames < 128 ? frames : 128);
    float* rows = sink != nullptr ? sink->take((int)grid, EMBED_ROW) : nullptr;
    if (dt == KASF_F32) hipLaunchKernelGGL(k_embed_bwd<float>, dim3(grid), dim3(256), 0, s, (const float*)g, in3, W, dW, db, dpos, din3, frames, rows);
    else hipLaunchKernelGGL(k_embed_bwd<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)g, in3, W, dW, db, dpos, din3, frames, rows);
    if (rows != nullptr) {
        sink->add(rows, EMBED_ROW, (int)grid, 17 * 128, dpos);
        sink->add(rows + 2176, EMBED_ROW, (int)grid, 384, dW);
        sink->add(rows + 2560, EMBED_ROW, (int)grid, 128, db);
    }
}
void kasf_launch_refusion_bwd(hipStream_t s, const float* x, const float* dlimb3, const float* params, float* grads, const KasfProOff* off,
                              int64_t frames, KasfColSink* sink, int64_t grad_base, int grad_len) {
    const unsigned grid = (unsigned)(frames < 256 ? frames : 256);
    float* rows = (sink != nullptr && grad_len > 0) ? sink->take((int)grid, grad_len) : nullptr;
    hipLaunchKernelGGL(k_refusion_bwd, dim3(grid), dim3(64), 0, s, x, dlimb3, params, grads, off, frames, rows, grad_base, grad_len);
    if (rows != nullptr) sink->add(rows, grad_len, (int)grid, grad_len, grads + grad_base);
}
void kasf_launch_gate_fwd(int dt, hipStream_t s, const void* xa, const void* xg, const void* xb, const float* W, const float* b, void* out,
                          float* alpha, int64_t M, int adaptive) {
    const unsigned grid = ew_grid(M * 16);
    if (dt == KASF_F32) hipLaunchKernelGGL(k_gate_fwd<float>, dim3(grid), dim3(256), 0, s, (const float*)xa, (const float*)xg, (const float*)xb, W, b, (float*)out, alpha, M, adaptive);
    else hipLaunchKernelGGL(k_gate_fwd<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)xa, (const bf16*)xg, (const bf16*)xb, W, b, (bf16*)out, alpha, M, adaptive);
}
void kasf_launch_gate_bwd(int dt, hipStream_t s, const void* g, const void* g1, const void* g2, const void* xa, const void* xg, const void* xb,
                          const float* W, const float* alpha, void* ga, void* gg, void* gb, float* dW, float* db, int64_t M, int adaptive, KasfColSink* sink) {
    unsigned grid = ew_grid(M * 16, 768);               // 3 workgroups per CU (168 VGPRs): enough loads in flight for an HBM stream of 7-9 tensors
    float* part = (sink != nullptr && adaptive) ? sink->take((int)grid, GATE_PART_LD) : nullptr;      // one row of dW[3][384] | db[3] per workgroup
    if (part == nullptr && grid > 256) grid = 256;      // atomics: few workgroups
    if (dt == KASF_F32) hipLaunchKernelGGL(k_gate_bwd<float>, dim3(grid), dim3(256), 0, s, (const float*)g, (const float*)g1, (const float*)g2, (const float*)xa, (const float*)xg, (const float*)xb, W, alpha, (float*)ga, (float*)gg, (float*)gb, dW, db, part, M, adaptive);
    else hipLaunchKernelGGL(k_gate_bwd<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)g, (const bf16*)g1, (const bf16*)g2, (const bf16*)xa, (const bf16*)xg, (const bf16*)xb, W, alpha, (bf16*)ga, (bf16*)gg, (bf16*)gb, dW, db, part, M, adaptive);
    if (part != nullptr) { sink->add(part, GATE_PART_LD, (int)grid, 1152, dW); sink->add(part + 1152, GATE_PART_LD, (int)grid, 3, db); }
}
void kasf_launch_head_fwd(int dt, hipStream_t s, const void* rep, const float* W, const float* b, float* out, int64_t M) {
    const unsigned grid = ew_grid(M * 16);
    if (dt == KASF_F32) hipLaunchKernelGGL(k_head_fwd<float>, dim3(grid), dim3(256), 0, s, (const float*)rep, W, b, out, M);
    else hipLaunchKernelGGL(k_head_fwd<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)rep, W, b, out, M);
}
// backward of forward(x, return_rep=True) (KASportsFormer.py:342-343): the output is tanh(fc(norm(x))), so dpre = drep * (1 - rep^2)
template <typename T>
static __global__ __launch_bounds__(256) void k_rep_bwd(const float* __restrict__ drep, const T* __restrict__ rep, T* __restrict__ dpre, int64_t n8) {
    for (int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x; item < n8; item += (int64_t)gridDim.x * 256) {
        float d[8], r[8];
        load8(drep + item * 8, d);
        load8(rep + item * 8, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] *= 1.0f - r[e] * r[e];
        store8(dpre + item * 8, d);
    }
}
void kasf_launch_rep_bwd(int dt, hipStream_t s, const float* drep, const void* rep, void* dpre, int64_t M) {
    const int64_t n8 = M * 64, blocks = (n8 + 255) / 256;
    const unsigned grid = (unsigned)(blocks > 4096 ? 4096 : blocks);
    if (dt == KASF_F32) hipLaunchKernelGGL(k_rep_bwd<float>, dim3(grid), dim3(256), 0, s, drep, (const float*)rep, (float*)dpre, n8);
    else hipLaunchKernelGGL(k_rep_bwd<bf16>, dim3(grid), dim3(256), 0, s, drep, (const bf16*)rep, (bf16*)dpre, n8);
}
void kasf_launch_head_bwd(int dt, hipStream_t s, const float* dy, const void* rep, const float* W, void* dpre, float* dW, float* db, int64_t M,
                          KasfColSink* sink) {
    const unsigned grid = ew_grid(M * 16, 256);
    float* rows = sink != nullptr ? sink->take((int)grid, HEAD_ROW) : nullptr;
    if (dt == KASF_F32) hipLaunchKernelGGL(k_head_bwd<float>, dim3(grid), dim3(256), 0, s, dy, (const float*)rep, W, (float*)dpre, dW, db, M, rows);
    else hipLaunchKernelGGL(k_head_bwd<bf16>, dim3(grid), dim3(256), 0, s, dy, (const bf16*)rep, W, (bf16*)dpre, dW, db, M, rows);
    if (rows != nullptr) { sink->add(rows, HEAD_ROW, (int)grid, 1536, dW); sink->add(rows + 1536, HEAD_ROW, (int)grid, 3, db); }
}
void kasf_launch_cast_to_f32(int dt, hipStream_t s, const void* src, float* dst, int64_t n) {
    if (dt == KASF_F32) hipLaunchKernelGGL(k_cast_to_f32<float>, dim3(ew_grid(n)), dim3(256), 0, s, (const float*)src, dst, n);
    else hipLaunchKernelGGL(k_cast_to_f32<bf16>, dim3(ew_grid(n)), dim3(256), 0, s, (const bf16*)src, dst, n);
}
void kasf_launch_cast_from_f32(int dt, hipStream_t s, const float* src, void* dst, int64_t n) {
    if (dt == KASF_F32) hipLaunchKernelGGL(k_cast_from_f32<float>, dim3(ew_grid(n)), dim3(256), 0, s, src, (float*)dst, n);
    else hipLaunchKernelGGL(k_cast_from_f32<bf16>, dim3(ew_grid(n)), dim3(256), 0, s, src, (bf16*)dst, n);
}
void kasf_launch_add_inplace(int dt, hipStream_t s, void* dst, const void* a, int64_t n) {
    if (dt == KASF_F32) hipLaunchKernelGGL(k_add_inplace<float>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (float*)dst, (const float*)a, n / 8);
    else hipLaunchKernelGGL(k_add_inplace<bf16>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (bf16*)dst, (const bf16*)a, n / 8);
}
void kasf_launch_add3(int dt, hipStream_t s, void* dst, const void* a, const void* b, const void* c, int64_t n) {
    if (dt == KASF_F32) hipLaunchKernelGGL(k_add3<float>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (float*)dst, (const float*)a, (const float*)b, (const float*)c, n / 8);
    else hipLaunchKernelGGL(k_add3<bf16>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (bf16*)dst, (const bf16*)a, (const bf16*)b, (const bf16*)c, n / 8);
}
void kasf_launch_finalize_ls(hipStream_t s, float* dW, const float* W, const float* bias, const float* ls, float* db, float* dls, int N, int K) {
    hipLaunchKernelGGL(k_finalize_ls, dim3(N), dim3(128), 0, s, dW, W, bias, ls, db, dls, K);
}
void kasf_launch_loss3(hipStream_t s, const float* pred, const float* tgt, float* dpred, float* losses, int B, int T, float lambda_n, float lambda_v,
                       float grad_scale) {
    hipLaunchKernelGGL(k_loss3, dim3(B), dim3(256), 3 * T * sizeof(float), s, pred, tgt, dpred, losses, B, T, lambda_n, lambda_v, grad_scale);
    hipLaunchKernelGGL(k_loss3_finish, dim3(1), dim3(256), 0, s, losses, B, T, lambda_n, lambda_v);
}
void kasf_launch_adamw(hipStream_t s, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
                       float bc1, float bc2, float grad_scale) {
    hipLaunchKernelGGL(k_adamw, dim3(ew_grid(n / 4, 2048)), dim3(256), 0, s, p, g, m, v, n / 4, lr, beta1, beta2, eps, wd, bc1, bc2, grad_scale);
}
