// Evaluation-side kernels: left/right flip, flip-TTA merge and the fused per-clip metric kernel.
// All of this is HBM-bound fp32 data of 51 floats per frame; one launch replaces the reference's per-clip numpy loop
// (train_and_evaluate_sp.py:55-127, utils/error_calc.py:5-48, utils/utilities.py:128-135).
#include "kernels.h"

namespace {

// utils/utilities.py:128-135: destination joint j takes source joint c_flip_src[j]; left [1,2,3,14,15,16] <-> right [4,5,6,11,12,13]
__constant__ int c_flip_src[17] = {0, 4, 5, 6, 1, 2, 3, 7, 8, 9, 10, 14, 15, 16, 11, 12, 13};

__global__ __launch_bounds__(256) void k_joint_flip(const float* __restrict__ src, float* __restrict__ dst, int64_t n /* rows*51 */) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / 51;
        const int r = (int)(i - row * 51), j = r / 3, c = r - 3 * j;
        const float v = src[row * 51 + 3 * c_flip_src[j] + c];
        dst[i] = c == 0 ? -v : v;
    }
}

// train_and_evaluate_sp.py:46-55: (model(x) + flip(model(flip(x)))) / 2, then the root joint is zeroed.  pf == nullptr: no TTA, root zeroing only.
__global__ __launch_bounds__(256) void k_tta_merge(const float* __restrict__ p, const float* __restrict__ pf, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / 51;
        const int r = (int)(i - row * 51), j = r / 3, c = r - 3 * j;
        float v = p[i];
        if (pf) {
            const float f = pf[row * 51 + 3 * c_flip_src[j] + c];
            v = (v + (c == 0 ? -f : f)) / 2;
        }
        out[i] = j == 0 ? 0.0f : v;
    }
}

// Batch assembly from the HBM-resident clip set (data/reader/sp_dataset.py:45-92): out[b] = clips[index[b]], left/right flipped when flip[b].
// Two arrays share the plan (train: inputs + labels; test: inputs + scaled labels).  Out-of-range indices produce zeros, never a fault.
__global__ __launch_bounds__(256) void k_gather_clips(const float* __restrict__ xa, const float* __restrict__ ya, const int64_t* __restrict__ index,
                                                       const unsigned char* __restrict__ flip, int64_t n_clips, int64_t clip_floats, int64_t n,
                                                       float* __restrict__ xo, float* __restrict__ yo) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / clip_floats, r = i - b * clip_floats, row = r / 51;
        const int q = (int)(r - row * 51), j = q / 3, c = q - 3 * j;
        const int64_t src = index[b];
        const bool f = flip && flip[b];
        float vx = 0.0f, vy = 0.0f;
        if (src >= 0 && src < n_clips) {
            const int64_t o = src * clip_floats + row * 51 + 3 * (f ? c_flip_src[j] : j) + c;
            vx = xa[o];
            if (ya) vy = ya[o];
            if (f && c == 0) { vx = -vx; vy = -vy; }
        }
        xo[i] = vx;
        if (ya) yo[i] = vy;
    }
}

// cyclic Jacobi on a symmetric 3x3 (fp64): A -> eigenvalues on the diagonal, V columns = eigenvectors
__device__ inline void jacobi3(double A[3][3], double V[3][3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) V[a][b] = a == b ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off < 1e-60) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = A[p][q];
            if (apq == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {          // A <- A J
                const double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq;
                A[k][q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {          // A <- J^T A
                const double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk;
                A[q][k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq;
                V[k][q] = s * vkp + c * vkq;
            }
        }
    }
}

constexpr int EV_THR = 256;            // >= T: one thread per frame
constexpr int EV_COLS = 22;            // mpjpe, p_mpjpe, accel, jpe[17], frames, accel frames

// One workgroup per clip.  pred [B,T,17,3] normalised model output; label_scaled [B,T,17,3] (mm); factor [B,T]; res [B,2] = (w,h);
// action [B] ids.  Outputs per frame + per-action sums (fp64 atomics).
__global__ __launch_bounds__(EV_THR) void k_eval_metrics(const float* __restrict__ pred, const float* __restrict__ label, const float* __restrict__ factor,
                                                          const float* __restrict__ res, const int* __restrict__ action, int T, int n_actions,
                                                          float* __restrict__ o_mpjpe, float* __restrict__ o_pmpjpe, float* __restrict__ o_acc,
                                                          float* __restrict__ o_jpe, double* __restrict__ action_sums) {
    extern __shared__ float sm[];
    float* P = sm;                  // [T][51] de-normalised, scaled, root-centred prediction
    float* G = sm + T * 51;         // [T][51] root-centred ground truth
    __shared__ double sred[EV_COLS];
    const int b = blockIdx.x;
    const double w = res[2 * b], h = res[2 * b + 1];
    if (threadIdx.x < EV_COLS) sred[threadIdx.x] = 0.0;
    for (int i = threadIdx.x; i < T * 51; i += EV_THR) {
        const int t = i / 51, r = i - 51 * t, j = r / 3, c = r - 3 * j;
        const double f = factor[(int64_t)b * T + t];
        const double shift = c == 0 ? 1.0 : (c == 1 ? h / w : 0.0);
        // sp:55 root zeroing, sp:62-63 de-normalisation (float64 temporaries stored back to float32), sp:67 x factor, sp:68 root-centring
        const float pv = j == 0 ? 0.0f : pred[(int64_t)b * T * 51 + i];
        const float dn = (float)(((double)pv + shift) * w / 2);
        const float rt = (float)((0.0 + shift) * w / 2);
        P[i] = (float)((double)dn * f) - (float)((double)rt * f);
        G[i] = label[(int64_t)b * T * 51 + i] - label[(int64_t)b * T * 51 + 51 * t + c];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < T) {
        const float* p = P + 51 * t;
        const float* g = G + 51 * t;
        // ---- MPJPE / per-joint error (error_calc.py:5-12) ----
        double msum = 0.0;
        for (int j = 0; j < 17; ++j) {
            const float dx = p[3 * j] - g[3 * j], dy = p[3 * j + 1] - g[3 * j + 1], dz = p[3 * j + 2] - g[3 * j + 2];
            const float e = sqrtf(dx * dx + dy * dy + dz * dz);
            o_jpe[((int64_t)b * T + t) * 17 + j] = e;
            atomicAdd(&sred[3 + j], (double)e);
            msum += e;
        }
        const float mp = (float)(msum / 17.0);
        o_mpjpe[(int64_t)b * T + t] = mp;
        atomicAdd(&sred[0], (double)mp);
        atomicAdd(&sred[20], 1.0);
        // ---- acceleration error (error_calc.py:15-19) ----
        if (t < T - 2) {
            double asum = 0.0;
            for (int j = 0; j < 17; ++j) {
                float d[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float ap = p[3 * j + c] - 2 * p[51 + 3 * j + c] + p[102 + 3 * j + c];
                    const float ag = g[3 * j + c] - 2 * g[51 + 3 * j + c] + g[102 + 3 * j + c];
                    d[c] = ap - ag;
                }
                asum += sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            }
            const float ae = (float)(asum / 17.0);
            o_acc[(int64_t)b * (T - 2) + t] = ae;
            atomicAdd(&sred[2], (double)ae);
            atomicAdd(&sred[21], 1.0);
        }
        // ---- Procrustes-aligned MPJPE (error_calc.py:21-48), fp64 ----
        double muX[3] = {0, 0, 0}, muY[3] = {0, 0, 0};
        for (int j = 0; j < 17; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) { muX[c] += g[3 * j + c]; muY[c] += p[3 * j + c]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) { muX[c] /= 17.0; muY[c] /= 17.0; }
        double nX = 0, nY = 0, H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int j = 0; j < 17; ++j) {
            double x0[3], y0[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { x0[c] = g[3 * j + c] - muX[c]; y0[c] = p[3 * j + c] - muY[c]; nX += x0[c] * x0[c]; nY += y0[c] * y0[c]; }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) H[a][c] += x0[a] * y0[c];
        }
        nX = sqrt(nX); nY = sqrt(nY);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) H[a][c] /= nX * nY;
        // H = U S V^T through the eigen-decomposition of H^T H; the reflection fix of error_calc.py:37-41 equals taking
        // u3 = u1 x u2, v3 = v1 x v2 (both bases right-handed) and weighting s3 by sign(det H).
        double A[3][3], V[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) A[a][c] = H[0][a] * H[0][c] + H[1][a] * H[1][c] + H[2][a] * H[2][c];
        jacobi3(A, V);
        double lam[3] = {A[0][0], A[1][1], A[2][2]};
        int i0 = 0, i1 = 1, i2 = 2;
        if (lam[i0] < lam[i1]) { const int s = i0; i0 = i1; i1 = s; }
        if (lam[i0] < lam[i2]) { const int s = i0; i0 = i2; i2 = s; }
        if (lam[i1] < lam[i2]) { const int s = i1; i1 = i2; i2 = s; }
        const double s1 = sqrt(fmax(lam[i0], 0.0)), s2 = sqrt(fmax(lam[i1], 0.0)), s3 = sqrt(fmax(lam[i2], 0.0));
        double v1[3], v2[3], v3[3], u1[3], u2[3], u3[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { v1[a] = V[a][i0]; v2[a] = V[a][i1]; }
        v3[0] = v1[1] * v2[2] - v1[2] * v2[1]; v3[1] = v1[2] * v2[0] - v1[0] * v2[2]; v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
        double n1 = 0, n2 = 0, d12 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { u1[a] = H[a][0] * v1[0] + H[a][1] * v1[1] + H[a][2] * v1[2]; n1 += u1[a] * u1[a]; }
        n1 = sqrt(n1);
#pragma unroll
        for (int a = 0; a < 3; ++a) u1[a] /= fmax(n1, 1e-300);
#pragma unroll
        for (int a = 0; a < 3; ++a) { u2[a] = H[a][0] * v2[0] + H[a][1] * v2[1] + H[a][2] * v2[2]; d12 += u2[a] * u1[a]; }
#pragma unroll
        for (int a = 0; a < 3; ++a) { u2[a] -= d12 * u1[a]; n2 += u2[a] * u2[a]; }
        n2 = sqrt(n2);
#pragma unroll
        for (int a = 0; a < 3; ++a) u2[a] /= fmax(n2, 1e-300);
        u3[0] = u1[1] * u2[2] - u1[2] * u2[1]; u3[1] = u1[2] * u2[0] - u1[0] * u2[2]; u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
        const double detH = H[0][0] * (H[1][1] * H[2][2] - H[1][2] * H[2][1]) - H[0][1] * (H[1][0] * H[2][2] - H[1][2] * H[2][0]) +
                            H[0][2] * (H[1][0] * H[2][1] - H[1][1] * H[2][0]);
        const double tr = s1 + s2 + (detH < 0 ? -s3 : s3);
        double R[3][3];                               // R = V U^T
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) R[a][c] = v1[a] * u1[c] + v2[a] * u2[c] + v3[a] * u3[c];
        const double sc = tr * nX / nY;
        double tv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) tv[c] = muX[c] - sc * (muY[0] * R[0][c] + muY[1] * R[1][c] + muY[2] * R[2][c]);
        double psum = 0.0;
        for (int j = 0; j < 17; ++j) {
            double e2 = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double al = sc * (p[3 * j] * R[0][c] + p[3 * j + 1] * R[1][c] + p[3 * j + 2] * R[2][c]) + tv[c] - g[3 * j + c];
                e2 += al * al;
            }
            psum += sqrt(e2);
        }
        const float pm = (float)(psum / 17.0);
        o_pmpjpe[(int64_t)b * T + t] = pm;
        atomicAdd(&sred[1], (double)pm);
    }
    __syncthreads();
    if (action_sums && threadIdx.x < EV_COLS) {
        const int a = action[b];
        if (a >= 0 && a < n_actions) atomicAdd(&action_sums[(int64_t)a * EV_COLS + threadIdx.x], sred[threadIdx.x]);
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

void kasf_launch_joint_flip(hipStream_t s, const float* src, float* dst, int64_t rows) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_joint_flip, dim3(grid_for(rows * 51)), dim3(256), 0, s, src, dst, rows * 51);
}
void kasf_launch_tta_merge(hipStream_t s, const float* p, const float* pf, float* out, int64_t rows) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_tta_merge, dim3(grid_for(rows * 51)), dim3(256), 0, s, p, pf, out, rows * 51);
}
void kasf_launch_gather_clips(hipStream_t s, const float* xa, const float* ya, const int64_t* index, const unsigned char* flip, int64_t n_clips, int B,
                              int T, float* xo, float* yo) {
    if (B <= 0) return;
    const int64_t cf = (int64_t)T * 51;
    hipLaunchKernelGGL(k_gather_clips, dim3(grid_for(B * cf)), dim3(256), 0, s, xa, ya, index, flip, n_clips, cf, B * cf, xo, yo);
}
void kasf_launch_eval_metrics(hipStream_t s, const float* pred, const float* label, const float* factor, const float* res, const int* action, int B, int T,
                              int n_actions, float* mpjpe, float* pmpjpe, float* acc, float* jpe, double* action_sums) {
    if (B <= 0) return;
    const size_t sh = (size_t)T * 51 * 2 * sizeof(float);      // the clip and its label staged in LDS: 99 KB at T = 243, i.e. above the 64 KB default limit
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_eval_metrics), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) { kasf_set_error(1000 + (int)e, "eval_metrics: cannot reserve the clip's LDS staging area"); return; }
    hipLaunchKernelGGL(k_eval_metrics, dim3(B), dim3(EV_THR), sh, s, pred, label, factor, res, action, T, n_actions, mpjpe, pmpjpe, acc, jpe, action_sums);
}
