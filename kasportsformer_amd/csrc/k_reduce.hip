// Bit-reproducible per-channel reductions (same seed, same bits).
//
// Every gradient that is a sum over tokens of a per-channel quantity -- LayerNorm gamma / beta, biases, layer scales, the small top-level tensors --
// used to leave its kernel as one fp32 atomic per channel per workgroup: the order of those additions changed from run to run, and with it the last
// bits of the gradient.  Now a workgroup STORES its per-channel sums as one row of a scratch matrix (KasfColSink::take) and a finishing launch per
// backward stage adds the rows of every such matrix in a fixed order (k_col_finish).  GEMM weight gradients already worked this way (per-split
// partial tiles + fixed-order reduce).  No inter-workgroup synchronisation inside any kernel: kernel boundaries order the two phases.
#include "common.h"
#include "kernels.h"

float* KasfColSink::take(int rows, int ld) {
    const int64_t need = ((int64_t)rows * ld + 63) & ~int64_t(63);
    if (scratch == nullptr || used + need > cap) { overflow = true; return nullptr; }
    float* p = scratch + used;
    used += need;
    return p;
}
void KasfColSink::add(const float* part, int ld, int rows, int ncols, float* dst, int mode, const float* a, const float* b, float* dst2) {
    if (part == nullptr || dst == nullptr || rows <= 0) return;
    if (njobs >= KASF_COLJOBS_MAX) { overflow = true; return; }
    jobs[njobs++] = KasfColJob{part, dst, a, b, dst2, rows, ncols, ld, mode};
}

namespace {

constexpr int CF_LAUNCH_JOBS = 48;
struct ColJobs {
    KasfColJob j[CF_LAUNCH_JOBS];
    int first[CF_LAUNCH_JOBS + 1];      // first workgroup of job k (a workgroup owns 16 columns of one job)
    int n;
};

// 256 threads = 16 row lanes x 16 columns (a workgroup owns 16 columns of one job).  Row lane l adds rows l, l + 16, l + 32, ... as eight independent
// chains (the loads of a pass are issued back to back: latency, not bandwidth, is what this kernel is made of) combined in a fixed tree, the sixteen
// lanes meet in LDS in a fixed tree: the result depends on (rows, values) only.
__global__ __launch_bounds__(256) void k_col_finish(const ColJobs js) {
    __shared__ float red[16][17];
    int j = 0;
    while (j + 1 < js.n && (int)blockIdx.x >= js.first[j + 1]) ++j;
    const KasfColJob jb = js.j[j];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = ((int)blockIdx.x - js.first[j]) * 16 + cl;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < jb.ncols) {
        const float* p = jb.part + c;
        int r = rl;
        for (; r + 7 * 16 < jb.rows; r += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += p[(int64_t)(r + 16 * u) * jb.ld];
        }
        for (; r < jb.rows; r += 16) s[0] += p[(int64_t)r * jb.ld];
    }
    red[rl][cl] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (rl == 0 && c < jb.ncols) {
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = red[k][cl];
        const float v = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) + (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15])));
        if (jb.mode == 0) jb.dst[c] += v;
        else {                          // fc2 finish: v = colsum(g);  db2 = ls2 . v;  dls2 += b2 . v   (the W2 . G term of dls2 is already there)
            jb.dst[c] = jb.b[c] * v;
            jb.dst2[c] += jb.a[c] * v;
        }
    }
}

}  // namespace

void kasf_col_flush(hipStream_t s, KasfColSink* const* sinks, int nsinks) {
    ColJobs js;
    js.n = 0;
    int wg = 0;
    auto launch = [&]() {
        if (js.n == 0) return;
        js.first[js.n] = wg;
        hipLaunchKernelGGL(k_col_finish, dim3((unsigned)wg), dim3(256), 0, s, js);
        js.n = 0;
        wg = 0;
    };
    for (int k = 0; k < nsinks; ++k) {
        KasfColSink* sk = sinks[k];
        if (sk == nullptr) continue;
        if (sk->overflow) kasf_set_error(6, "column-reduction scratch or job table too small (KasfColSink)");
        for (int q = 0; q < sk->njobs; ++q) {
            if (js.n == CF_LAUNCH_JOBS) launch();
            js.j[js.n] = sk->jobs[q];
            js.first[js.n] = wg;
            wg += (sk->jobs[q].ncols + 15) / 16;
            ++js.n;
        }
        sk->njobs = 0;
        sk->used = 0;
        sk->overflow = false;
    }
    launch();
}
