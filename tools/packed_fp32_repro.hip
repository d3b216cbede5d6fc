// Repro attempt for the round-3 finding: k_gcn_bwd2_* gave results that depended on what else shared the SIMD while hipcc's SLP
// vectoriser had turned its scalar BatchNorm-backward arithmetic into v_pk_add_f32 / v_pk_mul_f32 with op_sel / op_sel_hi broadcasts.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/pk_repro tools/packed_fp32_repro.hip && /tmp/pk_repro          (SLP on: the packed form)
//   hipcc ... -fno-slp-vectorize ...                                                                                              (the shipped flag: scalar form)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off --cuda-device-only -S -o - tools/packed_fp32_repro.hip | grep v_pk_      (the ISA in question)
// bn_bwd_like is dy_math / dy_chunk of csrc/k_gcn.hip statement for statement (per-node table in LDS, 8 channels per lane, bf16 in / out); mfma_hog keeps
// every SIMD's matrix pipe busy from a second stream with workgroups small enough to co-reside.  Expected: every launch of bn_bwd_like gives the bits of
// the host's scalar evaluation (no fma contraction on either side), alone and beside mfma_hog.  Observed: profiles/r4_packed_fp32_repro.txt.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NODES = 17, LD = 5;

__global__ __launch_bounds__(256) void bn_bwd_like(const __hip_bfloat16* __restrict__ rbuf, const __hip_bfloat16* __restrict__ y, const float* __restrict__ coefg,
                                                   __hip_bfloat16* __restrict__ out, int n_items) {
    __shared__ float sC[NODES * LD];
    if (threadIdx.x < NODES * LD) sC[threadIdx.x] = coefg[threadIdx.x];
    __syncthreads();
    const int sub = threadIdx.x & 15;
    for (int item = blockIdx.x * 256 + threadIdx.x; item < n_items; item += gridDim.x * 256) {
        const int tk = item >> 4, node = tk % NODES;
        float r[8], c[8], dy[8];
        for (int e = 0; e < 8; ++e) { r[e] = __bfloat162float(rbuf[(size_t)tk * 128 + sub * 8 + e]); c[e] = __bfloat162float(y[(size_t)tk * 128 + sub * 8 + e]); }
        const float sc = sC[node * LD], mean = sC[node * LD + 1], rstd = sC[node * LD + 2], c1 = sC[node * LD + 3], c2 = sC[node * LD + 4];
#pragma unroll
        for (int e = 0; e < 8; ++e) dy[e] = sc * (r[e] - c1 - (c[e] - mean) * rstd * c2);
        for (int e = 0; e < 8; ++e) out[(size_t)tk * 128 + sub * 8 + e] = __float2bfloat16(dy[e]);
    }
}

__global__ __launch_bounds__(64) void mfma_hog(float* sink, int reps) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (threadIdx.x + e)); b[e] = (__bf16)(0.02f * (threadIdx.x - e)); }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc1, 0, 0, 0);
    }
    sink[blockIdx.x * 64 + threadIdx.x] = acc0[0] + acc1[1];
}

static float bf(float v) { __hip_bfloat16 h = __float2bfloat16(v); return __bfloat162float(h); }
int main() {
    const int M = 117504, n_items = M * 16;
    std::vector<__hip_bfloat16> hr((size_t)M * 128), hy((size_t)M * 128), hout((size_t)M * 128), href((size_t)M * 128), first((size_t)M * 128);
    std::vector<float> coef(NODES * LD);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (size_t i = 0; i < hr.size(); ++i) { hr[i] = __float2bfloat16(rnd() * 1e-3f); hy[i] = __float2bfloat16(rnd() * 2.f); }
    for (int n = 0; n < NODES; ++n) { coef[n * LD] = 0.5f + rnd() * 0.3f; coef[n * LD + 1] = rnd() * 0.1f; coef[n * LD + 2] = 1.f + rnd() * 0.2f; coef[n * LD + 3] = rnd() * 1e-5f; coef[n * LD + 4] = rnd() * 1e-5f; }
    for (int tk = 0; tk < M; ++tk) {                                   // the host's scalar evaluation, same operation order, no contraction (-ffp-contract=off)
        const float* C = &coef[(tk % NODES) * LD];
        for (int e = 0; e < 128; ++e) {
            const float r = __bfloat162float(hr[(size_t)tk * 128 + e]), c = __bfloat162float(hy[(size_t)tk * 128 + e]);
            volatile float t1 = c - C[1]; volatile float t2 = t1 * C[2]; volatile float t3 = t2 * C[4]; volatile float t4 = r - C[3]; volatile float t5 = t4 - t3;
            href[(size_t)tk * 128 + e] = __float2bfloat16(C[0] * t5);
        }
    }
    __hip_bfloat16 *dr, *dyv, *dout; float *dcoef, *sink;
    hipMalloc(&dr, hr.size() * 2); hipMalloc(&dyv, hr.size() * 2); hipMalloc(&dout, hr.size() * 2); hipMalloc(&dcoef, coef.size() * 4); hipMalloc(&sink, 4096 * 64 * 4);
    hipMemcpy(dr, hr.data(), hr.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dyv, hy.data(), hr.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dcoef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int mode = 0; mode < 2; ++mode) {                             // 0: alone, 1: beside the MFMA kernel
        long bad_ref = 0, bad_first = 0, launches = 0;
        for (int it = 0; it < 20; ++it) {
            hipMemsetAsync(dout, 0, hr.size() * 2, s1);
            if (mode == 1) hipLaunchKernelGGL(mfma_hog, dim3(4096), dim3(64), 0, s2, sink, 20000);
            for (int k = 0; k < (mode == 1 ? 8 : 1); ++k) hipLaunchKernelGGL(bn_bwd_like, dim3(4096), dim3(256), 0, s1, dr, dyv, dcoef, dout, n_items);
            hipDeviceSynchronize();
            hipMemcpy(hout.data(), dout, hr.size() * 2, hipMemcpyDeviceToHost);
            if (mode == 0 && it == 0) first = hout;
            for (size_t i = 0; i < hout.size(); ++i) {
                bad_ref += memcmp(&hout[i], &href[i], 2) != 0;
                bad_first += memcmp(&hout[i], &first[i], 2) != 0;
            }
            ++launches;
        }
        printf("%s: %ld passes of %zu elements: %ld differ from the host's scalar evaluation, %ld differ from the first device pass\n",
               mode ? "beside mfma_hog" : "alone          ", launches, hout.size(), bad_ref, bad_first);
    }
    return 0;
}
