// Issue-port probe for gfx950, round 5: WHEN does another wave's vector work hide behind MFMAs?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/issue_probe tools/issue_probe.hip && /tmp/issue_probe
// One workgroup per CU (100 KB of LDS), 256 workgroups, threads / 256 = waves per SIMD.  Two families:
//   ROLE   waves 0-3 issue MFMAs only (optionally an s_nop after each), every other wave K vector instructions per MFMA of the first kind
//   SYM    every wave issues the same stream: 1 MFMA followed by K vector instructions
// Reported: wall time, and ns per MFMA per SIMD (the matrix pipe's floor is 6.9 ns for 16x16x32, 13.8 ns for 32x32x16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int REPS = 1000, GROUPS = 32;      // MFMAs per wave = REPS * GROUPS

enum { V_FMA = 0, V_PK16 = 1 };

template <int VK> __device__ __forceinline__ void vop(float& a, float k, float c) {
    if (VK == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(k), "v"(c));
    else asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a) : "v"(k), "v"(c));
}
template <int NOP> __device__ __forceinline__ void nop() {
    if (NOP == 1) asm volatile("s_nop 0");
    if (NOP == 2) asm volatile("s_nop 1");
    if (NOP == 3) asm volatile("s_nop 3");
    if (NOP == 4) asm volatile("s_nop 7");
    if (NOP == 5) asm volatile("s_nop 11");
}

// MODE 0: ROLE (16x16x32), MODE 1: SYM (16x16x32), MODE 2: SYM (32x32x16), MODE 3: ROLE (32x32x16), MODE 4: vector only (K per group, every wave)
template <int MODE, int K, int VK, int NOP> __global__ void probe(float* out) {
    extern __shared__ char pad[];
    if (threadIdx.x == 0 && out == nullptr) pad[0] = 1;
    float a[4] = {threadIdx.x * 1e-3f, 1.f, 2.f, 3.f};
    const float k = 0.999f, c = 1e-3f;
    f32x4 acc[4] = {};
    f32x16 big[2] = {};
    bf16x8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(a[0] + e); fb[e] = (__bf16)(a[1] - e); }
    const bool mfma_role = (threadIdx.x >> 8) == 0;
    if (MODE == 0 || MODE == 3) {                    // the role branch sits OUTSIDE the instruction stream (one branch per wave, not one per group)
        if (mfma_role) {
            for (int r = 0; r < REPS; ++r) {
#pragma unroll
                for (int g = 0; g < GROUPS; ++g) {
                    if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(fa), "v"(fb));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[g & 1]) : "v"(fa), "v"(fb));
                    nop<NOP>();
                }
            }
        } else {
            for (int r = 0; r < REPS; ++r) {
#pragma unroll
                for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
                    for (int j = 0; j < K; ++j) vop<VK>(a[j & 3], k, c);
                }
            }
        }
    } else {
        for (int r = 0; r < REPS; ++r) {
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                if (MODE == 1) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(fa), "v"(fb));
#pragma unroll
                    for (int j = 0; j < K; ++j) vop<VK>(a[j & 3], k, c);
                } else if (MODE == 2) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[g & 1]) : "v"(fa), "v"(fb));
#pragma unroll
                    for (int j = 0; j < K; ++j) vop<VK>(a[j & 3], k, c);
                } else {
#pragma unroll
                    for (int j = 0; j < K; ++j) vop<VK>(a[j & 3], k, c);
                }
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + big[0][0] + big[1][5];
}

template <int MODE, int K, int VK, int NOP> void run(const char* name, int threads, float* out) {
    const int blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, K, VK, NOP>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, K, VK, NOP>), dim3(blocks), dim3(threads), 100 * 1024, 0, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, K, VK, NOP>), dim3(blocks), dim3(threads), 100 * 1024, 0, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int wps = threads / 256;
    const double groups = (double)REPS * GROUPS;                     // per wave
    const double mfma_waves = (MODE == 0 || MODE == 3) ? 1 : (MODE == 4 ? 0 : wps);
    const double vec_waves = (MODE == 0 || MODE == 3) ? wps - 1 : wps;
    printf("%-58s %d waves/SIMD  wall %7.3f ms   per SIMD: %6.2f ns/MFMA  %6.3f ns/vector-instr   (MFMA:vector per SIMD = %g:%g)\n", name, wps, ms,
           mfma_waves > 0 ? ms * 1e6 / (groups * mfma_waves) : 0.0, K * vec_waves > 0 ? ms * 1e6 / (groups * K * vec_waves) : 0.0, mfma_waves, K * vec_waves);
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 1024 * 4);
    // ROLE: one MFMA wave, one vector wave
    run<0, 1, V_FMA, 0>("role16: mfma b2b | 1 wave x 1 fma", 512, out);
    run<0, 2, V_FMA, 0>("role16: mfma b2b | 1 wave x 2 fma", 512, out);
    run<0, 3, V_FMA, 0>("role16: mfma b2b | 1 wave x 3 fma", 512, out);
    run<0, 4, V_FMA, 0>("role16: mfma b2b | 1 wave x 4 fma", 512, out);
    run<0, 2, V_FMA, 0>("role16: mfma b2b | 3 waves x 2 fma", 1024, out);
    run<0, 3, V_FMA, 1>("role16: mfma + s_nop 0 | 1 wave x 3 fma", 512, out);
    run<0, 3, V_FMA, 2>("role16: mfma + s_nop 1 | 1 wave x 3 fma", 512, out);
    run<0, 3, V_FMA, 3>("role16: mfma + s_nop 3 | 1 wave x 3 fma", 512, out);
    run<0, 3, V_FMA, 4>("role16: mfma + s_nop 7 | 1 wave x 3 fma", 512, out);
    run<0, 2, V_FMA, 3>("role16: mfma + s_nop 3 | 3 waves x 2 fma", 1024, out);
    run<0, 2, V_FMA, 4>("role16: mfma + s_nop 7 | 3 waves x 2 fma", 1024, out);
    run<0, 3, V_PK16, 0>("role16: mfma b2b | 1 wave x 3 pk_f16", 512, out);
    run<0, 3, V_PK16, 3>("role16: mfma + s_nop 3 | 1 wave x 3 pk_f16", 512, out);
    run<3, 2, V_FMA, 0>("role32: mfma32 b2b | 1 wave x 2 fma", 512, out);
    run<3, 4, V_FMA, 0>("role32: mfma32 b2b | 1 wave x 4 fma", 512, out);
    run<3, 6, V_FMA, 0>("role32: mfma32 b2b | 1 wave x 6 fma", 512, out);
    run<3, 8, V_FMA, 0>("role32: mfma32 b2b | 1 wave x 8 fma", 512, out);
    run<3, 6, V_FMA, 4>("role32: mfma32 + s_nop 7 | 1 wave x 6 fma", 512, out);
    run<3, 6, V_FMA, 5>("role32: mfma32 + s_nop 11 | 1 wave x 6 fma", 512, out);
    run<3, 4, V_FMA, 0>("role32: mfma32 b2b | 3 waves x 4 fma", 1024, out);
    // vector only, for reference
    run<4, 4, V_FMA, 0>("fma only (4 per group)", 256, out);
    run<4, 4, V_FMA, 0>("fma only (4 per group)", 512, out);
    run<4, 4, V_FMA, 0>("fma only (4 per group)", 1024, out);
    run<4, 4, V_PK16, 0>("pk_f16 only (4 per group)", 512, out);
    run<4, 4, V_PK16, 0>("pk_f16 only (4 per group)", 1024, out);
    // SYM 16x16x32
    run<1, 0, V_FMA, 0>("sym16: mfma only", 256, out);
    run<1, 0, V_FMA, 0>("sym16: mfma only", 512, out);
    run<1, 1, V_FMA, 0>("sym16: 1 mfma : 1 fma", 512, out);
    run<1, 2, V_FMA, 0>("sym16: 1 mfma : 2 fma", 512, out);
    run<1, 3, V_FMA, 0>("sym16: 1 mfma : 3 fma", 512, out);
    run<1, 4, V_FMA, 0>("sym16: 1 mfma : 4 fma", 512, out);
    run<1, 5, V_FMA, 0>("sym16: 1 mfma : 5 fma", 512, out);
    run<1, 6, V_FMA, 0>("sym16: 1 mfma : 6 fma", 512, out);
    run<1, 8, V_FMA, 0>("sym16: 1 mfma : 8 fma", 512, out);
    run<1, 3, V_FMA, 0>("sym16: 1 mfma : 3 fma", 256, out);
    run<1, 5, V_FMA, 0>("sym16: 1 mfma : 5 fma", 256, out);
    run<1, 3, V_FMA, 0>("sym16: 1 mfma : 3 fma", 1024, out);
    run<1, 5, V_FMA, 0>("sym16: 1 mfma : 5 fma", 1024, out);
    run<1, 8, V_FMA, 0>("sym16: 1 mfma : 8 fma", 1024, out);
    run<1, 3, V_PK16, 0>("sym16: 1 mfma : 3 pk_f16", 512, out);
    run<1, 5, V_PK16, 0>("sym16: 1 mfma : 5 pk_f16", 512, out);
    run<1, 5, V_PK16, 0>("sym16: 1 mfma : 5 pk_f16", 1024, out);
    // SYM 32x32x16 (same FLOPs per MFMA pipe time: compare K with 2 x the 16x16x32 K)
    run<2, 0, V_FMA, 0>("sym32: mfma32 only", 512, out);
    run<2, 4, V_FMA, 0>("sym32: 1 mfma32 : 4 fma", 512, out);
    run<2, 6, V_FMA, 0>("sym32: 1 mfma32 : 6 fma", 512, out);
    run<2, 8, V_FMA, 0>("sym32: 1 mfma32 : 8 fma", 512, out);
    run<2, 10, V_FMA, 0>("sym32: 1 mfma32 : 10 fma", 512, out);
    run<2, 12, V_FMA, 0>("sym32: 1 mfma32 : 12 fma", 512, out);
    run<2, 10, V_FMA, 0>("sym32: 1 mfma32 : 10 fma", 256, out);
    run<2, 10, V_FMA, 0>("sym32: 1 mfma32 : 10 fma", 1024, out);
    run<2, 10, V_PK16, 0>("sym32: 1 mfma32 : 10 pk_f16", 512, out);
    run<2, 10, V_PK16, 0>("sym32: 1 mfma32 : 10 pk_f16", 1024, out);
    return 0;
}
