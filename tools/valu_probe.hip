// Issue-rate probe for gfx950: cycles per wave64 instruction of the VALU forms the MLP kernels are made of.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_probe tools/valu_probe.hip && /tmp/valu_probe
// Each case runs REPS x 64 instructions in a loop per wave.  Run as `valu_probe 256`: 256 workgroups pinned one per CU by 100 KB of LDS, so
// 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int REPS = 2000;

#define R16(x) x x x x x x x x x x x x x x x x
template <int CASE> __global__ void probe(float* out, long long* cyc) {
    extern __shared__ char pad[];                       // 100 KB of dynamic LDS: one workgroup per CU, so threads / 256 = waves per SIMD
    if (threadIdx.x == 0 && out == nullptr) pad[0] = 1;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, k = 0.999f, c = 1e-3f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0}, pk = {k, k}, pc = {c, c};
    f32x4 acc = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    bf16x8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(a0 + e); fb[e] = (__bf16)(a1 - e); }
    const long long t0 = clock64();
    for (int r = 0; r < REPS; ++r) {
        if (CASE == 0) { R16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
        if (CASE == 1) { R16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pk), "v"(pc));) }
        if (CASE == 2) { R16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(k), "v"(c));) }
        if (CASE == 3) { R16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pk), "v"(pc));) }
        if (CASE == 4) { R16(asm volatile("v_exp_f32 %0, %4\n v_exp_f32 %1, %4\n v_exp_f32 %2, %4\n v_exp_f32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));) }
        if (CASE == 5) { R16(asm volatile("v_rcp_f32 %0, %4\n v_rcp_f32 %1, %4\n v_rcp_f32 %2, %4\n v_rcp_f32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));) }
        if (CASE == 6) { R16(asm volatile("v_mul_f32 %0, |%0|, %4\n v_mul_f32 %1, %1, -%4\n v_min_f32 %2, |%2|, %4\n v_max_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));) }
        if (CASE == 7) { R16(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %1, %2\n v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %1, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (CASE == 8) {   // MFMA only, two independent accumulators
            R16(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+v"(acc), "+v"(acc1) : "v"(fa), "v"(fb));)
        }
        if (CASE == 9) {   // one MFMA to three packed FMAs (4 "instructions" per group, 16 groups)
            R16(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n v_pk_fma_f32 %1, %1, %7, %8\n v_pk_fma_f32 %2, %2, %7, %8\n v_pk_fma_f32 %3, %3, %7, %8" : "+v"(acc), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(fa), "v"(fb), "v"(pk), "v"(pc));)
        }
        if (CASE == 10) {  // one MFMA to seven VALU (8 per group, 8 groups)
#define G8 asm volatile("v_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n v_pk_fma_f32 %1, %1, %7, %8\n v_pk_fma_f32 %2, %2, %7, %8\n v_pk_fma_f32 %3, %3, %7, %8\n v_pk_fma_f32 %4, %4, %7, %8\n v_pk_fma_f32 %1, %1, %7, %8\n v_pk_fma_f32 %2, %2, %7, %8\n v_pk_fma_f32 %3, %3, %7, %8" : "+v"(acc), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(fa), "v"(fb), "v"(pk), "v"(pc));
            G8 G8 G8 G8 G8 G8 G8 G8
        }
        if (CASE == 12 || CASE == 13) {   // waves 0-3 of the workgroup: MFMA only; waves 4-7 (same SIMDs): VALU only
            if ((threadIdx.x >> 8) == 0) {
                R16(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+v"(acc), "+v"(acc1) : "v"(fa), "v"(fb));)
            } else if (CASE == 12) {
                R16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pk), "v"(pc));)
            } else {
                R16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));)
            }
        }
        if (CASE == 14) {  // one MFMA (4 independent accumulators in rotation) to seven plain FMAs
#define H8(A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n v_fma_f32 %1, %1, %7, %8\n v_fma_f32 %2, %2, %7, %8\n v_fma_f32 %3, %3, %7, %8\n v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %1, %1, %7, %8\n v_fma_f32 %2, %2, %7, %8\n v_fma_f32 %3, %3, %7, %8" : "+v"(A), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fa), "v"(fb), "v"(k), "v"(c));
            H8(acc) H8(acc1) H8(acc) H8(acc1) H8(acc) H8(acc1) H8(acc) H8(acc1)
        }
        if (CASE >= 15 && CASE <= 18) {   // round 4: waves 0-3 MFMA only (one per SIMD); EVERY other wave of the workgroup (2 or 3 more per SIMD) plain or packed fp32,
                                          // three times the instruction count each: how much vector issue is left beside a saturated matrix stream?
            if ((threadIdx.x >> 8) == 0) {
                R16(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+v"(acc), "+v"(acc1) : "v"(fa), "v"(fb));)
            } else if (CASE == 15 || CASE == 17) {
                for (int rr = 0; rr < 3; ++rr) { R16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
            } else {
                for (int rr = 0; rr < 3; ++rr) { R16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pk), "v"(pc));) }
            }
        }
        if (CASE == 24 || CASE == 25) {   // round 4: does PRIORITY buy the overlap?  MFMA wave at priority 0, vector waves at priority 3 (24) or the reverse (25)
            if ((threadIdx.x >> 8) == 0) {
                if (r == 0) { if (CASE == 25) __builtin_amdgcn_s_setprio(3); }
                R16(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+v"(acc), "+v"(acc1) : "v"(fa), "v"(fb));)
            } else {
                if (r == 0) { if (CASE == 24) __builtin_amdgcn_s_setprio(3); }
                for (int rr = 0; rr < 3; ++rr) { R16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
            }
        }
        if (CASE >= 20 && CASE <= 23) {   // round 4: the same with v_mfma_f32_32x32x16_bf16 (32 cycles in the pipe, vector issue held for 8 of them: 24 free against 8 of 16)
            static_assert(true, "");
            if ((threadIdx.x >> 8) == 0 || CASE == 20) {
                f32x16 c0 = {0}, c1 = {0};
                for (int q = 0; q < 1; ++q) {
                    R16(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(c0), "+v"(c1) : "v"(fa), "v"(fb));)
                }
                acc[0] += c0[0] + c1[5];
            } else if (CASE == 21 || CASE == 23) {
                for (int rr = 0; rr < 3; ++rr) { R16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
            } else {
                for (int rr = 0; rr < 3; ++rr) { R16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pk), "v"(pc));) }
            }
        }
        if (CASE == 19) {   // the vector side of cases 15 / 17 alone (no MFMA wave): 3 x the instructions per wave
            for (int rr = 0; rr < 3; ++rr) { R16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
        }
        // round 5: packed fp16 (v_pk_*_f16): two elements per lane -- at the plain-fp32 issue cost?  beside another wave's MFMAs?
        if (CASE == 30) { R16(asm volatile("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
        if (CASE == 31) { R16(asm volatile("v_pk_mul_f16 %0, %0, %4\n v_pk_min_f16 %1, %1, %5\n v_pk_max_f16 %2, %2, %4\n v_pk_add_f16 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
        if (CASE == 32) { R16(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_cvt_pk_f16_f32 %3, %1, %2\n v_cvt_pk_f16_f32 %0, %1, %2\n v_cvt_pk_f16_f32 %3, %1, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (CASE == 33 || CASE == 36 || CASE == 37) {   // waves 0-3 MFMA only; the other waves (1, 2 or 3 per SIMD) v_pk_fma_f16 (x3 the instructions in 36 / 37)
            if ((threadIdx.x >> 8) == 0) {
                R16(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+v"(acc), "+v"(acc1) : "v"(fa), "v"(fb));)
            } else {
                for (int rr = 0; rr < (CASE == 33 ? 1 : 3); ++rr) { R16(asm volatile("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
            }
        }
        if (CASE == 34) {   // f16 MFMA only
            R16(asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1" : "+v"(acc), "+v"(acc1) : "v"(fa), "v"(fb));)
        }
        if (CASE == 35) {  // one MFMA (2 accumulators in rotation) to seven v_pk_fma_f16 in the SAME wave
#define P8(A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n v_pk_fma_f16 %1, %1, %7, %8\n v_pk_fma_f16 %2, %2, %7, %8\n v_pk_fma_f16 %3, %3, %7, %8\n v_pk_fma_f16 %4, %4, %7, %8\n v_pk_fma_f16 %1, %1, %7, %8\n v_pk_fma_f16 %2, %2, %7, %8\n v_pk_fma_f16 %3, %3, %7, %8" : "+v"(A), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fa), "v"(fb), "v"(k), "v"(c));
            P8(acc) P8(acc1) P8(acc) P8(acc1) P8(acc) P8(acc1) P8(acc) P8(acc1)
        }
        if (CASE == 38) {   // the vector side of 36 / 37 alone
            for (int rr = 0; rr < 3; ++rr) { R16(asm volatile("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
        }
        if (CASE == 39) {   // the MLP forward's shape: one wave per SIMD MFMA-heavy (waves 0-3), one wave per SIMD issuing 1 MFMA : 9 v_pk_fma_f16 (waves 4-7)
            if ((threadIdx.x >> 8) == 0) {
                R16(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+v"(acc), "+v"(acc1) : "v"(fa), "v"(fb));)
            } else {
                P8(acc) P8(acc1) P8(acc) P8(acc1) P8(acc) P8(acc1) P8(acc) P8(acc1)
            }
        }
        if (CASE == 40) {   // ... and the same with packed fp32 (what k_mlp_fwd_s issues today)
            if ((threadIdx.x >> 8) == 0) {
                R16(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+v"(acc), "+v"(acc1) : "v"(fa), "v"(fb));)
            } else {
                G8 G8 G8 G8 G8 G8 G8 G8
            }
        }
        if (CASE == 41) { R16(asm volatile("v_fma_mix_f32 %0, %4, %0, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %1, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %4, %2, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %3, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
        if (CASE == 42) {   // waves 0-3 MFMA only; waves 4-7 v_fma_mix_f32
            if ((threadIdx.x >> 8) == 0) {
                R16(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+v"(acc), "+v"(acc1) : "v"(fa), "v"(fb));)
            } else {
                R16(asm volatile("v_fma_mix_f32 %0, %4, %0, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %1, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %4, %2, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %3, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));)
            }
        }
        if (CASE == 43) { R16(asm volatile("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16 %2, %5\n v_cvt_f32_f16_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c));) }
        if (CASE == 11) { R16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pk), "v"(pc));) }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0[0] + p1[1] + p2[0] + p3[1] + acc[0] + acc1[1];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static int g_blocks = 1024;
template <int CASE> void run(const char* name, int threads, float* out, long long* cyc) {
    const int blocks = g_blocks;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<CASE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<CASE>, dim3(blocks), dim3(threads), 100 * 1024, 0, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<CASE>, dim3(blocks), dim3(threads), 100 * 1024, 0, out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
    // 1024 workgroups over 256 CUs: 4 rounds if one workgroup per CU at a time is resident... report per-wave view from clock64 and the wall view
    const double ninstr = (double)REPS * 64;
    const double waves_per_simd = (double)blocks * threads / 64 / 1024;
    printf("%-34s blocks %4d thr %3d  clock64/instr/wave %6.2f   wall %7.3f ms   ns/instr/SIMD %6.3f  (%.1f waves/SIMD)\n", name, blocks, threads, avg / ninstr, ms,
           ms * 1e6 / (ninstr * waves_per_simd), waves_per_simd);
}
#define BOTH(C, N) run<C>(N, 256, out, cyc); run<C>(N, 512, out, cyc); run<C>(N, 1024, out, cyc);
int main(int argc, char** argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 1024 * 8);
    BOTH(0, "v_fma_f32 x4 chains")
    BOTH(1, "v_pk_fma_f32 x4 chains")
    BOTH(2, "v_fma_f32 one chain")
    BOTH(3, "v_pk_fma_f32 one chain")
    BOTH(4, "v_exp_f32")
    BOTH(5, "v_rcp_f32")
    BOTH(6, "v_mul/min/max with modifiers")
    BOTH(7, "v_cvt_pk_bf16_f32")
    BOTH(8, "mfma 16x16x32 bf16 only")
    BOTH(9, "1 mfma : 3 pk_fma")
    BOTH(10, "1 mfma : 7 pk_fma")
    BOTH(11, "v_pk_mul / v_pk_add")
    run<12>("waves 0-3 mfma | waves 4-7 pk_fma", 512, out, cyc);
    run<13>("waves 0-3 mfma | waves 4-7 fma", 512, out, cyc);
    BOTH(14, "1 mfma : 7 fma")
    // round 4: vector issue beside a saturated MFMA wave.  "ns/instr/SIMD" of these rows is per instruction of ALL waves; read the wall time:
    // MFMA wave alone 0.945 ms (row 'mfma only', 256 threads); vector waves alone: rows 'fma x3 alone'
    run<19>("fma x3 alone, 2 waves/SIMD", 512, out, cyc);
    run<19>("fma x3 alone, 3 waves/SIMD", 768, out, cyc);
    run<15>("w0-3 mfma | 2 waves/SIMD fma x3", 768, out, cyc);
    run<17>("w0-3 mfma | 3 waves/SIMD fma x3", 1024, out, cyc);
    run<16>("w0-3 mfma | 2 waves/SIMD pk_fma x3", 768, out, cyc);
    run<18>("w0-3 mfma | 3 waves/SIMD pk_fma x3", 1024, out, cyc);
    // 32x32x16: per loop iteration 32 MFMAs of 32 cycles = the pipe time of the 64 16x16x32 MFMAs above (same FLOPs)
    run<20>("mfma 32x32x16 only (32 per rep)", 256, out, cyc);
    run<21>("w0-3 mfma32 | 2 waves/SIMD fma x3", 768, out, cyc);
    run<23>("w0-3 mfma32 | 3 waves/SIMD fma x3", 1024, out, cyc);
    run<22>("w0-3 mfma32 | 2 waves/SIMD pk_fma x3", 768, out, cyc);
    run<24>("w0-3 mfma prio0 | 2 waves fma x3 PRIO 3", 768, out, cyc);
    run<25>("w0-3 mfma PRIO 3 | 2 waves fma x3 prio0", 768, out, cyc);
    // round 5: packed fp16
    BOTH(30, "v_pk_fma_f16 x4 chains")
    BOTH(31, "v_pk_mul/min/max/add_f16")
    BOTH(32, "v_cvt_pk_f16_f32")
    BOTH(34, "mfma 16x16x32 f16 only")
    BOTH(35, "1 mfma : 7 pk_fma_f16 (same wave)")
    run<33>("waves 0-3 mfma | waves 4-7 pk_fma_f16", 512, out, cyc);
    run<38>("pk_fma_f16 x3 alone, 2 waves/SIMD", 512, out, cyc);
    run<36>("w0-3 mfma | 2 waves/SIMD pk_fma_f16 x3", 768, out, cyc);
    run<37>("w0-3 mfma | 3 waves/SIMD pk_fma_f16 x3", 1024, out, cyc);
    run<39>("w0-3 mfma | w4-7 1 mfma : 7 pk_fma_f16", 512, out, cyc);
    run<40>("w0-3 mfma | w4-7 1 mfma : 7 pk_fma_f32", 512, out, cyc);
    BOTH(41, "v_fma_mix_f32 (f16 x f32 + f32)")
    run<42>("waves 0-3 mfma | waves 4-7 fma_mix", 512, out, cyc);
    BOTH(43, "v_cvt_f32_f16 (plain / sdwa hi)")
    return 0;
}
