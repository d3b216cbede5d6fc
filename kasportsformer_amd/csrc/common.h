// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the KASportsFormer path.
// Everything here is written for 64-wide wavefronts and the 16x16 MFMA shapes:
//   bf16 : v_mfma_f32_16x16x32_bf16  (8 k per lane)          -- fast mode
//   f32  : v_mfma_f32_16x16x4_f32    (exact f32, 1 k per lane) -- parity mode
// Both share the C/D map  col = lane&15, row = 4*(lane>>4)+reg.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define KASF_C 128          // dim_feat
#define KASF_J 17           // joints
#define KASF_H 8            // heads
#define KASF_D 16           // head dim
#define KASF_LN_EPS 1e-5f

__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }

// ---- 8 consecutive elements <-> 8 floats (16 B for bf16, 2 x 16 B for f32) ----
__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store4(bf16* p, const float (&v)[4]) {
    bf16x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (bf16)v[i];
    *reinterpret_cast<bf16x4*>(p) = t;
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = v[i];
    *reinterpret_cast<f32x4*>(p) = a;
}

// Sum over an aligned group of 16 lanes (one DPP "row"), result in every lane.  Pure VALU (v_add with DPP operand
// modifiers): quad xor-1, quad xor-2, mirror within 8, mirror within 16.  __shfl_xor would lower to ds_bpermute_b32,
// i.e. 4 dependent LDS round trips (~100+ cycles each) per reduction -- the LayerNorm paths do 2-6 reductions per row.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
    return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float reduce16(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    v = dpp_add<0x140>(v);     // row_mirror
    return v;
}
__device__ __forceinline__ float reduce64(float v) {
    v = reduce16(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// Exact-erf GELU (nn.GELU default), erf from Abramowitz-Stegun with ONE exponential: exp(-x^2/2) is both erf's tail factor
// for the argument x/sqrt(2) and the Gaussian density of the derivative.
//   s(x) = 1/2 (1 - erf(|x|/sqrt2)) = 1/2 poly(t) e,  t = 1/(1 + p |x|/sqrt2),  e = exp(-x^2/2)      (fp32 mode; bf16 mode below)
//   GELU(x) = max(x,0) - |x| s           Phi(x) = x >= 0 ? 1 - s : s           GELU'(x) = Phi + x e / sqrt(2 pi)
// FAST = false (fp32 parity mode): 7.1.26, five terms, |erf error| <= 1.5e-7 (fp32 rounding level).
// FAST = true  (bf16 mode):        no reciprocal: s = e . G(|x|) with G(a) = 1/2 erfcx(a / sqrt2) (the scaled Mills ratio, entire and slowly
//                                  varying) as a degree-6 polynomial fitted on [0, 6] in the e-weighted minimax sense, |s error| <= 1.6e-5 (the
//                                  three-term 7.1.25 it replaces: 1.1e-5), two orders below bf16 rounding of the result.  v_rcp_f32 and v_exp_f32
//                                  run at quarter rate; the six FMAs pair up into v_pk_fma_f32 and cost less than the reciprocal alone.
//                                  |x| is clamped at 8 (e(8) = 1e-14) so that the polynomial cannot overflow against e = 0.
// ~12-14 VALU instructions instead of ~35 for libm erff + expf; a '/' would expand to a 9-instruction IEEE sequence.
template <bool FAST> __device__ __forceinline__ void gelu_tail(float x, float& s, float& e) {
    if (FAST) {
        const float a = fminf(fabsf(x), 8.0f);
        e = __builtin_amdgcn_exp2f(a * a * -0.72134752044448170f);
        const float G = fmaf(a, fmaf(a, fmaf(a, fmaf(a, fmaf(a, fmaf(a, 7.042173346e-04f, -8.041790507e-03f), 3.967198035e-02f), -1.169407755e-01f),
                                             2.444233516e-01f), -3.982094769e-01f), 4.999843037e-01f);
        s = G * e;
    } else {
        const float ax = fabsf(x) * 0.70710678118654752f;
        e = __expf(-ax * ax);
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
        s = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 0.5307027145f, -0.7265760135f), 0.7107068705f), -0.142248368f), 0.127414796f) * e;
    }
}
template <typename T> struct GeluMode { static constexpr bool FAST = false; };
template <> struct GeluMode<bf16> { static constexpr bool FAST = true; };

template <typename T = float> __device__ __forceinline__ float gelu_f(float x) {
    float s, e;
    gelu_tail<GeluMode<T>::FAST>(x, s, e);
    return fmaf(-fabsf(x), s, fmaxf(x, 0.0f));
}
// Round 5, the forward pass's form in bf16 mode: the same odd polynomial evaluated in PACKED FP16 (v_pk_fma_f16: two elements per lane, the issue cost of a
// plain fp32 instruction, and -- unlike v_pk_*_f32 -- beside another wave's MFMAs instead of in the matrix pipe: tools/valu_probe.hip cases 30-40,
// profiles/r5_valu_probe.txt), its result kept as FP16: the hidden activation reaches GEMM2 (v_mfma_f32_16x16x32_f16 against an fp16 copy of W2) with
// 11 significant bits instead of the 8 of bf16.
//   xh = f16(x);  u = clamp01(xh^2 / 16);  Phi(x) - 1/2 = x . Q(u)  with Q of degree 6 in u = x^2 / 16 on |x| <= 4 (minimax in the GELU error under the
//   constraint 4 . Q(1) = 1/2);  Phi = clamp01(x . Q + 1/2)  continues it exactly beyond |x| = 4;  GELU = xh . Phi.
// 11 instructions per pair, all packed (the fp32 form above: 14 + one v_cvt_pk_bf16_f32).  Error against the exact erf GELU over every fp16 input in
// [-8, 8] and 4e5 normal deviates (tests/studies/f16_gelu_study.py): rms 4.7e-4 (a correctly rounded bf16 result: 1.3e-3), max 6e-3 (the fp16 spacing
// of results in [4, 8); bf16: 1.6e-2); relative error on the negative side, where GELU is a difference of O(1) terms, up to 1.2e-2 of values below 0.17.
// Coefficients are O(1) in u (no cancellation beyond one binade), which is what makes fp16 Horner usable: in t = x^2 they span nine decades.
#define KASF_GH0 3.9787025043e-01f
#define KASF_GH1 -1.0328702880e+00f
#define KASF_GH2 2.2435028997e+00f
#define KASF_GH3 -3.3267139471e+00f
#define KASF_GH4 3.1302451291e+00f
#define KASF_GH5 -1.6659993847e+00f
#define KASF_GH6 3.7896534064e-01f
template <int N> __device__ __forceinline__ void gelu_pairs_h(const f32x2 (&x)[N], f16x2 (&y)[N]) {
    auto C = [](float v) { return f16x2{(f16)v, (f16)v}; };
    auto fma2 = [](f16x2 a, f16x2 b, f16x2 c) { return __builtin_elementwise_fma(a, b, c); };
    auto clamp01 = [&](f16x2 a) { return __builtin_elementwise_min(__builtin_elementwise_max(a, C(0.0f)), C(1.0f)); };   // folds into the producing instruction's clamp bit
    f16x2 xh[N], u[N], q[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        xh[k] = __builtin_convertvector(x[k], f16x2);              // v_cvt_pk_f16_f32 (round to nearest even)
        u[k] = xh[k] * xh[k];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) u[k] = clamp01(u[k] * C(0.0625f));
#pragma unroll
    for (int k = 0; k < N; ++k) q[k] = fma2(u[k], C(KASF_GH6), C(KASF_GH5));
#pragma unroll
    for (int k = 0; k < N; ++k) q[k] = fma2(q[k], u[k], C(KASF_GH4));
#pragma unroll
    for (int k = 0; k < N; ++k) q[k] = fma2(q[k], u[k], C(KASF_GH3));
#pragma unroll
    for (int k = 0; k < N; ++k) q[k] = fma2(q[k], u[k], C(KASF_GH2));
#pragma unroll
    for (int k = 0; k < N; ++k) q[k] = fma2(q[k], u[k], C(KASF_GH1));
#pragma unroll
    for (int k = 0; k < N; ++k) q[k] = fma2(q[k], u[k], C(KASF_GH0));
#pragma unroll
    for (int k = 0; k < N; ++k) y[k] = xh[k] * clamp01(fma2(xh[k], q[k], C(0.5f)));
}
// The backward pass's form (round 5): Phi(z) AND GELU'(z) = Phi + z pdf(z) of N pairs in packed fp16, both as 1/2 + zc . poly(v) with zc = clamp(z, -4, 4) and
// v = zc^2 / 8 - 1 in [-1, 1]:  Phi - 1/2 = zc Q(v) (degree 6),  GELU' - 1/2 = zc R(v), R = Q + pdf (degree 7; its u = x^2/16 form has coefficients up to 45 and
// loses three digits in fp16 Horner, so both polynomials live in the shifted variable).  No exponential (the fp32 form: two quarter-rate v_exp_f32 per pair).
// The caller multiplies in fp32 (v_fma_mix_f32: fp32 z / dH times the fp16 factor), so H and dZ carry no fp16 range limit: gradients do underflow fp16.
// tests/studies/f16_gelu_study.py: GELU' rms error 3.3e-4 (max 1.4e-3; a correctly rounded bf16 GELU': 1.35e-3), H after its bf16 rounding 1.36e-3 (1.33e-3).
#define KASF_GQ0 1.7597076120e-01f
#define KASF_GQ1 -8.4424545240e-02f
#define KASF_GQ2 5.5395112154e-02f
#define KASF_GQ3 -3.5476099890e-02f
#define KASF_GQ4 2.4147918417e-02f
#define KASF_GQ5 -1.6534480087e-02f
#define KASF_GQ6 5.9213334475e-03f
#define KASF_GR0 1.8316959647e-01f
#define KASF_GR1 -1.1385186317e-01f
#define KASF_GR2 1.1756731017e-01f
#define KASF_GR3 -1.1265037252e-01f
#define KASF_GR4 8.2926297726e-02f
#define KASF_GR5 -7.5993324699e-02f
#define KASF_GR6 7.7233806846e-02f
#define KASF_GR7 -3.3267620593e-02f
template <int N> __device__ __forceinline__ void gelu_grad_pairs_h(const f32x2 (&z)[N], f16x2 (&phi)[N], f16x2 (&dg)[N]) {
    auto C = [](float v) { return f16x2{(f16)v, (f16)v}; };
    auto fma2 = [](f16x2 a, f16x2 b, f16x2 c) { return __builtin_elementwise_fma(a, b, c); };
    f16x2 zc[N], v[N], q[N], r[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        zc[k] = __builtin_convertvector(z[k], f16x2);
        zc[k] = __builtin_elementwise_min(__builtin_elementwise_max(zc[k], C(-4.0f)), C(4.0f));
    }
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = fma2(zc[k] * zc[k], C(0.125f), C(-1.0f));
#pragma unroll
    for (int k = 0; k < N; ++k) { q[k] = fma2(v[k], C(KASF_GQ6), C(KASF_GQ5)); r[k] = fma2(v[k], C(KASF_GR7), C(KASF_GR6)); }
#pragma unroll
    for (int k = 0; k < N; ++k) { q[k] = fma2(q[k], v[k], C(KASF_GQ4)); r[k] = fma2(r[k], v[k], C(KASF_GR5)); }
#pragma unroll
    for (int k = 0; k < N; ++k) { q[k] = fma2(q[k], v[k], C(KASF_GQ3)); r[k] = fma2(r[k], v[k], C(KASF_GR4)); }
#pragma unroll
    for (int k = 0; k < N; ++k) { q[k] = fma2(q[k], v[k], C(KASF_GQ2)); r[k] = fma2(r[k], v[k], C(KASF_GR3)); }
#pragma unroll
    for (int k = 0; k < N; ++k) { q[k] = fma2(q[k], v[k], C(KASF_GQ1)); r[k] = fma2(r[k], v[k], C(KASF_GR2)); }
#pragma unroll
    for (int k = 0; k < N; ++k) { q[k] = fma2(q[k], v[k], C(KASF_GQ0)); r[k] = fma2(r[k], v[k], C(KASF_GR1)); }
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = fma2(r[k], v[k], C(KASF_GR0));
#pragma unroll
    for (int k = 0; k < N; ++k) { phi[k] = fma2(zc[k], q[k], C(0.5f)); dg[k] = fma2(zc[k], r[k], C(0.5f)); }
}
template <typename T = float> __device__ __forceinline__ void gelu_and_grad(float x, float& y, float& dy) {
    float s, e;
    gelu_tail<GeluMode<T>::FAST>(x, s, e);
    const float Phi = x >= 0.0f ? 1.0f - s : s;
    y = x * Phi;
    dy = fmaf(x * 0.3989422804014327f, e, Phi);
}

// ------------------------------------------------------------------------------------------
// LDS tile of [rows][128] elements, 16-byte chunks XOR-swizzled by (row & 15): a ds_read_b128
// by 16 lanes reading the same chunk of 16 different rows is then bank-conflict free.
// ------------------------------------------------------------------------------------------
template <typename T> struct Tile {
    static constexpr int EPC = 16 / sizeof(T);      // elements per 16-B chunk (8 / 4)
    static constexpr int CPR = 128 / EPC;           // chunks per row (16 / 32)
    __device__ static __forceinline__ int chunk_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 15)) * EPC); }
    // offset of element (row, col), col a multiple of 4 (never crosses a chunk)
    __device__ static __forceinline__ int off4(int row, int col) { return chunk_off(row, col / EPC) + (col % EPC); }
};

// Write 8 consecutive elements (col0 multiple of 8) of one tile row.
__device__ __forceinline__ void tile_store8(bf16* s, int row, int col0, const float (&v)[8]) {
    store8(s + Tile<bf16>::chunk_off(row, col0 / 8), v);
}
__device__ __forceinline__ void tile_store8(float* s, int row, int col0, const float (&v)[8]) {
    float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
    store4(s + Tile<float>::chunk_off(row, col0 / 4), a);
    store4(s + Tile<float>::chunk_off(row, col0 / 4 + 1), b);
}
__device__ __forceinline__ void tile_load8(const bf16* s, int row, int col0, float (&v)[8]) {
    load8(s + Tile<bf16>::chunk_off(row, col0 / 8), v);
}
__device__ __forceinline__ void tile_load8(const float* s, int row, int col0, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(s + Tile<float>::chunk_off(row, col0 / 4));
    f32x4 b = *reinterpret_cast<const f32x4*>(s + Tile<float>::chunk_off(row, col0 / 4 + 1));
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}

// ------------------------------------------------------------------------------------------
// One 128-deep MFMA sweep.  sW rows are output features n (the MFMA "A" operand), sX rows are
// tokens m (the "B" operand); acc[nt][mt][r] = C[m = x_row0+16mt+(lane&15)][n = w_row0+16nt+4(lane>>4)+r].
// Each lane thus owns 4 consecutive output features of one token (vector stores in the epilogue).
// ------------------------------------------------------------------------------------------
template <int NT, int MT>
__device__ __forceinline__ void mma_k128(const bf16* sW, int w_row0, const bf16* sX, int x_row0, f32x4 (&acc)[NT][MT]) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        bf16x8 a[NT], b[MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[nt] = *reinterpret_cast<const bf16x8*>(sW + Tile<bf16>::chunk_off(w_row0 + nt * 16 + i, 4 * s + g));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[mt] = *reinterpret_cast<const bf16x8*>(sX + Tile<bf16>::chunk_off(x_row0 + mt * 16 + i, 4 * s + g));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], b[mt], acc[nt][mt], 0, 0, 0);
    }
}
template <int NT, int MT>
__device__ __forceinline__ void mma_k128(const float* sW, int w_row0, const float* sX, int x_row0, f32x4 (&acc)[NT][MT]) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < 8; ++s) {   // lane group g supplies k = 16s + 4g + r in MFMA r (same bijection on both operands)
        f32x4 a[NT], b[MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[nt] = *reinterpret_cast<const f32x4*>(sW + Tile<float>::chunk_off(w_row0 + nt * 16 + i, 4 * s + g));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[mt] = *reinterpret_cast<const f32x4*>(sX + Tile<float>::chunk_off(x_row0 + mt * 16 + i, 4 * s + g));
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt][r], b[mt][r], acc[nt][mt], 0, 0, 0);
    }
}

template <int NT, int MT> __device__ __forceinline__ void zero_acc(f32x4 (&acc)[NT][MT]) {
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------------------------------
// Stage BM token rows x 128 channels (global row stride ld) into a swizzled tile with an
// optional fused LayerNorm (two-pass variance, eps inside the sqrt).  256 threads: 16 lanes own
// one row (8 channels each); the global loads of up to 4 rows per thread are issued back to back
// before any of them is consumed.  Rows >= M are zero filled.
// ------------------------------------------------------------------------------------------
template <typename T, int BM, bool LN, int NTHR = 256>
__device__ __forceinline__ void stage_rows(T* sA, const T* X, int64_t ld, int64_t row0, int64_t M, const float* gamma, const float* beta,
                                           T* xn_out) {
    const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    constexpr int RS = NTHR / 16;                        // rows per sweep
    constexpr int NB = BM / RS >= 4 ? 4 : BM / RS;       // rows per thread per batch
    float gm[8], bt[8];
    if (LN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { gm[i] = gamma[sub * 8 + i]; bt[i] = beta[sub * 8 + i]; }
    }
#pragma unroll
    for (int r0 = 0; r0 < BM; r0 += RS * NB) {
        float v[NB][8];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int64_t row = row0 + r0 + rl + RS * b;
            if (row < M) load8(X + row * ld + sub * 8, v[b]);
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[b][i] = 0.f;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int r = r0 + rl + RS * b;
            const int64_t row = row0 + r;
            if (LN) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[b][i];
                const float mean = reduce16(s) * (1.0f / 128.0f);
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[b][i] -= mean; q += v[b][i] * v[b][i]; }
                const float rstd = rsqrtf(reduce16(q) * (1.0f / 128.0f) + KASF_LN_EPS);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[b][i] = (row < M) ? v[b][i] * rstd * gm[i] + bt[i] : 0.f;
                if (xn_out != nullptr && row < M) store8(xn_out + row * 128 + sub * 8, v[b]);
            }
            tile_store8(sA, r, sub * 8, v[b]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Asynchronous LDS-direct copy (global_load_lds_dwordx4) of ROWS x 128 elements, global row stride
// ld, into a swizzled tile.  The LDS destination of one wave-instruction is linear (base + 16 B x
// lane), so the XOR swizzle is applied to each lane's SOURCE chunk.  Rows >= nvalid re-read row
// nvalid-1 (never out of bounds; their results are discarded by the caller).  256 threads.
// Consumers must execute  wait_async(); __syncthreads();  before reading the tile.
// ------------------------------------------------------------------------------------------
// One LDS-direct 16-byte-per-lane load: LDS[lds_base + 16*lane] <- *gsrc.  Issued from inline asm so that hipcc's
// waitcnt pass does not see a pending LDS write (it would otherwise drain vmcnt(0) before the next ds_read of the
// staging array and collapse any multi-tile ring to depth one).  M0 carries the wave-uniform LDS byte address and
// is saved / restored inside the statement (cdna_hip_programming.md 5.7).  Completion: wait_async*() + a barrier.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}
// Same, with the address split into a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: per tile only the base moves
// (two scalar adds) instead of a 64-bit vector address per lane.
__device__ __forceinline__ void glds16_s(const void* ubase, unsigned lane_off, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(ubase), "s"(lds_base)
                 : "memory");
}
__device__ __forceinline__ const void* uniform_ptr(const void* p) {      // tells the compiler the pointer is wave-uniform (lives in SGPRs)
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

template <typename T, int ROWS, int NTHR = 256>
__device__ __forceinline__ void stage_tile_async(T* sT, const T* src, int64_t ld, int nvalid) {
    constexpr int EPC = Tile<T>::EPC, CPR = Tile<T>::CPR;
    constexpr int RPI = 64 / CPR;                        // rows per wave-instruction (4 bf16, 2 f32)
    constexpr int IPW = ROWS / RPI / (NTHR / 64);        // instructions per wave
    static_assert(IPW >= 1, "tile too small for this workgroup size");
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(sT));
    const void* ub = uniform_ptr(src);                   // every caller passes a per-tile (wave-uniform) base
    const unsigned ldb = (unsigned)ld * (unsigned)sizeof(T);
#pragma unroll
    for (int t = 0; t < IPW; ++t) {
        const int inst = w * IPW + t;
        const int row = inst * RPI + lane / CPR, pc = lane % CPR, c = pc ^ (row & 15);
        const int srow = row < nvalid ? row : nvalid - 1;
        glds16_s(ub, (unsigned)srow * ldb + (unsigned)(c * EPC) * (unsigned)sizeof(T), base + (unsigned)inst * 1024u);
    }
}
__device__ __forceinline__ void wait_async() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's vector-memory operations (LDS-direct loads included, in issue order) are outstanding
// A register USE of a value loaded by an ordinary (compiler-tracked) load.  Placed after a prologue's wait_async(), in front of a loop that keeps LDS-direct
// requests in flight: hipcc's own s_waitcnt for the first use of a pre-loop load otherwise lands INSIDE the loop -- as vmcnt(1) / vmcnt(0) per iteration, because
// it cannot see the inline-asm requests issued since -- and drains the look-ahead every tile (round 5: 155 -> 188 us on k_mlp_bwd_s; DESIGN section 4, compiler traps).
#ifdef KASF_NO_TOUCH_LOADED          // negative control of tests/test_vmcnt_guard_cpu.py only
template <typename V> __device__ __forceinline__ void touch_loaded(const V&) {}
#else
template <typename V> __device__ __forceinline__ void touch_loaded(const V& v) { asm volatile("" ::"v"(v)); }
#endif
template <int N> __device__ __forceinline__ void wait_async_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier WITHOUT the vmcnt(0)/lgkmcnt(0) drain __syncthreads() adds while LDS-direct loads are in flight
__device__ __forceinline__ void barrier_keep_async() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void load4(const bf16* p, float (&v)[4]) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = t[i];
}

#define HIP_OK(x)                                                                 \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) return kasf_set_error(1000 + (int)e_, hipGetErrorString(e_)); \
    } while (0)

// raw 8-element register images of a thread's 16-byte (bf16) / 32-byte (f32) chunk: kept as loaded, widened at the point of use (as fp32 a bf16 chunk
// would hold 8 VGPRs instead of 4: k_gcn_agg_temporal loads a whole track ahead, k_gate_bwd keeps three branch rows across two passes)
template <typename T> struct Raw8;
template <> struct Raw8<bf16> {
    bf16x8 v;          // (typed, widened element by element like load8: a __builtin_bit_cast of the words of an f32x4 image was compiled into eight copies of word 0)
    __device__ __forceinline__ void load(const bf16* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ void get(float (&o)[8]) const {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (float)v[k];
    }
};
template <> struct Raw8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
    __device__ __forceinline__ void get(float (&o)[8]) const {
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = a[k]; o[4 + k] = b[k]; }
    }
};
