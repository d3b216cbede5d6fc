// Second-generation row-GEMM kernels (bf16): persistent workgroups, weights resident in registers for the whole launch,
// token tiles streamed through an LDS ring by LDS-direct loads issued two tiles ahead (counted vmcnt, raw barriers).
// Per tile there is no L2 traffic besides the tokens themselves, so these kernels are bound by the HBM stream.
//
//   k_dgrad_r<KC,...>   out = [resid] + [out] + LNbwd( dY[M x 128KC] . Wt[128 x 128KC]^T [+ add] ; x, gamma ),
//                       dgamma/dbeta reductions, optional LN(x) output for the matching weight-gradient GEMM
//                       (the data gradient of every LN-fused linear of the mixers: qkv, q, kv, U|V)
//   k_linear_r<NC,LN,RES>   C[M x 128NC] = LN?(A) . W^T + bias           (qkv / q / kv / U|V / d_o), optional LN(A) output
//                           RES: C = resid + ls * (A . W^T + bias)        (attention proj + layer-scale + residual)
#include "common.h"
#include "kernels.h"
#include "tile_ops.h"
#include "mlp_fin.h"
#ifndef KASF_LINEAR_ISSUE_AT        // k_linear_r: the look-ahead loads of tile t+2 behind B1 (1, shipped since round 5) or behind B2, back to back with the LayerNorm / copy-out stores (0): 29.3 -> 28.6 us in step
#define KASF_LINEAR_ISSUE_AT 1
#endif

namespace {

constexpr int R_BM = 32, R_NW = 8, R_THR = R_NW * 64;
constexpr int R_TILE = R_BM * 128;                      // elements of one [32][128] tile

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ bf16x8 tok_frag(const bf16* s, int row, int ks) {
    const int g = (threadIdx.x & 63) >> 4;
    return *reinterpret_cast<const bf16x8*>(s + Tile<bf16>::chunk_off(row, 4 * ks + g));
}

// Ring slot layout (tiles of [32][128] bf16, 8 KB each): dY chunk 0..KC-1 | x | resid? | add? | out(accumulate)?
// WG (round 4): the weight gradient dW[128 KC][128] = dY^T . LN(x) of the SAME linear is accumulated here as well -- the dY tile is in the ring and LN(x) is
// formed by the LayerNorm-backward phase anyway, so the separate streaming launch (k_wgrad_ring_jobs: dY and LN(x) read once more, LN(x) written for it) goes
// away.  Tile t - 1 is multiplied while tile t's data gradient runs (transposed LDS fragments, tokens = the MFMA's reduction axis); wave w keeps rows
// [16 KC w, +16 KC) of dW in 8 KC accumulator registers for the whole token range and leaves ONE bf16 partial tile per workgroup (fixed-order finish).
// BIAS (with WG; the GCN's U | V linear has one): the bias gradient = column sums of dY over the tokens, accumulated from the ring tile by the LayerNorm-backward
// phase's (row, 8-column) threads and left as 128 KC more columns of the workgroup's row in `part`.
// PROJ (with WG and RESID; the bone block's q linear): the block's OUTPUT-PROJECTION weight gradient G = g_mid^T . o rides here as well -- g_mid is this
// kernel's residual operand, the attention output o arrives as one more ring stream -- with the column sums of g_mid (the proj bias / layer-scale terms);
// the block then has no streaming weight-gradient launch at all, and o and g_mid are not read a second time.
// MLPFIN (round 6, KC = 4 with RESID): the second launch of the MLP backward.  dY = dZ [M][512] as k_mlp_bwd_s<DZOUT> leaves it, Wt = fc1.weight^T: out = g + LNbwd(dZ W1; x_mid,
// norm2) -- the dA product that used to be four bf16 partials out of the issue-bound kernel, formed here on an idle matrix pipe with ONE fp32 accumulation over the 512 hidden
// units; the workgroup's row of `part` is dgamma | dbeta | colsum(g) (the fc2 bias / layer-scale terms), and workgroups past `ngrid` (128 of them, two 256-thread fin blocks each)
// run the fixed-order sum of the weight-gradient partial tiles (mlp_fin.h) -- the roles k_lnbwd_sum4_fin had.
template <int KC, bool RESID, bool ADD, bool ACC, bool XN, int RING, bool WG = false, bool BIAS = false, bool PROJ = false, bool MLPFIN = false>
__global__ __launch_bounds__(R_THR) void k_dgrad_r(const bf16* __restrict__ dY, const bf16* __restrict__ Wt, const bf16* __restrict__ dxn_add,
                                                   const bf16* __restrict__ X, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const bf16* __restrict__ resid, bf16* __restrict__ out, float* __restrict__ dgamma,
                                                   float* __restrict__ dbeta, bf16* __restrict__ xn_out, int64_t M, float* __restrict__ part,
                                                   bf16* __restrict__ wpart, float* __restrict__ dbias, const bf16* __restrict__ Oin,
                                                   bf16* __restrict__ ppart, float* __restrict__ pbrow, const MlpFinArgs fa, int ngrid) {
    static_assert(!MLPFIN || (RESID && !WG && !BIAS && !PROJ), "the MLP form: residual operand, no fused weight gradient of its own");
    if (MLPFIN && (int)blockIdx.x >= ngrid) {            // weight-gradient finish role: fin blocks 2 b, 2 b + 1 (both on the same side of 128: same barrier count)
        __shared__ f32x4 sHalf[2][128];
        __shared__ float sDot[2][2];
        const int sb = threadIdx.x >> 8;
        mlp_wfinish_body(fa, 2 * ((int)blockIdx.x - ngrid) + sb, (int)(threadIdx.x & 255), sHalf[sb], sDot[sb]);
        return;
    }
    const int nwg = MLPFIN ? ngrid : (int)gridDim.x;     // workgroups that walk token tiles
    static_assert(!WG || XN, "the fused weight gradient multiplies by LN(x)");
    static_assert(!PROJ || (WG && RESID), "the proj gradient multiplies the residual operand g_mid by o");
    static_assert(!BIAS || WG, "the bias gradient rides with the fused weight gradient");
    constexpr int PLD = 256 + (BIAS ? 128 * KC : 0) + (MLPFIN ? 128 : 0);    // floats per workgroup row of `part`: dgamma | dbeta [| dbias] [| colsum(resid)]
    constexpr int Kd = 128 * KC;
    constexpr int NSTREAM = KC + 1 + (RESID ? 1 : 0) + (ADD ? 1 : 0) + (ACC ? 1 : 0) + (PROJ ? 1 : 0);     // LDS-direct loads per wave per tile
    constexpr int SLOT = NSTREAM * R_TILE;
    constexpr int O_X = KC * R_TILE, O_RES = O_X + R_TILE, O_ADD = O_RES + (RESID ? R_TILE : 0), O_ACC = O_ADD + (ADD ? R_TILE : 0);
    constexpr int O_O = O_ACC + (ACC ? R_TILE : 0);     // PROJ: the attention output rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sRing = reinterpret_cast<bf16*>(smem);        // [RING][SLOT]
    bf16* sD = sRing + RING * SLOT;                     // [32][128] dxn of the current tile
    bf16* sXn = sD + R_TILE;                            // WG: [2][32][128] LN(x) of tiles t - 1 / t (rows past M zero)
    float* sRed = reinterpret_cast<float*>(smem);       // [2][32][128] end-of-kernel dgamma/dbeta partials: reuses the (then dead) ring
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, sub = lane & 15, rl = threadIdx.x >> 4;
    const int64_t ntiles_total = (M + R_BM - 1) / R_BM;
    const int64_t per = (ntiles_total + nwg - 1) / nwg;
    const int64_t tile0 = (int64_t)blockIdx.x * per;
    int64_t ntiles = ntiles_total - tile0;
    if (ntiles > per) ntiles = per;
    if (ntiles <= 0) return;

    bf16x8 wf[4 * KC];                                  // this wave's 16 output features over the whole reduction axis
#pragma unroll
    for (int ks = 0; ks < 4 * KC; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(Wt + (int64_t)(16 * w + i) * Kd + 32 * ks + 8 * g);
    float gm[8], bt[8], dg[8], db[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[e] = gamma[sub * 8 + e]; bt[e] = XN ? beta[sub * 8 + e] : 0.f; dg[e] = 0.f; db[e] = 0.f; }
    float* sGB = reinterpret_cast<float*>(sXn + 2 * R_TILE);     // WG: gamma | beta in LDS, re-read per tile (16 registers the KC = 3 form does not have)
    if (WG) {
        if (threadIdx.x < 128) { sGB[threadIdx.x] = gamma[threadIdx.x]; sGB[128 + threadIdx.x] = beta[threadIdx.x]; }
        __syncthreads();
    }

    f32x4 accW[WG ? KC : 1][WG ? 8 : 1];
    if (WG) zero_acc(accW);
    float dbs[BIAS ? KC : 1][8];
#pragma unroll
    for (int kc = 0; kc < (BIAS ? KC : 1); ++kc)
#pragma unroll
        for (int e = 0; e < 8; ++e) dbs[kc][e] = 0.f;
    f32x4 accP[PROJ ? 8 : 1];                            // PROJ: rows [16 w, +16) of G = g_mid^T . o
    float gcol[8];                                       // PROJ: column sums of g_mid (this thread's 8 columns over its rows)
#pragma unroll
    for (int b = 0; b < (PROJ ? 8 : 1); ++b) accP[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) gcol[e] = 0.f;
    auto wgrad_tile = [&](const bf16* slot_prev, const bf16* xnT) {      // dW rows [16 KC w, +16 KC) += dY(tile)^T . LN(x)(tile): one k-step of 32 tokens
        bf16x8 ra[WG ? KC : 1];
#pragma unroll
        for (int a = 0; a < (WG ? KC : 0); ++a) {
            const int R0 = 16 * KC * w + 16 * a;         // a 16-row group never crosses a 128-column chunk of dY
            ra[a] = frag_tr(slot_prev + (R0 >> 7) * R_TILE, 8 * g, R0 & 127);
        }
#pragma unroll
        for (int b = 0; b < (WG ? 8 : 0); ++b) {
            const bf16x8 cb = frag_tr(xnT, 8 * g, 16 * b);
#pragma unroll
            for (int a = 0; a < (WG ? KC : 0); ++a) accW[a][b] = mfma16(ra[a], cb, accW[a][b]);
        }
        if (PROJ) {
            const bf16x8 rp = frag_tr(slot_prev + O_RES, 8 * g, 16 * w);
#pragma unroll
            for (int b = 0; b < (PROJ ? 8 : 0); ++b) accP[b] = mfma16(rp, frag_tr(slot_prev + O_O, 8 * g, 16 * b), accP[b]);
        }
    };
    auto nxr = [](int sl) { return sl == RING - 1 ? 0 : sl + 1; };   // ring slots roll (no 64-bit modulo in the loop)
    auto issue = [&](int64_t t, int sl) {                // exactly NSTREAM LDS-direct loads per wave per call
        const int64_t tt = t < ntiles ? t : ntiles - 1;
        const int64_t row0 = (tile0 + tt) * R_BM;
        const int nvalid = (int)((M - row0) < R_BM ? (M - row0) : R_BM);
        bf16* slot = sRing + sl * SLOT;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) stage_tile_async<bf16, R_BM, R_THR>(slot + kc * R_TILE, dY + row0 * Kd + kc * 128, Kd, nvalid);
        stage_tile_async<bf16, R_BM, R_THR>(slot + O_X, X + row0 * 128, 128, nvalid);
        if (RESID) stage_tile_async<bf16, R_BM, R_THR>(slot + O_RES, resid + row0 * 128, 128, nvalid);
        if (ADD) stage_tile_async<bf16, R_BM, R_THR>(slot + O_ADD, dxn_add + row0 * 128, 128, nvalid);
        if (ACC) stage_tile_async<bf16, R_BM, R_THR>(slot + O_ACC, out + row0 * 128, 128, nvalid);
        if (PROJ) stage_tile_async<bf16, R_BM, R_THR>(slot + O_O, Oin + row0 * 128, 128, nvalid);
    };
    issue(0, 0);
    if (RING == 3) { issue(1, 1); wait_async_le<NSTREAM>(); }   // tile 0 landed, tile 1 in flight
    int sl = 0, psl = 0;
    for (int64_t t = 0; t < ntiles; ++t, psl = sl, sl = nxr(sl)) {
        const bf16* slot = sRing + sl * SLOT;
        const int64_t row0 = (tile0 + t) * R_BM;
        if (RING == 2) wait_async();
        barrier_keep_async();                            // B1: tile t visible to every wave; everyone is past tile t-1
        if (RING == 2) issue(t + 1, nxr(sl));
        if (WG && t >= 1) wgrad_tile(sRing + psl * SLOT, sXn + (int)((t - 1) & 1) * R_TILE);    // slot (t-1) % RING is refilled only after B2
        {   // ---- GEMM: 16 features x 32 tokens per wave ----
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            bf16x8 fb[2][2];
            fb[0][0] = tok_frag(slot, i, 0);
            fb[0][1] = tok_frag(slot, 16 + i, 0);
#pragma unroll
            for (int ks = 0; ks < 4 * KC; ++ks) {
                if (ks + 1 < 4 * KC) {
                    const bf16* nx = slot + ((ks + 1) >> 2) * R_TILE;
                    fb[(ks + 1) & 1][0] = tok_frag(nx, i, (ks + 1) & 3);
                    fb[(ks + 1) & 1][1] = tok_frag(nx, 16 + i, (ks + 1) & 3);
                }
                acc[0] = mfma16(wf[ks], fb[ks & 1][0], acc[0]);
                acc[1] = mfma16(wf[ks], fb[ks & 1][1], acc[1]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float v[4] = {acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]};
                store4(sD + Tile<bf16>::off4(mt * 16 + i, 16 * w + 4 * g), v);
            }
        }
        barrier_keep_async();                            // B2: the [32][128] dxn tile is complete
        if (RING == 3) issue(t + 2, nxr(nxr(sl)));       // slot (t+2)%3 held tile t-1, which everyone finished before B1
        {   // ---- LayerNorm backward: one 16-lane group per token row ----
            const int64_t row = row0 + rl;
            float d[8], x[8], o[8];
            tile_load8(sD, rl, sub * 8, d);
            tile_load8(slot + O_X, rl, sub * 8, x);
            if (ADD) {
                float a[8];
                tile_load8(slot + O_ADD, rl, sub * 8, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] += a[e];
            }
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[e];
            const float mean = reduce16(s) * (1.0f / 128.0f);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[e] -= mean; q += x[e] * x[e]; }
            const float rstd = rsqrtf(reduce16(q) * (1.0f / 128.0f) + KASF_LN_EPS);
            float xn[8];
            float s1 = 0.f, s2 = 0.f;
            const bool live = row < M;                   // rows past M are clamped copies of the last row: keep them out of the sums
            if (WG) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(sGB + sub * 8), g1 = *reinterpret_cast<const f32x4*>(sGB + sub * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(sGB + 128 + sub * 8), b1 = *reinterpret_cast<const f32x4*>(sGB + 128 + sub * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { gm[e] = g0[e]; gm[4 + e] = g1[e]; bt[e] = b0[e]; bt[4 + e] = b1[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (XN) xn[e] = x[e] * rstd * gm[e] + bt[e];
                x[e] *= rstd;
                if (live) { dg[e] += d[e] * x[e]; db[e] += d[e]; }
                d[e] *= gm[e];
                s1 += d[e];
                s2 += d[e] * x[e];
            }
            s1 = reduce16(s1) * (1.0f / 128.0f);
            s2 = reduce16(s2) * (1.0f / 128.0f);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd * (d[e] - s1 - x[e] * s2);
            if (RESID) {
                float a[8];
                tile_load8(slot + O_RES, rl, sub * 8, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += a[e];
                if (MLPFIN) {
                    if (live) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) gcol[e] += a[e];
                    }
                }
                if (PROJ) {
                    if (live) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) gcol[e] += a[e];
                    } else {                             // a clamped copy of the last row: its o row must not reach G (this tile's slot is multiplied next iteration)
                        const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        tile_store8(const_cast<bf16*>(slot) + O_O, rl, sub * 8, z8);
                    }
                }
            }
            if (ACC) {
                float a[8];
                tile_load8(slot + O_ACC, rl, sub * 8, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += a[e];
            }
            // tile t+1 has landed when only this wave's youngest NSTREAM requests (tile t+2) are outstanding.  Waiting BEFORE this
            // tile's stores keeps them out of that count: they get the whole next tile to drain.
            if (RING == 3) wait_async_le<NSTREAM>();
            if (WG) {
                if (!live) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xn[e] = 0.f;             // clamped copies of the last row must not reach the weight gradient
                }
                tile_store8(sXn + (int)(t & 1) * R_TILE, rl, sub * 8, xn);
            }
            if (BIAS && live) {
#pragma unroll
                for (int kc = 0; kc < (BIAS ? KC : 0); ++kc) {
                    float v[8];
                    tile_load8(slot + kc * R_TILE, rl, sub * 8, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dbs[kc][e] += v[e];
                }
            }
            if (live) {
                store8(out + row * 128 + sub * 8, o);
                if (XN && !WG) store8(xn_out + row * 128 + sub * 8, xn);
            }
        }
    }
    wait_async();
    __syncthreads();
    if (WG) {
        wgrad_tile(sRing + psl * SLOT, sXn + (int)((ntiles - 1) & 1) * R_TILE);         // the last tile (psl: its slot after the loop's final step)
        __syncthreads();                                 // the ring is read for the last time: it becomes sRed / the partial's staging tile below
    }
    // ---- dgamma / dbeta: 32 row groups -> one value per channel per workgroup: row blockIdx.x of `part` (fixed-order finish, k_reduce.hip) ----
#pragma unroll
    for (int e = 0; e < 8; ++e) { sRed[rl * 128 + sub * 8 + e] = dg[e]; sRed[R_TILE + rl * 128 + sub * 8 + e] = db[e]; }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int c = threadIdx.x & 127, which = threadIdx.x >> 7;
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) s += sRed[which * R_TILE + k * 128 + c];
        if (part != nullptr) part[(int64_t)blockIdx.x * PLD + threadIdx.x] = s;
        else atomicAdd((which == 0 ? dgamma : dbeta) + c, s);
    }
    if (BIAS) {
#pragma unroll
        for (int kc = 0; kc < (BIAS ? KC : 0); ++kc) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) sRed[rl * 128 + sub * 8 + e] = dbs[kc][e];
            __syncthreads();
            if (threadIdx.x < 128) {
                float sb = 0.f;
#pragma unroll 8
                for (int k = 0; k < 32; ++k) sb += sRed[k * 128 + threadIdx.x];
                if (part != nullptr) part[(int64_t)blockIdx.x * PLD + 256 + kc * 128 + threadIdx.x] = sb;
                else atomicAdd(dbias + kc * 128 + threadIdx.x, sb);
            }
        }
    }
    if (PROJ || MLPFIN) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) sRed[rl * 128 + sub * 8 + e] = gcol[e];
        __syncthreads();
        if (threadIdx.x < 128) {
            float sb = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) sb += sRed[k * 128 + threadIdx.x];
            if (MLPFIN) {                                // colsum(g): the third block of the workgroup's row (without scratch: an atomic on the gsum slot, passed as `dbias`)
                if (part != nullptr) part[(int64_t)blockIdx.x * PLD + 256 + threadIdx.x] = sb;
                else atomicAdd(dbias + threadIdx.x, sb);
            }
            else pbrow[(int64_t)blockIdx.x * 128 + threadIdx.x] = sb;          // one row of colsum(g_mid) per workgroup: the finish launch's `brow`
        }
    }
    if (WG) {
        // this workgroup's partial dW tile: accumulators -> bf16 image [128 KC][128] in the dead ring -> whole 256-byte rows out (16 bytes per lane)
        __syncthreads();
        bf16* sW = sRing;
#pragma unroll
        for (int a = 0; a < (WG ? KC : 0); ++a)
#pragma unroll
            for (int b = 0; b < (WG ? 8 : 0); ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) sW[(16 * KC * w + 16 * a + 4 * g + r) * 128 + 16 * b + i] = (bf16)accW[a][b][r];
        __syncthreads();
        bf16* sWp = sW + 128 * KC * 128;                 // PROJ: the [128][128] tile of G behind it
        if (PROJ) {
#pragma unroll
            for (int b = 0; b < (PROJ ? 8 : 0); ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) sWp[(16 * w + 4 * g + r) * 128 + 16 * b + i] = (bf16)accP[b][r];
        }
        __syncthreads();
        bf16* dst = wpart + (int64_t)blockIdx.x * (128 * KC * 128);
        for (int c = threadIdx.x; c < 128 * KC * 16; c += R_THR) *reinterpret_cast<f32x4*>(dst + c * 8) = *reinterpret_cast<const f32x4*>(sW + c * 8);
        if (PROJ) {
            bf16* dstp = ppart + (int64_t)blockIdx.x * (128 * 128);
            for (int c = threadIdx.x; c < 128 * 16; c += R_THR) *reinterpret_cast<f32x4*>(dstp + c * 8) = *reinterpret_cast<const f32x4*>(sWp + c * 8);
        }
    }
}

template <int KC, bool RESID, bool ADD, bool ACC, bool XN, bool WG = false, bool BIAS = false, bool PROJ = false>
int launch_dgrad_r(hipStream_t s, const void* dY, const void* Wt, const void* add, const void* X, const float* gamma, const float* beta, const void* resid,
                   void* out, float* dgamma, float* dbeta, void* xn_out, int64_t M, KasfColSink* sink, void* wpart = nullptr, float* dbias = nullptr,
                   const void* o_in = nullptr, void* ppart = nullptr, float* pbrow = nullptr) {
    constexpr int PLD = 256 + (BIAS ? 128 * KC : 0);
    constexpr int NSTREAM = KC + 1 + (RESID ? 1 : 0) + (ADD ? 1 : 0) + (ACC ? 1 : 0) + (PROJ ? 1 : 0);
    constexpr size_t fixed = (size_t)R_TILE * 2 * (WG ? 3 : 1) + (WG ? 1024 : 0);                         // sD (+ the two LN(x) tiles of the fused weight gradient); the end-of-kernel reductions reuse the ring: >= 32 KB
    constexpr bool ring3 = 3 * NSTREAM * R_TILE * 2 + fixed <= 160 * 1024;
    constexpr int RING = ring3 ? 3 : 2;
    static_assert(!WG || RING * NSTREAM >= 4 * (KC + (PROJ ? 1 : 0)), "the partial weight-gradient tiles are staged in the ring");
    const size_t sh = (size_t)RING * NSTREAM * R_TILE * 2 + fixed;
    auto kern = k_dgrad_r<KC, RESID, ADD, ACC, XN, RING, WG, BIAS, PROJ>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    const int64_t tiles = (M + R_BM - 1) / R_BM;
    const int64_t gcap = kasf_narrow_grid(KASF_NG_DGRAD, 256, M);
    const unsigned grid = (unsigned)(tiles < gcap ? tiles : gcap);
    const int64_t per = (tiles + grid - 1) / grid;
    const int active = (int)((tiles + per - 1) / per);                                  // workgroups that own at least one tile (the others return at once)
    float* part = sink != nullptr ? sink->take(active, PLD) : nullptr;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(R_THR), sh, s, (const bf16*)dY, (const bf16*)Wt, (const bf16*)add, (const bf16*)X, gamma, beta,
                       (const bf16*)resid, (bf16*)out, dgamma, dbeta, (bf16*)xn_out, M, part, (bf16*)wpart, dbias, (const bf16*)o_in, (bf16*)ppart, pbrow, MlpFinArgs{}, (int)grid);
    if (part != nullptr) {
        sink->add(part, PLD, active, 128, dgamma);
        sink->add(part + 128, PLD, active, 128, dbeta);
        if (BIAS) sink->add(part + 256, PLD, active, 128 * KC, dbias);
    }
    return active;                                       // WG: this many bf16 partial tiles of [128 KC][128] were written to wpart
}

template <int NC, bool LN, bool RES>
__global__ __launch_bounds__(R_THR) void k_linear_r(const bf16* __restrict__ A, const bf16* __restrict__ W, const float* __restrict__ bias,
                                                    const float* __restrict__ ln_g, const float* __restrict__ ln_b, bf16* __restrict__ xn_out,
                                                    const float* __restrict__ ls, const bf16* __restrict__ resid, bf16* __restrict__ C, int64_t M) {
    static_assert(!RES || NC == 1, "the residual form is 128 -> 128");
    constexpr int N = 128 * NC, NSTREAM = 1 + (RES ? 1 : 0), SLOT = NSTREAM * R_TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sRing = reinterpret_cast<bf16*>(smem);        // [3][SLOT]  raw A tile (| resid tile)
    bf16* sA = sRing + 3 * SLOT;                        // [2][32][128] LN(A) (LN only)
    bf16* sO = sA + (LN ? 2 * R_TILE : 0);              // [NC][32][128] output tile
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, sub = lane & 15, rl = threadIdx.x >> 4;
    const int64_t ntiles_total = (M + R_BM - 1) / R_BM;
    const int64_t per = (ntiles_total + gridDim.x - 1) / gridDim.x;
    const int64_t tile0 = (int64_t)blockIdx.x * per;
    int64_t ntiles = ntiles_total - tile0;
    if (ntiles > per) ntiles = per;
    if (ntiles <= 0) return;

    bf16x8 wf[NC][4];
    f32x4 bv[NC], lsv = f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[nt][ks] = *reinterpret_cast<const bf16x8*>(W + (int64_t)(16 * NC * w + 16 * nt + i) * 128 + 32 * ks + 8 * g);
        bv[nt] = bias != nullptr ? *reinterpret_cast<const f32x4*>(bias + 16 * NC * w + 16 * nt + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (RES) lsv = *reinterpret_cast<const f32x4*>(ls + 16 * w + 4 * g);
    float gm[8], bt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[e] = LN ? ln_g[sub * 8 + e] : 1.f; bt[e] = LN ? ln_b[sub * 8 + e] : 0.f; }

    auto nx3 = [](int sl) { return sl == 2 ? 0 : sl + 1; };
    auto issue = [&](int64_t t, int sl) {
        const int64_t tt = t < ntiles ? t : ntiles - 1;
        const int64_t row0 = (tile0 + tt) * R_BM;
        const int nvalid = (int)((M - row0) < R_BM ? (M - row0) : R_BM);
        bf16* slot = sRing + sl * SLOT;
        stage_tile_async<bf16, R_BM, R_THR>(slot, A + row0 * 128, 128, nvalid);
        if (RES) stage_tile_async<bf16, R_BM, R_THR>(slot + R_TILE, resid + row0 * 128, 128, nvalid);
    };
    auto layernorm = [&](int64_t t, int sl) {            // row rl of tile t: raw slot -> sA[t & 1] (+ xn_out)
        float v[8];
        tile_load8(sRing + sl * SLOT, rl, sub * 8, v);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        const float mean = reduce16(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] -= mean; q += v[e] * v[e]; }
        const float rstd = rsqrtf(reduce16(q) * (1.0f / 128.0f) + KASF_LN_EPS);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * rstd * gm[e] + bt[e];
        tile_store8(sA + (int)(t & 1) * R_TILE, rl, sub * 8, v);
        const int64_t row = (tile0 + t) * R_BM + rl;
        if (xn_out != nullptr && row < M) store8(xn_out + row * 128 + sub * 8, v);
    };
    issue(0, 0);
    issue(1, 1);
    wait_async_le<NSTREAM>();
    if (LN) layernorm(0, 0);
    int sl = 0;
    for (int64_t t = 0; t < ntiles; ++t, sl = nx3(sl)) {
        const bf16* slot = sRing + sl * SLOT;
        const bf16* cA = LN ? sA + (int)(t & 1) * R_TILE : slot;
        const int64_t row0 = (tile0 + t) * R_BM;
        barrier_keep_async();                            // B1: operand tile t complete and visible; everyone is past the copy-out of tile t-1
#if KASF_LINEAR_ISSUE_AT == 1
        issue(t + 2, nx3(nx3(sl)));                      // slot (t+2)%3 held tile t-1: its last reader (the copy-out of t-1) is behind B1
#endif
        {
            f32x4 acc[NC][2];
            zero_acc(acc);
            bf16x8 fb[2][2];
            fb[0][0] = tok_frag(cA, i, 0);
            fb[0][1] = tok_frag(cA, 16 + i, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) { fb[(ks + 1) & 1][0] = tok_frag(cA, i, ks + 1); fb[(ks + 1) & 1][1] = tok_frag(cA, 16 + i, ks + 1); }
#pragma unroll
                for (int nt = 0; nt < NC; ++nt) {
                    acc[nt][0] = mfma16(wf[nt][ks], fb[ks & 1][0], acc[nt][0]);
                    acc[nt][1] = mfma16(wf[nt][ks], fb[ks & 1][1], acc[nt][1]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NC; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int col = 16 * NC * w + 16 * nt + 4 * g;          // feature index inside [0, 128 NC)
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[nt][mt][r] + bv[nt][r];
                    if (RES) {
                        float x[4];
                        load4(slot + R_TILE + Tile<bf16>::off4(mt * 16 + i, col), x);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = x[r] + lsv[r] * v[r];
                    }
                    store4(sO + (col >> 7) * R_TILE + Tile<bf16>::off4(mt * 16 + i, col & 127), v);
                }
        }
        barrier_keep_async();                            // B2: output tile complete
#if KASF_LINEAR_ISSUE_AT == 0
        issue(t + 2, nx3(nx3(sl)));                      // slot (t+2)%3 held tile t-1: its last reader (epilogue of t-1) is behind B1
#endif
        wait_async_le<NSTREAM>();                        // tile t+1 landed (only tile t+2 outstanding); this tile's stores come after
        if (LN && t + 1 < ntiles) layernorm(t + 1, nx3(sl));
#pragma unroll
        for (int c = 0; c < NC; ++c) {                   // 32 rows x 16 NC chunks of 16 bytes over 512 threads
            const int chunk = c * R_THR + threadIdx.x, r = chunk / (16 * NC), cc = chunk % (16 * NC);
            const int64_t row = row0 + r;
            if (row < M)
                *reinterpret_cast<f32x4*>(C + row * N + cc * 8) = *reinterpret_cast<const f32x4*>(sO + (cc >> 4) * R_TILE + Tile<bf16>::chunk_off(r, cc & 15));
        }
    }
    wait_async();
}

template <int NC, bool LN, bool RES>
void launch_linear_r(hipStream_t s, const void* A, const void* W, const float* bias, const float* ln_g, const float* ln_b, void* xn_out, const float* ls,
                     const void* resid, void* C, int64_t M) {
    const size_t sh = (size_t)(3 * (1 + (RES ? 1 : 0)) + (LN ? 2 : 0) + NC) * R_TILE * 2;
    auto kern = k_linear_r<NC, LN, RES>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    const int64_t tiles = (M + R_BM - 1) / R_BM;
    const int64_t gcap = kasf_narrow_grid(KASF_NG_LINEAR, 256, M);
    const unsigned grid = (unsigned)(tiles < gcap ? tiles : gcap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(R_THR), sh, s, (const bf16*)A, (const bf16*)W, bias, ln_g, ln_b, (bf16*)xn_out, ls, (const bf16*)resid,
                       (bf16*)C, M);
}

}  // namespace

// Returns false when the combination is not one of the instantiated ones (the caller then uses k_dgrad_lnbwd).
bool kasf_launch_dgrad_r(hipStream_t s, const void* dY, int Kd, const void* Wt, const void* dxn_add, const void* X, const float* gamma, const void* resid,
                         void* out, int accumulate, float* dgamma, float* dbeta, int64_t M, void* xn_out, const float* beta, KasfColSink* sink) {
    const bool R = resid != nullptr, A = dxn_add != nullptr, C = accumulate != 0, XN = xn_out != nullptr;
    if (M <= 0) return true;
    if (Kd == 384 && R && !A && !C && XN) launch_dgrad_r<3, true, false, false, true>(s, dY, Wt, dxn_add, X, gamma, beta, resid, out, dgamma, dbeta, xn_out, M, sink);
    else if (Kd == 128 && R && !A && !C && XN) launch_dgrad_r<1, true, false, false, true>(s, dY, Wt, dxn_add, X, gamma, beta, resid, out, dgamma, dbeta, xn_out, M, sink);
    else if (Kd == 256 && !R && !A && C && XN) launch_dgrad_r<2, false, false, true, true>(s, dY, Wt, dxn_add, X, gamma, beta, resid, out, dgamma, dbeta, xn_out, M, sink);
    else if (Kd == 256 && R && A && !C && !XN) launch_dgrad_r<2, true, true, false, false>(s, dY, Wt, dxn_add, X, gamma, beta, resid, out, dgamma, dbeta, xn_out, M, sink);
    else return false;
    return true;
}

// The data gradient of an LN-fused linear WITH its weight gradient (dW[Kd][128] = dY^T LN(x)) in one streaming launch.  The weight gradient leaves it as
// `return value` bf16 partial tiles of Kd x 128 in wpart (room for 256 of them), which the block's k_wgrad_finish_jobs launch adds in a fixed order
// (KasfBf16Reduce).  Returns 0 for combinations that are not instantiated (the caller runs the two-kernel sequence).
// Which (shape, operand) combinations kasf_launch_dgrad_wg has an instantiation for.  The engine asks BEFORE it launches the first of a block's two fused
// kernels: a first launch that has already registered its dgamma / dbeta rows in the column sink cannot be taken back if the second one then declines.
bool kasf_dgrad_wg_supported(int Kd, bool resid, bool accumulate, bool dxn_add, bool dbias, bool proj, int64_t M, int64_t wpart_bytes) {
    if (M <= 0 || wpart_bytes < (int64_t)256 * Kd * 128 * 2) return false;
    if (proj) return Kd == 128 && resid && !accumulate && !dxn_add && !dbias;
    if (dxn_add || dbias) return Kd == 256 && resid && !accumulate && dxn_add && dbias;
    return ((Kd == 384 || Kd == 128) && resid && !accumulate) || (Kd == 256 && !resid && accumulate);
}

int kasf_launch_dgrad_wg(hipStream_t s, const void* dY, int Kd, const void* Wt, const void* X, const float* gamma, const float* beta, const void* resid, void* out,
                         int accumulate, float* dgamma, float* dbeta, int64_t M, KasfColSink* sink, void* wpart, int64_t wpart_bytes, const void* dxn_add,
                         float* dbias, const void* proj_o, void* proj_part, float* proj_brow) {
    const bool R = resid != nullptr, C = accumulate != 0;
    if (wpart == nullptr || (proj_o != nullptr && (proj_part == nullptr || proj_brow == nullptr)) ||
        !kasf_dgrad_wg_supported(Kd, R, C, dxn_add != nullptr, dbias != nullptr, proj_o != nullptr, M, wpart_bytes)) return 0;
    if (proj_o != nullptr) {                             // the bone block's q linear carrying the block's proj weight gradient: g_mid (= resid) ^T . o
        if (Kd == 128 && R && !C && dxn_add == nullptr && dbias == nullptr && proj_part != nullptr && proj_brow != nullptr)
            return launch_dgrad_r<1, true, false, false, true, true, false, true>(s, dY, Wt, nullptr, X, gamma, beta, resid, out, dgamma, dbeta, nullptr, M, sink, wpart,
                                                                                  nullptr, proj_o, proj_part, proj_brow);
        return 0;
    }
    if (dxn_add != nullptr || dbias != nullptr) {        // the GCN's U | V linear: direct LN(x) gradient added in, bias gradient = column sums of dY
        if (Kd == 256 && R && !C && dxn_add != nullptr && dbias != nullptr)
            return launch_dgrad_r<2, true, true, false, true, true, true>(s, dY, Wt, dxn_add, X, gamma, beta, resid, out, dgamma, dbeta, nullptr, M, sink, wpart, dbias);
        return 0;
    }
    if (Kd == 384 && R && !C) return launch_dgrad_r<3, true, false, false, true, true>(s, dY, Wt, nullptr, X, gamma, beta, resid, out, dgamma, dbeta, nullptr, M, sink, wpart);
    if (Kd == 128 && R && !C) return launch_dgrad_r<1, true, false, false, true, true>(s, dY, Wt, nullptr, X, gamma, beta, resid, out, dgamma, dbeta, nullptr, M, sink, wpart);
    if (Kd == 256 && !R && C) return launch_dgrad_r<2, false, false, true, true, true>(s, dY, Wt, nullptr, X, gamma, beta, resid, out, dgamma, dbeta, nullptr, M, sink, wpart);
    return 0;
}

// Second launch of the bf16 MLP backward (k_mlp_bwd_s<DZOUT> before it): g_in = g + LNbwd(dZ W1; x, gamma) + per-workgroup rows dgamma | dbeta | colsum(g) in the sink +
// the weight-gradient finish role.  fin: MlpFinArgs of kasf_launch_mlp_bwd_q (as void* to keep mlp_fin.h out of kernels.h).
void kasf_launch_mlp_dgrad_fin(hipStream_t s, const void* dZ, const void* W1t, const void* X, const float* gamma, const void* g, void* g_in, float* dgamma, float* dbeta,
                               float* gsum, int64_t M, KasfColSink* sink, const void* fin_args, const float* b2, const float* ls2, float* dls2, bool have_w2) {
    constexpr int KC = 4, NSTREAM = KC + 2, PLD = 384;
    constexpr size_t fixed = (size_t)R_TILE * 2;
    constexpr int RING = (3 * NSTREAM * R_TILE * 2 + fixed <= 160 * 1024) ? 3 : 2;
    const size_t sh = (size_t)RING * NSTREAM * R_TILE * 2 + fixed;
    auto kern = k_dgrad_r<KC, true, false, false, false, RING, false, false, false, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    const int64_t tiles = (M + R_BM - 1) / R_BM;
    const int64_t gcap = kasf_narrow_grid(KASF_NG_DGRAD, 256, M);          // (full width measured 1 % slower in step: profiles/r6_mlp_bwd_dz_form_ab.md)
    const int grid = (int)(tiles < gcap ? tiles : gcap);
    const int64_t per = (tiles + grid - 1) / grid;
    const int active = (int)((tiles + per - 1) / per);
    float* part = sink != nullptr ? sink->take(active, PLD) : nullptr;
    hipLaunchKernelGGL(kern, dim3(grid + 128), dim3(R_THR), sh, s, (const bf16*)dZ, (const bf16*)W1t, (const bf16*)nullptr, (const bf16*)X, gamma, (const float*)nullptr,
                       (const bf16*)g, (bf16*)g_in, dgamma, dbeta, (bf16*)nullptr, M, part, (bf16*)nullptr, gsum, (const bf16*)nullptr, (bf16*)nullptr,
                       (float*)nullptr, *reinterpret_cast<const MlpFinArgs*>(fin_args), grid);
    if (part != nullptr) {
        sink->add(part, PLD, active, 128, dgamma);
        sink->add(part + 128, PLD, active, 128, dbeta);
        if (have_w2) sink->add(part + 256, PLD, active, 128, gsum, 1, b2, ls2, dls2);      // gsum is the fc2 bias gradient slot: db2 = ls2 . colsum(g)
        else sink->add(part + 256, PLD, active, 128, gsum);
    }
}

// y = LN?(a) W^T + bias, a [M,128] dense, W [N,128] dense, y [M,N] dense, N in {128, 256, 384}
bool kasf_launch_linear_r(hipStream_t s, const void* A, const void* W, const float* bias, void* C, int64_t M, int N, const float* ln_g, const float* ln_b,
                          void* xn_out) {
    if (M <= 0) return true;
    const bool ln = ln_g != nullptr;
    if (N == 384 && ln) launch_linear_r<3, true, false>(s, A, W, bias, ln_g, ln_b, xn_out, nullptr, nullptr, C, M);
    else if (N == 256 && ln) launch_linear_r<2, true, false>(s, A, W, bias, ln_g, ln_b, xn_out, nullptr, nullptr, C, M);
    else if (N == 128 && ln) launch_linear_r<1, true, false>(s, A, W, bias, ln_g, ln_b, xn_out, nullptr, nullptr, C, M);
    else if (N == 128 && !ln && xn_out == nullptr) launch_linear_r<1, false, false>(s, A, W, bias, nullptr, nullptr, nullptr, nullptr, nullptr, C, M);
    else return false;
    return true;
}
// x_mid = resid + ls * (o Wproj^T + b)
void kasf_launch_linear_res_r(hipStream_t s, const void* A, const void* W, const float* bias, const float* ls, const void* resid, void* C, int64_t M) {
    if (M <= 0) return;
    launch_linear_r<1, false, true>(s, A, W, bias, nullptr, nullptr, nullptr, ls, resid, C, M);
}
