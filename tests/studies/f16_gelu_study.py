"""Packed-fp16 GELU of the MLP forward (csrc/common.h: gelu_pairs_h): coefficient fit and error study, numpy emulation of the fp16 instruction sequence.

    python tests/studies/f16_gelu_study.py            # prints the coefficients (KASF_GH0..6) and the error table quoted in common.h / DESIGN.md

Phi(x) - 1/2 = x Q(u), u = min(x^2 / 16, 1).  Q is fitted in the shifted variable v = 2u - 1 (well conditioned), minimax in the GELU error x^2 (Q - target)
under the constraint 4 Q(1) = 1/2, then expanded in u: its coefficients there are O(1), so fp16 Horner loses no more than a binade.
Every fp16 operation is emulated as exact arithmetic (float64 holds fp16 products and sums exactly) followed by ONE rounding to fp16, i.e. v_pk_fma_f16.
"""
import numpy as np
from numpy.polynomial import polynomial as Pl
from scipy.special import erf

f16 = np.float16


def Phi(x):
    return 0.5 * (1 + erf(x / np.sqrt(2)))


def r16(a):
    return np.asarray(a, dtype=np.float64).astype(f16).astype(np.float64)


def fma16(a, b, c):
    return r16(a * b + c)


def fit_u(deg, iters=80):
    v = np.cos(np.linspace(0, np.pi, 6001))
    t = 8 * (v + 1)
    x = np.sqrt(t)
    m = x > 1e-4
    v, t, x = v[m], t[m], x[m]
    target = (Phi(x) - 0.5) / x
    w = np.ones_like(x)
    best = None
    for _ in range(iters):                       # Lawson iteration towards the minimax fit of the GELU error
        A = np.stack([(v - 1) * v ** k for k in range(deg)], 1)
        b = target - 0.125
        W = w * t
        q, *_ = np.linalg.lstsq(A * W[:, None], b * W, rcond=None)
        err = t * (A @ q - b)
        mm = np.abs(err).max()
        if best is None or mm < best[0]:
            best = (mm, q.copy())
        w = w * (1 + 4 * np.abs(err) / mm)
        w /= w.max()
    mm, q = best
    Pv = np.zeros(deg + 1)
    Pv[0] += 0.125
    for k in range(deg):
        Pv[k + 1] += q[k]
        Pv[k] -= q[k]
    return mm, Pl.Polynomial(Pv)(Pl.Polynomial([-1, 2])).coef     # monomials in u


def gelu_h(x32, coef):
    """The instruction sequence of gelu_pairs_h."""
    xh = r16(x32)
    u = np.clip(r16(r16(xh * xh) * 0.0625), 0, 1)
    c = r16(coef)
    q = fma16(u, c[-1], c[-2])
    for ck in c[-3::-1]:
        q = fma16(q, u, ck)
    phi = np.clip(fma16(xh, q, 0.5), 0, 1)
    return r16(xh * phi)


def bf16_round(a):
    b = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
    b = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return b.astype(np.uint32).view(np.float32).astype(np.float64)


def main():
    allh = np.arange(65536, dtype=np.uint16).view(f16).astype(np.float64)
    allh = allh[np.isfinite(allh) & (np.abs(allh) <= 8)]
    rng = np.random.default_rng(0)
    z = rng.normal(0, 1.2, 400000)
    x = np.concatenate([allh, z])
    ref = x * Phi(x)
    n0 = len(allh)
    rms = lambda e: float(np.sqrt(np.mean(e[n0:] ** 2)))
    eb = np.abs(bf16_round(ref) - ref)
    print(f"exact GELU rounded to bf16 (rounds 1-4): max {eb.max():.2e}  rms over N(0, 1.2) {rms(eb):.2e}")
    for deg in (5, 6, 7):
        fit_err, coef = fit_u(deg)
        e = np.abs(gelu_h(x, coef) - ref)
        neg = (ref < -0.02)
        print(f"degree {deg}: fit error {fit_err:.1e} | fp16 sequence: max {e.max():.2e}  rms {rms(e):.2e}  worst relative error where GELU < -0.02: {np.max(e[neg] / -ref[neg]):.2e}")
        print("   " + "  ".join(f"#define KASF_GH{k} {c:.10e}f" for k, c in enumerate(coef)))


if __name__ == "__main__":
    main()
