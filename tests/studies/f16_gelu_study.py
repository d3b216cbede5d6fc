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


def pdf(x):
    return np.exp(-x * x / 2) / np.sqrt(2 * np.pi)


def fit_v(deg, derivative, iters=80):
    """Coefficients in v = x^2 / 8 - 1 (|v| <= 1) of Q (derivative = False: Phi - 1/2 = x Q, weight x^2: the GELU error) or of R (derivative = True:
    GELU' - 1/2 = x R, R = Q + pdf, weight |x|), with the value at |x| = 4 held exactly (the clamp of x continues both)."""
    v = np.cos(np.linspace(0, np.pi, 8001))
    t = 8 * (v + 1)
    x = np.sqrt(t)
    m = x > 1e-4
    v, t, x = v[m], t[m], x[m]
    target = (Phi(x) - 0.5) / x + (pdf(x) if derivative else 0.0)
    end = 0.125 + (pdf(4.0) if derivative else 0.0)
    w = np.ones_like(x)
    best = None
    for _ in range(iters):
        A = np.stack([(v - 1) * v ** k for k in range(deg)], 1)
        b = target - end
        W = w * (x if derivative else t)
        q, *_ = np.linalg.lstsq(A * W[:, None], b * W, rcond=None)
        err = (x if derivative else t) * (A @ q - b)
        mm = np.abs(err).max()
        if best is None or mm < best[0]:
            best = (mm, q.copy())
        w = w * (1 + 4 * np.abs(err) / mm)
        w /= w.max()
    mm, q = best
    Pv = np.zeros(deg + 1)
    Pv[0] += end
    for k in range(deg):
        Pv[k + 1] += q[k]
        Pv[k] -= q[k]
    return mm, Pv


def gelu_grad_h(x32, cq, cr):
    """The instruction sequence of gelu_grad_pairs_h (backward pass): Phi and GELU' as fp16, the products with z / dH are fp32 (v_fma_mix_f32)."""
    xh = r16(x32)
    zc = np.clip(xh, -4, 4)
    v = fma16(r16(zc * zc), 0.125, -1.0)

    def horner(c):
        c = r16(c)
        q = fma16(v, c[-1], c[-2])
        for ck in c[-3::-1]:
            q = fma16(q, v, ck)
        return q
    phi = fma16(zc, horner(cq), 0.5)
    dg = fma16(zc, horner(cr), 0.5)
    return np.asarray(x32, np.float32).astype(np.float64) * phi, dg


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


def main_backward():
    allh = np.arange(65536, dtype=np.uint16).view(f16).astype(np.float64)
    allh = allh[np.isfinite(allh) & (np.abs(allh) <= 8)]
    rng = np.random.default_rng(0)
    x = np.concatenate([allh, rng.normal(0, 1.2, 400000)])
    n0 = len(allh)
    rms = lambda e: float(np.sqrt(np.mean(e[n0:] ** 2)))
    h_ref, d_ref = x * Phi(x), Phi(x) + x * pdf(x)
    print(f"backward: exact GELU / GELU' rounded to bf16: rms {rms(np.abs(bf16_round(h_ref) - h_ref)):.2e} / {rms(np.abs(bf16_round(d_ref) - d_ref)):.2e}")
    fq, cq = fit_v(6, False)
    fr, cr = fit_v(7, True)
    h, d = gelu_grad_h(x, cq, cr)
    eh, ed = np.abs(bf16_round(h) - h_ref), np.abs(d - d_ref)
    print(f"backward, Q degree 6 / R degree 7 in v = x^2/8 - 1 (fit errors {fq:.1e} / {fr:.1e}): H (fp32 z . fp16 Phi, rounded to bf16) max {eh.max():.2e} rms {rms(eh):.2e};"
          f"  GELU' (fp16) max {ed.max():.2e} rms {rms(ed):.2e}")
    print("   " + "  ".join(f"#define KASF_GQ{k} {c:.10e}f" for k, c in enumerate(cq)))
    print("   " + "  ".join(f"#define KASF_GR{k} {c:.10e}f" for k, c in enumerate(cr)))


if __name__ == "__main__":
    main()
    main_backward()
    main()
