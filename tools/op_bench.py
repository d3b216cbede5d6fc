"""Isolated timings of the non-MLP GEMM-family kernels at the bench size, against their HBM floor (6 TB/s achievable)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from kasportsformer_amd import _lib

lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 27 * 17
bf, dev = torch.bfloat16, "cuda"
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
r = lambda *s: torch.randn(*s, device=dev).to(bf)
f = lambda *s: torch.randn(*s, device=dev)
x, gam, bet = r(M, 128), f(128), f(128)
out = {}


def rec(name, fn, nbytes):
    t = bench.time_kernel(fn)
    out[name] = {"us": round(t * 1e6, 1), "hbm_floor_us": round(nbytes / 6e12 * 1e6, 1), "GBps": round(nbytes / t / 1e9)}


for N, ln in ((384, True), (256, True), (128, True), (256, False), (512, True)):
    w, b, y, xn = r(N, 128), f(N), torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, 128, device=dev, dtype=bf)
    rec(f"linear N={N} ln={int(ln)}", lambda: lib.kasf_op_linear(1, p(x), p(w), p(b), p(y), M, N, p(gam) if ln else None, p(bet) if ln else None, p(xn) if ln else None, 0, st()),
        M * (128 + N + (128 if ln else 0)) * 2)
for Kd in (128, 256, 384):
    dy, wt, res, o = r(M, Kd), r(128, Kd), r(M, 128), torch.empty(M, 128, device=dev, dtype=bf)
    dg, db = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
    xo = torch.empty(M, 128, device=dev, dtype=bf)
    rec(f"dgrad_lnbwd Kd={Kd} +xn", lambda: lib.kasf_op_dgrad_lnbwd(1, p(dy), Kd, p(wt), None, p(x), p(gam), p(res), p(o), 0, p(dg), p(db), M, p(xo), p(bet), st()),
        M * (Kd + 4 * 128) * 2)
    if Kd == 256:
        rec(f"dgrad_lnbwd Kd={Kd} gcn", lambda: lib.kasf_op_dgrad_lnbwd(1, p(dy), Kd, p(wt), p(res), p(x), p(gam), p(res), p(o), 0, p(dg), p(db), M, None, None, st()),
            M * (Kd + 4 * 128) * 2)
        rec(f"dgrad_lnbwd Kd={Kd} kv", lambda: lib.kasf_op_dgrad_lnbwd(1, p(dy), Kd, p(wt), None, p(x), p(gam), None, p(o), 1, p(dg), p(db), M, p(xo), p(bet), st()),
            M * (Kd + 4 * 128) * 2)
part = torch.empty(256 * 128 * 128, device=dev)
for N, K in ((384, 128), (128, 128), (256, 128), (128, 384)):
    g, xx, dw, dbv = r(M, N), r(M, K), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    rec(f"wgrad N={N} K={K}", lambda: lib.kasf_op_wgrad(1, p(g), N, p(xx), K, None, None, p(dw), p(dbv), M, p(part), part.numel(), st()), M * (N + K) * 2)
B, T = M // (27 * 17), 27
qkv, o = r(M, 384), torch.empty(M, 128, device=dev, dtype=bf)
do, dqkv = r(M, 128), torch.empty(M, 384, device=dev, dtype=bf)
for mode in (0, 1):
    rec(f"attn_fwd mode={mode}", lambda: lib.kasf_op_attention_fwd(1, p(qkv), 384, C.c_void_p(qkv.data_ptr() + 256), C.c_void_p(qkv.data_ptr() + 512), 384, p(o), B, T, mode, st()), M * 512 * 2)
    rec(f"attn_bwd mode={mode}", lambda: lib.kasf_op_attention_bwd(1, p(qkv), 384, C.c_void_p(qkv.data_ptr() + 256), C.c_void_p(qkv.data_ptr() + 512), 384, p(do), p(dqkv), 384,
                                                                   C.c_void_p(dqkv.data_ptr() + 256), C.c_void_p(dqkv.data_ptr() + 512), 384, B, T, mode, st()), M * (384 + 128 + 384) * 2)
for k, v in out.items():
    print(f"{k:28s} {v['us']:8.1f} us   floor {v['hbm_floor_us']:6.1f} us   {v['GBps']:6d} GB/s")
