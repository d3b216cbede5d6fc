"""Reads the clock64 segment timers of a -DKASF_PROBE_TIMERS build of k_mlp3.hip (workgroup 7, lane 0 of a producer and a consumer wave)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kasportsformer_amd import _lib
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_longlong * 32)()
M = 256 * 27 * 17
dev, bf = "cuda", torch.bfloat16
x = torch.randn(M, 128, device=dev).to(bf); out = torch.empty_like(x); xn = torch.empty_like(x)
w1 = (torch.randn(512, 128, device=dev) * 0.05).to(bf); w2 = (torch.randn(128, 512, device=dev) * 0.05).to(torch.float16)   # ABI 7: fp16 copy of fc2.weight
b1 = torch.zeros(512, device=dev); b2 = torch.zeros(128, device=dev); ls = torch.ones(128, device=dev); gam = torch.ones(128, device=dev); bet = torch.zeros(128, device=dev)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
bnames = {19: "P issue loads", 16: "P frags+gemm(0)", 20: "P gemm(1)+act(0) [fine]", 17: "P act(1) (or gemm1+act0+act1)", 21: "P vmcnt wait", 18: "P barrier", 24: "C loop top", 25: "C dA", 26: "C wgrad", 27: "C wait vmcnt", 28: "C dA stores", 29: "C barrier"}
names = {0: "P frag+gemm1(0)", 1: "P slices", 2: "P barrier", 8: "C issue", 9: "C gemm2", 10: "C wait vmcnt", 11: "C layernorm", 12: "C epilogue", 13: "C barrier"}
for label, xo in (("with xn store", xn), ("no xn store", None)):
    lib.kasf_op_mlp_fwd(1, p(x), p(gam), p(bet), p(w1), p(b1), p(w2), p(b2), p(ls), p(out), M, p(xo), st())
    raw.kasf_debug_read_prof(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.kasf_op_mlp_fwd(1, p(x), p(gam), p(bet), p(w1), p(b1), p(w2), p(b2), p(ls), p(out), M, p(xo), st())
    e1.record(); torch.cuda.synchronize()
    raw.kasf_debug_read_prof(buf, 0)
    v = list(buf)
    tiles = 10 * ((M + 31) // 32 + 255) // 256
    print(label, "us/launch", e0.elapsed_time(e1) * 100)
    for k, n in names.items():
        print(f"   {n:20s} {v[k] / tiles:9.0f} cycles/tile")

# fused backward (k_mlp_bwd_s): same timers, indices 16..29
g = torch.randn(M, 128, device=dev).to(bf); w2ts = (torch.randn(512, 128, device=dev) * 0.05).to(bf); w1t = w1.t().contiguous()
dap = torch.empty(4 * M * 128, device=dev, dtype=bf); part2 = torch.empty(2 * 64 * 65536 + 2048, device=dev)
dW1 = torch.zeros(512, 128, device=dev); dW2 = torch.zeros(128, 512, device=dev); db1 = torch.zeros(512, device=dev); gs = torch.zeros(128, device=dev)
gin = torch.empty_like(x); dg_ = torch.zeros(128, device=dev); db_ = torch.zeros(128, device=dev)
bwd = lambda: lib.kasf_op_mlp_bwd_fused(p(x), p(xn), p(g), p(gam), p(w1), p(b1), p(w2ts), p(w1t), p(dap), p(part2), p(dW1), p(dW2), p(db1), p(gs), p(gin), p(dg_), p(db_), M, st())
bwd(); raw.kasf_debug_read_prof(buf, 1)
for _ in range(10): bwd()
raw.kasf_debug_read_prof(buf, 0)
v = list(buf)
tiles = 10 * 58          # 3,672 tiles / 64 ranges = 57.4 per workgroup
print("fused backward, cycles per tile (workgroup 7)")
for k, n in bnames.items():
    print(f"   {n:20s} {v[k] / tiles:9.0f}")
