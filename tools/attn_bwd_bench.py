"""Attention backward with in-kernel d_o (kasf_op_attention_bwd_fused_do) at the benchmark's shape: persistent form (0) against the
one-group-per-workgroup form (1).  python tools/attn_bwd_bench.py [B] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kasportsformer_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 27
lib = _lib.load()
torch.manual_seed(0)
M = B * T * 17
qkv = torch.randn(M, 384, device="cuda").bfloat16()
g = torch.randn(M, 128, device="cuda").bfloat16()
w = (torch.randn(128, 128, device="cuda") * 0.1).bfloat16()
dqkv = torch.empty_like(qkv)
st = torch.cuda.current_stream().cuda_stream
p, d = qkv.data_ptr(), dqkv.data_ptr()
for mode in (0, 1):
    for form in (1, 0):
        def run():
            _lib.check(lib.kasf_op_attention_bwd_fused_do(p, 384, p + 256, p + 512, 384, g.data_ptr(), w.data_ptr(), d, 384, d + 256, d + 512, 384, B, T, mode, form, None, None, st))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        us = min(ts)
        print(f"mode {mode} ({'spatial' if mode == 0 else 'temporal'}) form {form} ({'persistent' if form == 0 else 'one group per workgroup'}): "
              f"{us:7.1f} us  {M * 7 * 256 / us / 1e3:7.0f} GB/s of algorithmic bytes (q|k|v, g_mid in; dq|dk|dv out)", flush=True)
