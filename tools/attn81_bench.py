import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from kasportsformer_amd import _lib
lib = _lib.load()
B, T = 128, 81
M = B * T * 17
bf = torch.bfloat16
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
qkv = torch.randn(M, 384, device="cuda").to(bf); do = torch.randn(M, 128, device="cuda").to(bf); dqkv = torch.empty_like(qkv); o = torch.empty(M, 128, device="cuda", dtype=bf)
for mode in (0, 1):
    f = lambda: lib.kasf_op_attention_fwd(1, p(qkv), 384, C.c_void_p(qkv.data_ptr() + 256), C.c_void_p(qkv.data_ptr() + 512), 384, p(o), B, T, mode, st())
    b = lambda: lib.kasf_op_attention_bwd(1, p(qkv), 384, C.c_void_p(qkv.data_ptr() + 256), C.c_void_p(qkv.data_ptr() + 512), 384, p(do), p(dqkv), 384, C.c_void_p(dqkv.data_ptr() + 256), C.c_void_p(dqkv.data_ptr() + 512), 384, B, T, mode, st())
    print("mode", mode, "fwd us", round(bench.time_kernel(f) * 1e6, 1), "bwd us", round(bench.time_kernel(b) * 1e6, 1))

# temporal backward with d_o formed in-kernel (what the T = 81 training step runs): self-contained kernel vs the key-tile-outer one (lse + o from the forward)
g = torch.randn(M, 128, device="cuda").to(bf); w = (torch.randn(128, 128, device="cuda") * 0.1).to(bf)
lse = torch.zeros(M, 8, device="cuda") + 5.0
for name, oo, ll in (("k_attn_bwd_long<3,true> (self-contained)", None, None), ("k_attn_bwd_kt<3> (lse + o from the forward)", p(o), p(lse))):
    f = lambda: lib.kasf_op_attention_bwd_fused_do(p(qkv), 384, C.c_void_p(qkv.data_ptr() + 256), C.c_void_p(qkv.data_ptr() + 512), 384, p(g), p(w), p(dqkv), 384,
                                                   C.c_void_p(dqkv.data_ptr() + 256), C.c_void_p(dqkv.data_ptr() + 512), 384, B, T, 1, 0, oo, ll, st())
    print(name, "us", round(bench.time_kernel(f) * 1e6, 1))
