"""Phase timers of k_attn_bwd_kt (temporal attention backward at T = 81, B = 128) through the operator entry point: the library built with -DKT_PROF prints clock64
differences per phase for workgroups 777 and 2000, waves 0 and 3.   make -C kasportsformer_amd/csrc BUILD=build_ktprof LIB=../libkasf_hip_prof.so EXTRA=-DKT_PROF ;
KASF_LIB=$PWD/kasportsformer_amd/libkasf_hip_prof.so python tools/kt_prof.py   (profiles/r6_kt_phase_timers.txt)"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from kasportsformer_amd import _lib
lib = _lib.load()
B, T = 128, 81
M = B * T * 17
bf = torch.bfloat16
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
qkv = torch.randn(M, 384, device="cuda").to(bf); dqkv = torch.empty_like(qkv); o = torch.randn(M, 128, device="cuda").to(bf)
g = torch.randn(M, 128, device="cuda").to(bf); w = (torch.randn(128, 128, device="cuda") * 0.1).to(bf)
lse = torch.zeros(M, 8, device="cuda") + 5.0
for _ in range(2):
    lib.kasf_op_attention_bwd_fused_do(p(qkv), 384, C.c_void_p(qkv.data_ptr() + 256), C.c_void_p(qkv.data_ptr() + 512), 384, p(g), p(w), p(dqkv), 384,
                                       C.c_void_p(dqkv.data_ptr() + 256), C.c_void_p(dqkv.data_ptr() + 512), 384, B, T, 1, 0, p(o), p(lse), st())
    torch.cuda.synchronize()
