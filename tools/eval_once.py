"""Forward-only passes (for rocprofv3): python tools/eval_once.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kasportsformer_amd as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(114514)
model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().eval()
x, _ = K.synthetic_clips(B, 27, seed=1234)
x = x.cuda()
with torch.no_grad():
    for _ in range(4):
        model(x)
torch.cuda.synchronize()
