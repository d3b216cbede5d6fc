"""A few training steps at a given (T, B) for rocprofv3: python tools/train_once.py [T] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kasportsformer_amd as K
T = int(sys.argv[1]) if len(sys.argv) > 1 else 81
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.manual_seed(114514)
model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=T, compute_dtype="bf16").cuda().train()
model.attach_param_grads = False
opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
x, y = (t.cuda() for t in K.synthetic_clips(B, T, seed=1234))
for _ in range(3):
    opt.zero_grad()
    loss, _ = K.loss3(model(x), y)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
