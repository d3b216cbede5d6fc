import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kasportsformer_amd as K
torch.manual_seed(114514)
m = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
m.attach_param_grads = False
opt = K.FusedAdamW(m, lr=5e-4, weight_decay=0.01)
x, y = (t.cuda() for t in K.synthetic_clips(64, 27, seed=5))
for step in range(int(os.environ.get("STEPS", "301"))):
    opt.zero_grad()
    loss, parts = K.loss3(m(x), y)
    loss.backward()
    opt.step()
    if step % int(os.environ.get("EVERY", "50")) == 0:
        print(step, [round(float(v), 4) for v in parts], flush=True)
print("finite params:", bool(torch.isfinite(m._flat).all()))
