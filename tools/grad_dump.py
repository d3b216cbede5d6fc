"""Writes the flat gradient of one training step (26 layers, bf16) to a file: for comparing kernel variants selected by environment switches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kasportsformer_amd as K
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(114514)
m = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
m.attach_param_grads = False
opt = K.FusedAdamW(m, lr=5e-4, weight_decay=0.01)
x, y = (t.cuda() for t in K.synthetic_clips(B, 27, seed=5))
reps = []
for rep in range(3):
    opt.zero_grad()
    loss, parts = K.loss3(m(x), y)
    loss.backward()
    torch.cuda.synchronize()
    g = m.flat_grad[:m.n_live].clone()
    reps.append(g)
    print(rep, float(loss), "finite:", bool(torch.isfinite(g).all()), "nan count:", int((~torch.isfinite(g)).sum()), "|g|max", float(g[torch.isfinite(g)].abs().max()))
torch.save(torch.stack(reps).cpu(), sys.argv[1])
