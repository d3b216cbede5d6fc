"""Phase timers of the fused attention-block backward (library built with -DBWDH_PROF, loaded through KASF_LIB): one layer, one step, one stream;
the kernel prints the clock64 sums of workgroup 77's waves 0 and 5 per phase.   KASF_LIB=.../libkasf_hip_prof.so python tools/bwdh_prof.py [T] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KASF_SINGLE_STREAM", "1")
import torch
import kasportsformer_amd as K
T = int(sys.argv[1]) if len(sys.argv) > 1 else 27
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.manual_seed(114514)
model = K.KASportsFormer(n_layers=1, num_heads=8, n_frames=T, compute_dtype="bf16").cuda().train()
model.attach_param_grads = False
x, y = (t.cuda() for t in K.synthetic_clips(B, T, seed=1234))
for _ in range(2):
    model.flat_grad = None
    loss, _ = K.loss3(model(x), y)
    loss.backward()
    torch.cuda.synchronize()
