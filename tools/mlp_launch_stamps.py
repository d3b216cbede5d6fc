"""UN-PROFILED timeline of the backward's MLP launches (library built with -DKASF_LSTAMP, loaded through KASF_LIB): every k_mlp_bwd_s launch stamps the start and end of
its first and last workgroup with the constant 100 MHz clock.  Prints, per backward layer of the last training step, when each of the six launches started and ended
relative to the layer's first one -- whether the last-enqueued branch's first launch waits for its CUs (a rocprofv3 trace cannot tell: the profiler slows the host's
enqueue loop by an order of magnitude, which delays exactly that launch).
    KASF_LIB=.../libkasf_hip_ls.so python tools/mlp_launch_stamps.py [T] [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kasportsformer_amd as K
from kasportsformer_amd import _lib
T = int(sys.argv[1]) if len(sys.argv) > 1 else 27
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
raw = C.CDLL(_lib.LIB_PATH)
raw.kasf_debug_read_lstamps.restype = C.c_int
buf = (C.c_longlong * 16384)()
torch.manual_seed(114514)
model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=T, compute_dtype="bf16").cuda().train()
model.attach_param_grads = False
opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
x, y = (t.cuda() for t in K.synthetic_clips(B, T, seed=1234))
def step():
    opt.zero_grad()
    loss, _ = K.loss3(model(x), y)
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
raw.kasf_debug_read_lstamps(buf, 1)
step()
n = raw.kasf_debug_read_lstamps(buf, 0)
ent = [(buf[i], buf[i + 1], buf[i + 2], buf[i + 3]) for i in range(0, min(n, 16384), 4)]
launches = {}
for w1, blk, t0, t1 in ent:                      # group the (first, last) workgroup stamps of one launch: same weight pointer, nearest in time
    launches.setdefault(w1, []).append((blk, t0, t1))
ls = []
for w1, v in launches.items():
    ls.append((min(t0 for _, t0, _ in v), max(t1 for _, _, t1 in v), max(t0 for _, t0, _ in v), w1))
ls.sort()
ptrs = sorted({w1 for *_, w1 in ls})
kind = {p: i % 6 for i, p in enumerate(ptrs)}   # arena order within a layer: att_s, att_t, graph_s, graph_t, bone_s, bone_t
names = ["att_s", "att_t", "graph_s", "graph_t", "bone_s", "bone_t"]
print(f"{len(ls)} k_mlp_bwd_s launches in the step (156 expected); times in us relative to the layer's first launch: start(first WG) / start(last WG) - end")
tot = []
for li in range(0, len(ls), 6):
    seg = ls[li:li + 6]
    if len(seg) < 6:
        break
    t00 = seg[0][0]
    tot.append((max(e for _, e, _, _ in seg) - t00) / 100.0)
    if li // 6 in (0, 1, 5, 12, 25):
        print(f"layer {25 - li // 6:2d}: " + "  ".join(f"{names[kind[w1]]} {(a - t00) / 100.0:.0f}/{(c - t00) / 100.0:.0f}-{(e - t00) / 100.0:.0f}" for a, e, c, w1 in seg))
print(f"mean span from the first MLP start to the last MLP end of a layer: {sum(tot) / len(tot):.0f} us")
