"""Does it matter which stream the caller hands the engine?  T = 27 training step rate with the step issued on torch's default stream against a fresh
non-default stream (the engine's two side streams are process-wide; which hardware queues the three end up on decides how well the branches overlap:
DESIGN section 6, round 4).   python tools/caller_stream_probe.py [B=256]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kasportsformer_amd as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(114514)
model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
model.attach_param_grads = False
opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
x, y = (t.cuda() for t in K.synthetic_clips(B, 27, seed=1234))


def step():
    opt.zero_grad()
    loss, _ = K.loss3(model(x), y)
    loss.backward()
    opt.step()


def rate(steps=8, warmup=3):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


out = {"B": B, "default_stream": [], "fresh_streams": []}
for rep in range(3):
    out["default_stream"].append(round(rate()))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out["fresh_streams"].append(round(rate()))
    torch.cuda.current_stream().wait_stream(s)
print(json.dumps(out))
