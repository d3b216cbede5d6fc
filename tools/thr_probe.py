"""ms per training step (T = 27, 26 layers, bf16) over a batch sweep: where the fused data + weight gradient kernels start to pay (engine.hip WG_FUSE_MIN_TOKENS).
    python tools/thr_probe.py;  KASF_LIB=<a build with -DKASF_WG_FUSE_MIN_TOKENS=1> python tools/thr_probe.py
Round 4, one box: unfused / fused  B=48 18.17 / 18.80, B=64 21.29 / 21.91, B=80 24.50 / 24.69, B=96 28.23 / 28.06 ms: the crossover sits at 37-44 k tokens."""
import sys, os, json, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import kasportsformer_amd as K
def run(B):
    torch.manual_seed(114514)
    m = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
    m.attach_param_grads = False
    opt = K.FusedAdamW(m, lr=5e-4, weight_decay=0.01)
    x, y = (t.cuda() for t in K.synthetic_clips(B, 27, seed=1234))
    def step():
        opt.zero_grad(); loss, _ = K.loss3(m(x), y); loss.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3
print(json.dumps({"lib": os.environ.get("KASF_LIB", "default (threshold 40000)"), **{f"B={B}": round(run(B), 2) for B in (48, 64, 80, 96, 128, 160)}}))
