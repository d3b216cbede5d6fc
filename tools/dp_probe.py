"""Where the data-parallel code path's time goes at small per-rank batches (BASELINE configs[2]: 32 clips per rank): per-step wall times of the same
training step through (a) the plain path, (b) the stage-sliced backward with NO collective, (c) DataParallel(overlap=True) with single-rank RCCL,
(d) DataParallel(overlap=False): one all-reduce after the backward.   python tools/dp_probe.py [B=32] [steps=30]"""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import kasportsformer_amd as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29537"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(k, v)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x, y = (t.cuda() for t in K.synthetic_clips(B, 27, seed=1234, res=(1920, 1080), det_conf=True))


def run(mode):
    torch.manual_seed(114514)
    m = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
    m.attach_param_grads = False
    opt = K.FusedAdamW(m, lr=5e-4, weight_decay=0.01)
    dp = None
    if mode == "sliced":                       # stage-sliced kasf_backward, hook that does nothing
        m.grad_stage_hook = lambda st, g: None
        m.grad_stage_group = 7
    elif mode in ("dp", "dp_no_overlap"):
        dp = K.DataParallel(m, optimizer=opt, overlap=(mode == "dp"))
    times = []
    for s in range(STEPS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = K.loss3(m(x), y)
        loss.backward()
        if dp is not None:
            dp.finish_gradients()
        opt.step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    # and unsynchronised, back to back (what bench.py times)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(10):
        opt.zero_grad()
        loss, _ = K.loss3(m(x), y)
        loss.backward()
        if dp is not None:
            dp.finish_gradients()
        opt.step()
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps({"mode": mode, "batch": B, "ms_back_to_back": round(b2b, 3), "ms_per_step_synchronised": [round(t, 2) for t in times]}), flush=True)


for mode in ("plain", "sliced", "dp", "dp_no_overlap", "plain", "dp"):
    run(mode)
dist.destroy_process_group()
