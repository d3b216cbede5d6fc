"""State dependence of the step time next to RCCL (round 4): scenarios in FRESH processes.   python tools/dp_probe2.py <scenario>
Every line: what ran, ms per step (10 unsynchronised steps after 3 warm-up steps)."""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import kasportsformer_amd as K

scen = sys.argv[1]
torch.cuda.set_device(0)


def pg():
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29539"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))


def make(B=32):
    torch.manual_seed(114514)
    m = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
    m.attach_param_grads = False
    opt = K.FusedAdamW(m, lr=5e-4, weight_decay=0.01)
    x, y = (t.cuda() for t in K.synthetic_clips(B, 27, seed=1234))
    return m, opt, x, y


def timeit(tag, m, opt, x, y, dp=None, n=10):
    def step():
        opt.zero_grad()
        loss, _ = K.loss3(m(x), y)
        loss.backward()
        if dp is not None:
            dp.finish_gradients()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print(json.dumps({"scenario": scen, "what": tag, "ms_per_step": round((time.perf_counter() - t0) / n * 1e3, 2)}), flush=True)


if scen == "A":          # no process group at all: two models one after the other
    a = make(); timeit("plain, model 1", *a)
    b = make(); timeit("plain, model 2 (model 1 alive)", *b)
    del a; timeit("plain, model 2 (model 1 freed)", *b)
elif scen == "B":        # one big synchronous single-rank all-reduce in the middle
    pg()
    a = make(); timeit("plain after init_process_group", *a)
    t = torch.zeros(29_000_000, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dist.all_reduce(t); torch.cuda.synchronize()
    print(json.dumps({"scenario": scen, "what": "one single-rank all_reduce of 116 MB, synchronous", "ms": round((time.perf_counter() - t0) * 1e3, 2)}), flush=True)
    t0 = time.perf_counter(); dist.all_reduce(t); torch.cuda.synchronize()
    print(json.dumps({"scenario": scen, "what": "the same again", "ms": round((time.perf_counter() - t0) * 1e3, 2)}), flush=True)
    timeit("plain, same model, after the all_reduce", *a)
    b = make(); timeit("plain, new model", *b)
elif scen == "C":
    pg()
    a = make(); timeit("plain after init_process_group", *a)
    dp = K.DataParallel(a[0], optimizer=a[1]); timeit("DataParallel(overlap=True), same model", *a, dp=dp)
    a[0].grad_stage_hook = None; timeit("plain again, same model", *a)
elif scen == "D":        # the order tools/bench_configs.py small ran in
    a = make(32); timeit("plain B=32 (no process group yet)", *a); del a
    a = make(64); timeit("plain B=64", *a); del a
    pg()
    a = make(32); dp = K.DataParallel(a[0], optimizer=a[1]); timeit("DataParallel B=32", *a, dp=dp); del a, dp
    a = make(32); timeit("plain B=32 after", *a)
elif scen == "E":        # process group first, then DataParallel straight away (what bench.py --gpus N does)
    pg()
    a = make(32); dp = K.DataParallel(a[0], optimizer=a[1]); timeit("DataParallel B=32 (first thing after init)", *a, dp=dp)
    timeit("DataParallel B=32 again", *a, dp=dp)
elif scen == "F":        # is it the process or the model?  model 1 timed, model 2 created, model 1 timed AGAIN, model 2 timed, model 2 on one stream
    a = make(); timeit("plain, model 1", *a)
    b = make()
    timeit("plain, model 1 again (model 2 exists now)", *a)
    timeit("plain, model 2", *b)
    K.set_single_stream(True); timeit("model 2, three branches on ONE stream", *b); timeit("model 1, ONE stream", *a); K.set_single_stream(False)
    timeit("plain, model 1 once more", *a)
elif scen == "G":        # third and fourth model, each created with all earlier ones alive
    ms = []
    for i in range(4):
        ms.append(make()); timeit(f"plain, model {i + 1} ({i} earlier models alive)", *ms[-1])
    timeit("plain, model 1 at the end", *ms[0])
if dist.is_initialized():
    dist.destroy_process_group()
