"""Throughput of the other SURVEY §8(d) configurations (bench.py reports configs[1] only).

    python tools/bench_configs.py train81        # config 4: T=81, B=128, bf16 training step
    python tools/bench_configs.py eval           # config 5: forward only, T=27, B/GPU sweep, with and without flip-TTA
    python tools/bench_configs.py small          # config 3 as one rank sees it: T=27 training at B=32 / 64, detector-confidence input, plain and data-parallel path
One JSON line per measurement.
"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime loads: RCCL's streams next to the engine's three (kasportsformer_amd/parallel.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kasportsformer_amd as K

class _Flop(dict):
    """Forward GFLOP per clip: 17 T tokens x (26 layers x (2,252,288 + 3 x 4 T x 128) + head 134,144 + embeddings 2,304)."""
    def __missing__(self, T):
        return 17 * T * (26 * (2252288 + 1536 * T) + 136448) / 1e9


FWD_GFLOP = _Flop()                     # 27 -> 27.44, 81 -> 85.28


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _single_rank_rccl():
    import torch.distributed as dist
    if not dist.is_initialized():
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))


def train(T, B, steps=5, warmup=2, cd="bf16", det_conf=False, dp=False):
    """`det_conf`: WorldPose-det style input (BASELINE configs[2]); `dp`: the data-parallel code path (stage-sliced backward, bucketed RCCL all-reduce) with
    one rank -- B = 32 with both is what every rank of configs[2] runs (one global batch of 256 scattered over 8 replicas, train_and_evaluate_wp.py:236-238)."""
    torch.manual_seed(114514)
    model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=T, compute_dtype=cd).cuda().train()
    model.attach_param_grads = False
    opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
    wrap = None
    if dp:
        _single_rank_rccl()
        wrap = K.DataParallel(model, optimizer=opt)
    x, y = (t.cuda() for t in K.synthetic_clips(B, T, seed=1234, **({"res": (1920, 1080), "det_conf": True} if det_conf else {})))

    def step():
        opt.zero_grad()
        loss, _ = K.loss3(model(x), y)
        loss.backward()
        if wrap is not None:
            wrap.finish_gradients()
        opt.step()
    dt = timed(step, steps, warmup)
    print(json.dumps({"config": f"train T={T} B={B} {cd}" + (", detector-confidence input" if det_conf else "") + (", data-parallel path (single-rank RCCL)" if dp else ""),
                      "clips_per_s": B / dt, "ms_per_step": dt * 1e3, "mfma_frac": B / dt * 3 * FWD_GFLOP[T] * 1e9 / 2.5e15}), flush=True)


def train_dropin(T=27, B=256, steps=5, warmup=2):
    """INTEGRATION.md path A, exactly as train_and_evaluate_sp.py:270-272,208-243 drives the reference module: a stock torch.optim.AdamW over
    model.parameters() (2,611 parameter views of the flat array) and per-parameter .grad tensors (attach_param_grads=True, the default)."""
    torch.manual_seed(114514)
    model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=T, compute_dtype="bf16").cuda().train()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.01)
    x, y = (t.cuda() for t in K.synthetic_clips(B, T, seed=1234))

    def step():
        pred = model(x)
        opt.zero_grad()
        loss, _ = K.loss3(pred, y)
        loss.backward()
        opt.step()
    dt = timed(step, steps, warmup)
    print(json.dumps({"config": f"train T={T} B={B} bf16, drop-in path A: torch.optim.AdamW(model.parameters()), attach_param_grads=True", "clips_per_s": B / dt,
                      "ms_per_step": dt * 1e3, "mfma_frac": B / dt * 3 * FWD_GFLOP[T] * 1e9 / 2.5e15}), flush=True)
    opt2 = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.01, foreach=True)
    opt, dt2 = opt2, None
    dt2 = timed(step, steps, warmup)
    print(json.dumps({"config": f"train T={T} B={B} bf16, drop-in path A with foreach=True", "clips_per_s": B / dt2, "ms_per_step": dt2 * 1e3}), flush=True)


def evaluate(T=27, batches=(32, 64, 128, 256, 512), steps=10, warmup=3):
    torch.manual_seed(114514)
    model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=T, compute_dtype="bf16").cuda().eval()
    for B in batches:
        x, _ = K.synthetic_clips(B, T, seed=1234)
        x = x.cuda()
        with torch.no_grad():
            dt = timed(lambda: model(x), steps, warmup)
            row = {"config": f"eval T={T} B={B} bf16", "clips_per_s": B / dt, "ms_per_batch": dt * 1e3, "mfma_frac": B / dt * FWD_GFLOP[T] * 1e9 / 2.5e15}
            if hasattr(K, "predict_flip_tta"):
                dt2 = timed(lambda: K.predict_flip_tta(model, x), steps, warmup)
                row["clips_per_s_flip_tta"] = B / dt2
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "eval"
    if what == "train81":
        train(81, 128)
    elif what == "train27":
        train(27, 256)
    elif what == "train27fp32":          # the parity mode (exact-f32 MFMA), same workload: the mode the <= 1e-3 / 0.1 mm claims are made in
        train(27, 256, steps=5, warmup=1, cd="fp32")
    elif what == "small":                # the per-rank regime of BASELINE configs[2]: 256 clips over 8 replicas = 32 per rank (64 at 4 ranks), detector-confidence input
        _single_rank_rccl()             # the process group first, as every multi-rank run has it (bench.py --gpus N): RCCL's streams exist before the engine's
        for B in (32, 64):
            train(27, B, steps=10, warmup=3, det_conf=True)
        for B in (32, 64):
            train(27, B, steps=10, warmup=3, det_conf=True, dp=True)
        train(27, 32, steps=10, warmup=3)
    elif what == "dropin":
        train_dropin()
    elif what == "train243":             # the long-clip configuration of the model family (generic temporal kernels)
        train(243, 32, steps=3, warmup=1)
    else:
        evaluate()
