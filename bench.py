#!/usr/bin/env python3
"""Headline benchmark of the KASportsFormer hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...          # starts N rank processes itself (one per GPU, RCCL), relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...   # or under a launcher

A "step" is one full training pass of the path over one batch already resident in HBM: forward, fused
3-term loss, backward, (N>1: bucketed RCCL all-reduce of the flat gradient, overlapped with backward),
fused AdamW.  Workload (BASELINE.json configs[1]): SportsPose-GT 27-frame clips, bf16 compute, batch 256
per GPU, the shipped 26-layer model, synthetic inputs, reference default init under seed 114514.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

# The engine runs the three branches of a layer on three HIP streams.  RCCL adds streams of its own, and with the runtime's default of 4
# hardware queues per process the branch streams then share queues and serialise (measured: -6 % with a process group merely alive).
# Must be set before the HIP runtime loads, i.e. before `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")          # no effect on the single-process run (measured), needed as soon as RCCL is alive
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver

torch = None          # imported by main() AFTER the launcher decision: the process that only starts the ranks never loads the HIP runtime

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH_PER_GPU, T, LAYERS = 256, 27, 26
PEAK_BF16_TFLOPS = 2500.0            # dense bf16 MFMA peak (MI355X_MICROARCH.md)
FLOP_PER_CLIP_TRAIN = 82.31e9        # 3 x 27.44 GFLOP forward
MLP_FLOP_PER_TOKEN_FWD = 262144      # fc1 + fc2 (2 x 2 x 128 x 512)


def _need_torch():
    """torch is imported lazily (the launcher process must not load the HIP runtime); tools that import this module call straight into the helpers."""
    global torch
    if torch is None:
        import torch as _torch
        torch = _torch


def time_kernel(fn, iters=20, warmup=3):
    _need_torch()
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def kernel_rooflines(M):
    """Times the MLP kernels (the dominant launches: ~69 % of the path's FLOPs) in isolation on torch's
    current stream, the stream the library launches on."""
    _need_torch()
    from kasportsformer_amd import _lib
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)
    bf = torch.bfloat16
    x = torch.randn(M, 128, generator=g).to(dev, bf)
    gout = torch.randn(M, 128, generator=g).to(dev, bf)
    w1 = (torch.randn(512, 128, generator=g) / 11.3).to(dev, bf)
    w2 = (torch.randn(128, 512, generator=g) / 22.6).to(dev, bf)
    w2ts, w1t = w2.t().contiguous(), w1.t().contiguous()
    w2h = w2.to(torch.float16)           # the forward's GEMM2 operand: an FP16 copy of fc2.weight (kasf.h, ABI 7)
    b1, b2 = torch.zeros(512, device=dev), torch.zeros(128, device=dev)
    ls, gam, bet = torch.ones(128, device=dev), torch.ones(128, device=dev), torch.zeros(128, device=dev)
    out, gin = torch.empty_like(x), torch.empty_like(x)
    H, dZ = torch.empty(M, 512, device=dev, dtype=bf), torch.empty(M, 512, device=dev, dtype=bf)
    dg, db = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
    dW1, db1 = torch.zeros(512, 128, device=dev), torch.zeros(512, device=dev)
    dW2, gs = torch.zeros(128, 512, device=dev), torch.zeros(128, device=dev)
    part = torch.empty(256 * 128 * 128, device=dev)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    xn = torch.empty_like(x)
    fwd = lambda: lib.kasf_op_mlp_fwd(1, p(x), p(gam), p(bet), p(w1), p(b1), p(w2h), p(b2), p(ls), p(out), M, p(xn), st())    # training-mode forward: also stores LN(x)
    dap = torch.empty(4 * M * 128, device=dev, dtype=bf)
    part2 = torch.empty(2 * 64 * 65536 + 2048, device=dev)
    # the engine's bf16 backward: data gradient + both weight gradients + LayerNorm backward in k_mlp_bwd_s / k_lnbwd_sum4_fin (the dA-partial stream and the weight-gradient finish in one launch)
    bwd = lambda: lib.kasf_op_mlp_bwd_fused(p(x), p(xn), p(gout), p(gam), p(w1), p(b1), p(w2ts), p(w1t), p(dap), p(part2), p(dW1), p(dW2), p(db1),
                                            p(gs), p(gin), p(dg), p(db), M, st())
    wg1 = lambda: lib.kasf_op_wgrad(1, p(dZ), 512, p(x), 128, None, None, p(dW1), p(db1), M, p(part), part.numel(), st())   # engine path: X = LN(x) emitted by k_mlp_bwd
    wg2 = lambda: lib.kasf_op_wgrad(1, p(gout), 128, p(H), 512, None, None, p(dW2), p(gs), M, p(part), part.numel(), st())
    res = {}
    # algorithmic FLOP: forward 2 GEMMs; fused backward = dgrad (2 GEMMs) + wgrad (2 GEMMs) = 2x forward (the Z recompute is not counted)
    for name, fn, flop in (("k_mlp_fwd_s", fwd, MLP_FLOP_PER_TOKEN_FWD * M), ("k_mlp_bwd_s(+lnbwd_sum4_fin)", bwd, 2 * MLP_FLOP_PER_TOKEN_FWD * M),
                           ("k_wgrad_ring[512x128]", wg1, MLP_FLOP_PER_TOKEN_FWD // 2 * M), ("k_wgrad_ring[128x512]", wg2, MLP_FLOP_PER_TOKEN_FWD // 2 * M)):
        t = time_kernel(fn)
        res[name] = {"seconds": t, "achieved_tflops": flop / t / 1e12, "algorithmic_flop": flop}
    return res


PROFILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")


def _profile(name):
    """Newest committed round of a profile file (profiles/r6_<name>, else r5_ / r4_ / r3_ / r2_<name>): counters cannot be read from inside the process, so the bench
    line QUOTES the committed summaries of separate rocprofv3 runs of the same launches (tools/refresh_profiles_r6.sh) and says so (`*_source`)."""
    for rnd in ("r6", "r5", "r4", "r3", "r2"):
        f = os.path.join(PROFILES, f"{rnd}_{name}")
        if os.path.exists(f):
            return f
    return os.path.join(PROFILES, f"r6_{name}")


STALE = []           # quoted profile files whose stamp is not this tree's (tools/stamp.py): reported as "profile_stale" and NOT quoted


def fresh(path):
    """True if `path` was measured on the sources this library is built from (its stamp's source_sha256 = sha256 over csrc/ + include/kasf.h of this tree).
    A kernel edited after the last refresh, or a pre-round-5 file without a stamp, is stale: the bench line then omits the figure instead of shipping last
    round's traffic / in-step numbers beside this round's kernels."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import stamp
    finally:
        sys.path.pop(0)
    ok = os.path.exists(path) and stamp.is_fresh(path, ROOT)
    if not ok and os.path.exists(path) and os.path.basename(path) not in STALE:
        STALE.append(os.path.basename(path))
    return ok


SQ_FILE = _profile("mlp_sq_counters.json")             # tools/pmc_kernel.sh (six --pmc passes over tools/mlp_bench.py), per-launch means + mfma_busy
TRAFFIC_FILE = _profile("pmc_traffic.json")            # tools/pmc_traffic.py (two --pmc passes over tools/mlp_bench.py)
IN_STEP_STATS = _profile("train_kernel_stats.csv")     # rocprofv3 --kernel-trace --stats of tools/train_once.py 27 256 (three streams overlap)
ISOLATED_STATS = _profile("single_stream_kernel_stats.csv")   # the same steps with the three branches on ONE stream (tools/prof27.sh): every launch alone on the chip
ISOLATED_FULL_STATS = _profile("single_stream_fullwidth_kernel_stats.csv")   # ... with every persistent launch at its full grid (KASF_NARROW_PCTS=100,...): comparable with rounds 1-3
MLP_HALF_CHIP_BELOW = 1 << 40                          # tokens (csrc/kernels.h: kasf_narrow_grid, KASF_HALF_CHIP_ALWAYS since round 6): an MLP launch of the engine takes 128 of the 256 CUs
STEP_TRAFFIC_FILE = _profile("pmc_step.json")          # tools/pmc_step.sh: FETCH_SIZE / WRITE_SIZE summed over whole training steps
TRAFFIC_PARTS = {"k_mlp_fwd_s": {"k_mlp_fwd_s": 1}, "k_mlp_bwd_s(+lnbwd_sum4_fin)": {"k_mlp_bwd_s": 1, "k_lnbwd_sum4_fin": 1}}


def pmc_traffic(entry, M):
    """HBM bytes per launch of a micro-benchmark entry, from the committed rocprofv3 --pmc summary (tools/pmc_traffic.py; the
    counters cannot be read from inside the process).  Only valid for the token count the summary was collected at."""
    if entry not in TRAFFIC_PARTS or M != BATCH_PER_GPU * T * 17 or not fresh(TRAFFIC_FILE):
        return None
    ks = json.load(open(TRAFFIC_FILE))["kernels"]
    if any(k not in ks for k in TRAFFIC_PARTS[entry]):
        return None
    return sum(ks[k]["hbm_bytes"] * n for k, n in TRAFFIC_PARTS[entry].items())


def in_step_duration(kernel_names, stats_file=None):
    """Average in-step launch duration (seconds) of the named kernels from the committed kernel trace of whole training steps: there three
    streams overlap, so a launch shares the chip with the other two branches -- the honest figure next to the isolated micro-benchmark."""
    import csv
    stats_file = stats_file or IN_STEP_STATS
    if not fresh(stats_file):
        return None
    total = 0.0
    for r in csv.DictReader(open(stats_file)):
        for k in kernel_names:
            if k + "(" in r["Name"] or k + "<" in r["Name"] or ("N_1" + str(len(k)) + k) in r["Name"]:
                total += float(r["AverageNs"]) * 1e-9
                break
    return total or None


def fp32_mode_rate(batch, steps=5):
    """The parity mode (exact-f32 MFMA: the mode the <= 1e-3 / <= 0.1 mm claims are made in) on the same workload, after the timed region."""
    _need_torch()
    import kasportsformer_amd as K
    torch.manual_seed(114514)
    m = K.KASportsFormer(n_layers=LAYERS, num_heads=8, n_frames=T, compute_dtype="fp32").cuda().train()
    m.attach_param_grads = False
    opt = K.FusedAdamW(m, lr=5e-4, weight_decay=0.01)
    x, y = (t.cuda() for t in K.synthetic_clips(batch, T, seed=1234))

    def step():
        opt.zero_grad()
        loss, _ = K.loss3(m(x), y)
        loss.backward()
        opt.step()
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"clips_per_sec": batch / dt, "ms_per_step": dt * 1e3, "steps": steps,
            "note": "same workload in the fp32 parity mode (exact-f32 MFMA); not part of value"}


def cpu_baseline(batch=8, steps=8):
    """Bounded sample of the SAME workload on the host cores: the CPU oracle (PyTorch fp32 restatement,
    verified equal to the reference on the golden fixtures) doing forward + 3-term loss + backward + AdamW."""
    _need_torch()
    from oracle import kasf_oracle as O
    torch.manual_seed(114514)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:                                     # cgroup quota (the GPU box shows every host CPU but grants a share)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: {cores} threads (os.cpu_count()={os.cpu_count()})", file=sys.stderr, flush=True)
    m = O.KASportsFormerOracle(n_layers=LAYERS, num_heads=8, n_frames=T).train()
    opt = torch.optim.AdamW(m.parameters(), lr=5e-4, weight_decay=0.01)
    x, y = O.synthetic_clips(batch, T)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = O.loss_total(m(x), y)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline step {i}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
    dt = sum(times[1:]) / steps
    return {"value": batch / dt, "unit": "pose-clips/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed train steps (1 warm-up) of the 26-layer fp32 CPU oracle at batch {batch}, T={T}"}


def workload_name(args, world, strong):
    data = "WorldPose-det style (detector-confidence input)" if args.det_conf else "SportsPose-GT"
    Tn = args.frames
    if args.eval_only:
        tag = " = BASELINE.json configs[4]" if (strong and args.global_batch == 2048 and world == 8 and Tn == 27) else ""
        return f"synthetic [B={args.batch * world}, T={Tn}, J=17] inference only (forward, evaluation mode, no flip), {args.batch} clips per GPU on {world} GPU(s){tag}"
    if strong:
        tag = " = BASELINE.json configs[2]" if (args.det_conf and args.global_batch == 256 and world == 8 and Tn == 27) else ""
        return f"{data} {Tn}-frame bf16 training, ONE global batch of {args.global_batch} split over {world} GPU(s) ({args.batch} clips per rank; strong scaling){tag}"
    tag = " (BASELINE.json configs[1])" if (args.batch == BATCH_PER_GPU and not args.det_conf and Tn == 27) else ""
    tag = " (BASELINE.json configs[3])" if (args.batch == 128 and not args.det_conf and Tn == 81 and world == 1) else tag
    return f"{data} {Tn}-frame bf16 training, batch={args.batch} per GPU{tag}"


def parity_summary():
    """What the parity tests OBSERVED on the shipped kernels, read from the files tests/test_gpu_model.py::test_full_depth_26_layers_against_oracle writes
    (gpurun_out/r6_parity_26layers_<mode>.json, committed as profiles/...): nothing in this string is typed in."""
    out = {}
    for cd in ("fp32", "bf16"):
        f = _profile(f"parity_26layers_{cd}.json")
        if not fresh(f):
            out[cd] = None
            continue
        r = json.load(open(f))
        out[cd] = {"samples": len(r["samples"]), "forward_rel_err_median": r["forward_rel_err_median"], "forward_rel_err_max": r["forward_rel_err_max"],
                   "gradient_cosine_median": r["gradient_cosine_median"], "gradient_cosine_min": r["gradient_cosine_min"],
                   "topk_rows_identical_pct": r["topk_rows_identical_pct"], "source": "committed file profiles/" + os.path.basename(f)}
        if "forward_rel_err_free_running_max" in r:      # the oracle taking its OWN top-4 decisions instead of following the HIP path's: the flipped near-ties are O(1) changes of a token
            out[cd]["forward_rel_err_free_running"] = r["forward_rel_err_free_running_max"]
            out[cd]["topk_rows_differ"] = r["topk_rows_differ"]
            out[cd]["topk_rows_differ_not_near_tie"] = r["topk_rows_differ_not_near_tie"]
    out["what"] = ("26-layer HIP model vs the CPU oracle following the same top-4 neighbour decisions, B = 2, de-identitied weights "
                   "(tests/test_gpu_model.py::test_full_depth_26_layers_against_oracle; bf16 = this line's mode: (input seed, weight salt) samples)")
    f = _profile("train_fidelity.json")
    if os.path.exists(f):                    # (a 1,000-step training run, not a per-build measurement: quoted by name whatever its stamp)
        out["training_fidelity_source"] = "committed file profiles/" + os.path.basename(f) + " (tools/train_fidelity.py: 1,000 steps at full depth, bf16 vs fp32 mode)"
    return out


FLOP_PER_CLIP_FWD = {27: 27.44e9, 81: 85.28e9}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="number of ranks (one process per GPU).  Without a launcher's WORLD_SIZE in the environment "
                    "and N > 1, this process starts the N ranks itself and relays rank 0's line")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="clips per GPU (weak scaling: the default)")
    ap.add_argument("--frames", type=int, default=T, help="clip length (27: configs[1]; 81: configs[3])")
    ap.add_argument("--global-batch", type=int, default=0, help="STRONG scaling: one global batch of this many clips split over the ranks (BASELINE configs[2] is "
                    "--gpus 8 --global-batch 256 --det-conf: 32 clips per rank, as nn.DataParallel scatters one batch, train_and_evaluate_wp.py:236-238)")
    ap.add_argument("--det-conf", action="store_true", help="WorldPose-det style input: detector confidence ~U(0,1) in the third channel, 1920x1080 frames (configs[2])")
    ap.add_argument("--eval-only", action="store_true", help="forward only (evaluation mode, no flip): BASELINE configs[4] is --gpus 8 --eval-only --global-batch 2048")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 parity-mode throughput leg")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the legs for BASELINE configs[2] (per-rank shape), [3] and [4] that follow the headline at N = 1")
    ap.add_argument("--grad-dtype", choices=("fp32", "bf16"), default="fp32", help="wire format of the gradient all-reduce (bf16: 58.7 MB per step instead of 117.4)")
    ap.add_argument("--force-dp", action="store_true", help="testing: take the data-parallel code path (stage-sliced backward, RCCL all-reduce) with one rank")
    ap.add_argument("--launch-only", action="store_true", help="testing: the ranks rendezvous, all-reduce one number and leave (gloo without GPUs); no model")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`bench.py --gpus N` without a launcher: start N rank processes (one per GPU) and relay rank 0's JSON line.  This process never
    touches the GPU (torch is not even imported here), so nothing is exec'ed from a process that has initialised HIP; the ranks are
    plain children.  The reference's counterpart is `torch.nn.DataParallel(model)` inside ONE process (train_and_evaluate_sp.py:262-264)."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    with socket.socket() as sk:                          # a free port now; rank 0 binds it a moment later (another process could take it in between: the ranks then
        sk.bind(("127.0.0.1", 0))                        # fail at rendezvous, this launcher returns non-zero and prints no line -- run again)
        port = sk.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line = []
    reader = threading.Thread(target=lambda: line.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    alive = set(range(n))
    kill_at = None                                       # a rank that ignores SIGTERM (stuck in a collective) is killed a few seconds later
    while alive:
        for r in sorted(alive):
            c = procs[r].poll()
            if c is None:
                continue
            alive.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                print(f"[bench] rank {r} exited with code {c}: stopping the other ranks", file=sys.stderr, flush=True)
                for q in alive:
                    procs[q].terminate()             # exactly the processes started above
                kill_at = time.monotonic() + 5.0
        if kill_at is not None and alive and time.monotonic() > kill_at:
            for q in alive:
                procs[q].kill()
            kill_at = time.monotonic() + 1e9
        time.sleep(0.05)
    reader.join(timeout=10)
    out = (line[0] if line else b"").decode()
    if rc == 0:                                          # a failed run prints no line: a rank 0 that finished before another rank failed must not look like a result
        sys.stdout.write(out)
        sys.stdout.flush()
        if not out.strip():
            rc = 1
    return rc


def launch_only(args, rank, world, local, real_stdout):
    """Rendezvous check without a model: what `tests/test_bench_launch_cpu.py` runs with two ranks on CPU (gloo)."""
    import torch.distributed as dist
    gpu = torch.cuda.device_count() >= world and torch.cuda.is_available()
    if gpu:
        torch.cuda.set_device(local)
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)
    dist.init_process_group("nccl" if gpu else "gloo", **({"device_id": torch.device("cuda", local)} if gpu else {}))
    t = torch.ones(1, device="cuda" if gpu else "cpu") * (rank + 1)
    dist.all_reduce(t)
    ok = int(t.item()) == world * (world + 1) // 2
    if rank == 0:
        os.write(real_stdout, (json.dumps({"launch_only": True, "n_gpus": world, "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                                           "allreduce_ok": ok, "gpus_arg": args.gpus}) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def extra_config_legs(log):
    """Driver-timed figures for the other BASELINE configurations, after the headline's timed region (N = 1 only): configs[3] (T = 81, B = 128 training),
    configs[4] at one GPU (B = 2,048 forward only in one pass) and the per-rank shape of configs[2] (32 detector-confidence clips through the
    data-parallel code path with single-rank RCCL -- in a child process: the process group has to exist before the first model)."""
    import subprocess
    import kasportsformer_amd as K
    legs = {}

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def leg_train81():
        torch.manual_seed(114514)
        m = K.KASportsFormer(n_layers=LAYERS, num_heads=8, n_frames=81, compute_dtype="bf16").cuda().train()
        m.attach_param_grads = False
        opt = K.FusedAdamW(m, lr=5e-4, weight_decay=0.01)
        x, y = (t.cuda() for t in K.synthetic_clips(128, 81, seed=1234))

        def step():
            opt.zero_grad()
            loss, _ = K.loss3(m(x), y)
            loss.backward()
            opt.step()
        dt = timed(step, 10, 2)
        log(f"configs[3] leg done: {128 / dt:.0f} clips/s")
        return {"workload": "SportsPose-GT 81-frame bf16 training, batch=128, 1 GPU", "clips_per_sec": 128 / dt, "ms_per_step": dt * 1e3, "steps": 10,
                "model_mfma_frac": 128 / dt * 3 * FLOP_PER_CLIP_FWD[81] / (PEAK_BF16_TFLOPS * 1e12)}

    def leg_eval2048():
        torch.manual_seed(114514)
        m = K.KASportsFormer(n_layers=LAYERS, num_heads=8, n_frames=T, compute_dtype="bf16").cuda().eval()
        x, _ = K.synthetic_clips(2048, T, seed=1234)
        x = x.cuda()
        with torch.no_grad():
            dt = timed(lambda: m(x), 5, 2)
        log(f"configs[4] leg done: {2048 / dt:.0f} clips/s")
        return {"workload": "synthetic [B=2048, T=27, J=17] inference only, the whole global batch in ONE pass on 1 GPU (8 GPUs: 256 per rank, no communication)",
                "clips_per_sec": 2048 / dt, "ms_per_step": dt * 1e3, "steps": 5,
                "model_mfma_frac": 2048 / dt * FLOP_PER_CLIP_FWD[27] / (PEAK_BF16_TFLOPS * 1e12)}

    for name, leg in (("configs[3]", leg_train81), ("configs[4]", leg_eval2048)):
        try:                                         # a failed leg (out of memory on a smaller part, a kernel error) must not cost the headline its line
            legs[name] = leg()
        except Exception as e:
            legs[name] = {"error": repr(e)}
        torch.cuda.empty_cache()

    cmd = [sys.executable, os.path.abspath(__file__), "--force-dp", "--global-batch", "32", "--det-conf", "--steps", "20", "--warmup", "5",
           "--no-cpu-baseline", "--no-fp32", "--no-kernel-roofline", "--no-extra-configs"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=300)
        j = json.loads(r.stdout.decode().strip().splitlines()[-1])
        legs["configs[2] per-rank shape"] = {"workload": j["config"]["workload"] + " -- ONE of the 8 ranks, data-parallel code path with single-rank RCCL",
                                             "clips_per_sec": j["value"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "model_mfma_frac": j["model_mfma_frac"]}
        log(f"configs[2] per-rank leg done: {j['value']:.0f} clips/s")
    except Exception as e:                       # a failed leg must not cost the headline its line
        legs["configs[2] per-rank shape"] = {"error": repr(e)}
    return legs


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))       # BEFORE anything loads the HIP runtime
    _need_torch()
    # stdout carries exactly ONE line (the JSON): everything else any library prints there (RCCL's version banner on communicator
    # creation, for instance) is sent to stderr by pointing file descriptor 1 at it for the duration of the run.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): reporting n_gpus = {world}", file=sys.stderr, flush=True)
    if args.launch_only:
        sys.exit(launch_only(args, rank, world, local, real_stdout))
    Tn = args.frames
    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} does not divide over {world} ranks")
        args.batch = args.global_batch // world
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1 or args.force_dp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))      # BEFORE the first model: RCCL's streams exist before the engine's
    import kasportsformer_amd as K

    torch.manual_seed(114514)                # configs/*.yaml:17
    model = K.KASportsFormer(n_layers=LAYERS, num_heads=8, n_frames=Tn, compute_dtype="bf16").cuda().train()
    model.attach_param_grads = False
    opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
    dp = None
    if (world > 1 or args.force_dp) and not args.eval_only:
        dp = K.DataParallel(model, overlap=os.environ.get("KASF_DP_OVERLAP", "1") != "0", optimizer=opt, grad_dtype=args.grad_dtype)     # sets opt.grad_scale = 1 / world
        if os.environ.get("KASF_DP_SKIP_ALLREDUCE") == "1":          # diagnosis only: process group alive, no collective in the step
            model.grad_stage_hook = None
            dp.finish_gradients = lambda *a: None
    x, y = K.synthetic_clips(args.batch, Tn, seed=1234 + rank, **({"res": (1920, 1080), "det_conf": True} if args.det_conf else {}))
    x, y = x.cuda(), y.cuda()

    def train_step():
        opt.zero_grad()
        loss, parts = K.loss3(model(x), y)
        loss.backward()
        if dp is not None:
            dp.finish_gradients()
        opt.step()
        return parts

    def eval_step():                         # configs[4]: inference only, no flip, no communication
        with torch.no_grad():
            return model(x)

    if args.eval_only:
        model.eval()
    step = eval_step if args.eval_only else train_step

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log(f"model on GPU; ws(B={args.batch}) = {model._lib.kasf_workspace_bytes(model._device_handle(), args.batch, 0 if args.eval_only else 1) / 1e9:.1f} GB")
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        parts = step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0        # this rank's own K steps (reported per rank); the headline clock stops after the barrier below
    fence()
    dt = time.perf_counter() - t0
    # a step that produced NaN / inf is not a measurement (checked after the clock has stopped: .item() synchronises)
    if args.eval_only:
        if not bool(torch.isfinite(parts).all()):
            raise RuntimeError("bench: non-finite predictions after the timed passes")
    elif not all(math.isfinite(float(v)) for v in parts) or not bool(torch.isfinite(model._flat[:model.n_live]).all()):
        raise RuntimeError(f"bench: non-finite loss terms or parameters after the timed steps: {[float(v) for v in parts]}")
    per_rank = [args.batch * args.steps / dt_own]
    if world > 1:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        own = [torch.zeros(1, device="cuda", dtype=torch.float64) for _ in range(world)]
        dist.all_gather(own, torch.tensor([dt_own], device="cuda", dtype=torch.float64))
        per_rank = [args.batch * args.steps / float(t.item()) for t in own]
    loss_val = None if args.eval_only else float(parts[0].item())
    log(f"timed {args.steps} steps in {dt:.3f} s")

    # BASELINE's metric is "train+eval": the forward-only rate of the same model and batch (evaluation mode, with and without flip-TTA) is
    # reported next to the headline; it is NOT part of `value` and runs after the timed region.
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            model(x)
        fence()
        t0 = time.perf_counter()
        for _ in range(5):
            model(x)
        fence()
        dt_eval = (time.perf_counter() - t0) / 5
        K.predict_flip_tta(model, x)
        fence()
        t0 = time.perf_counter()
        for _ in range(3):
            K.predict_flip_tta(model, x)
        fence()
        dt_tta = (time.perf_counter() - t0) / 3
    if world > 1:
        te = torch.tensor([dt_eval, dt_tta], device="cuda", dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        dt_eval, dt_tta = float(te[0]), float(te[1])
    model.train()

    if rank == 0:
        clips = args.batch * world * args.steps
        value = clips / dt
        flop_per_clip = (1 if args.eval_only else 3) * FLOP_PER_CLIP_FWD.get(Tn, 0.0)
        headline = args.batch == BATCH_PER_GPU and Tn == T and not args.eval_only and not strong
        out = {
            "metric": ("pose-clips/sec (27f x 17j) inference: forward only" if args.eval_only else
                       f"pose-clips/sec ({Tn}f x 17j) training step: fwd + 3-term loss + bwd + AdamW"), "value": value, "unit": "pose-clips/sec",
            "n_gpus": world, "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload_name(args, world, strong), "n_layers": LAYERS,
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "n_frames": Tn, "tokens_per_step_per_gpu": args.batch * Tn * 17,
                       "input": "detector confidence ~U(0,1), 1920x1080" if args.det_conf else "ground-truth 2-D (confidence 1), 1312x1216",
                       "parallelism": f"dp{world}" if world > 1 else "single", "init": "reference default init, seed 114514",
                       **({"gradient_allreduce": "bf16 on the wire, fp32 master gradient"} if (dp is not None and args.grad_dtype == "bf16") else {})},
            "per_rank_clips_per_sec": per_rank,
            "final_loss": loss_val,
            "eval": {"clips_per_sec": args.batch * world / dt_eval, "clips_per_sec_flip_tta": args.batch * world / dt_tta,
                     "note": "forward only, same model and batch per GPU, evaluation mode; not part of value"},
            "parity": parity_summary(),
            "model_mfma_frac": value / world * flop_per_clip / (PEAK_BF16_TFLOPS * 1e12),
        }
        if not args.no_kernel_roofline:
            ks = kernel_rooflines(args.batch * Tn * 17)
            log("kernel rooflines done")
            dom = max(ks, key=lambda k: ks[k]["seconds"])
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ks[dom]["achieved_tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": ks[dom]["achieved_tflops"] / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(dom, args.batch * Tn * 17),
                               "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)",
                               "traffic_source": "committed file profiles/" + os.path.basename(TRAFFIC_FILE) + " (separate rocprofv3 --pmc passes over the same launches: "
                                                 "tools/mlp_bench.py), not measured in this run",
                               "algorithmic_bytes_per_launch": 3 * 128 * 2 * args.batch * Tn * 17 if dom.startswith("k_mlp_bwd") else None,
                               "launch_ms": ks[dom]["seconds"] * 1e3, "algorithmic_flop_per_launch": ks[dom]["algorithmic_flop"]}
            if dom in TRAFFIC_PARTS and headline:
                t_in = in_step_duration(list(TRAFFIC_PARTS[dom]))
                if t_in:                 # the same launches inside whole training steps (committed trace), where they share the chip with two other streams
                    out["roofline"]["in_step"] = {"launch_ms": t_in * 1e3, "achieved": ks[dom]["algorithmic_flop"] / t_in / 1e12,
                                                  "frac": ks[dom]["algorithmic_flop"] / t_in / 1e12 / PEAK_BF16_TFLOPS,
                                                  "source": "committed file profiles/" + os.path.basename(IN_STEP_STATS) + " (rocprofv3 --kernel-trace --stats -- python3 tools/train_once.py 27 256)"}
                half = args.batch * Tn * 17 < MLP_HALF_CHIP_BELOW
                out["roofline"]["note"] = ("event-timed in a hot loop through the operator entry points: the kernels on all 256 CUs.  " +
                                           ("Inside the engine's step each of these launches takes 128 CUs and two branches' launches run side by side "
                                            "(csrc/kernels.h kasf_narrow_grid; profiles/r4_grid_width_probe.txt: +3 % step throughput at this batch, +8 % at 128): "
                                            "in_step / in_step_single_stream are durations of those half-chip launches, frac is of the WHOLE chip's peak." if half else ""))
                t_iso = in_step_duration(list(TRAFFIC_PARTS[dom]), ISOLATED_STATS)
                if t_iso:                # ... and inside whole training steps run on ONE stream: nothing else in flight, between the step's memory-bound launches instead of in a hot loop of itself
                    fr = ks[dom]["algorithmic_flop"] / t_iso / 1e12 / PEAK_BF16_TFLOPS
                    out["roofline"]["in_step_single_stream"] = {"launch_ms": t_iso * 1e3, "achieved": ks[dom]["algorithmic_flop"] / t_iso / 1e12, "frac": fr,
                                                                "cus": 128 if half else 256, "frac_of_its_cus": fr * (2 if half else 1),
                                                                "source": "committed file profiles/" + os.path.basename(ISOLATED_STATS) + " (KASF_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -- python3 tools/train_once.py 27 256): the engine's own launches, one at a time"}
                t_full = in_step_duration(list(TRAFFIC_PARTS[dom]), ISOLATED_FULL_STATS)
                if t_full:               # the one-stream step with full-width launches: what rounds 1-3 reported under this name
                    out["roofline"]["in_step_single_stream_full_width"] = {"launch_ms": t_full * 1e3, "achieved": ks[dom]["algorithmic_flop"] / t_full / 1e12,
                                                                           "frac": ks[dom]["algorithmic_flop"] / t_full / 1e12 / PEAK_BF16_TFLOPS, "cus": 256,
                                                                           "source": "committed file profiles/" + os.path.basename(ISOLATED_FULL_STATS) + " (the same with KASF_NARROW_PCTS=100,100,100,100,100,100,100); "
                                                                                     "the event-timed figure above is a back-to-back loop of this chain alone, where the same two kernels take longer per launch"}
            if headline and fresh(SQ_FILE):            # matrix-pipe busy fraction and LDS bank conflicts of the dominant kernel's launches
                sq = json.load(open(SQ_FILE))
                part = {k: v for k, v in sq.items() if isinstance(v, dict) and any(k.startswith(p) for p in TRAFFIC_PARTS.get(dom, (dom,)))}
                if part:
                    out["roofline"]["mfma_busy"] = {k: v.get("mfma_busy") for k, v in part.items()}
                    out["roofline"]["lds_bank_conflict_frac"] = {k: v.get("lds_bank_conflict_frac") for k, v in part.items()}
                    out["roofline"]["mfma_busy_source"] = ("committed file profiles/" + os.path.basename(SQ_FILE) + " (six rocprofv3 --pmc passes over tools/mlp_bench.py, tools/pmc_kernel.sh): "
                                                           "SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_CYCLES), the share of the launch during which a SIMD's matrix pipe is busy, recomputed Z included")
            if headline and fresh(STEP_TRAFFIC_FILE):
                out["step_hbm_GB"] = json.load(open(STEP_TRAFFIC_FILE))["hbm_GB_per_step"]
                out["step_hbm_GB_source"] = "committed file profiles/" + os.path.basename(STEP_TRAFFIC_FILE)
            out["kernels"] = {k: {"ms": v["seconds"] * 1e3, "tflops": v["achieved_tflops"]} for k, v in ks.items()}
        if world == 1:
            del model, opt, x, y
            torch.cuda.empty_cache()
            if headline and not args.force_dp and not args.no_extra_configs:
                out["configs"] = extra_config_legs(log)
            if not args.no_fp32 and not args.eval_only and Tn == T:
                out["fp32_mode"] = fp32_mode_rate(args.batch)
                log("fp32 parity-mode leg done")
            if not args.no_cpu_baseline and not args.eval_only and Tn == T:
                out["cpu_baseline"] = cpu_baseline()
        out["profile_stale"] = bool(STALE)       # a quoted committed file was measured on other sources than this library's: its figures are omitted above
        if STALE:
            out["profile_stale_files"] = sorted(STALE)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or args.force_dp:
        dist.barrier()                       # rank 0 may still be in its kernel micro-benchmark: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
