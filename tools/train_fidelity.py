"""Training fidelity of the arithmetic modes at FULL depth on a LEARNABLE task.

    python tools/train_fidelity.py [steps=1000] [modes=fp32,bf16,bf16@1,bf16@2] > profiles/r3_train_fidelity.json

The shipped 26-layer model, batch 256, T = 27, reference default init under the yaml seed, the reference's optimiser / warm-up
(train_and_evaluate_sp.py:270-272,325-329 with an "epoch" = one pass over the 16-batch clip pool), labels = ``teacher_labels`` (a fixed seeded
map of the 2-D pose: MPJPE falls by an order of magnitude, so a gap between modes would show).  The validated fp32 HIP mode (4e-6 of the CPU
oracle over 26 layers, tests/test_gpu_model.py) is the oracle's proxy: the CPU oracle itself needs ~50 s per step at this size.
GPU only; every mode trains on the same clips in the same order from the same weights.  Gradients are bit-reproducible since round 3, so a mode listed twice gives
the same run twice; `bf16@3` perturbs the initial weights by 1e-6 relative noise (seed 3): an independent sample of the same training.  The test MPJPE of a SINGLE
checkpoint moves by several mm from one checkpoint to the next in every mode (lr 5e-4, BatchNorm running statistics lagging): the statistic to compare modes on is the
mean over the last checkpoints (`tail`: every 10 steps over the last 100), with its spread, next to the training loss of the last 50 steps.
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kasportsformer_amd as K

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
MODES = (sys.argv[2] if len(sys.argv) > 2 else "fp32,bf16,bf16@1,bf16@2").split(",")
L, T, B, POOL, NTEST = int(os.environ.get("FID_LAYERS", 26)), int(os.environ.get("FID_T", 27)), int(os.environ.get("FID_BATCH", 256)), 16, 512
LR, WARM_EPOCHS = 5e-4, 10
EVAL_EVERY = max(1, STEPS // 10)

xs, ys = K.teacher_clips(B * POOL, T, seed=77)
xt, yt = K.teacher_clips(NTEST, T, seed=78)
label_scaled, factor, res, actions = K.synthetic_test_extras(yt, seed=79, noise_mm=2.0)
xs, ys, xt = xs.cuda(), ys.cuda(), xt.cuda()
torch.manual_seed(114514)
init = {k: v.clone() for k, v in K.KASportsFormer(n_layers=L, num_heads=8, n_frames=T, compute_dtype="fp32").state_dict().items()}


def evaluate(model):
    r = K.evaluate_one_epoch(model, [(xt[i:i + 256], label_scaled[i:i + 256], factor[i:i + 256], actions[i:i + 256], res[i:i + 256]) for i in range(0, NTEST, 256)], flip=True)
    model.train()
    return {"mpjpe_mm": r["mpjpe"], "p_mpjpe_mm": r["p_mpjpe"], "accel_mm": r["acceleration_error"]}


def run(mode):
    cd, _, pseed = mode.partition("@")
    model = K.KASportsFormer(n_layers=L, num_heads=8, n_frames=T, compute_dtype=cd)
    sd = init
    if pseed:
        g = torch.Generator().manual_seed(int(pseed))
        sd = {k: (v * (1.0 + 1e-6 * torch.randn(v.shape, generator=g)) if v.is_floating_point() else v) for k, v in init.items()}
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    model.attach_param_grads = False
    opt = K.FusedAdamW(model, lr=LR, weight_decay=0.01)
    losses = torch.zeros(STEPS, 4, device="cuda")
    evals = {0: evaluate(model)}
    tail = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(STEPS):
        K.apply_warmup(opt, s // POOL, LR, WARM_EPOCHS)
        o = (s % POOL) * B
        pred = model(xs[o:o + B])
        opt.zero_grad()
        loss, parts = K.loss3(pred, ys[o:o + B])
        losses[s] = parts
        loss.backward()
        opt.step()
        if s + 1 > STEPS - 100 and (s + 1) % 10 == 0:
            tail.append(evaluate(model))
        if (s + 1) % EVAL_EVERY == 0 or s + 1 == STEPS:
            evals[s + 1] = tail[-1] if tail and (s + 1) % 10 == 0 else evaluate(model)
            print(f"[{mode}] step {s + 1}: loss {float(parts[0]):.4f}  MPJPE {evals[s + 1]['mpjpe_mm']:.2f} mm", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lc = losses.cpu()
    every = max(1, STEPS // 100)
    tm = [e["mpjpe_mm"] for e in tail]
    tp = [e["p_mpjpe_mm"] for e in tail]
    return {"compute_dtype": mode, "seconds_incl_eval": round(dt, 1), "eval": {str(k): v for k, v in evals.items()},
            "tail": {"checkpoints": len(tail), "mpjpe_mm_mean": sum(tm) / max(1, len(tm)), "mpjpe_mm_min": min(tm), "mpjpe_mm_max": max(tm),
                     "p_mpjpe_mm_mean": sum(tp) / max(1, len(tp)), "p_mpjpe_mm_min": min(tp), "p_mpjpe_mm_max": max(tp)},
            "loss_total_every_%d_steps" % every: [round(float(v), 5) for v in lc[::every, 0]],
            "loss_mean_last_50_steps": {n: float(lc[-50:, i].mean()) for i, n in enumerate(("total", "mpjpe", "n_mpjpe", "velocity"))}}


out = {"steps": STEPS, "layers": L, "batch": B, "n_frames": T, "task": "teacher_labels(seed 2024) on synthetic_clips(seed 77), 16-batch pool; test 512 clips (seed 78), label noise 2 mm",
       "init": "reference default init, seed 114514", "optimizer": "AdamW lr 5e-4 wd 0.01, warm-up lr/100 -> lr over 10 pool passes", "runs": []}
for cd in MODES:
    out["runs"].append(run(cd))
ref = next((r for r in out["runs"] if r["compute_dtype"] == "fp32"), None)
if ref is not None:
    final = str(STEPS)
    out["final_mpjpe_gap_vs_fp32_mm"] = [{"compute_dtype": r["compute_dtype"], "mpjpe_mm": r["eval"][final]["mpjpe_mm"],
                                          "gap_mm": r["eval"][final]["mpjpe_mm"] - ref["eval"][final]["mpjpe_mm"],
                                          "tail_mean_gap_mm": r["tail"]["mpjpe_mm_mean"] - ref["tail"]["mpjpe_mm_mean"],
                                          "tail_p_mpjpe_gap_mm": r["tail"]["p_mpjpe_mm_mean"] - ref["tail"]["p_mpjpe_mm_mean"]} for r in out["runs"]]
print(json.dumps(out, indent=1))
