"""Grid widths of the persistent launches (kernels.h: kasf_narrow_grid): training throughput at several batch sizes per setting.
    [PROBE_BATCHES=32,64,256] python tools/smallb_probe.py [pcts ...]      (one process per setting: the knobs are read once per process)"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(HERE))
    import torch, time
    import kasportsformer_amd as K
    out = {}
    for B in [int(b) for b in os.environ.get("PROBE_BATCHES", "32,64,256").split(",")]:
        torch.manual_seed(1)
        model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
        model.attach_param_grads = False
        opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
        x, y = (t.cuda() for t in K.synthetic_clips(B, 27, seed=1234))
        def step():
            opt.zero_grad(); K.loss3(model(x), y)[0].backward(); opt.step()
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 12 if B < 256 else 6
        for _ in range(n): step()
        torch.cuda.synchronize(); out[B] = round(B * n / (time.perf_counter() - t0), 1)
        del model, opt
    print(json.dumps(out))
    sys.exit(0)
SETTINGS = sys.argv[1:] or ["100,100,100,100,100,100,100", "50,50,100,100,100,100,100", "50,50,50,50,100,100,100", "33,33,100,100,100,100,100"]
# each: seven comma-separated percentages of the full grid (mlp fwd, mlp bwd, dgrad, linear, attn fwd, attn bwd, wgrad jobs), applied at every batch size
# (per-branch widths were tried too -- the last MLP of each branch, or the bone branch, at full width; a narrower graph branch: all worse than 50 % everywhere,
#  profiles/r4_grid_width_probe.txt)
for pcts in SETTINGS:
    env = dict(os.environ, KASF_NARROW_PCTS=pcts, KASF_NARROW_BELOW=os.environ.get("KASF_NARROW_BELOW", "1000000000"))
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    print(f"pcts {pcts:30s} clips/s by batch: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
