"""Where INTEGRATION path A (stock torch.optim.AdamW over the 2,611 parameter views, per-parameter .grad) spends its step: host and device time per phase.
    python tools/dropin_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kasportsformer_amd as K

torch.manual_seed(114514)
model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
x, y = (t.cuda() for t in K.synthetic_clips(256, 27, seed=1234))
for name, make in (("AdamW default", lambda: torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.01)),
                   ("AdamW fused=True", lambda: torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.01, fused=True))):
    opt = make()
    acc = {}
    def phase(tag, fn):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        a = acc.setdefault(tag, [0.0, 0.0]); a[0] += t1 - t0; a[1] += t2 - t0
        return r
    N = 6
    for it in range(N + 2):
        if it == 2:
            acc.clear()
        pred = phase("forward", lambda: model(x))
        phase("zero_grad", lambda: opt.zero_grad())
        loss = phase("loss", lambda: K.loss3(pred, y)[0])
        phase("backward (+ .grad views)", lambda: loss.backward())
        phase("optimizer.step", lambda: opt.step())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        pred = model(x); opt.zero_grad(); loss, _ = K.loss3(pred, y); loss.backward(); opt.step()
    torch.cuda.synchronize(); b2b = (time.perf_counter() - t0) / 5
    print(json.dumps({"optimizer": name, "ms_per_step_back_to_back": round(b2b * 1e3, 2),
                      "phases_ms_host_then_host_plus_device": {k: [round(v[0] / N * 1e3, 2), round(v[1] / N * 1e3, 2)] for k, v in acc.items()}}), flush=True)
