"""Which tensors carry the bf16 gradient deviation of a 26-layer sample?  HIP bf16 vs the CPU oracle (same neighbour decisions), grouped by parameter family
and by layer: |ref|, |diff| / |ref| and each group's share of the squared deviation.   python tools/parity_breakdown.py [seed=7] [salt=2002]"""
import os, sys, re, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import kasf_oracle as O
from tests.gpu_util import make_pair, forced_adjacency

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
salt = int(sys.argv[2]) if len(sys.argv) > 2 else 2002
cd = sys.argv[3] if len(sys.argv) > 3 else "bf16"
oracle, model = make_pair(26, 27, cd, salt=salt)
x, y = O.synthetic_clips(2, 27, seed=seed)
oracle.train(); model.train()
with forced_adjacency(model, x) as fa:
    ref = oracle(x)
    O.loss_total(ref, y)[0].backward()
pred = model(x.cuda())
O.loss_total(pred, y.cuda())[0].backward()
torch.cuda.synchronize()
fam, lay = collections.defaultdict(lambda: [0.0, 0.0, 0.0]), collections.defaultdict(lambda: [0.0, 0.0, 0.0])
tot = [0.0, 0.0, 0.0]
for (n, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
    if q.grad is None:
        continue
    g, r = p.grad.detach().double().cpu(), q.grad.double()
    d = (float((g * r).sum()), float((g * g).sum()), float((r * r).sum()), float(((g - r) ** 2).sum()))
    m = re.match(r"layers_with_bone\.(\d+)\.(.*)", n)
    f = re.sub(r"\.\d+\.", ".N.", m.group(2)) if m else re.sub(r"\.\d+\.", ".N.", n)
    for dst in (fam[f], lay[int(m.group(1)) if m else -1]):
        dst[0] += d[2]; dst[1] += d[3]; dst[2] += d[0]
    tot[0] += d[0]; tot[1] += d[1]; tot[2] += d[2]
cos = tot[0] / (tot[1] ** 0.5 * tot[2] ** 0.5)
err2 = sum(v[1] for v in fam.values())
print(f"sample (seed {seed}, salt {salt}), {cd}: gradient cosine {cos:.4f}; |g|/|r| = {(tot[1] / tot[2]) ** 0.5:.4f}; forward err {float((pred.detach().cpu() - ref.detach()).abs().max()) / max(1.0, float(ref.abs().max())):.3e}")
print("by family (sorted by share of squared deviation):")
for f, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {f:55s} |ref| {v[0] ** 0.5:10.3e}  |diff|/|ref| {(v[1] / max(v[0], 1e-300)) ** 0.5:7.3f}  share {100 * v[1] / err2:5.1f} %")
print("by layer (-1 = top level):")
for l, v in sorted(lay.items()):
    print(f"  layer {l:3d} |ref| {v[0] ** 0.5:10.3e}  |diff|/|ref| {(v[1] / max(v[0], 1e-300)) ** 0.5:7.3f}  share {100 * v[1] / err2:5.1f} %")
