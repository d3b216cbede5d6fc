"""Compare scratch tensors of the graph branch between two identical backward passes (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kasportsformer_amd as K
from kasportsformer_amd import _lib
from oracle import kasf_oracle as O
from tests.gpu_util import make_pair, ws_tensor

_, model = make_pair(1, 27, "bf16")
x, y = (t.cuda() for t in O.synthetic_clips(16, 27, seed=91))
model.train()
buffers, nbt = model._flat_buffers.clone(), model._nbt.clone()
snaps = []
names = ["rep", "L0.graph_temporal.bn_bwd_stats", "L0.graph_temporal.bn_coef", "L0.graph_temporal.xn", "L0.graph_temporal.y", "L0.graph_temporal.adj_mask", "mlp_h.graph", "gcn_duv.graph", "gcn_r.graph", "g_tmp1.graph", "g_tmp2.graph", "g_in.graph", "L0.graph_spatial.xn", "L0.graph_spatial.y", "L0.graph_spatial.bn_bwd_stats",
         "L0.graph_spatial.bn_coef", "g_att", "g_graph", "g_bonebr", "wgrad_partials.graph", "column_partials.graph"]
for rep in range(2):
    model._flat_buffers.copy_(buffers); model._nbt.copy_(nbt)
    model.zero_grad()
    with torch.enable_grad():
        out, ws, flags = model._launch_forward(x, False, keep=True)
    dp = torch.empty_like(out); scratch = torch.empty(4 + 4 * 16, device="cuda")
    import ctypes as C
    _lib.check(model._lib.kasf_loss3(out.data_ptr(), y.data_ptr(), dp.data_ptr(), scratch.data_ptr(), scratch.numel(), 16, 27, 0.5, 20.0, 1.0, model._stream()))
    model._launch_backward(ws, dp, 16, flags)
    torch.cuda.synchronize()
    snaps.append({n: ws_tensor(model, ws, 16, n, flags=_lib.FLAG_TRAIN).clone() for n in names})
for n in names:
    a, b = snaps[0][n], snaps[1][n]
    if a.dtype == torch.int32: a, b = a.float(), b.float()
    neq = (a != b) & ~(a.isnan() & b.isnan())
    print(f"{n:34s} {int(neq.sum()):9d} of {a.numel():9d} differ", end="")
    if n in ("gcn_duv.graph", "rep"):
        d = neq.view(-1, 256)
        print(f"   U-part {int(d[:, :128].sum())}, V-part {int(d[:, 128:].sum())}; tokens {d.any(1).nonzero().flatten()[:10].tolist()}", end="")
    print()
# magnitude of the duv differences and which pass deviates from a host recomputation of dU = sc (r - c1 - (y - mean) rstd c2)
a, b = snaps[0]["gcn_duv.graph"].float().view(-1, 256), snaps[1]["gcn_duv.graph"].float().view(-1, 256)
d = (a - b).abs()
print("duv max abs diff", float(d.max()), "scale", float(a.abs().max()), "U", float(d[:, :128].max()), "V", float(d[:, 128:].max()))
idx = (d[:, :128] > 0).nonzero()[:12]
r = snaps[0]["gcn_r.graph"].float().view(-1, 128)
yv = snaps[0]["L0.graph_spatial.y"].float().view(-1, 128)
coef = snaps[0]["L0.graph_spatial.bn_coef"].view(-1, 8)
bst = snaps[0]["L0.graph_spatial.bn_bwd_stats"].view(4, 512)
tot = bst.sum(0)
count = 16 * 27 * 128
for t, c in idx.tolist():
    n = t % 17
    sc, mean, rstd = float(coef[n, 0]), float(coef[n, 2]), float(coef[n, 3])
    c1, c2 = float(torch.tensor(float(tot[2 * n]) / count, dtype=torch.float32)), float(torch.tensor(float(tot[2 * n + 1]) / count, dtype=torch.float32))
    want = sc * (float(r[t, c]) - c1 - (float(yv[t, c]) - mean) * rstd * c2)
    print(f"tok {t} ch {c}: pass0 {float(a[t, c]):.6e} pass1 {float(b[t, c]):.6e} host {want:.6e}")
