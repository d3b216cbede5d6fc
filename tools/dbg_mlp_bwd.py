import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from kasportsformer_amd import _lib
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
bf = torch.bfloat16
x = torch.randn(M, 128); g = torch.randn(M, 128)
W1 = torch.randn(512, 128) * 0.09; W2 = torch.randn(128, 512) * 0.045; b1 = torch.randn(512) * 0.1; b2 = torch.randn(128) * 0.1
ls = torch.rand(128) + 0.5; gam = torch.rand(128) + 0.5; bet = torch.randn(128) * 0.1
d = lambda t: t.cuda().to(bf).contiguous(); f = lambda t: t.cuda().float().contiguous()
p = lambda t: C.c_void_p(t.data_ptr()); st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
xd, gd, w1, w2 = d(x), d(g), d(W1), d(W2)
w2ts, w1t = d((ls[:, None] * W2).T), d(W1.T)
out, xn = torch.empty_like(xd), torch.empty_like(xd)
lib.kasf_op_mlp_fwd(1, p(xd), p(f(gam)), p(f(bet)), p(w1), p(f(b1)), p(w2), p(f(b2)), p(f(ls)), p(out), M, p(xn), st())
dap = torch.zeros(4 * M * 128, device="cuda", dtype=bf); part = torch.zeros(2 * 64 * 65536 + 2048, device="cuda")
z = lambda *s: torch.zeros(*s, device="cuda")
dW1, dW2, db1, gsum, dg, db, gin = z(512, 128), z(128, 512), z(512), z(128), z(128), z(128), torch.empty_like(xd)
_lib.check(lib.kasf_op_mlp_bwd_fused(p(xd), p(xn), p(gd), p(f(gam)), p(w1), p(f(b1)), p(w2ts), p(w1t), p(dap), p(part), p(dW1), p(dW2), p(db1), p(gsum), p(gin), p(dg), p(db), M, st()))
torch.cuda.synchronize()
xr = xd.float().cpu().requires_grad_(True)
h = F.gelu(F.layer_norm(xr, (128,), gam, bet) @ w1.float().cpu().T + b1)
o = xr + ls * (h @ w2.float().cpu().T + b2)
o.backward(gd.float().cpu())
err = (gin.float().cpu() - xr.grad).abs()
print("max err", float(err.max()))
print("per-channel max err (x100):", (err.max(0)[0] * 100).int().tolist())
print("per-row max err (x100):", (err.max(1)[0] * 100).int().tolist()[:40])
# the partials themselves: sum of 4 slabs vs reference dA
dA = dap.view(4, M, 128).float().sum(0).cpu()
xn_ref = F.layer_norm(xr, (128,), gam, bet).detach().requires_grad_(True)
h2 = F.gelu(xn_ref @ w1.float().cpu().T + b1); o2 = (ls * (h2 @ w2.float().cpu().T + b2)); o2.backward(gd.float().cpu())
e2 = (dA - xn_ref.grad).abs()
print("partials: per-channel max err (x100):", (e2.max(0)[0] * 100).int().tolist())
