import sys, torch
sys.path.insert(0, '.')
from tests.gpu_util import DT, ptr, stream
from kasportsformer_amd import _lib
from oracle.kasf_oracle import attention_core, _heads
lib = _lib.load()
for cd in ("fp32", "bf16"):
  for mode, T in ((0, 27), (1, 27), (1, 9), (1, 81)):
    B = 2
    g = torch.Generator().manual_seed(14)
    qkv = torch.randn(B, T, 17, 384, generator=g); do = torch.randn(B, T, 17, 128, generator=g)
    qd, dod = qkv.cuda().to(DT[cd][1]).contiguous(), do.cuda().to(DT[cd][1]).contiguous()
    dqkv = torch.full_like(qd, float('nan'))
    es = qd.element_size(); base = qd.data_ptr(); db = dqkv.data_ptr()
    _lib.check(lib.kasf_op_attention_bwd(DT[cd][0], base, 384, base+128*es, base+256*es, 384, ptr(dod), db, 384, db+128*es, db+256*es, 384, B, T, mode, stream()))
    torch.cuda.synchronize()
    qr = qd.float().cpu().requires_grad_(True)
    q, k, v = _heads(qr, 3, 8)
    ref = attention_core(q, k, v, "spatial" if mode == 0 else "temporal", 0.25)
    ref.backward(dod.float().cpu())
    got = dqkv.float().cpu(); r = qr.grad
    err = (got - r).abs()
    print(f"{cd} mode{mode} T{T}: nan={int(torch.isnan(got).sum())}", end=" ")
    for nm, sl in (("dq", slice(0,128)), ("dk", slice(128,256)), ("dv", slice(256,384))):
        e = err[..., sl]
        print(f"{nm}={float(torch.nan_to_num(e, nan=9.0).max()):.3e}", end=" ")
    e = torch.nan_to_num(err, nan=9.0)
    bad = (e > 1e-2 * float(r.abs().max()))
    if bad.any():
        idx = bad.nonzero()
        print("\n   bad count", int(bad.sum()), "first", idx[:5].tolist(), "per-head bad:", [int(bad[..., c*16:(c+1)*16].sum()) for c in range(24)],
              "\n   per-joint bad:", [int(bad[:, :, j].sum()) for j in range(17)], "per-frame bad:", [int(bad[:, t].sum()) for t in range(T)][:30])
    else:
        print("OK")
