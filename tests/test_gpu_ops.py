"""GPU parity of the single-operator C-ABI entry points (include/kasf.h, kasf_op_*) against plain fp32
torch math of the same op (the per-module oracle).  fp32 mode must hold 1e-3 (north_star tolerance,
observed ~1e-6); bf16 mode is checked against the same math on bf16-rounded operands."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import DT, ptr, rel_err, stream

pytestmark = pytest.mark.gpu
TOL = {"fp32": 1e-4, "bf16": 3e-2}


@pytest.fixture(scope="module")
def lib():
    from kasportsformer_amd import _lib
    return _lib.load()


def _rand(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


_KEEP = []


@pytest.fixture(autouse=True)
def _keepalive():
    """Device copies made by _f32()/_dev() must outlive the (asynchronous) kernel launches."""
    _KEEP.clear()
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def _dev(t, cd):
    d = t.cuda().to(DT[cd][1]).contiguous()
    _KEEP.append(d)
    return d


def _f32(t):
    d = t.float().cuda().contiguous()
    _KEEP.append(d)
    return d


def _back(t):
    return t.float().cpu()


def _ln(x, g, b):
    return F.layer_norm(x, (128,), g, b, 1e-5)


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
@pytest.mark.parametrize("use_ln,act,N", [(True, 0, 384), (False, 0, 128), (True, 1, 512)])
@pytest.mark.parametrize("M", [300, 40003])       # 40003: five 32-token tiles per persistent workgroup (3-slot ring reuse) + a ragged tail
def test_linear(lib, cd, use_ln, act, N, M):
    from kasportsformer_amd import _lib
    a, w, bias = _rand(M, 128, seed=1), _rand(N, 128, seed=2, scale=1 / math.sqrt(128)), _rand(N, seed=3, scale=0.1)
    g, b = torch.rand(128) + 0.5, _rand(128, seed=4, scale=0.1)
    ad, wd = _dev(a, cd), _dev(w, cd)
    y = torch.empty(M, N, device="cuda", dtype=DT[cd][1])
    xn = torch.empty(M, 128, device="cuda", dtype=DT[cd][1])
    _lib.check(lib.kasf_op_linear(DT[cd][0], ptr(ad), ptr(wd), ptr(_f32(bias)), ptr(y), M, N, ptr(_f32(g)) if use_ln else None,
                                  ptr(_f32(b)) if use_ln else None, ptr(xn) if use_ln else None, act, stream()))
    torch.cuda.synchronize()
    ar, wr = _back(ad), _back(wd)
    h = _ln(ar, g, b) if use_ln else ar
    ref = h @ wr.T + bias
    if act:
        ref = torch.tanh(ref)
    assert rel_err(_back(y), ref) < TOL[cd]
    if use_ln:
        assert rel_err(_back(xn), h) < TOL[cd]


def _mlp_params(seed=10):
    return dict(W1=_rand(512, 128, seed=seed, scale=1 / math.sqrt(128)), b1=_rand(512, seed=seed + 1, scale=0.1),
                W2=_rand(128, 512, seed=seed + 2, scale=1 / math.sqrt(512)), b2=_rand(128, seed=seed + 3, scale=0.1),
                ls=torch.rand(128) + 0.5, g=torch.rand(128) + 0.5, b=_rand(128, seed=seed + 4, scale=0.1))


def _mlp_ref(x, p):
    return x + p["ls"] * (F.gelu(_ln(x, p["g"], p["b"]) @ p["W1"].T + p["b1"]) @ p["W2"].T + p["b2"])


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
@pytest.mark.parametrize("M", [459, 1000, 50001])      # 50001: six 32-token tiles per persistent workgroup + a ragged tail (ring reuse, look-ahead waits)
def test_mlp_forward(lib, cd, M):
    from kasportsformer_amd import _lib
    p = _mlp_params()
    x = _rand(M, 128, seed=5)
    xd, w1 = _dev(x, cd), _dev(p["W1"], cd)
    w2 = p["W2"].to("cuda", torch.float16) if cd == "bf16" else _dev(p["W2"], cd)      # bf16 mode: GEMM2 runs on FP16 operands (kasf.h, ABI 7)
    out = torch.empty_like(xd)
    xn = torch.empty_like(xd) if cd == "bf16" else None          # training mode also stores LN(x) for the fused backward
    _lib.check(lib.kasf_op_mlp_fwd(DT[cd][0], ptr(xd), ptr(_f32(p["g"])), ptr(_f32(p["b"])), ptr(w1), ptr(_f32(p["b1"])), ptr(w2),
                                   ptr(_f32(p["b2"])), ptr(_f32(p["ls"])), ptr(out), M, ptr(xn), stream()))
    torch.cuda.synchronize()
    if xn is not None:
        assert rel_err(_back(xn), _ln(_back(xd), p["g"], p["b"])) < TOL[cd]
    pr = dict(p, W1=_back(w1), W2=_back(w2))
    assert rel_err(_back(out), _mlp_ref(_back(xd), pr)) < TOL[cd]


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
def test_mlp_backward_and_wgrad(lib, cd):
    from kasportsformer_amd import _lib
    M = 459
    p = _mlp_params(seed=20)
    x, gout = _rand(M, 128, seed=6), _rand(M, 128, seed=7)
    xd, gd, w1 = _dev(x, cd), _dev(gout, cd), _dev(p["W1"], cd)
    w2ts = _dev((p["ls"][:, None] * p["W2"]).T.contiguous(), cd)       # [512,128]
    w1t = _dev(p["W1"].T.contiguous(), cd)                              # [128,512]
    H = torch.empty(M, 512, device="cuda", dtype=DT[cd][1])
    dZ, gin = torch.empty_like(H), torch.empty_like(xd)
    dg, db = torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda")
    _lib.check(lib.kasf_op_mlp_bwd(DT[cd][0], ptr(xd), ptr(gd), ptr(_f32(p["g"])), ptr(_f32(p["b"])), ptr(w1), ptr(_f32(p["b1"])), ptr(w2ts),
                                   ptr(w1t), ptr(H), ptr(dZ), ptr(gin), ptr(dg), ptr(db), M, stream()))
    dW1, db1 = torch.zeros(512, 128, device="cuda"), torch.zeros(512, device="cuda")
    dW2, gsum = torch.zeros(128, 512, device="cuda"), torch.zeros(128, device="cuda")
    part = torch.empty(256 * 128 * 128, device="cuda")
    _lib.check(lib.kasf_op_wgrad(DT[cd][0], ptr(dZ), 512, ptr(xd), 128, ptr(_f32(p["g"])), ptr(_f32(p["b"])), ptr(dW1), ptr(db1), M, None, 0, stream()))
    _lib.check(lib.kasf_op_wgrad(DT[cd][0], ptr(gd), 128, ptr(H), 512, None, None, ptr(dW2), ptr(gsum), M, ptr(part), part.numel(), stream()))   # split-partials + reduce path
    torch.cuda.synchronize()
    # reference by autograd on the rounded operands
    xr = _back(xd).requires_grad_(True)
    pr = {k: (v.clone().requires_grad_(True) if k in ("W1", "b1", "W2", "g", "b") else v) for k, v in p.items()}
    pr["W1"] = _back(w1).requires_grad_(True)
    gr = _back(gd)
    z = _ln(xr, pr["g"], pr["b"]) @ pr["W1"].T + pr["b1"]
    h = F.gelu(z)
    out = xr + p["ls"] * (h @ pr["W2"].T + p["b2"])
    out.backward(gr)
    tol = TOL[cd]
    assert rel_err(_back(H), h) < tol
    assert rel_err(_back(gin), xr.grad) < tol
    assert rel_err(_back(dg), pr["g"].grad) < tol and rel_err(_back(db), pr["b"].grad) < tol
    assert rel_err(_back(dW1), pr["W1"].grad) < tol and rel_err(_back(db1), pr["b1"].grad) < tol
    # unscaled G2 = g^T H: dW2 = ls[:,None] * G2
    assert rel_err(p["ls"][:, None] * _back(dW2), pr["W2"].grad) < tol
    assert rel_err(_back(gsum), gr.sum(0)) < tol


@pytest.mark.parametrize("M", [459, 3000, 20011, 117504, 33])     # 20011: ten tiles per token range (steady-state ring slots and vmcnt accounting) + a ragged tail;
def test_mlp_backward_fused_bf16(lib, M):                          # 117504: the benchmark's launch (58 tiles per range); 33: two tiles, one of them a single row
    """k_mlp_bwd_s + k_lnbwd_sum4_fin (the bf16 engine path; here without a column sink: per-channel sums by fp32 atomics) against autograd.
    Launched three times on the same scratch with a different upstream gradient each time."""
    from kasportsformer_amd import _lib
    cd = "bf16"
    p = _mlp_params(seed=30)
    x, gout = _rand(M, 128, seed=31), _rand(M, 128, seed=32)
    xd, gd, w1 = _dev(x, cd), _dev(gout, cd), _dev(p["W1"], cd)
    w2ts = _dev((p["ls"][:, None] * p["W2"]).T.contiguous(), cd)
    w1t = _dev(p["W1"].T.contiguous(), cd)
    dap = torch.empty(4 * M * 128, device="cuda", dtype=torch.bfloat16)
    part = torch.empty(2 * 64 * 65536, device="cuda")
    gin = torch.empty_like(xd)
    z = lambda *s: torch.zeros(*s, device="cuda")
    dW1, dW2, db1, gsum, dg, db = z(512, 128), z(128, 512), z(512), z(128), z(128), z(128)
    out, xn = torch.empty_like(xd), torch.empty_like(xd)           # the forward pass leaves LN(x) behind for the backward kernel
    _lib.check(lib.kasf_op_mlp_fwd(1, ptr(xd), ptr(_f32(p["g"])), ptr(_f32(p["b"])), ptr(w1), ptr(_f32(p["b1"])), ptr(p["W2"].to("cuda", torch.float16)), ptr(_f32(p["b2"])),
                                   ptr(_f32(p["ls"])), ptr(out), M, ptr(xn), stream()))
    g_final = gd
    for rep in range(3):           # a different upstream gradient every time: anything stale from the previous launch would be off by a factor
        for t in (dW1, dW2, db1, gsum, dg, db):
            t.zero_()
        gin.fill_(float("nan"))
        gd = (g_final.float() * (0.25, -0.5, 1.0)[rep]).to(g_final.dtype)
        _lib.check(lib.kasf_op_mlp_bwd_fused(ptr(xd), ptr(xn), ptr(gd), ptr(_f32(p["g"])), ptr(w1), ptr(_f32(p["b1"])), ptr(w2ts), ptr(w1t),
                                             ptr(dap), ptr(part), ptr(dW1), ptr(dW2), ptr(db1), ptr(gsum), ptr(gin), ptr(dg), ptr(db), M, stream()))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(gin).all()), rep
    xr = _back(xd).requires_grad_(True)
    pr = {k: (v.clone().requires_grad_(True) if k in ("b1", "W2", "g", "b") else v) for k, v in p.items()}
    pr["W1"] = _back(w1).requires_grad_(True)
    gr = _back(gd)
    h = F.gelu(_ln(xr, pr["g"], pr["b"]) @ pr["W1"].T + pr["b1"])
    out = xr + p["ls"] * (h @ pr["W2"].T + p["b2"])
    out.backward(gr)
    tol = TOL[cd]
    assert rel_err(_back(gin), xr.grad) < tol
    assert rel_err(_back(dg), pr["g"].grad) < tol and rel_err(_back(db), pr["b"].grad) < tol
    assert rel_err(_back(dW1), pr["W1"].grad) < tol and rel_err(_back(db1), pr["b1"].grad) < tol
    assert rel_err(p["ls"][:, None] * _back(dW2), pr["W2"].grad) < tol
    assert rel_err(_back(gsum), gr.sum(0)) < tol


# (Kd, dxn_add, resid, accumulate, xn_out): the four combinations the engine uses (bf16: persistent k_dgrad_r) + the generic path (Kd=512 head)
DGRAD_CASES = [(384, False, True, False, True), (128, False, True, False, True), (256, False, False, True, True), (256, True, True, False, False),
               (512, True, True, False, False), (128, True, False, True, False)]


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
@pytest.mark.parametrize("M", [333, 8200, 40003])
@pytest.mark.parametrize("Kd,use_add,use_resid,accumulate,want_xn", DGRAD_CASES)
def test_dgrad_lnbwd(lib, cd, M, Kd, use_add, use_resid, accumulate, want_xn):
    from kasportsformer_amd import _lib
    x, dy, add, resid, prev = _rand(M, 128, seed=8), _rand(M, Kd, seed=9), _rand(M, 128, seed=10), _rand(M, 128, seed=11), _rand(M, 128, seed=14)
    W = _rand(Kd, 128, seed=12, scale=1 / math.sqrt(128))      # forward weight [out=Kd, in=128]
    g, b = torch.rand(128) + 0.5, _rand(128, seed=13, scale=0.1)
    xd, dyd, addd, rd = _dev(x, cd), _dev(dy, cd), _dev(add, cd), _dev(resid, cd)
    wt = _dev(W.T.contiguous(), cd)                              # [128, Kd]
    out = _dev(prev, cd)
    xn_out = torch.empty_like(xd) if want_xn else None
    dg, db = torch.ones(128, device="cuda"), torch.ones(128, device="cuda")          # accumulated into
    _lib.check(lib.kasf_op_dgrad_lnbwd(DT[cd][0], ptr(dyd), Kd, ptr(wt), ptr(addd) if use_add else None, ptr(xd), ptr(_f32(g)),
                                       ptr(rd) if use_resid else None, ptr(out), int(accumulate), ptr(dg), ptr(db), M, ptr(xn_out),
                                       ptr(_f32(b)) if want_xn else None, stream()))
    torch.cuda.synchronize()
    xr = _back(xd).requires_grad_(True)
    gp, bp = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xn = _ln(xr, gp, bp)
    dxn = _back(dyd) @ _back(wt).T + (_back(addd) if use_add else 0)
    xn.backward(dxn)
    want = xr.grad + (_back(rd) if use_resid else 0) + (_back(_dev(prev, cd)) if accumulate else 0)
    tol = TOL[cd] * (3 if M > 1000 else 1)                      # column sums over more rows: more bf16 rounding noise in dgamma/dbeta
    assert rel_err(_back(out), want) < TOL[cd]
    assert rel_err(_back(dg) - 1, gp.grad) < tol and rel_err(_back(db) - 1, bp.grad) < tol
    if want_xn:
        assert rel_err(_back(xn_out), xn.detach()) < TOL[cd]


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
@pytest.mark.parametrize("mode,T", [(0, 27), (1, 27), (1, 81), (1, 9),
                                    (1, 97), (1, 128), (1, 129), (1, 192), (1, 193), (1, 243), (1, 256),   # bf16: 4 / 6 / 8 score tiles in registers, two-pass backward
                                    (1, 257)])                                                              # past the MFMA cores: the LDS-resident fp32 kernels
def test_attention_core(lib, cd, mode, T):
    from kasportsformer_amd import _lib
    from oracle.kasf_oracle import attention_core, _heads
    B = 2
    qkv = _rand(B, T, 17, 384, seed=14)
    do = _rand(B, T, 17, 128, seed=15)
    qd, dod = _dev(qkv, cd), _dev(do, cd)
    o = torch.empty(B, T, 17, 128, device="cuda", dtype=DT[cd][1])
    dqkv = torch.zeros_like(qd)
    es = qd.element_size()
    base = qd.data_ptr()
    _lib.check(lib.kasf_op_attention_fwd(DT[cd][0], base, 384, base + 128 * es, base + 256 * es, 384, ptr(o), B, T, mode, stream()))
    db = dqkv.data_ptr()
    _lib.check(lib.kasf_op_attention_bwd(DT[cd][0], base, 384, base + 128 * es, base + 256 * es, 384, ptr(dod), db, 384, db + 128 * es,
                                         db + 256 * es, 384, B, T, mode, stream()))
    torch.cuda.synchronize()
    qr = _back(qd).requires_grad_(True)
    q, k, v = _heads(qr, 3, 8)
    ref = attention_core(q, k, v, "spatial" if mode == 0 else "temporal", 0.25)
    ref.backward(_back(dod))
    assert rel_err(_back(o), ref) < TOL[cd]
    assert rel_err(_back(dqkv), qr.grad) < TOL[cd]


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
@pytest.mark.parametrize("heads,mode,T", [(4, 0, 27), (4, 1, 27), (4, 1, 9), (4, 1, 32), (4, 1, 33), (4, 1, 81), (4, 1, 96), (4, 1, 97), (4, 1, 130), (4, 1, 243), (4, 1, 256),
                                          (4, 1, 257), (16, 1, 27), (2, 0, 27)])
def test_attention_core_other_head_counts(lib, cd, heads, mode, T):
    """num_heads = 4 is the reference constructor's default (KASportsFormer.py:293; head dimension 32, scale 32 ** -0.5): bf16 mode runs MFMA cores of its own
    (k_attn_fwd_mfma32 / k_attn_bwd_2p32: 1, 3, 4, 6, 8 key tiles; 257 positions fall back to the LDS-resident cores, like 2 and 16 heads and fp32 mode)."""
    from kasportsformer_amd import _lib
    from oracle.kasf_oracle import attention_core, _heads
    B = 2
    qkv = _rand(B, T, 17, 384, seed=24)
    do = _rand(B, T, 17, 128, seed=25)
    qd, dod = _dev(qkv, cd), _dev(do, cd)
    o = torch.empty(B, T, 17, 128, device="cuda", dtype=DT[cd][1])
    dqkv = torch.zeros_like(qd)
    es, base, db = qd.element_size(), qd.data_ptr(), dqkv.data_ptr()
    _lib.check(lib.kasf_op_attention_fwd_heads(DT[cd][0], base, 384, base + 128 * es, base + 256 * es, 384, ptr(o), B, T, mode, heads, stream()))
    _lib.check(lib.kasf_op_attention_bwd_heads(DT[cd][0], base, 384, base + 128 * es, base + 256 * es, 384, ptr(dod), db, 384, db + 128 * es, db + 256 * es, 384,
                                               B, T, mode, heads, stream()))
    torch.cuda.synchronize()
    qr = _back(qd).requires_grad_(True)
    q, k, v = _heads(qr, 3, heads)
    ref = attention_core(q, k, v, "spatial" if mode == 0 else "temporal", (128 // heads) ** -0.5)
    ref.backward(_back(dod))
    assert rel_err(_back(o), ref) < TOL[cd]
    assert rel_err(_back(dqkv), qr.grad) < TOL[cd]


def _random_attention_cases(n=14, seed=20261004):
    import random
    rnd = random.Random(seed)
    cases = []
    for _ in range(n):
        heads = rnd.choice([4, 8])
        mode = rnd.choice([0, 1, 1, 1])                       # temporal groups have the shape-dependent kernels
        T = rnd.choice([rnd.randint(4, 32), rnd.randint(33, 96), rnd.randint(97, 256)])
        cases.append((heads, mode, T, rnd.randint(1, 3)))
    return cases


@pytest.mark.parametrize("heads,mode,T,B", _random_attention_cases())
def test_attention_core_random_shapes_bf16(lib, heads, mode, T, B):
    """Seeded random (head count, mode, clip length, batch) over everything the MFMA cores take: 1 / 3 / 4 / 6 / 8 key tiles, ragged last tiles, 4 and 8 heads."""
    from kasportsformer_amd import _lib
    from oracle.kasf_oracle import attention_core, _heads
    cd = "bf16"
    qkv = _rand(B, T, 17, 384, seed=100 + T)
    do = _rand(B, T, 17, 128, seed=200 + T)
    qd, dod = _dev(qkv, cd), _dev(do, cd)
    o = torch.empty(B, T, 17, 128, device="cuda", dtype=DT[cd][1])
    dqkv = torch.zeros_like(qd)
    es, base, db = qd.element_size(), qd.data_ptr(), dqkv.data_ptr()
    _lib.check(lib.kasf_op_attention_fwd_heads(DT[cd][0], base, 384, base + 128 * es, base + 256 * es, 384, ptr(o), B, T, mode, heads, stream()))
    _lib.check(lib.kasf_op_attention_bwd_heads(DT[cd][0], base, 384, base + 128 * es, base + 256 * es, 384, ptr(dod), db, 384, db + 128 * es, db + 256 * es, 384,
                                               B, T, mode, heads, stream()))
    torch.cuda.synchronize()
    qr = _back(qd).requires_grad_(True)
    q, k, v = _heads(qr, 3, heads)
    ref = attention_core(q, k, v, "spatial" if mode == 0 else "temporal", (128 // heads) ** -0.5)
    ref.backward(_back(dod))
    assert rel_err(_back(o), ref) < TOL[cd]
    assert rel_err(_back(dqkv), qr.grad) < TOL[cd]


@pytest.mark.parametrize("bone", [False, True])
@pytest.mark.parametrize("mode,T,B", [(0, 27, 2), (1, 27, 3), (1, 9, 2), (0, 27, 41), (1, 32, 67), (0, 4, 1)])     # 41 x 27 = 1,107 groups: three per persistent
def test_attention_backward_fused_do(lib, bone, mode, T, B):                                                     # workgroup, last range short; 67 x 17 = 1,139
    """The training step's attention backward: d_o = g_mid . (ls1 . Wproj) formed inside the kernel (attention.py's proj + layer-scale data gradient
    folded in).  Persistent form against the oracle's attention core, and bit for bit against the one-group-per-workgroup form it replaced."""
    from kasportsformer_amd import _lib
    from oracle.kasf_oracle import attention_core, _heads
    qkv = _rand(B, T, 17, 384, seed=24)
    g_mid = _rand(B, T, 17, 128, seed=25)
    W = _rand(128, 128, seed=26) * 0.2
    ls1 = _rand(128, seed=27)
    wts = (ls1[:, None] * W).t().contiguous()                     # [c][n] = ls1[n] W[n][c]
    qd, gd, wd = _dev(qkv, "bf16"), _dev(g_mid, "bf16"), _dev(wts, "bf16")
    if bone:                                                      # q in its own [M,128] tensor, k|v in a [M,256] one (bone_crossattention.py)
        qsep, kv = qd[..., :128].contiguous(), qd[..., 128:].contiguous()
        args = (qsep.data_ptr(), 128, kv.data_ptr(), kv.data_ptr() + 256, 256)
    else:
        args = (qd.data_ptr(), 384, qd.data_ptr() + 256, qd.data_ptr() + 512, 384)
    outs = []
    for form in (0, 1):
        torch.manual_seed(0)
        if bone:
            dq = torch.full((B, T, 17, 128), float("nan"), device="cuda", dtype=torch.bfloat16)
            dkv = torch.full((B, T, 17, 256), float("nan"), device="cuda", dtype=torch.bfloat16)
            oargs = (dq.data_ptr(), 128, dkv.data_ptr(), dkv.data_ptr() + 256, 256)
        else:
            dqkv = torch.full((B, T, 17, 384), float("nan"), device="cuda", dtype=torch.bfloat16)
            oargs = (dqkv.data_ptr(), 384, dqkv.data_ptr() + 256, dqkv.data_ptr() + 512, 384)
        _lib.check(lib.kasf_op_attention_bwd_fused_do(*args, ptr(gd), ptr(wd), *oargs, B, T, mode, form, None, None, stream()))
        torch.cuda.synchronize()
        outs.append(torch.cat((dq, dkv), dim=-1).float().cpu() if bone else dqkv.float().cpu())
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
    d_o = (_back(gd) @ _back(wd).t()).to(torch.bfloat16).float()
    qr = _back(qd).requires_grad_(True)
    q, k, v = _heads(qr, 3, 8)
    attention_core(q, k, v, "spatial" if mode == 0 else "temporal", 0.25).backward(d_o)
    assert rel_err(outs[0], qr.grad) < TOL["bf16"]


@pytest.mark.parametrize("bone", [False, True])
@pytest.mark.parametrize("T,B", [(81, 3), (33, 2), (64, 2), (96, 1), (50, 5)])       # 33: a one-row last tile; 64 / 96: two / three full tiles
def test_attention_backward_long_groups_key_tile_outer(lib, bone, T, B):
    """Temporal groups of 33..96 frames: k_attn_bwd_kt (statistics from the forward's log-sum-exp, delta from the saved attention output, un-transposed
    scores) against the oracle's attention core, and against the self-contained kernel it replaces in the engine (same operands, no lse / o)."""
    from kasportsformer_amd import _lib
    from oracle.kasf_oracle import attention_core, _heads
    mode = 1
    qkv = _rand(B, T, 17, 384, seed=34)
    g_mid = _rand(B, T, 17, 128, seed=35)
    W = _rand(128, 128, seed=36) * 0.2
    ls1 = _rand(128, seed=37)
    wts = (ls1[:, None] * W).t().contiguous()
    qd, gd, wd = _dev(qkv, "bf16"), _dev(g_mid, "bf16"), _dev(wts, "bf16")
    # what the training forward leaves behind: o (bf16) and lse = log sum_j exp(q_i . k_j / 4) per (token, head), from the stored bf16 q, k, v
    qf = _back(qd)
    q, k, v = _heads(qf, 3, 8)                                    # [B,H,T,J,d]
    s = (q.transpose(2, 3) @ k.transpose(2, 3).transpose(-2, -1)) * 0.25          # [B,H,J,T,T]
    lse = torch.logsumexp(s, dim=-1).permute(0, 3, 2, 1).contiguous()             # [B,T,J,H]
    o = attention_core(q, k, v, "temporal", 0.25)                                 # [B,T,J,128]
    od, lsed = _dev(o, "bf16"), lse.float().cuda()
    if bone:
        qsep, kv = qd[..., :128].contiguous(), qd[..., 128:].contiguous()
        args = (qsep.data_ptr(), 128, kv.data_ptr(), kv.data_ptr() + 256, 256)
    else:
        args = (qd.data_ptr(), 384, qd.data_ptr() + 256, qd.data_ptr() + 512, 384)
    outs = []
    for with_stats in (True, False):
        if bone:
            dq = torch.full((B, T, 17, 128), float("nan"), device="cuda", dtype=torch.bfloat16)
            dkv = torch.full((B, T, 17, 256), float("nan"), device="cuda", dtype=torch.bfloat16)
            oargs = (dq.data_ptr(), 128, dkv.data_ptr(), dkv.data_ptr() + 256, 256)
        else:
            dqkv = torch.full((B, T, 17, 384), float("nan"), device="cuda", dtype=torch.bfloat16)
            oargs = (dqkv.data_ptr(), 384, dqkv.data_ptr() + 256, dqkv.data_ptr() + 512, 384)
        _lib.check(lib.kasf_op_attention_bwd_fused_do(*args, ptr(gd), ptr(wd), *oargs, B, T, mode, 0, ptr(od) if with_stats else None,
                                                      ptr(lsed) if with_stats else None, stream()))
        torch.cuda.synchronize()
        outs.append(torch.cat((dq, dkv), dim=-1).float().cpu() if bone else dqkv.float().cpu())
    assert torch.isfinite(outs[0]).all() and torch.isfinite(outs[1]).all()
    d_o = (_back(gd) @ _back(wd).t()).to(torch.bfloat16).float()
    qr = _back(qd).requires_grad_(True)
    q, k, v = _heads(qr, 3, 8)
    attention_core(q, k, v, "temporal", 0.25).backward(d_o)
    assert rel_err(outs[0], qr.grad) < TOL["bf16"]
    assert rel_err(outs[1], qr.grad) < TOL["bf16"]
    assert rel_err(outs[0], outs[1]) < TOL["bf16"]


def test_loss3_and_adamw(lib):
    import kasportsformer_amd as K
    from oracle import kasf_oracle as O
    pred = _rand(3, 27, 17, 3, seed=16).cuda().requires_grad_(True)
    tgt = _rand(3, 27, 17, 3, seed=17).cuda()
    total, parts = K.loss3(pred, tgt)
    total.backward()
    pr = pred.detach().cpu().requires_grad_(True)
    ref, ref_parts = O.loss_total(pr, tgt.cpu())
    ref.backward()
    assert abs(total.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    for a, b in zip(parts[1:].cpu().tolist(), ref_parts):
        assert abs(a - b.item()) < 1e-5
    assert rel_err(pred.grad, pr.grad) < 1e-4
    # AdamW vs torch.optim.AdamW, 3 steps
    from kasportsformer_amd import _lib
    n = 4096
    p0, gs = _rand(n, seed=18), [_rand(n, seed=19 + i) for i in range(3)]
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=5e-4, weight_decay=0.01)
    pd, m, v = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for i, g in enumerate(gs):
        pt.grad = g.clone()
        opt.step()
        _lib.check(lib.kasf_adamw_step(ptr(pd), ptr(_f32(g)), ptr(m), ptr(v), n, 5e-4, 0.9, 0.999, 1e-8, 0.01, i + 1, 1.0, stream()))
    torch.cuda.synchronize()
    assert (pd.cpu() - pt.detach()).abs().max() < 1e-6
