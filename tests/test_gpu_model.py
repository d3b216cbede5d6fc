"""Whole-path GPU parity: the HIP KASportsFormer (through the C-ABI) against the CPU oracle and against
golden fixtures captured from the real reference.  Bars: fp32 mode <= 1e-3 (north_star), bit-exact
bone gathers; bf16 mode reported against a looser, documented bound."""
import os

import numpy as np
import pytest
import torch

from oracle import kasf_oracle as O
from tests.gpu_util import compare_grads, decode_masks, forced_adjacency, make_pair, oracle_stage_hooks, rel_err, ws_tensor

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _abs_err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def test_library_is_loaded_and_product_path_has_no_fallback():
    import kasportsformer_amd as K
    from kasportsformer_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    maps = open("/proc/self/maps").read()
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libkasf_hip.so" in maps
    m = K.KASportsFormer(n_layers=1, num_heads=8)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 27, 17, 3))            # CPU tensors are refused, there is no CPU path


@pytest.mark.parametrize("name,tol", [("model_L2_T27_B2.npz", 1e-3), ("model_L1_T81_B1.npz", 1e-3)])
def test_fp32_forward_matches_reference_golden(name, tol):
    fx = np.load(os.path.join(GOLDEN, name))
    T, L = int(fx["T"]), int(fx["n_layers"])
    _, model = make_pair(L, T, "fp32")
    x = torch.from_numpy(fx["x"]).cuda()
    x_copy = x.clone()
    model.eval()
    with torch.no_grad():
        pred = model(x)
        rep = model(x, return_rep=True)
    assert torch.equal(x, x_copy), "forward must not modify its input"
    assert _abs_err(pred, torch.from_numpy(fx["pred_eval"])) < tol
    assert _abs_err(rep[0], torch.from_numpy(fx["rep_eval_b0"])) < tol
    pred[:, :, 0, :] = 0                        # callers mutate the output in place (train_and_evaluate_sp.py:55)
    model.train()
    with torch.no_grad():
        pred_t = model(x)
    assert _abs_err(pred_t, torch.from_numpy(fx["pred_train"])) < tol
    sd = model.state_dict()
    for k in fx.files:
        if k.startswith("buf/"):                # BatchNorm running statistics after one training forward
            ref = torch.from_numpy(fx[k])
            assert _abs_err(sd[k[4:]].to(ref.dtype), ref) < 1e-4, k


def test_stage_outputs_equal_forward_hooks_on_the_reference_modules():
    """`model.stage_outputs(x)` (the replacement for hook-based inspection: the sub-modules are parameter holders, the forward is one library call) returns, under the
    reference's module names, what forward hooks on those modules capture in the oracle -- in training and in evaluation mode -- and the same `out` as `model(x)`."""
    oracle, model = make_pair(2, 27, "fp32")
    x, _ = O.synthetic_clips(2, 27, seed=5)
    for train in (True, False):
        oracle.train(train); model.train(train)
        cap = {}
        for li, layer in enumerate(oracle.layers_with_bone):
            for kind in O.BLOCK_KINDS:
                getattr(layer, kind).register_forward_hook(lambda m, i, o, k=f"layers_with_bone.{li}.{kind}": cap.__setitem__(k, o.detach()))
            layer.register_forward_hook(lambda m, i, o, k=f"layers_with_bone.{li}": cap.__setitem__(k, o.detach()))
        oracle.rep_logit.register_forward_hook(lambda m, i, o: cap.__setitem__("rep_logit", o.detach()))
        buf = model._flat_buffers.clone()
        with forced_adjacency(model, x):
            ref = oracle(x)
        out, stages = model.stage_outputs(x.cuda())
        if train:
            model._flat_buffers.copy_(buf)
        assert set(cap) <= set(stages) and {"joints_embed", "bone_embed", "limb_embed"} <= set(stages)
        for name, r in cap.items():
            assert _abs_err(stages[name], r) / max(1.0, float(r.abs().max())) < 1e-3, (name, train)
        assert _abs_err(out, ref) / max(1.0, float(ref.abs().max())) < 1e-3
        with torch.no_grad():
            assert torch.equal(out, model(x.cuda())) or train      # (training mode: the second forward sees updated running statistics only in its buffers, not in `out`)
        with pytest.raises(RuntimeError):
            model.layers_with_bone[0].att_spatial(x.cuda())


@pytest.mark.parametrize("cd,tol", [("fp32", 1e-3), ("bf16", 0.05)])      # bf16: observed 2.5e-2 (x 2)
def test_stage_by_stage_against_oracle(cd, tol):
    """Every FormerModule / layer output of a 2-layer model (layer 0 exercises the bone-embedding start)."""
    oracle, model = make_pair(2, 27, cd)
    cap = oracle_stage_hooks(oracle)
    x, _ = O.synthetic_clips(2, 27)
    oracle.train()
    ref = oracle(x)
    model.train()
    with torch.no_grad():
        out, ws, _ = model._launch_forward(x.cuda(), False, keep=True)
    torch.cuda.synchronize()
    worst = {}
    for name, r in cap.items():
        got = ws_tensor(model, ws, 2, name).float().view(r.shape)
        worst[name] = _abs_err(got, r) / max(1.0, float(r.abs().max()))
    worst["pred"] = _abs_err(out, ref) / max(1.0, float(ref.abs().max()))
    print(f"[stages, {cd}] worst stage error {max(worst.values()):.3e} ({max(worst, key=worst.get)})")
    bad = {k: v for k, v in worst.items() if not v < tol}
    assert not bad, f"stages above {tol}: {bad}\nall: {worst}"
    # prologue: bone decomposition is a gather + fp32 arithmetic -> compare exactly-rounded values tightly
    bone = ws_tensor(model, ws, 2, "bone3").view(2, 27, 17, 3)
    assert _abs_err(bone, O.bone_decompose(x)) < 1e-6
    limb = ws_tensor(model, ws, 2, "limb3").view(2, 27, 17, 3)
    assert _abs_err(limb, oracle.bone_refusion(x)) < 1e-5


MID_COS_MANY_TOKENS, MID_COS_FEW_CLIPS = 0.99, 0.9      # bf16, per-tensor cosine of the 65..255-element gradients: with >= 15k tokens behind every sum / with a handful of clips
BF16_TENSOR_TOL = 0.35      # bf16, per-tensor bar for tensors of >= 256 elements with a HANDFUL of clips behind every sum (smaller tensors are judged pooled: tests/gpu_util.py
                            # compare_grads): observed up to 0.30, always on the 128 x 128 U / V weights of the spatial GCN (BatchNorm backward subtracts batch means of bf16-stored
                            # operands: a cancellation); with >= 15k tokens behind every sum the bar is 0.04 (observed 0.021)


@pytest.mark.parametrize("L,T,B", [(2, 27, 2), (1, 81, 2), (1, 9, 3), (1, 27, 37), (1, 27, 101)])      # B=37: 531 tiles, two per persistent workgroup; B=101: 1,449 tiles (six per workgroup: ring slots reused, steady-state look-ahead waits), both with a ragged last tile; B=101 (46,359 tokens) is also past the engine's threshold (40,000 tokens) for the FUSED data + weight gradient kernels (k_dgrad_r<..., WG>: qkv / q / kv / U|V weight gradients and the U|V bias gradient from bf16 partial tiles), the smaller cases run the two-kernel sequence
@pytest.mark.parametrize("cd,tol", [("fp32", 2e-3), ("bf16", BF16_TENSOR_TOL)])
def test_backward_matches_oracle(cd, tol, L, T, B):
    oracle, model = make_pair(L, T, cd)
    x, y = O.synthetic_clips(B, T)
    oracle.train()
    loss_ref, _ = O.loss_total(oracle(x), y)
    loss_ref.backward()
    model.train()
    pred = model(x.cuda())
    loss, _ = O.loss_total(pred, y.cuda())       # torch autograd for the loss, as the reference harness does
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) < (1e-4 if cd == "fp32" else 5e-2) * max(1.0, abs(loss_ref.item()))
    # fp32 floor: gradients that are sums of cancelling terms (constant-confidence limb MLPs) carry fp32 summation-order noise on both sides (at B=101 these
    # sums run over 46k tokens and the noise of the cancelling limb-MLP gradients, which sit at 1e-4..1e-2 of gmax, grows with it: the floor is raised there)
    rep = compare_grads(model, oracle, cd, floor_rel=(1e-3 if B <= 37 else 2e-2) if cd == "fp32" else 0.05)
    assert not rep["none_mismatch"], rep["none_mismatch"]
    assert sum(1 for p in model.parameters() if p.grad is None) == 8 * L   # 8 dead norm1_limb tensors per layer
    print(f"[backward, {cd}, L={L} T={T} B={B}] gradient cosine {rep['cosine']:.7f}, worst per-tensor error {rep['worst']:.3e} ({rep['worst_name']}), "
          f"pooled cosine of the tensors below 256 elements {rep['pooled_small_cosine']:.5f}")
    assert rep["cosine"] > (0.999999 if cd == "fp32" else 0.999), rep["cosine"]                  # bf16: observed >= 0.9995
    assert rep["pooled_small_cosine"] > 0.999, rep["pooled_small_cosine"]           # bf16: observed >= 0.99914
    if cd == "bf16":     # the 128-element tensors (LayerNorm gamma / beta, biases, layer scales: per-workgroup rows + k_col_finish / kasf_launch_proj_finish) one by one
        print(f"    worst 65..255-element tensor: cosine {rep['mid_min_cosine']:.5f} ({rep['mid_min_name']}); below 0.999: "
              f"{sorted((round(v, 4), k) for k, v in rep['mid_cosines'].items() if v < 0.999)[:6]}")
        assert rep["mid_min_cosine"] > (MID_COS_MANY_TOKENS if B * T >= 900 else MID_COS_FEW_CLIPS), (rep["mid_min_cosine"], rep["mid_min_name"])
    if cd == "bf16" and B * T >= 900:
        tol = 0.04       # per-tensor bf16 error with >= 15k tokens behind every sum: observed 2.1e-2 (x 2)
    bad = sorted(((v, k) for k, v in rep["errors"].items() if not v < tol), reverse=True)
    assert not bad, f"{len(bad)} gradients above {tol}; worst (err, name): {bad[:12]}"


@pytest.mark.parametrize("T,B", [(27, 256), (81, 128)])          # BASELINE.json configs[1] and configs[3] (KASportsFormer.py:320-347, configs/sportspose-gt-kasportsformer.yaml:61,67)
def test_backward_matches_oracle_at_benchmark_shapes(T, B):
    """The ENGINE against the oracle at the benchmark's own token counts (one layer, bf16): 117,504 / 176,256 tokens put every persistent launch in its
    grid classes of the persistent launches at these token counts (half the chip for the MLP / data-gradient launches), give the MLP ranges >= 50 tiles and run k_attn_bwd_kt / the fused
    attention-block backward over the whole batch -- regimes the smaller oracle cases above never reach (missing #2)."""
    import psutil
    if psutil.virtual_memory().available < 24 * 2**30:
        pytest.skip("the CPU oracle peaks at 8-13 GB at this shape (15-20 s); less than 24 GB available on this box")
    oracle, model = make_pair(1, T, "bf16")
    x, y = O.synthetic_clips(B, T)
    oracle.train()
    loss_ref, _ = O.loss_total(oracle(x), y)
    loss_ref.backward()
    model.train()
    pred = model(x.cuda())
    loss, _ = O.loss_total(pred, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) < 5e-2 * max(1.0, abs(loss_ref.item()))
    rep = compare_grads(model, oracle, "bf16", floor_rel=0.05)
    assert not rep["none_mismatch"], rep["none_mismatch"]
    print(f"[backward at the benchmark shape, bf16, L=1 T={T} B={B}] gradient cosine {rep['cosine']:.7f}, worst per-tensor error {rep['worst']:.3e} "
          f"({rep['worst_name']}), worst 65..255-element tensor cosine {rep['mid_min_cosine']:.5f} ({rep['mid_min_name']})")
    assert rep["cosine"] > 0.999, rep["cosine"]
    assert rep["pooled_small_cosine"] > 0.999, rep["pooled_small_cosine"]
    assert rep["mid_min_cosine"] > MID_COS_MANY_TOKENS, (rep["mid_min_cosine"], rep["mid_min_name"])
    bad = sorted(((v, k) for k, v in rep["errors"].items() if not v < 0.04), reverse=True)      # the many-token bar of test_backward_matches_oracle
    assert not bad, f"{len(bad)} gradients above 0.04; worst (err, name): {bad[:12]}"


@pytest.mark.parametrize("cd,tol", [("fp32", 2e-3), ("bf16", BF16_TENSOR_TOL)])
def test_detected_keypoints_confidence_channel(cd, tol):
    """SURVEY config 3 ("WorldPose-det"): the third input channel is a detector confidence ~U(0,1) instead of the constant 1 of ground-truth 2-D
    input, so the confidence-channel limb MLPs and the third embedding column see real data: forward and every gradient against the oracle."""
    oracle, model = make_pair(2, 27, cd)
    x, y = O.synthetic_clips(3, 27, seed=77, res=(1920, 1080), det_conf=True)
    oracle.train()
    loss_ref, _ = O.loss_total(oracle(x), y)
    loss_ref.backward()
    model.train()
    pred = model(x.cuda())
    loss, _ = O.loss_total(pred, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) < (1e-4 if cd == "fp32" else 5e-2) * max(1.0, abs(loss_ref.item()))
    rep = compare_grads(model, oracle, cd)
    assert not rep["none_mismatch"], rep["none_mismatch"]
    print(f"[detector confidence, {cd}] gradient cosine {rep['cosine']:.7f}, worst per-tensor error {rep['worst']:.3e} ({rep['worst_name']}), pooled cosine of the "
          f"tensors below 256 elements {rep['pooled_small_cosine']:.5f}")
    bad = sorted(((v, k) for k, v in rep["errors"].items() if not v < tol), reverse=True)
    assert not bad, bad[:10]
    assert rep["pooled_small_cosine"] > 0.999 and rep["cosine"] > (0.999999 if cd == "fp32" else 0.9995)       # bf16 observed: 0.99978 / 0.99984


def test_zero_length_bones_and_static_clip():
    """Two inputs the reference handles by special cases.  (1) Coincident joints: zero-length bones get length 1 and direction 0
    (KASportsFormer.py:51) -- forward and backward against the oracle.  (2) A static clip (every frame the same): all temporal similarities tie
    exactly, `topk` + `ge` keeps every neighbour (graph.py:109-111); whatever the reference's BLAS does with the ties, the result must be finite
    and the same for every frame, because nothing in the model distinguishes frames but their content."""
    oracle, model = make_pair(1, 27, "fp32")
    x, y = O.synthetic_clips(2, 27, seed=11)
    x[:, :, 1, :2] = x[:, :, 0, :2]            # bone 0-1 has zero length in every frame
    x[0, 5, 9, :2] = x[0, 5, 8, :2]            # and one more in a single frame
    oracle.train()
    ref = oracle(x)
    l_ref, _ = O.loss_total(ref, y)
    l_ref.backward()
    model.train()
    pred = model(x.cuda())
    loss, _ = O.loss_total(pred, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all()
    assert float((pred.cpu() - ref).abs().max()) < 1e-3
    ref_grads = dict(oracle.named_parameters())
    gmax = max(float(q.grad.abs().max()) for q in ref_grads.values() if q.grad is not None)
    for n, p in model.named_parameters():
        r = ref_grads[n].grad
        if r is not None:
            assert float((p.grad.cpu() - r).abs().max()) < 2e-3 * max(float(r.abs().max()), 1e-3 * gmax), n
    import kasportsformer_amd as K
    for cd in ("fp32", "bf16"):
        # default initialisation (BatchNorm1d(T) carries per-FRAME affine parameters: random ones would tell the frames apart), layer scales raised
        # from 1e-5 to 1 so that the mixers actually contribute
        torch.manual_seed(5)
        m2 = K.KASportsFormer(n_layers=2, num_heads=8, n_frames=27, compute_dtype=cd)
        sd = m2.state_dict()
        for k in sd:
            if "layer_scale" in k:
                sd[k] = torch.ones_like(sd[k])
        m2.load_state_dict(sd)
        m2 = m2.cuda().eval()
        frame = O.synthetic_clips(3, 1, seed=12)[0]
        static = frame.expand(3, 27, 17, 3).contiguous().cuda()
        with torch.no_grad():
            out = m2(static)
        assert torch.isfinite(out).all()
        spread = float((out - out[:, :1]).abs().max())
        assert spread <= (1e-5 if cd == "fp32" else 2e-2) * max(1.0, float(out.abs().max())), (cd, spread)


def test_empty_batch():
    """B = 0: an empty result in both modes, as the reference returns (no kernel runs on zero clips)."""
    _, model = make_pair(1, 27, "bf16")
    x = torch.empty(0, 27, 17, 3, device="cuda")
    for mode in (model.eval, model.train):
        mode()
        with torch.no_grad():
            assert tuple(model(x).shape) == (0, 27, 17, 3)
            assert tuple(model(x, return_rep=True).shape) == (0, 27, 17, 512)
    assert tuple(model(x).shape) == (0, 27, 17, 3)          # training mode with autograd enabled


def test_large_batches():
    """SURVEY config 5's global batch on one GPU (B = 2048, T = 27, 26 layers, bf16: 940k tokens, 29k token tiles): finite, and every clip's
    evaluation output bit-identical to what the same clip gives inside a batch of 256 (no index arithmetic wraps, no tile is skipped);
    one training step at B = 512 (twice the benchmark batch) stays finite."""
    import kasportsformer_amd as K
    torch.manual_seed(3)
    m = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().eval()
    x = O.synthetic_clips(2048, 27, seed=21)[0].cuda()
    with torch.no_grad():
        big = m(x)
        assert torch.isfinite(big).all()
        for lo in (0, 1024, 1792):
            assert torch.equal(big[lo:lo + 256], m(x[lo:lo + 256])), lo
    del big
    m.train()
    m.attach_param_grads = False
    opt = K.FusedAdamW(m, lr=5e-4, weight_decay=0.01)
    xb, yb = (t.cuda() for t in O.synthetic_clips(512, 27, seed=22))
    opt.zero_grad()
    loss, parts = K.loss3(m(xb), yb)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert all(float(v) == float(v) for v in parts) and bool(torch.isfinite(m._flat).all()) and bool(torch.isfinite(m.flat_grad[:m.n_live]).all())


def test_fp32_backward_matches_reference_golden():
    fx = np.load(os.path.join(GOLDEN, "model_L2_T27_B2.npz"))
    _, model = make_pair(2, 27, "fp32")
    x, y = torch.from_numpy(fx["x"]).cuda(), torch.from_numpy(fx["y"]).cuda()
    model.train()
    import kasportsformer_amd as K
    pred = model(x)
    total, parts = K.loss3(pred, y)              # fused loss kernel
    total.backward()
    torch.cuda.synchronize()
    assert np.abs(parts.cpu().numpy().astype(np.float64) - fx["losses"]).max() < 1e-4
    bad = []
    for n, p in model.named_parameters():
        if "gnone/" + n in fx.files:
            assert p.grad is None, n
            continue
        g = p.grad.reshape(-1).cpu()
        step = max(1, g.numel() // 256)
        ref = fx["gsmp/" + n]
        err = np.abs(g[::step][:256].numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        if not err < 2e-3:
            bad.append((float(err), n))
    assert not bad, sorted(bad, reverse=True)[:12]


def test_training_step_with_fused_adamw_tracks_oracle():
    """Three optimisation steps: HIP fp32 path + FusedAdamW vs oracle + torch.optim.AdamW from the same init."""
    import kasportsformer_amd as K
    oracle, model = make_pair(1, 27, "fp32")
    opt_ref = torch.optim.AdamW(oracle.parameters(), lr=5e-4, weight_decay=0.01)
    opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
    model.attach_param_grads = False
    oracle.train(); model.train()
    for step in range(3):
        x, y = O.synthetic_clips(2, 27, seed=100 + step)
        opt_ref.zero_grad()
        l_ref, _ = O.loss_total(oracle(x), y)
        l_ref.backward()
        opt_ref.step()
        opt.zero_grad()
        l, _ = K.loss3(model(x.cuda()), y.cuda())
        l.backward()
        opt.step()
        assert abs(l.item() - l_ref.item()) < 2e-3 * max(1.0, abs(l_ref.item())), (step, l.item(), l_ref.item())
    sd = model.state_dict()
    worst = max(_abs_err(sd[n], p) for n, p in oracle.named_parameters())
    assert worst < 5e-4, worst
    dead = [n for n, p in oracle.named_parameters() if p.grad is None]
    fresh = O.name_seeded_fill(oracle.state_dict())
    for n in dead:                                # never-touched parameters are not decayed either
        assert torch.equal(sd[n].cpu(), fresh[n])


def test_state_dict_roundtrip_and_module_prefix():
    import kasportsformer_amd as K
    m = K.KASportsFormer(n_layers=1, num_heads=8).cuda()
    sd = {("module." + k): v.clone() for k, v in m.state_dict().items()}          # DataParallel-saved checkpoints
    m2 = K.KASportsFormer(n_layers=1, num_heads=8).cuda()
    m2.load_state_dict({k[len("module."):]: v for k, v in sd.items()}, strict=True)
    x, _ = O.synthetic_clips(1, 27)
    m.eval(); m2.eval()
    with torch.no_grad():
        assert torch.equal(m(x.cuda()), m2(x.cuda()))


def test_plain_mean_fusion_forward_backward():
    """use_adaptive_fusion=False (KASportsFormer.py:284): plain mean of the three branches, no gate parameters touched."""
    import kasportsformer_amd as K
    oracle = O.KASportsFormerOracle(n_layers=2, num_heads=8, n_frames=27, use_adaptive_fusion=False)
    sd = O.name_seeded_fill(oracle.state_dict())
    oracle.load_state_dict(sd, strict=True)
    model = K.KASportsFormer(n_layers=2, num_heads=8, n_frames=27, use_adaptive_fusion=False, compute_dtype="fp32")
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    oracle.train()
    x, y = O.synthetic_clips(3, 27, seed=21)
    ref = oracle(x)
    O.loss_total(ref, y)[0].backward()
    pred = model(x.cuda())
    K.loss3(pred, y.cuda())[0].backward()
    torch.cuda.synchronize()
    assert _abs_err(pred, ref) / max(1.0, float(ref.abs().max())) < 1e-3
    gmax = max(float(q.grad.abs().max()) for q in oracle.parameters() if q.grad is not None)
    for (n, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if q.grad is not None:
            assert float((p.grad.cpu() - q.grad).abs().max()) / max(float(q.grad.abs().max()), 1e-3 * gmax) < 2e-3, n


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
def test_clips_are_independent_in_eval_mode(cd):
    """Evaluation-mode output of a clip does not depend on what else is in the batch, bit for bit (BatchNorm uses running statistics;
    every kernel computes a token / a track with the same operation order whatever tile it lands in).  Covers B=1 and a ragged last tile."""
    _, model = make_pair(2, 27, cd)
    model.eval()
    x, _ = O.synthetic_clips(7, 27, seed=33)
    x = x.cuda()
    with torch.no_grad():
        full = model(x)
        one = model(x[3:4])
        pair = model(x[5:7])
        rep_full = model(x, return_rep=True)
        rep_one = model(x[3:4], return_rep=True)
    assert torch.equal(full[3:4], one) and torch.equal(full[5:7], pair)
    assert rep_full.shape == (7, 27, 17, 512) and torch.equal(rep_full[3:4], rep_one)


def test_single_clip_training_step_runs_and_matches_oracle():
    """B = 1: 459 tokens, fewer tiles than workgroups in every persistent kernel."""
    import kasportsformer_amd as K
    oracle, model = make_pair(1, 27, "fp32")
    oracle.train(); model.train()
    x, y = O.synthetic_clips(1, 27, seed=41)
    O.loss_total(oracle(x), y)[0].backward()
    K.loss3(model(x.cuda()), y.cuda())[0].backward()
    torch.cuda.synchronize()
    gmax = max(float(q.grad.abs().max()) for q in oracle.parameters() if q.grad is not None)
    for (n, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        if q.grad is not None:
            assert float((p.grad.cpu() - q.grad).abs().max()) / max(float(q.grad.abs().max()), 1e-3 * gmax) < 2e-3, n


@pytest.mark.parametrize("T,B", [(27, 256), (81, 128)])          # BASELINE.json configs[1] and configs[3]
def test_full_size_backward_is_invariant_to_clip_order(T, B):
    """BASELINE.json configs[1] / configs[3] size (26 layers, T=27, B=256 / T=81, B=128, bf16): the parameter gradient of the 3-term loss does not depend on the order of
    the clips in the batch (BatchNorm statistics and every weight-gradient reduction are sums over clips) -- up to summation-order noise.
    Exercises every backward kernel at the benchmark's shape, ragged tiles excluded; also checks the never-updated tensors stay untouched."""
    import kasportsformer_amd as K
    torch.manual_seed(114514)
    model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=T, compute_dtype="bf16").cuda().train()
    model.attach_param_grads = False
    x, y = O.synthetic_clips(B, T, seed=1234)
    x, y = x.cuda(), y.cuda()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    grads = []
    for xs, ys in ((x, y), (x[perm].contiguous(), y[perm].contiguous())):
        model._nbt.zero_()
        model.flat_grad = None                     # (no optimizer here: a second backward would otherwise accumulate, as p.grad does)
        loss, parts = K.loss3(model(xs), ys)
        loss.backward()
        torch.cuda.synchronize()
        g = model.flat_grad
        assert torch.isfinite(g).all() and torch.isfinite(parts).all()
        assert float(g[model.n_live:].abs().max()) == 0.0          # 208 norm1_limb tensors: no gradient
        grads.append((g[:model.n_live].clone(), float(loss.detach())))
    (g1, l1), (g2, l2) = grads
    assert abs(l1 - l2) < 1e-4 * abs(l1)
    cos = float((g1.double() @ g2.double()) / (g1.double().norm() * g2.double().norm()))
    assert cos > 0.9999, cos
    # per layer bucket: relative difference of the gradient vectors (bf16 activations: permuting rows changes which tile / partial sum a clip lands in)
    assert float((g1 - g2).norm() / g1.norm()) < 2e-2


# ---------------------------------------------------------------------------------------------------------------
# Discrete parts of the path: index gathers and the top-k adjacency (north_star: "bit-exact for the bone adjacency / index gathers")
# ---------------------------------------------------------------------------------------------------------------
def _mask_agreement(cd, L, T, B, seed):
    oracle, model = make_pair(L, T, cd)
    cap = {}
    for li, layer in enumerate(oracle.layers_with_bone):
        layer.graph_temporal.mixer.register_forward_pre_hook(lambda m, inp, k=li: cap.__setitem__(k, inp[0].detach()))
    x, _ = O.synthetic_clips(B, T, seed=seed)
    oracle.train(); model.train()
    oracle(x)
    with torch.no_grad():
        _, ws, _ = model._launch_forward(x.cuda(), False, keep=True)
    torch.cuda.synchronize()
    rows = mismatched = near_tie_rows = unexplained = bits = bits_equal = 0
    for li in range(L):
        g = cap[li].transpose(1, 2).reshape(B * 17, T, 128)                    # graph.py:104-112 on the oracle's own LN(x)
        want = O.temporal_topk_adjacency(g, 4).bool()
        got = decode_masks(ws_tensor(model, ws, B, f"L{li}.graph_temporal.adj_mask"), B * 17, T)
        sim = g @ g.transpose(1, 2)
        top = sim.topk(5, dim=-1)[0]
        near = (top[..., 3] - top[..., 4]).abs() <= 1e-5 * sim.abs().amax(dim=-1)          # 4th and 5th largest closer than fp32 summation-order noise
        bits += want.numel(); bits_equal += int((want == got).sum())
        bad = (want != got).any(dim=-1)
        rows += bad.numel(); mismatched += int(bad.sum()); near_tie_rows += int(near.sum()); unexplained += int((bad & ~near).sum())
        assert bool((got.sum(-1) >= 4).all()), "every row keeps at least its 4 largest similarities"
    return rows, mismatched, near_tie_rows, unexplained, 100.0 * bits_equal / bits


@pytest.mark.parametrize("L,T,B", [(2, 27, 3), (1, 81, 2), (1, 9, 4), (1, 243, 1), (1, 100, 1), (1, 5, 2), (1, 50, 2), (1, 256, 1)])
def test_temporal_topk_adjacency_masks_fp32_bit_exact(L, T, B):
    """fp32 mode: the stored adjacency bit masks equal torch's `sim >= topk(sim, 4)[..., -1:]` (graph.py:104-112) row for row.  The only rows
    allowed to differ are exact near-ties of the ORACLE's own similarities (4th and 5th largest within summation-order noise)."""
    rows, mismatched, near, unexplained, _ = _mask_agreement("fp32", L, T, B, seed=61)
    print(f"[fp32 T={T}] adjacency rows {rows}: {rows - mismatched} identical ({100.0 * (rows - mismatched) / rows:.4f} %), {near} near-tie rows")
    assert unexplained == 0 and mismatched <= near


@pytest.mark.parametrize("L,T,B", [(2, 27, 3), (1, 81, 2)])
def test_temporal_topk_adjacency_masks_bf16_agreement(L, T, B):
    """bf16 mode: similarities come from bf16-rounded LN(x).  The LN(x) rows of one joint track are nearly parallel -- the similarities of a row
    differ in the 3rd-4th digit -- so the 4th / 5th neighbour swap wherever they are closer than bf16 resolution: observed 65-67 % of the
    rows keep exactly the oracle's neighbour set, 97-98 % of the adjacency entries are equal.  SURVEY §8(d) asks for the agreement to be
    REPORTED; the floors asserted leave half of what is missing today."""
    rows, mismatched, near, _, entries = _mask_agreement("bf16", L, T, B, seed=61)
    agree = 100.0 * (rows - mismatched) / rows
    print(f"[bf16 T={T}] adjacency rows {rows}: {agree:.2f} % identical to the fp32 oracle's; entries {entries:.3f} % equal")
    assert agree >= 50.0 and entries >= 95.0


def test_bone_gather_is_bit_exact_on_integer_coordinates():
    """Joint j sits at the integer point (j + 1, 3 * (j + 1) * (j + 1)) times a per-frame power of two: every bone vector is then an exact
    integer pair that identifies its (child, parent) pair uniquely, and sqrt / divide are correctly rounded on both sides, so rows 0..15 of
    the bone tensor must equal the oracle's bit for bit; row 16 (a 16-term mean) to summation order."""
    _, model = make_pair(1, 27, "fp32")
    j = torch.arange(17, dtype=torch.float32) + 1
    frame = torch.stack((j, 3 * j * j, torch.ones(17)), dim=-1)                     # [17,3]
    scale = torch.tensor([2.0 ** (t % 5 - 2) for t in range(27)]).view(1, 27, 1, 1)
    x = (frame.view(1, 1, 17, 3) * torch.cat((scale, scale, torch.ones_like(scale)), dim=-1).expand(2, 27, 17, 3)).contiguous()
    model.train()
    with torch.no_grad():
        _, ws, _ = model._launch_forward(x.cuda(), False, keep=True)
    torch.cuda.synchronize()
    bone = ws_tensor(model, ws, 2, "bone3").view(2, 27, 17, 3).cpu()
    want = O.bone_decompose(x)
    assert torch.equal(bone[:, :, :16], want[:, :, :16])
    assert float((bone[:, :, 16] - want[:, :, 16]).abs().max()) < 1e-6 * float(want.abs().max())


# ---------------------------------------------------------------------------------------------------------------
# Full depth: the shipped 26-layer model against the oracle, both compute modes (every other oracle comparison is <= 2 layers)
# ---------------------------------------------------------------------------------------------------------------
def _full_depth_sample(cd, seed, salt):
    """One (input seed, weight salt) sample of the 26-layer comparison: (1) every top-4 neighbour decision of the 26 temporal GCN blocks row by row,
    (2) forward, per-layer activations, loss and all gradients with the oracle following the HIP path's neighbour decisions (tests/gpu_util.py
    forced_adjacency says why), (3) the free-running end-to-end deviation, reported."""
    oracle, model = make_pair(26, 27, cd, salt=salt)
    cap = oracle_stage_hooks(oracle)
    x, y = O.synthetic_clips(2, 27, seed=seed)
    oracle.train(); model.train()
    with torch.no_grad():
        free = oracle(x)                                          # the oracle's own decisions: for the report only
    with forced_adjacency(model, x) as fa:
        ref = oracle(x)
        l_ref, _ = O.loss_total(ref, y)
        l_ref.backward()
    pred = model(x.cuda())
    loss, _ = O.loss_total(pred, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    with torch.no_grad():
        buf = model._flat_buffers.clone()
        _, ws, _ = model._launch_forward(x.cuda(), False, keep=True)
        model._flat_buffers.copy_(buf)
    torch.cuda.synchronize()
    drift = [_abs_err(ws_tensor(model, ws, 2, f"L{li}.gate_out").float().view(cap[f"L{li}.gate_out"].shape), cap[f"L{li}.gate_out"]) /
             max(1.0, float(cap[f"L{li}.gate_out"].abs().max())) for li in range(26)]
    err = _abs_err(pred, ref) / max(1.0, float(ref.abs().max()))
    err_free = _abs_err(pred, free) / max(1.0, float(free.abs().max()))
    dots = [0.0, 0.0, 0.0]
    for (n, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if q.grad is not None:
            g, r = p.grad.detach().double().cpu(), q.grad.double()
            dots[0] += float((g * r).sum()); dots[1] += float((g * g).sum()); dots[2] += float((r * r).sum())
    cosine = dots[0] / (dots[1] ** 0.5 * dots[2] ** 0.5)
    print(f"[26 layers, {cd}, input seed {seed}, weight salt {salt}] {fa.summary()}\n    same neighbour decisions: forward rel err {err:.3e}, loss {loss.item():.6f} vs "
          f"{l_ref.item():.6f}, gradient cosine {cosine:.7f}, per-layer drift max {max(drift):.3e} (layer {drift.index(max(drift))}), first/last "
          f"{drift[0]:.2e}/{drift[-1]:.2e};  free-running oracle: forward rel err {err_free:.3e}")
    return {"seed": seed, "salt": salt, "forward_rel_err": err, "gradient_cosine": cosine, "loss": float(loss.item()), "loss_oracle": float(l_ref.item()),
            "drift_max": max(drift), "forward_rel_err_free_running": err_free, "topk_rows": fa.rows, "topk_rows_identical": fa.rows - fa.mismatched,
            "topk_rows_differ_not_near_tie": fa.unexplained, "topk_entries_equal_pct": 100.0 * fa.bits_equal / max(fa.bits, 1)}


def _write_parity_report(cd, samples):
    """The observed figures go to gpurun_out/r6_parity_26layers_<cd>.json (merged back from the GPU box; the copy committed under profiles/ is what
    bench.py quotes in its `parity` field instead of a typed-in string)."""
    import json, os, statistics
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    rep = {"test": "tests/test_gpu_model.py::test_full_depth_26_layers_against_oracle", "compute_dtype": cd, "n_layers": 26, "n_frames": 27, "batch": 2, "samples": samples,
           "forward_rel_err_median": statistics.median(s["forward_rel_err"] for s in samples), "forward_rel_err_max": max(s["forward_rel_err"] for s in samples),
           "gradient_cosine_median": statistics.median(s["gradient_cosine"] for s in samples), "gradient_cosine_min": min(s["gradient_cosine"] for s in samples),
           "topk_rows_identical_pct": 100.0 * sum(s["topk_rows_identical"] for s in samples) / max(1, sum(s["topk_rows"] for s in samples))}
    rep["forward_rel_err_free_running_max"] = max(s["forward_rel_err_free_running"] for s in samples)      # the oracle taking its OWN top-4 decisions
    rep["topk_rows_differ"] = sum(s["topk_rows"] - s["topk_rows_identical"] for s in samples)
    rep["topk_rows_differ_not_near_tie"] = sum(s["topk_rows_differ_not_near_tie"] for s in samples)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(out), "tools"))
    try:
        import stamp
        rep["stamp"] = stamp.stamp()             # which build these figures were observed on (bench.py quotes the file only if it matches the library it times)
    finally:
        sys.path.pop(0)
    with open(os.path.join(out, f"r6_parity_26layers_{cd}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    return rep


FULL_DEPTH_BF16_SAMPLES = ((5, 0), (6, 1001), (7, 2002), (8, 3003), (9, 4004), (10, 5005))         # (input seed, weight salt); (5, 0) is the single sample of rounds 1-3


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
def test_full_depth_26_layers_against_oracle(cd):
    """The shipped depth, de-identitied weights, HIP vs oracle.  fp32 mode: one sample at the arithmetic tolerance.  bf16 mode: SIX independent (input seed,
    weight salt) samples, all printed, judged on their median and their worst -- with O(1) layer scales this 26-layer network amplifies rounding noise
    chaotically, so one sample cannot tell a kernel regression from luck."""
    if cd == "fp32":
        s = _full_depth_sample(cd, 5, 0)
        _write_parity_report(cd, [s])
        assert s["topk_rows_differ_not_near_tie"] == 0, "a neighbour decision differs where the oracle's 4th and 5th similarities are NOT a near-tie"
        assert s["forward_rel_err"] < 1e-3 and s["drift_max"] < 1e-3 and s["gradient_cosine"] > 0.99999
        assert abs(s["loss"] - s["loss_oracle"]) < 1e-4 * max(1.0, abs(s["loss_oracle"]))
        return
    samples = [_full_depth_sample(cd, seed, salt) for seed, salt in FULL_DEPTH_BF16_SAMPLES]
    # what bf16 arithmetic ALONE does to the same samples: the rounding points of the bf16 mode emulated on the CPU oracle, no HIP kernel involved
    # (tests/studies/mixed_precision_study.py samples > tests/golden/bf16_emulation_26layers.json)
    import json, os
    emu = {(e["seed"], e["salt"]): e for e in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bf16_emulation_26layers.json")))["samples"]}
    for s_ in samples:
        e = emu[(s_["seed"], s_["salt"])]
        s_["emulated_forward_rel_err"], s_["emulated_gradient_cosine"] = e["forward_rel_err"], e["gradient_cosine"]
    rep = _write_parity_report(cd, samples)
    import statistics
    emu_cos = statistics.median(s_["emulated_gradient_cosine"] for s_ in samples)
    print(f"[26 layers, bf16, {len(samples)} samples] forward rel err median {rep['forward_rel_err_median']:.3e} / worst {rep['forward_rel_err_max']:.3e}; gradient cosine "
          f"median {rep['gradient_cosine_median']:.4f} / worst {rep['gradient_cosine_min']:.4f}; top-4 rows identical to the fp32 oracle's {rep['topk_rows_identical_pct']:.2f} %;  "
          f"CPU emulation of the same roundings: forward {statistics.median(s_['emulated_forward_rel_err'] for s_ in samples):.3e}, cosine {emu_cos:.4f} "
          f"({min(s_['emulated_gradient_cosine'] for s_ in samples):.4f} worst)")
    # bf16: activations are re-rounded to bf16 by every one of the 156 blocks; with O(1) layer scales this network amplifies rounding noise chaotically (DESIGN
    # section 10).  Judged against the emulation, sample by sample: forward error within 2 x its emulated one, gradient cosine no
    # more than 0.1 below its emulated one; the median cosine within 0.02 of the emulated median; and absolutely: median / worst forward 0.15 / 0.3, median /
    # worst cosine 0.9 / 0.85.  Observed in round 5 (the MLP's GELU / GELU' in fp16 instead of bf16-rounded: H carries 11 significant bits into GEMM2):
    # forward 0.081 / 0.225, cosine 0.955 / 0.936 -- every sample at or above its emulation (0.906 ... 0.942) except (7, 2002): 0.9356 against 0.9400.
    # Round 4: 0.910 / 0.716, that sample 0.22 below its emulation.
    for s_ in samples:
        assert s_["forward_rel_err"] < 2.0 * s_["emulated_forward_rel_err"], s_
        assert s_["gradient_cosine"] > s_["emulated_gradient_cosine"] - 0.1, s_
    assert rep["gradient_cosine_median"] > emu_cos - 0.02, (rep["gradient_cosine_median"], emu_cos)
    assert rep["forward_rel_err_median"] < 0.15 and rep["forward_rel_err_max"] < 0.3
    assert rep["gradient_cosine_median"] > 0.9 and rep["gradient_cosine_min"] > 0.85


@pytest.mark.parametrize("T,B", [(243, 1), (100, 2), (33, 2), (5, 3), (4, 8), (64, 2), (96, 2), (32, 2), (130, 1), (200, 1), (256, 1), (50, 2)])      # 64 / 96: the three-tile fused temporal forward with an empty / a full last tile; 32: the largest one-tile group
@pytest.mark.parametrize("cd,tol", [("fp32", 2e-3), ("bf16", BF16_TENSOR_TOL)])      # bf16, a handful of clips: per-tensor bar for tensors >= 256 elements, the tiny ones pooled; cosine >= 0.99977, forward 1.2e-2
def test_arbitrary_clip_lengths(cd, tol, T, B):
    """The reference builds for any n_frames (KASportsFormer.py:291-295, README.md:59): 243 is the long configuration of this model family;
    100 is past the MFMA attention cores (<= 96), 33 past the fused attention block (<= 32), 5 and 4 are the shortest clips whose rows
    still have four similarities to pick from (torch.topk(k=4) raises below that).  Forward, loss and every gradient against the oracle
    following the same neighbour decisions (tests/gpu_util.py forced_adjacency)."""
    oracle, model = make_pair(1, T, cd)
    x, y = O.synthetic_clips(B, T, seed=7)
    oracle.train(); model.train()
    with forced_adjacency(model, x) as fa:
        ref = oracle(x)
        l_ref, _ = O.loss_total(ref, y)
        l_ref.backward()
    pred = model(x.cuda())
    loss, _ = O.loss_total(pred, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    err = _abs_err(pred, ref) / max(1.0, float(ref.abs().max()))
    assert err < (1e-3 if cd == "fp32" else 0.025), err
    if cd == "fp32":
        assert fa.unexplained == 0, fa.summary()
    rep = compare_grads(model, oracle, cd)
    assert not rep["none_mismatch"], rep["none_mismatch"]
    print(f"[clip length T={T} B={B}, {cd}] forward err {err:.3e}, gradient cosine {rep['cosine']:.6f}, worst per-tensor error {rep['worst']:.3e} ({rep['worst_name']}), "
          f"pooled cosine of the tensors below 256 elements {rep['pooled_small_cosine']:.5f}")
    assert rep["cosine"] > (0.999999 if cd == "fp32" else 0.9995), rep["cosine"]
    assert rep["pooled_small_cosine"] > 0.999, rep["pooled_small_cosine"]             # bf16: observed >= 0.99969
    bad = sorted(((v, k) for k, v in rep["errors"].items() if not v < tol), reverse=True)
    assert not bad, bad[:8]
    model.eval()
    with torch.no_grad():
        assert torch.isfinite(model(x.cuda())).all()


@pytest.mark.parametrize("k,T", [(1, 27), (2, 27), (3, 81), (2, 100)])
def test_neighbour_num_other_than_four(k, T):
    """`neighbour_num` is a constructor argument of the reference (KASportsFormer.py:295 -> graph.py:104-112, torch.topk(k)); every yaml uses 4.  1..3 keep the
    k-th largest similarity of a row as the threshold (ties kept): the stored masks must equal the oracle's decision row for row in fp32 mode, and forward, loss and
    gradients must follow at the arithmetic tolerance."""
    oracle, model = make_pair(1, T, "fp32", neighbour_num=k)
    x, y = O.synthetic_clips(2, T, seed=23)
    oracle.train(); model.train()
    with forced_adjacency(model, x) as fa:
        ref = oracle(x)
        l_ref, _ = O.loss_total(ref, y)
        l_ref.backward()
    assert fa.unexplained == 0, fa.summary()
    assert all(bool((m.sum(-1) >= k).all()) for m in fa.masks) and any(bool((m.sum(-1) == k).any()) for m in fa.masks)
    pred = model(x.cuda())
    loss, _ = O.loss_total(pred, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert _abs_err(pred, ref) / max(1.0, float(ref.abs().max())) < 1e-3
    ref_grads = dict(oracle.named_parameters())
    gmax = max(float(q.grad.abs().max()) for q in ref_grads.values() if q.grad is not None)
    for n, p in model.named_parameters():
        r = ref_grads[n].grad
        assert (r is None) == (p.grad is None), n
        if r is not None:
            assert float((p.grad.cpu() - r).abs().max()) <= 2e-3 * max(float(r.abs().max()), 1e-3 * gmax), n
    import kasportsformer_amd as K
    with pytest.raises(NotImplementedError):
        K.KASportsFormer(n_layers=1, num_heads=8, neighbour_num=5)


@pytest.mark.parametrize("cd,tol", [("fp32", 2e-3), ("bf16", 0.4)])
def test_without_layer_scale(cd, tol):
    """use_layer_scale=False (KASportsFormer.py:98-101: x + mixer(norm1(x)), x + mlp(norm2(x)); no layer_scale_* entries in the state_dict).  The library keeps its
    layer-scale slices as the constant 1; the module exposes exactly the reference's parameters, and two optimizer steps (flat FusedAdamW) leave the constant alone."""
    import kasportsformer_amd as K
    oracle, model = make_pair(2, 27, cd, use_layer_scale=False)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys()) and not any("layer_scale" in k for k in model.state_dict())
    assert sum(p.numel() for p in model.parameters()) == sum(p.numel() for p in oracle.parameters())
    x, y = O.synthetic_clips(3, 27, seed=31)
    oracle.train(); model.train()
    with forced_adjacency(model, x):
        ref = oracle(x)
        l_ref, _ = O.loss_total(ref, y)
        l_ref.backward()
    pred = model(x.cuda())
    loss, _ = O.loss_total(pred, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    err = _abs_err(pred, ref) / max(1.0, float(ref.abs().max()))
    assert err < (1e-3 if cd == "fp32" else 0.03), err
    ref_grads = dict(oracle.named_parameters())
    gmax = max(float(q.grad.abs().max()) for q in ref_grads.values() if q.grad is not None)
    worst = 0.0
    for n, p in model.named_parameters():
        r = ref_grads[n].grad
        assert (r is None) == (p.grad is None), n
        if r is not None:
            worst = max(worst, float((p.grad.cpu() - r).abs().max() / max(float(r.abs().max()), (1e-3 if cd == "fp32" else 0.05) * gmax)))
    print(f"[no layer scale, {cd}] forward err {err:.3e}, worst per-tensor gradient error {worst:.3e}")
    assert worst < tol
    if cd == "fp32":
        model.attach_param_grads = False
        opt, ropt = K.FusedAdamW(model, lr=1e-3, weight_decay=0.01), torch.optim.AdamW(oracle.parameters(), lr=1e-3, weight_decay=0.01)
        for _ in range(2):
            opt.zero_grad(); ropt.zero_grad()
            with forced_adjacency(model, x):
                O.loss_total(oracle(x), y)[0].backward()
            K.loss3(model(x.cuda()), y.cuda())[0].backward()
            opt.step(); ropt.step()
        torch.cuda.synchronize()
        assert bool((model._flat[model._const_index] == 1.0).all())
        for (n, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
            assert float((p.detach().cpu() - q.detach()).abs().max()) <= 2e-3 * max(1.0, float(q.abs().max())), n


@pytest.mark.parametrize("heads", [4, 16, 2])
@pytest.mark.parametrize("cd,tol", [("fp32", 2e-3), ("bf16", 0.4)])      # bf16, two clips: observed 0.12-0.29 on tensors of >= 256 elements (the GCN U|V weights of the last layer)
def test_other_head_counts(cd, tol, heads):
    """`KASportsFormer()` without arguments builds with num_heads=4 (KASportsFormer.py:293; every yaml overrides it with 8): head dimension 32.
    Other head counts run the generic attention kernels in both modes; forward and all gradients against the oracle."""
    import kasportsformer_amd as K
    oracle = O.KASportsFormerOracle(n_layers=2, num_heads=heads, n_frames=27)
    sd = O.name_seeded_fill(oracle.state_dict())
    oracle.load_state_dict(sd, strict=True)
    model = K.KASportsFormer(n_layers=2, num_heads=heads, n_frames=27, compute_dtype=cd)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    oracle.train()
    x, y = O.synthetic_clips(2, 27, seed=17)
    with forced_adjacency(model, x):
        ref = oracle(x)
        O.loss_total(ref, y)[0].backward()
    pred = model(x.cuda())
    O.loss_total(pred, y.cuda())[0].backward()
    torch.cuda.synchronize()
    assert _abs_err(pred, ref) / max(1.0, float(ref.abs().max())) < (1e-3 if cd == "fp32" else 0.12)
    ref_grads = dict(oracle.named_parameters())
    gmax = max(float(q.grad.abs().max()) for q in ref_grads.values() if q.grad is not None)
    # bf16: tensors of a few elements at the end of the longest path (the 16-element fc2 weights of the limb MLPs) carry a gradient that is a small difference of
    # large noisy terms: their individual error is a sample of bf16 rounding noise (observed 0.23-0.47 from one equally accurate kernel formulation to the next), so
    # they are checked POOLED (cosine over all of them) and the per-tensor bar applies to tensors of at least 256 elements
    bad, worst_e, small = [], 0.0, [0.0, 0.0, 0.0]
    for n, p in model.named_parameters():
        r = ref_grads[n].grad
        assert (r is None) == (p.grad is None), n
        if r is not None:
            g = p.grad.detach().double().cpu()
            if cd == "bf16" and r.numel() < 256:
                small[0] += float((g * r.double()).sum()); small[1] += float((g * g).sum()); small[2] += float((r.double() ** 2).sum())
                continue
            e = float((g - r.double()).abs().max() / max(float(r.abs().max()), (1e-3 if cd == "fp32" else 0.05) * gmax))
            worst_e = max(worst_e, e)
            if not e < tol:
                bad.append((e, n))
    pooled = small[0] / max(1e-30, small[1] ** 0.5 * small[2] ** 0.5) if cd == "bf16" else 1.0
    print(f"[heads={heads}, {cd}] worst per-tensor gradient error {worst_e:.3e}, pooled cosine of the small tensors {pooled:.5f}")
    assert not bad, sorted(bad, reverse=True)[:8]
    assert pooled > 0.99, pooled


def test_bare_constructor_runs():
    """The reference's own defaults (26 layers, num_heads=4, T=27): a bare drop-in construction trains one step."""
    import kasportsformer_amd as K
    torch.manual_seed(1)
    m = K.KASportsFormer().cuda().train()
    x, y = (t.cuda() for t in O.synthetic_clips(4, 27, seed=2))
    loss, _ = K.loss3(m(x), y)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
