"""SURVEY §8(d) "MPJPE vs ref": train both backends from identical weights on the same clip order, run the evaluation procedure on
both, compare MPJPE in millimetres.  Also the checkpoint hand-over in both directions in the middle of a run (§8(f) row 3)."""
import numpy as np
import pytest
import torch

from oracle import kasf_oracle as O
from gpu_util import make_pair

pytestmark = pytest.mark.gpu
L, T, B, STEPS = 2, 27, 8, 16
_ORACLE_RUNS = {}        # init -> (oracle state after STEPS, its evaluation): shared by the fp32 and bf16 cases


def _data():
    # learnable labels (a fixed seeded map of the 2-D pose): independent-noise labels keep MPJPE at the label scale whatever the model does
    xs, ys = O.teacher_clips(B * 4, T, seed=77)
    xt, yt = O.teacher_clips(6, T, seed=78)
    label_scaled, factor, res, actions = O.synthetic_test_extras(yt, seed=79, noise_mm=2.0)
    return xs, ys, xt, (label_scaled, factor, res, actions)


def _oracle_eval(oracle, xt, extras):
    label_scaled, factor, res, actions = extras
    oracle.eval()
    with torch.no_grad():
        pred = O.predict_flip_tta(oracle, xt)
    oracle.train()
    return O.evaluate_batches([(pred.numpy(), label_scaled.numpy(), factor.numpy(), actions, res.numpy())])


def _hip_eval(K, model, xt, extras):
    label_scaled, factor, res, actions = extras
    r = K.evaluate_one_epoch(model, [(xt, label_scaled, factor, actions, res)], flip=True)
    model.train()
    return r


def _default_init_pair(cd):
    """The reference's own initialisation (layer-scale 1e-5, zero gate weights ...) under the yaml seed: the regime real training runs in."""
    import kasportsformer_amd as K
    torch.manual_seed(114514)
    oracle = O.KASportsFormerOracle(n_layers=L, num_heads=8, n_frames=T)
    model = K.KASportsFormer(n_layers=L, num_heads=8, n_frames=T, compute_dtype=cd)
    model.load_state_dict(oracle.state_dict(), strict=True)
    return oracle, model.cuda()


# Labels are LEARNABLE (teacher_labels), so MPJPE moves (208 -> 173 mm in these 16 steps) and a deviation shows; on the independent-noise labels of
# rounds 1-2 it sat at the label scale whatever the model did.  fp32 from the reference's default init: the SURVEY §8(d) bar, |dMPJPE| <= 0.1 mm
# (observed 1e-4).  With de-identitied (fully random, O(1) layer scales) weights the temporal top-4 neighbour choice sits on near-ties, one flipped
# neighbour changes a token by O(1) and the runs drift apart chaotically (the reference on two machines would too): fp32 observed 0.43 mm of 207.
# bf16 (bars = observed x 2..3): default init 0.55 mm of 173 -- the regime real training runs in; profiles/r3_train_fidelity.json follows it for
# 1,000 steps at full depth: final MPJPE 27.9 mm against fp32's 28.0 -- ; de-identitied weights 20 mm of 207: a third of the temporal GCN rows pick
# another 4th neighbour in bf16 (test_temporal_topk_adjacency_masks_bf16_agreement), which this untrained random network amplifies.
@pytest.mark.parametrize("init,cd,tol_mm", [("default", "fp32", 0.1), ("seeded", "fp32", 1.0), ("default", "bf16", 1.5), ("seeded", "bf16", 40.0)])
def test_training_then_evaluation_tracks_oracle(init, cd, tol_mm):
    import kasportsformer_amd as K
    oracle, model = make_pair(L, T, cd) if init == "seeded" else _default_init_pair(cd)
    xs, ys, xt, extras = _data()
    opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
    model.train()
    for s in range(STEPS):
        o = (s % 4) * B
        opt.zero_grad()
        K.loss3(model(xs[o:o + B].cuda()), ys[o:o + B].cuda())[0].backward()
        opt.step()
    # the CPU side (half a minute per run) is the same for both arithmetic modes of one initialisation: train the oracle once per init
    if init not in _ORACLE_RUNS:
        topt = torch.optim.AdamW(oracle.parameters(), lr=5e-4, weight_decay=0.01)
        oracle.train()
        for s in range(STEPS):
            o = (s % 4) * B
            topt.zero_grad()                                       # train_and_evaluate_sp.py:208-243 order
            O.loss_total(oracle(xs[o:o + B]), ys[o:o + B])[0].backward()
            topt.step()
        _ORACLE_RUNS[init] = ({k: v.clone() for k, v in oracle.state_dict().items()}, _oracle_eval(oracle, xt, extras))
    else:
        oracle.load_state_dict(_ORACLE_RUNS[init][0], strict=True)
    ref, got = _ORACLE_RUNS[init][1], _hip_eval(K, model, xt, extras)
    d = {k: abs(got[k] - float(ref[k])) for k in ("mpjpe", "p_mpjpe", "acceleration_error")}
    print(f"[{init} init, {cd}] after {STEPS} steps: MPJPE oracle {float(ref['mpjpe']):.4f} mm, HIP {got['mpjpe']:.4f} mm, deltas {d}")
    # P-MPJPE this early aligns near-collapsed predictions: the fitted rotation amplifies small differences, so it gets ten times the room
    assert d["mpjpe"] <= tol_mm and d["p_mpjpe"] <= 10 * tol_mm, (d, float(ref["mpjpe"]))
    osd, msd = oracle.state_dict(), model.state_dict()
    for n in osd:      # BatchNorm running statistics drive the evaluation-mode forward (a missed or mis-weighted update would be an O(1) error)
        if n.endswith("running_var") or n.endswith("running_mean"):
            # seeded weights: a flipped top-4 neighbour (see above) shifts the statistics of that layer's temporal mixer by ~10 %; run-to-run noise of
            # the fp32 atomics decides whether it happens, so that case gets the bf16 room
            loose = cd == "bf16" or init == "seeded"
            assert torch.allclose(msd[n].cpu(), osd[n], rtol=0.1 if loose else 5e-2, atol=0.1 if loose else 5e-3), n
        if n.endswith("num_batches_tracked"):
            assert int(msd[n]) == int(osd[n]) == STEPS


def test_checkpoint_handover_both_directions(tmp_path):
    """HIP run -> checkpoint -> torch AdamW continues on the CPU oracle; oracle run -> reference-style checkpoint -> HIP continues."""
    import kasportsformer_amd as K
    oracle, model = _default_init_pair("fp32")            # reference default init: no near-tie neighbour flips between the two backends
    xs, ys, _, _ = _data()
    opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
    topt = torch.optim.AdamW(oracle.parameters(), lr=5e-4, weight_decay=0.01)
    model.train()
    oracle.train()

    def hip_step(m, o, s):
        o.zero_grad()
        K.loss3(m(xs[s * B:(s + 1) * B].cuda()), ys[s * B:(s + 1) * B].cuda())[0].backward()
        o.step()

    def cpu_step(m, o, s):
        o.zero_grad()
        O.loss_total(m(xs[s * B:(s + 1) * B]), ys[s * B:(s + 1) * B])[0].backward()
        o.step()

    for s in range(2):
        hip_step(model, opt, s)
        cpu_step(oracle, topt, s)
    slots = {id(p): i for i, p in enumerate(model.parameters())}
    # direction 1: HIP -> file -> fresh oracle + torch AdamW (what the reference's resume does, sp:285-297): the state arrives exactly
    path = tmp_path / "hip_latest.pth"
    K.checkpoint_save(str(path), 0, 5e-4, opt, model, 99.0, "w1")
    ck = torch.load(str(path), map_location="cpu", weights_only=True)
    oracle2 = O.KASportsFormerOracle(n_layers=L, num_heads=8, n_frames=T).train()
    oracle2.load_state_dict(K.strip_module_prefix(ck["model"]), strict=True)
    topt2 = torch.optim.AdamW(oracle2.parameters(), lr=1e-3)
    topt2.load_state_dict(ck["optimizer"])
    for (n, a), (_, b) in zip(oracle2.state_dict().items(), model.state_dict().items()):
        assert torch.equal(a, b.cpu()), n
    st = topt2.state_dict()["state"]
    for p, off, numel, shape in model._live:
        e = st[slots[id(p)]]
        assert float(e["step"]) == 2 and torch.equal(e["exp_avg"], opt.exp_avg[off:off + numel].view(shape).cpu())
        assert torch.equal(e["exp_avg_sq"], opt.exp_avg_sq[off:off + numel].view(shape).cpu())
    assert topt2.param_groups[0]["lr"] == 5e-4
    # ... and the next step from that state agrees with the HIP run continuing (never further apart than one AdamW step, lr)
    cpu_step(oracle2, topt2, 2)
    hip_step(model, opt, 2)
    msd = model.state_dict()
    worst = max((float((p.detach() - msd[n].cpu()).abs().max()), n) for n, p in oracle2.named_parameters())
    assert worst[0] < 5e-4, worst
    # direction 2: oracle -> reference-style file -> fresh HIP model + FusedAdamW
    ref_path = tmp_path / "ref_latest.pth"
    torch.save({"epoch": 3, "learning_rate": 5e-4, "optimizer": topt.state_dict(), "model": {"module." + k: v for k, v in oracle.state_dict().items()},
                "min_mpjpe": 42.0, "wandb_id": "w2"}, str(ref_path))
    model2 = K.KASportsFormer(n_layers=L, num_heads=8, n_frames=T, compute_dtype="fp32").cuda().train()
    opt2 = K.FusedAdamW(model2, lr=1e-3)
    info = K.checkpoint_load(str(ref_path), model2, opt2, resume=True)
    assert info["epoch"] == 3 and opt2.step_index == 2 and opt2.param_groups[0]["lr"] == 5e-4
    for (n, a), (_, b) in zip(oracle.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b.cpu()), n
    st = topt.state_dict()["state"]
    slots2 = {id(p): i for i, p in enumerate(model2.parameters())}
    for p, off, numel, shape in model2._live:
        assert torch.equal(st[slots2[id(p)]]["exp_avg"], opt2.exp_avg[off:off + numel].view(shape).cpu())
    hip_step(model2, opt2, 3)
    cpu_step(oracle, topt, 3)
    ref = dict(oracle.named_parameters())
    worst = max((float((p.detach().cpu() - ref[n].detach()).abs().max()), n) for n, p in model2.named_parameters())
    assert worst[0] < 5e-4, worst
