"""The CPU oracle (oracle/kasf_oracle.py) against fixtures captured from the real reference
(tests/golden/make_golden.py).  This is what pins parity: the GPU tests compare the HIP path
with this oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import kasf_oracle as O

TOL = 1e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _build(fx):
    T, L = int(fx["T"]), int(fx["n_layers"])
    m = O.KASportsFormerOracle(n_layers=L, num_heads=8, n_frames=T)
    m.load_state_dict(O.name_seeded_fill(m.state_dict()), strict=True)
    return m


@pytest.mark.parametrize("name", ["model_L2_T27_B2.npz", "model_L1_T81_B1.npz"])
def test_model_forward_backward_matches_reference(golden_dir, name):
    fx = _load(golden_dir, name)
    m = _build(fx)
    x, y = torch.from_numpy(fx["x"]), torch.from_numpy(fx["y"])
    # synthetic generator is deterministic
    x2, y2 = O.synthetic_clips(x.shape[0], x.shape[1], seed=1234)
    assert torch.equal(x, x2) and torch.equal(y, y2)

    m.eval()
    with torch.no_grad():
        assert np.abs(m(x).numpy() - fx["pred_eval"]).max() < TOL
        assert np.abs(m(x, return_rep=True)[0].numpy() - fx["rep_eval_b0"]).max() < TOL

    m.train()
    stages = {}
    if "stage/limb" in fx:
        L0 = m.layers_with_bone[0]
        for kind in O.BLOCK_KINDS:
            getattr(L0, kind).register_forward_hook(
                lambda mod, i, o, k=kind: stages.__setitem__("stage/L0." + k, o[0].detach().numpy()))
        for li, layer in enumerate(m.layers_with_bone):
            layer.register_forward_hook(lambda mod, i, o, k=li: stages.__setitem__("stage/layer%d" % k, o[0].detach().numpy()))
        m.bone_refusion.register_forward_hook(lambda mod, i, o: stages.__setitem__("stage/limb", o.detach().numpy()))
    pred = m(x)
    loss, parts = O.loss_total(pred, y)
    loss.backward()
    assert np.abs(pred.detach().numpy() - fx["pred_train"]).max() < TOL
    for k, v in stages.items():
        assert np.abs(v - fx[k]).max() < 5e-5, k
    got = np.array([loss.item()] + [p.item() for p in parts])
    assert np.abs(got - fx["losses"]).max() < 1e-5
    # gradients: None-ness, sums and strided samples for every parameter
    n_none = 0
    for n, p in m.named_parameters():
        if "gnone/" + n in fx:
            assert p.grad is None, n
            n_none += 1
            continue
        g = p.grad.reshape(-1)
        step = max(1, g.numel() // 256)
        smp = g[::step][:256].numpy()
        ref = fx["gsmp/" + n]
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(smp - ref).max() < 2e-4 * scale, (n, np.abs(smp - ref).max())
        s = fx["gsum/" + n]
        assert abs(g.double().abs().sum().item() - s[1]) <= 1e-3 * max(1.0, s[1]), n
    assert n_none == 8 * int(fx["n_layers"])
    for n, b in m.named_buffers():
        assert np.abs(b.numpy().astype(np.float64) - fx["buf/" + n].astype(np.float64)).max() < TOL, n


def test_state_dict_manifest_matches_reference(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))
    m = O.KASportsFormerOracle(n_layers=26, num_heads=8, n_frames=27)
    sd = m.state_dict()
    mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
    assert len(mine) == 2975
    assert sorted(map(tuple, map(lambda e: (e[0], tuple(e[1]), e[2]), mine))) == \
        sorted((e[0], tuple(e[1]), e[2]) for e in man["entries"])
    assert [e[0] for e in mine] == [e[0] for e in man["entries"]], "key order differs"
    assert sum(p.numel() for p in m.parameters()) == man["n_params"] == 29365668
    assert len(man["grad_none"]) == 208


def test_functional_pins(golden_dir):
    fx = _load(golden_dir, "functional.npz")
    out = O.bone_decompose(torch.from_numpy(fx["bone_in"])).numpy()
    assert np.abs(out - fx["bone_out"]).max() < 1e-6
    assert np.allclose(out[0, 0, 0], [0, 0, 1])              # zero-length bone -> (0,0,1)
    for name, fn in (("mpjpe", O.loss_mpjpe), ("n_mpjpe", O.loss_n_mpjpe), ("velocity", O.loss_velocity)):
        p = torch.from_numpy(fx["loss_pred"]).requires_grad_(True)
        v = fn(p, torch.from_numpy(fx["loss_tgt"]))
        v.backward()
        assert abs(v.item() - float(fx["loss_" + name])) < 1e-6
        assert np.abs(p.grad.numpy() - fx["loss_" + name + "_grad"]).max() < 1e-7
    a, b = fx["met_pred"], fx["met_tgt"]
    assert np.allclose(O.mpjpe(a, b), fx["met_mpjpe"])
    assert np.allclose(O.jpe(a, b), fx["met_jpe"])
    assert np.allclose(O.acc_error(a, b), fx["met_acc"])
    assert np.allclose(O.p_mpjpe(a.copy(), b.copy()), fx["met_pmpjpe"])
    assert np.array_equal(O.joint_flip(torch.from_numpy(fx["flip_in"])).numpy(), fx["flip_out"])
    f = torch.from_numpy(fx["flip_in"])
    assert torch.equal(O.joint_flip(O.joint_flip(f)), f)


def test_evaluation_procedure_matches_reference_fixture(golden_dir):
    """Reference model + reference metric functions (fixture) vs oracle model + oracle metrics + restated evaluation loop
    (train_and_evaluate_sp.py:27-149); SURVEY §8(d) config 1 at 2 layers."""
    fx = _load(golden_dir, "eval_L2_T27_B4.npz")
    m = O.KASportsFormerOracle(n_layers=2, num_heads=8, n_frames=27)
    m.load_state_dict(O.name_seeded_fill(m.state_dict()), strict=True)
    m.eval()
    x, y = O.synthetic_clips(4, 27, seed=99)
    label_scaled, factor, res, actions = O.synthetic_test_extras(y, seed=98)
    assert np.array_equal(x.numpy(), fx["x"]) and np.array_equal(label_scaled.numpy(), fx["label_scaled"])
    assert actions == list(fx["actions"]) and np.array_equal(res.numpy(), fx["res"])
    with torch.no_grad():
        for tag, flip in (("tta", True), ("plain", False)):
            pred = O.predict_flip_tta(m, x, flip=flip)
            assert np.abs(pred.numpy() - fx[f"pred_{tag}"]).max() < TOL
            r = O.evaluate_batches([(fx[f"pred_{tag}"], fx["label_scaled"], fx["factor"], actions, fx["res"])])
            assert r["activity_name_sequence"] == list(fx["activity_name_sequence"])
            for key, name in (("mpjpe", "mpjpe"), ("p_mpjpe", "p_mpjpe"), ("acceleration_error", "acc")):
                assert abs(float(r[key]) - float(fx[f"{tag}_{name}"])) < 1e-5 * float(fx[f"{tag}_{name}"])
            assert np.allclose(r["mpjpe_joint"], fx[f"{tag}_mpjpe_joint"], rtol=1e-5)
            assert np.allclose(r["mpjpe_activity"], fx[f"{tag}_mpjpe_activity"], rtol=1e-5)
            # ... and against what the reference's OWN loop (evaluate_one_epoch_new, imported by make_golden.py `evalloop`) returned for the same batches
            for key, name in (("mpjpe", "mpjpe"), ("p_mpjpe", "p_mpjpe"), ("acceleration_error", "acc")):
                assert abs(float(r[key]) - float(fx[f"refloop_{tag}_{name}"])) < 1e-5 * float(fx[f"refloop_{tag}_{name}"])
            assert np.allclose(r["mpjpe_joint"], fx[f"refloop_{tag}_mpjpe_joint"], rtol=1e-5)
    per = [O.clip_metrics(fx["pred_tta"][i], fx["label_scaled"][i], fx["factor"][i], tuple(int(v) for v in fx["res"][i])) for i in range(4)]
    for k, name in enumerate(("clip_mpjpe", "clip_jpe", "clip_acc", "clip_pmpjpe")):
        assert np.allclose(np.stack([q[k] for q in per]), fx[name], rtol=1e-5, atol=1e-4)
