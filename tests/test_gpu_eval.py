"""Evaluation side on the GPU (kasf_joint_flip / kasf_tta_merge / kasf_eval_metrics through the C-ABI) against the oracle's numpy
restatement and the fixture captured from the reference (tests/golden/eval_L2_T27_B4.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import kasf_oracle as O
from gpu_util import make_pair

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_joint_flip_bit_exact():
    import kasportsformer_amd as K
    fx = np.load(os.path.join(GOLDEN, "functional.npz"))
    got = K.joint_flip(torch.from_numpy(fx["flip_in"]).cuda())
    assert np.array_equal(got.cpu().numpy(), fx["flip_out"])
    x = torch.randn(7, 27, 17, 3)
    xg = x.cuda()
    assert torch.equal(K.joint_flip(xg).cpu(), O.joint_flip(x))
    assert torch.equal(xg.cpu(), x)                                   # deep copy: input untouched (the eval loop reuses it)
    assert torch.equal(K.joint_flip(K.joint_flip(xg)).cpu(), x)       # involution
    K.joint_flip(xg, deep_copy=False)
    assert torch.equal(xg.cpu(), O.joint_flip(x))


@pytest.mark.parametrize("T", [27, 81, 9, 243, 256])      # 243 / 256: the clip staging area exceeds the 64 KB default LDS limit (99 / 104 KB)
def test_clip_metrics_match_oracle(T):
    import kasportsformer_amd as K
    B = 6
    _, y = O.synthetic_clips(B, T, seed=5)
    label_scaled, factor, res, _ = O.synthetic_test_extras(y, seed=6)
    pred = y + 0.05 * torch.randn(B, T, 17, 3)                      # root NOT zeroed: the kernel applies sp:55 itself
    mp, jp, ac, pm = (t.cpu().numpy() for t in K.clip_metrics(pred.cuda(), label_scaled, factor, res))
    pz = pred.clone()
    pz[:, :, 0] = 0
    for i in range(B):
        m, j, a, p = O.clip_metrics(pz[i].numpy(), label_scaled[i].numpy(), factor[i].numpy(), (int(res[i][0]), int(res[i][1])))
        assert np.allclose(mp[i], m, rtol=1e-5, atol=1e-4)
        assert np.allclose(jp[i], j, rtol=1e-5, atol=1e-4)
        assert np.allclose(ac[i], a, rtol=1e-5, atol=1e-4)
        assert np.allclose(pm[i], p, rtol=1e-4, atol=1e-3)             # numpy runs the 3x3 SVD in float32


def test_p_mpjpe_reflection_and_planar_cases():
    """det(R) < 0 (mirrored prediction) exercises the reflection fix; planar poses have a zero singular value."""
    import kasportsformer_amd as K
    B, T = 3, 9
    g = torch.Generator().manual_seed(3)
    gt = torch.randn(B, T, 17, 3, generator=g) * 200
    gt = gt - gt[:, :, :1]
    pred = gt.clone() / 656.0                                          # w = 1312: de-normalisation multiplies by 656
    pred[0, ..., 2] *= -1                                              # clip 0: mirrored in z
    pred[1, ..., 2] = 0                                                # clip 1: planar prediction
    pred = pred + 0.01 * torch.randn(B, T, 17, 3, generator=g)
    factor = torch.ones(B, T)
    res = torch.tensor([[1312, 1216]] * B)
    _, _, _, pm = K.clip_metrics(pred.cuda(), gt, factor, res)
    pz = pred.clone()
    pz[:, :, 0] = 0
    for i in range(B):
        pd, gtc = pz[i].numpy().astype(np.float64).copy(), gt[i].numpy().astype(np.float64)
        pd[:, :, :2] = (pd[:, :, :2] + np.array([1, 1216 / 1312])) * 656
        pd[:, :, 2:] *= 656
        pd = pd - pd[:, :1]
        ref = O.p_mpjpe(pd, gtc)                                       # float64 SVD
        assert np.allclose(pm[i].cpu().numpy(), ref, rtol=1e-4, atol=1e-3), (i, pm[i].cpu().numpy()[:3], ref[:3])


def test_evaluation_matches_reference_fixture_and_oracle():
    """Full procedure: HIP model (fp32 mode) + flip-TTA + fused metrics vs the fixture made with the reference model and metric functions."""
    import kasportsformer_amd as K
    fx = np.load(os.path.join(GOLDEN, "eval_L2_T27_B4.npz"))
    _, model = make_pair(2, 27, "fp32")
    model.eval()
    x = torch.from_numpy(fx["x"]).cuda()
    x_before = x.clone()
    for tag, flip in (("tta", True), ("plain", False)):
        pred = K.predict_flip_tta(model, x, flip=flip)
        assert torch.equal(x, x_before)
        assert np.abs(pred.cpu().numpy() - fx[f"pred_{tag}"]).max() < 1e-3
        assert float(pred[:, :, 0].abs().max()) == 0.0
        pred[:, :, 0, :] = 0                                           # callers edit the prediction in place (sp:55): must be writable
    # two batches of two clips through the loader-style entry point
    batches = [(torch.from_numpy(fx["x"][i:i + 2]), torch.from_numpy(fx["label_scaled"][i:i + 2]), torch.from_numpy(fx["factor"][i:i + 2]),
                list(fx["actions"][i:i + 2]), torch.from_numpy(fx["res"][i:i + 2])) for i in (0, 2)]
    r = K.evaluate_one_epoch(model, batches, flip=True)
    assert r["activity_name_sequence"] == list(fx["activity_name_sequence"])
    for key, name in (("mpjpe", "mpjpe"), ("p_mpjpe", "p_mpjpe"), ("acceleration_error", "acc")):
        assert abs(r[key] - float(fx[f"tta_{name}"])) < 1e-4 * float(fx[f"tta_{name}"]) + 2e-3 * 656, (key, r[key], float(fx[f"tta_{name}"]))
    # metrics alone on the reference's own prediction: tight
    ev = K.Evaluator()
    ev.update(torch.from_numpy(fx["pred_tta"]).cuda(), torch.from_numpy(fx["label_scaled"]), torch.from_numpy(fx["factor"]), list(fx["actions"]),
              fx["res"])
    r2 = ev.result()
    for key, name in (("mpjpe", "mpjpe"), ("p_mpjpe", "p_mpjpe"), ("acceleration_error", "acc")):
        assert abs(r2[key] - float(fx[f"tta_{name}"])) < 2e-5 * float(fx[f"tta_{name}"]), (key, r2[key], float(fx[f"tta_{name}"]))
    assert np.allclose(r2["mpjpe_joint"], fx["tta_mpjpe_joint"], rtol=2e-5)
    assert np.allclose(r2["mpjpe_activity"], fx["tta_mpjpe_activity"], rtol=2e-5)
    # the same against the reference's own evaluation loop (evaluate_one_epoch_new run by tests/golden/make_golden.py `evalloop`)
    for key, name in (("mpjpe", "mpjpe"), ("p_mpjpe", "p_mpjpe"), ("acceleration_error", "acc")):
        assert abs(r2[key] - float(fx[f"refloop_tta_{name}"])) < 2e-5 * float(fx[f"refloop_tta_{name}"]), (key, r2[key])
    assert np.allclose(r2["mpjpe_joint"], fx["refloop_tta_mpjpe_joint"], rtol=2e-5)


def test_stacked_tta_forward_equals_two_forwards_and_train_mode_is_kept_separate():
    import kasportsformer_amd as K
    _, model = make_pair(1, 27, "bf16")
    x, _ = O.synthetic_clips(3, 27, seed=8)
    x = x.cuda()
    model.eval()
    with torch.no_grad():
        a = K.predict_flip_tta(model, x)
        b = (model(x) + K.joint_flip(model(K.joint_flip(x)))) / 2
        b[:, :, 0] = 0
    assert torch.equal(a, b)                                          # eval-mode rows are independent: stacking changes nothing, bit for bit


def test_full_size_properties_at_baseline_config():
    """BASELINE.json configs[1] size (26 layers, T=27, B=256, bf16), through properties that need no oracle run:
    (1) a clip's evaluation-mode prediction does not depend on its batch (bit-exact), (2) flip-TTA commutes with the flip:
    tta(flip(x)) == flip(tta(x)) bit for bit, root joint exactly zero, (3) the metric kernel is invariant to the same symmetry."""
    import kasportsformer_amd as K
    torch.manual_seed(114514)
    model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().eval()
    x, y = O.synthetic_clips(256, 27, seed=1234)
    x = x.cuda()
    with torch.no_grad():
        full = model(x)
        for i in (0, 101, 255):
            assert torch.equal(full[i:i + 1], model(x[i:i + 1])), i
        a = K.predict_flip_tta(model, x)
        b = K.predict_flip_tta(model, K.joint_flip(x))
    assert torch.isfinite(a).all() and float(a[:, :, 0].abs().max()) == 0.0
    assert torch.equal(K.joint_flip(a), b)
    label_scaled, factor, res, _ = O.synthetic_test_extras(y, seed=7)
    m1 = K.clip_metrics(a, label_scaled, factor, res)
    m2 = K.clip_metrics(b, K.joint_flip(label_scaled.cuda()), factor, res)          # mirrored prediction vs mirrored ground truth
    perm = torch.tensor([0, 4, 5, 6, 1, 2, 3, 7, 8, 9, 10, 14, 15, 16, 11, 12, 13], device="cuda")
    assert torch.allclose(m1[0], m2[0], rtol=1e-5, atol=1e-4)                         # MPJPE
    assert torch.allclose(m1[1][:, :, perm], m2[1], rtol=1e-5, atol=1e-4)             # per-joint errors, left/right swapped
    assert torch.allclose(m1[2], m2[2], rtol=1e-5, atol=1e-4)                         # acceleration error
    assert torch.allclose(m1[3], m2[3], rtol=1e-4, atol=1e-3)                         # P-MPJPE: a reflection is undone by ... the reflection fix
