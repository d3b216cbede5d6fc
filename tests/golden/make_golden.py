#!/usr/bin/env python3
"""Generate golden fixtures from the REAL reference (run in the build container only).

    python tests/golden/make_golden.py          # needs /root/reference

Imports jw0r1n/KASportsFormer from /root/reference (CPU, fp32) with a stand-in for
the one missing import (``timm.models.layers.DropPath`` -- inert at drop_path=0,
model/KASportsFormer.py:12,96), fills every parameter/buffer BY NAME from
``oracle.kasf_oracle.name_seeded_fill`` (so no constructor-RNG order matters and
the 'identity at init' vacuity is removed), and stores inputs / outputs /
sampled gradients as ``.npz`` data.  Only tensors are written: no reference
source, bytecode or pickled modules ever enter the repository.
"""
import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import kasf_oracle as O  # noqa: E402  (fill + synthetic inputs are shared test infra)


def import_reference():
    timm, models, layers = (types.ModuleType(n) for n in ("timm", "timm.models", "timm.models.layers"))

    class DropPath(nn.Module):
        def __init__(self, p=None):
            super().__init__()
            self.drop_prob = p

        def forward(self, x):
            assert not (self.training and self.drop_prob)
            return x

    layers.DropPath = DropPath
    timm.models = models
    models.layers = layers
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    sys.path.insert(0, "/root/reference")
    from model.KASportsFormer import KASportsFormer, bone_decomposer
    import utils.loss_calc as LC
    import utils.error_calc as EC
    return KASportsFormer, bone_decomposer, LC, EC


def grad_samples(named_grads):
    """Per-parameter compact gradient pin: sum, abs-sum (f64) and a strided sample (<=256)."""
    out = {}
    for name, g in named_grads:
        if g is None:
            out["gnone/" + name] = np.zeros(0, np.float32)
            continue
        f = g.detach().reshape(-1)
        step = max(1, f.numel() // 256)
        out["gsum/" + name] = np.array([f.double().sum().item(), f.double().abs().sum().item()])
        out["gsmp/" + name] = f[::step][:256].numpy().copy()
    return out


def model_fixture(Ref, LC, path, n_layers, B, T, stages):
    torch.manual_seed(0)
    m = Ref(n_layers=n_layers, dim_in=3, dim_feat=128, dim_rep=512, dim_out=3, mlp_ratio=4,
            num_heads=8, n_frames=T)
    m.load_state_dict(O.name_seeded_fill(m.state_dict()), strict=True)
    x, y = O.synthetic_clips(B, T, seed=1234)
    out = {"x": x.numpy(), "y": y.numpy(), "n_layers": np.array(n_layers), "T": np.array(T)}

    # ---- eval-mode forward (running-stat BN) ----
    m.eval()
    with torch.no_grad():
        out["pred_eval"] = m(x).numpy()
        out["rep_eval_b0"] = m(x, return_rep=True)[0].numpy()

    # ---- train-mode forward + 3-term loss + backward ----
    m.train()
    captured = {}
    hooks = []
    if stages:
        L0 = m.layers_with_bone[0]
        for kind in O.BLOCK_KINDS:
            hooks.append(getattr(L0, kind).register_forward_hook(
                lambda mod, inp, o, k=kind: captured.__setitem__("stage/L0." + k, o[0].detach().numpy().copy())))
        for li, layer in enumerate(m.layers_with_bone):
            hooks.append(layer.register_forward_hook(
                lambda mod, inp, o, k=li: captured.__setitem__("stage/layer%d" % k, o[0].detach().numpy().copy())))
        hooks.append(m.bone_refusion.register_forward_hook(
            lambda mod, inp, o: captured.__setitem__("stage/limb", o.detach().numpy().copy())))
    pred = m(x)      # the reference mutates an intermediate in place (KASportsFormer.py:51): x cannot require grad
    l1 = LC.mpjpe_loss_calc(pred, y)
    l2 = LC.n_mpjpe_loss_calc(pred, y)
    l3 = LC.velocity_loss_calc(pred, y)
    loss = l1 + 0.5 * l2 + 20.0 * l3           # train_and_evaluate_sp.py:222, yaml :30-31
    loss.backward()
    for h in hooks:
        h.remove()
    out.update(captured)
    out["pred_train"] = pred.detach().numpy()
    out["losses"] = np.array([loss.item(), l1.item(), l2.item(), l3.item()], dtype=np.float64)
    out.update(grad_samples((n, p.grad) for n, p in m.named_parameters()))
    for n, b in m.named_buffers():              # updated BN running statistics
        out["buf/" + n] = b.detach().numpy().copy()
    np.savez_compressed(path, **out)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


def eval_fixture(Ref, EC):
    left, right = [1, 2, 3, 14, 15, 16], [4, 5, 6, 11, 12, 13]
    # 4. evaluation procedure (SURVEY §8(d) config 1 at 2 layers): REFERENCE model + REFERENCE metric functions; the loop around them
    #    (train_and_evaluate_sp.py:27-149) cannot be imported (wandb, easydict) and is the restatement in oracle.evaluate_batches.
    torch.manual_seed(0)
    ref = Ref(n_layers=2, dim_in=3, dim_feat=128, dim_rep=512, dim_out=3, mlp_ratio=4, num_heads=8, n_frames=27)
    sd = O.name_seeded_fill(ref.state_dict())
    ref.load_state_dict(sd)
    ref.eval()
    x, y = O.synthetic_clips(4, 27, seed=99)
    label_scaled, factor, res, actions = O.synthetic_test_extras(y, seed=98)

    def flip(t):
        t = t.clone()
        t[..., 0] *= -1
        t[..., left + right, :] = t[..., right + left, :]
        return t
    with torch.no_grad():
        p0 = ref(x.clone())
        p1 = flip(ref(flip(x)))
        pred_tta, pred_plain = (p0 + p1) / 2, p0.clone()
    pred_tta[:, :, 0, :] = 0
    pred_plain[:, :, 0, :] = 0
    saved = (O.mpjpe, O.jpe, O.acc_error, O.p_mpjpe)
    O.mpjpe, O.jpe, O.acc_error, O.p_mpjpe = EC.mpjpe_calc, EC.jpe_calc, EC.acc_error_calc, EC.p_mpjpe_calc
    ev = {}
    for tag, pr in (("tta", pred_tta), ("plain", pred_plain)):
        r = O.evaluate_batches([(pr.numpy(), label_scaled.numpy(), factor.numpy(), actions, res.numpy())])
        ev.update({f"{tag}_mpjpe": np.float64(r["mpjpe"]), f"{tag}_p_mpjpe": np.float64(r["p_mpjpe"]), f"{tag}_acc": np.float64(r["acceleration_error"]),
                   f"{tag}_mpjpe_joint": np.asarray(r["mpjpe_joint"], np.float64), f"{tag}_mpjpe_activity": np.asarray(r["mpjpe_activity"], np.float64)})
        ev["activity_name_sequence"] = np.array(r["activity_name_sequence"])
    per = [O.clip_metrics(pred_tta[i].numpy(), label_scaled[i].numpy(), factor[i].numpy(), (int(res[i][0]), int(res[i][1]))) for i in range(4)]
    O.mpjpe, O.jpe, O.acc_error, O.p_mpjpe = saved
    ev.update(x=x.numpy(), label_scaled=label_scaled.numpy(), factor=factor.numpy(), res=res.numpy(), actions=np.array(actions),
              pred_tta=pred_tta.numpy(), pred_plain=pred_plain.numpy(), clip_mpjpe=np.stack([q[0] for q in per]), clip_jpe=np.stack([q[1] for q in per]),
              clip_acc=np.stack([q[2] for q in per]), clip_pmpjpe=np.stack([q[3] for q in per]))
    np.savez_compressed(os.path.join(HERE, "eval_L2_T27_B4.npz"), **ev)
    print("wrote eval_L2_T27_B4.npz:", {k: float(v) for k, v in ev.items() if np.ndim(v) == 0})


def clips_fixture():
    """Clip files written by the REFERENCE writers (data/preprocessor/clip_generate_{sp,wp}.py:28-79) and read back by the REFERENCE
    datasets (data/reader/{sp,wp}_dataset.py:45-92).  The .pkl files are data (dicts of numpy arrays); what the datasets return is
    stored next to them as the expected result of the product loader."""
    import importlib
    import random
    import shutil
    import io
    import contextlib
    sys.path.insert(0, "/root/reference/data")
    T = 9
    out_root = os.path.join(HERE, "clips")
    shutil.rmtree(out_root, ignore_errors=True)
    exp = {}
    for tag, gen_name, ds_mod, ds_cls, set_name in (("sp", "preprocessor.clip_generate_sp", "reader.sp_dataset", "SportsPose3DDataset", f"SPgt-{T}"),
                                                    ("wp", "preprocessor.clip_generate_wp", "reader.wp_dataset", "WorldPose3DDataset", f"WPdete-{T}")):
        gen = importlib.import_module(gen_name)
        ds = getattr(importlib.import_module(ds_mod), ds_cls)
        n_train, n_test = 5, 4
        x, y = O.synthetic_clips(n_train + n_test, T, seed=31 if tag == "sp" else 32, res=(1312, 1216) if tag == "sp" else (1920, 1080),
                                 det_conf=(tag == "wp"))
        y_abs = y + torch.randn(n_train + n_test, T, 1, 3, generator=torch.Generator().manual_seed(5)) * 0.1   # labels before root-relative
        label_scaled, factor, res, actions = O.synthetic_test_extras(y[n_train:], seed=33, res_choices=((1312, 1216), (1216, 1936)) if tag == "sp" else ((1920, 1080),))
        root = os.path.join(out_root, set_name)
        with contextlib.redirect_stderr(io.StringIO()):
            gen.save_clips_train(root_path=root, input_set=x[:n_train].numpy(), label_set=y_abs[:n_train].numpy())
            kw = dict(root_path=root, input_set=x[n_train:].numpy(), label_set=y_abs[n_train:].numpy(), label_scaled_set=label_scaled.numpy().astype(np.float64),
                      action_set=[[a] * T for a in actions], factor_set=factor.numpy().astype(np.float64), hw_set=res.numpy().astype(np.float64))
            if tag == "sp":
                kw["envtag_set"] = [["indoors" if i % 2 else "outdoors"] * T for i in range(n_test)]
            gen.save_clips_test(**kw)
        args = types.SimpleNamespace(model_name="KASportsFormer", input_channel_number=3, data_root=out_root, flip=False, clip_set_name=set_name)
        tr = ds(args, "train")
        exp[f"{tag}_train_x"] = np.stack([tr[i][0].numpy() for i in range(len(tr))])
        exp[f"{tag}_train_y"] = np.stack([tr[i][1].numpy() for i in range(len(tr))])
        args.flip = True
        keep = random.random
        random.random = lambda: 0.9                                  # "> 0.5": every clip flipped
        trf = ds(args, "train")
        exp[f"{tag}_train_x_flip"] = np.stack([trf[i][0].numpy() for i in range(len(trf))])
        exp[f"{tag}_train_y_flip"] = np.stack([trf[i][1].numpy() for i in range(len(trf))])
        random.random = keep
        te = ds(args, "test")
        items = [te[i] for i in range(len(te))]
        exp[f"{tag}_test_x"] = np.stack([it[0].numpy() for it in items])
        exp[f"{tag}_test_label_scaled"] = np.stack([np.asarray(it[1]) for it in items])
        exp[f"{tag}_test_factor"] = np.stack([np.asarray(it[2]) for it in items])
        exp[f"{tag}_test_action"] = np.array([it[3] for it in items])
        exp[f"{tag}_test_res"] = np.stack([np.asarray(it[4]) for it in items])
    np.savez_compressed(os.path.join(HERE, "clips_expected.npz"), **exp)
    print("wrote clips/ and clips_expected.npz:", {k: v.shape for k, v in exp.items()})


def synthetic_source(dataset, seed):
    """A miniature of the monolithic source pickles the reference slices (sp_reader.py:25-103 field names): per-frame arrays plus a video id
    per frame.  Video lengths cover every branch of the two slicers at T = 9: exact multiples, a tail >= T/2, a tail < T/2, a video shorter
    than T, one shorter than T/2, an id that re-appears later, and a short LAST video (which both slicers drop)."""
    rng = np.random.RandomState(seed)

    def split(lengths, ids, with_test):
        n = int(sum(lengths))
        d = {"joint_2d": (rng.rand(n, 17, 3) * 1000).astype(np.float64), "joint3d_image": (rng.rand(n, 17, 3) * 1000).astype(np.float64),
             "source": np.concatenate([np.full(k, v) for k, v in zip(lengths, ids)])}
        if dataset == "sp":
            d["camera_name"] = np.concatenate([np.full(k, "outdoors" if i % 2 == 0 else "indoors") for i, k in enumerate(lengths)])
        if seed % 2 == 1:
            d["confidence"] = rng.rand(n, 17).astype(np.float64)
        if with_test:
            acts = ["soccer", "tennis", "jump"]
            d["action"] = np.concatenate([np.full(k, acts[i % 3]) for i, k in enumerate(lengths)])
            d["2.5d_factor"] = (rng.rand(n) * 0.4 + 0.8).astype(np.float64)
            d["joints_2.5d_image"] = (rng.randn(n, 17, 3) * 300).astype(np.float64)
        return d
    return {"train": split([27, 20, 5, 9, 14, 3, 31, 6], ["a", "b", "c", "d", "e", "f", "c", "g"], False),
            "test": split([18, 13, 4, 9, 22, 7], ["t0", "t1", "t2", "t3", "t1", "t5"], True)}


def slicing_fixture():
    """Offline clip slicing: the REFERENCE readers (data/reader/sp_reader.py, wp_reader.py) run on miniature source
    files; the source arrays and everything the readers return are stored as data.  ``np.random.seed`` is set right before the slicing
    call: ``resample`` draws from numpy's global generator."""
    import importlib
    import pickle
    import tempfile
    sys.path.insert(0, "/root/reference/data")
    T = 9
    out = {}
    for tag, mod, cls, getter in (("sp", "reader.sp_reader", "DataReaderSportsPose", "get_sliced_data_sp"),
                                  ("wp", "reader.wp_reader", "DataReaderWorldPose", "get_sliced_data_wp")):
        for variant, seed in (("a", 10), ("b", 11)):                    # b: with a 'confidence' field (detector input)
            src = synthetic_source(tag, seed)
            with tempfile.NamedTemporaryFile(suffix=".pkl", delete=False) as f:
                pickle.dump(src, f)
            reader = getattr(importlib.import_module(mod), cls)(n_frames=T, sample_stride=1, data_stride_train=T // 3, data_stride_test=T,
                                                                source_file_path=f.name)
            os.remove(f.name)
            np.random.seed(1000 + seed)
            train_dict, test_dict = getattr(reader, getter)()
            ids_train, ids_test = reader.get_split_id()
            k = f"{tag}_{variant}_"
            for split in ("train", "test"):
                for name, arr in src[split].items():
                    out[k + f"src_{split}_{name}"] = np.asarray(arr)
            out[k + "ids_train"] = np.stack([np.asarray(list(c), dtype=np.int64) for c in ids_train])
            out[k + "ids_test"] = np.stack([np.asarray(list(c), dtype=np.int64) for c in ids_test])
            for name, arr in train_dict.items():
                out[k + "train_" + name] = np.asarray(arr)
            for name, arr in test_dict.items():
                out[k + "test_" + name] = np.asarray(arr)
    # resample on its own, every branch (sp_reader.py:129-150)
    np.random.seed(77)
    rs = importlib.import_module("reader.sp_reader").DataReaderSportsPose.resample
    for i, (a, b, kw) in enumerate(((5, 9, {}), (20, 9, {}), (9, 9, {}), (4, 27, {}), (7, 9, {"randomness": False}), (30, 9, {"replay": True}), (4, 9, {"replay": True}))):
        out[f"resample_{i}"] = np.asarray(list(rs(None, a, b, **kw)), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "slicing.npz"), **out)
    print("wrote slicing.npz:", len(out), "arrays;", {k: v.shape for k, v in out.items() if "ids_" in k})


def import_reference_harness():
    """`train_and_evaluate_sp.py` itself (for its `evaluate_one_epoch_new`, sp:27-149): importing the script drags in the logging / baseline-model
    imports of the whole repository.  Absent third-party modules get inert stand-ins -- `wandb`, `easydict` (a dict with attribute access),
    `torchprofile`, and the few `timm` names the vendored baselines import at module level; none of them is on the evaluated path."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):
        def __init__(self, p=None):
            super().__init__()
            self.drop_prob = p

        def forward(self, x):
            assert not (self.training and self.drop_prob)
            return x

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    layers = mod("timm.models.layers", DropPath=DropPath, trunc_normal_=nn.init.trunc_normal_, to_2tuple=lambda x: (x, x))
    helpers = mod("timm.models.helpers", load_pretrained=lambda *a, **k: None)
    registry = mod("timm.models.registry", register_model=lambda f: f)
    models = mod("timm.models", layers=layers, helpers=helpers, registry=registry)
    data = mod("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    timm = mod("timm", models=models, data=data)
    timm.__path__ = []
    mod("easydict", EasyDict=EasyDict)
    wutil = mod("wandb.util", generate_id=lambda: "offline")
    w = mod("wandb", util=wutil, log=lambda *a, **k: None, init=lambda *a, **k: None)
    w.__path__ = []
    mod("torchprofile", profile_macs=lambda *a, **k: 0)
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    import train_and_evaluate_sp as S
    return S, EasyDict


def eval_loop_fixture():
    """Re-pins the evaluation procedure with the reference's OWN loop: `evaluate_one_epoch_new` is run on the reference model
    and the batches of eval_L2_T27_B4.npz, with and without flip-TTA; the results must equal what that fixture stores (they came out of the
    restated loop `oracle.evaluate_batches` with the reference's metric functions), and are written next to it."""
    import logging
    S, EasyDict = import_reference_harness()
    from model.KASportsFormer import KASportsFormer as Ref
    fx = dict(np.load(os.path.join(HERE, "eval_L2_T27_B4.npz")))
    ref = Ref(n_layers=2, dim_in=3, dim_feat=128, dim_rep=512, dim_out=3, mlp_ratio=4, num_heads=8, n_frames=27)
    ref.load_state_dict(O.name_seeded_fill(ref.state_dict()), strict=True)
    ref.eval()
    x, label_scaled, factor, res = (torch.from_numpy(fx[k]) for k in ("x", "label_scaled", "factor", "res"))
    actions = [str(a) for a in fx["actions"]]
    loader = [(x[:2], label_scaled[:2], factor[:2], actions[:2], res[:2]), (x[2:], label_scaled[2:], factor[2:], actions[2:], res[2:])]   # two batches
    log = logging.getLogger("golden")
    out = {}
    for tag, flip in (("tta", True), ("plain", False)):
        args = EasyDict(num_joints=17, flip=flip, eval_only=True)
        r = S.evaluate_one_epoch_new(args, ref, loader, "cpu", 0, log)
        order = np.argsort(np.array(r["activity_name_sequence"]))             # the reference iterates a set: order is arbitrary
        want_order = np.argsort(fx["activity_name_sequence"])
        got = {"mpjpe": float(r["mpjpe"]), "p_mpjpe": float(r["p_mpjpe"]), "acc": float(r["acceleration_error"]),
               "mpjpe_joint": np.asarray(r["mpjpe_joint"], np.float64), "mpjpe_activity": np.asarray(r["mpjpe_activity"], np.float64)[order]}
        # float32 means of per-action float32 means: the reference walks a Python set, the restatement first-seen order -> one or two float32 ulps
        assert abs(got["mpjpe"] - float(fx[f"{tag}_mpjpe"])) < 1e-6 * abs(got["mpjpe"]), (tag, got["mpjpe"], float(fx[f"{tag}_mpjpe"]))
        assert abs(got["p_mpjpe"] - float(fx[f"{tag}_p_mpjpe"])) < 1e-6 * abs(got["p_mpjpe"])
        assert abs(got["acc"] - float(fx[f"{tag}_acc"])) < 1e-6 * abs(got["acc"])
        assert np.allclose(got["mpjpe_joint"], fx[f"{tag}_mpjpe_joint"], rtol=1e-6, atol=0), np.abs(got["mpjpe_joint"] - fx[f"{tag}_mpjpe_joint"]).max()   # float32 means over actions in set order
        assert np.allclose(got["mpjpe_activity"], fx[f"{tag}_mpjpe_activity"][want_order], rtol=1e-6, atol=0)
        out.update({f"refloop_{tag}_mpjpe": np.float64(got["mpjpe"]), f"refloop_{tag}_p_mpjpe": np.float64(got["p_mpjpe"]), f"refloop_{tag}_acc": np.float64(got["acc"]),
                    f"refloop_{tag}_mpjpe_joint": got["mpjpe_joint"]})
    fx.update(out)
    np.savez_compressed(os.path.join(HERE, "eval_L2_T27_B4.npz"), **fx)
    print("evaluate_one_epoch_new (reference loop) reproduces eval_L2_T27_B4.npz:", {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


def main():
    Ref, bone_decomposer, LC, EC = import_reference()
    torch.set_num_threads(8)
    if sys.argv[1:] == ["clips"]:                            # only the clip-file fixture
        return clips_fixture()
    if sys.argv[1:] == ["eval"]:                             # only the evaluation fixture
        return eval_fixture(Ref, EC)
    if sys.argv[1:] == ["slicing"]:                          # only the offline clip-slicing fixture
        return slicing_fixture()
    if sys.argv[1:] == ["evalloop"]:                         # re-pin the evaluation fixture with the reference's own loop
        return eval_loop_fixture()

    # 1. state_dict manifest of the full 26-layer model (names/shapes/dtypes only)
    full = Ref(n_layers=26, dim_in=3, dim_feat=128, dim_rep=512, dim_out=3, mlp_ratio=4, num_heads=8, n_frames=27)
    man = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in full.state_dict().items()]
    full.train()
    x, y = O.synthetic_clips(1, 27)
    LC.mpjpe_loss_calc(full(x), y).backward()
    dead = [n for n, p in full.named_parameters() if p.grad is None]
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump({"entries": man, "grad_none": dead,
                   "n_params": sum(p.numel() for p in full.parameters())}, f, separators=(",", ":"))
    print("manifest:", len(man), "entries;", len(dead), "never-grad params")
    del full

    # 2. model fixtures
    model_fixture(Ref, LC, os.path.join(HERE, "model_L2_T27_B2.npz"), n_layers=2, B=2, T=27, stages=True)
    model_fixture(Ref, LC, os.path.join(HERE, "model_L1_T81_B1.npz"), n_layers=1, B=1, T=81, stages=False)

    # 3. small functional pins: bone decomposition (incl. a zero-length bone), losses, metrics
    g = torch.Generator().manual_seed(7)
    xb = torch.randn(2, 5, 17, 3, generator=g)
    xb[0, 0, 1, :2] = xb[0, 0, 0, :2]                       # bone 0 (joints 0-1) has zero length
    p = torch.randn(2, 27, 17, 3, generator=g).requires_grad_(True)
    t = torch.randn(2, 27, 17, 3, generator=g)
    fx = {"bone_in": xb.numpy(), "bone_out": bone_decomposer(xb.clone()).numpy(), "loss_pred": p.detach().numpy(),
          "loss_tgt": t.numpy()}
    for name, fn in (("mpjpe", LC.mpjpe_loss_calc), ("n_mpjpe", LC.n_mpjpe_loss_calc), ("velocity", LC.velocity_loss_calc)):
        p.grad = None
        v = fn(p, t)
        v.backward()
        fx["loss_" + name] = np.array(v.item())
        fx["loss_" + name + "_grad"] = p.grad.numpy().copy()
    a = torch.randn(27, 17, 3, generator=g).numpy().astype(np.float64) * 100
    b = a + torch.randn(27, 17, 3, generator=g).numpy().astype(np.float64) * 20
    fx.update(met_pred=a, met_tgt=b, met_mpjpe=EC.mpjpe_calc(a.copy(), b.copy()), met_jpe=EC.jpe_calc(a.copy(), b.copy()),
              met_acc=EC.acc_error_calc(a.copy(), b.copy()), met_pmpjpe=EC.p_mpjpe_calc(a.copy(), b.copy()))
    # joint_flip (utils/utilities.py:128-135) cannot be imported (easydict); its index lists are data:
    xf = torch.randn(2, 3, 17, 3, generator=g)
    fl = xf.clone()
    fl[..., 0] *= -1
    left, right = [1, 2, 3, 14, 15, 16], [4, 5, 6, 11, 12, 13]
    fl[..., left + right, :] = fl[..., right + left, :]
    fx.update(flip_in=xf.numpy(), flip_out=fl.numpy())
    np.savez_compressed(os.path.join(HERE, "functional.npz"), **fx)
    print("wrote functional.npz")
    eval_fixture(Ref, EC)
    clips_fixture()
    slicing_fixture()
    eval_loop_fixture()


if __name__ == "__main__":
    main()
