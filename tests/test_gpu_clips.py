"""On-device batch assembly (kasf_gather_clips) against what the reference's Dataset classes return for the same clip files
(tests/golden/clips_expected.npz, produced by the reference code in tests/golden/make_golden.py).  Bit-exact: this is data movement."""
import os

import numpy as np
import pytest
import torch

from oracle import kasf_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CLIPS = os.path.join(HERE, "golden", "clips")
EXPECTED = os.path.join(HERE, "golden", "clips_expected.npz")


@pytest.mark.parametrize("tag,set_name", [("sp", "SPgt-9"), ("wp", "WPdete-9")])
def test_train_loader_matches_reference_dataset(tag, set_name):
    import kasportsformer_amd as K
    exp = np.load(EXPECTED)
    clips = K.pack_clip_directory(os.path.join(CLIPS, set_name, "train"))
    plain = list(K.DeviceClipLoader(clips, batch_size=2, shuffle=False, flip=False))
    assert [b[0].shape[0] for b in plain] == [2, 2, 1]                                  # ragged last batch kept, like DataLoader(drop_last=False)
    assert np.array_equal(torch.cat([b[0] for b in plain]).cpu().numpy(), exp[f"{tag}_train_x"])
    assert np.array_equal(torch.cat([b[1] for b in plain]).cpu().numpy(), exp[f"{tag}_train_y"])
    # shuffled + flipped epoch: every clip once; each one equals the reference's flipped or unflipped version (coin per clip)
    ld = K.DeviceClipLoader(clips, batch_size=3, shuffle=True, flip=True, seed=11)
    n_flipped = 0
    for epoch in range(4):
        ld.set_epoch(epoch)
        xs = torch.cat([b[0] for b in ld]).cpu().numpy()
        ld.set_epoch(epoch)
        ys = torch.cat([b[1] for b in ld]).cpu().numpy()                               # same epoch -> same plan
        order = K.shard_indices(len(clips), 11 + epoch, True, 0, 1).tolist()
        assert sorted(order) == list(range(5))
        for k, src in enumerate(order):
            is_plain = np.array_equal(xs[k], exp[f"{tag}_train_x"][src]) and np.array_equal(ys[k], exp[f"{tag}_train_y"][src])
            is_flip = np.array_equal(xs[k], exp[f"{tag}_train_x_flip"][src]) and np.array_equal(ys[k], exp[f"{tag}_train_y_flip"][src])
            assert is_plain != is_flip
            n_flipped += is_flip
    assert 3 <= n_flipped <= 17                                                         # p = 1/2 over 20 draws


def test_test_loader_feeds_the_evaluator_and_ranks_partition_the_set():
    import kasportsformer_amd as K
    exp = np.load(EXPECTED)
    clips = K.pack_clip_directory(os.path.join(CLIPS, "SPgt-9", "test"))
    batches = list(K.DeviceClipLoader(clips, batch_size=3))
    x = torch.cat([b[0] for b in batches]).cpu().numpy()
    assert np.array_equal(x, exp["sp_test_x"])                                          # test split: sequential, never flipped
    assert sum((b[3] for b in batches), []) == list(exp["sp_test_action"])
    assert np.array_equal(torch.cat([b[4] for b in batches]).cpu().numpy(), exp["sp_test_res"].astype(np.float32))
    # metrics through the loader == oracle restatement on the raw fixture
    pred = torch.randn(4, 9, 17, 3, generator=torch.Generator().manual_seed(2)) * 0.2
    ev = K.Evaluator(action_names=clips.action_names)
    o = 0
    for xb, ls, fac, act, res in batches:
        ev.update(pred[o:o + xb.shape[0]].cuda(), ls, fac, act, res)
        o += xb.shape[0]
    got = ev.result()
    pz = pred.clone()
    pz[:, :, 0] = 0
    ref = O.evaluate_batches([(pz.numpy(), exp["sp_test_label_scaled"].astype(np.float32), exp["sp_test_factor"], list(exp["sp_test_action"]),
                               exp["sp_test_res"])])
    assert abs(got["mpjpe"] - float(ref["mpjpe"])) < 2e-5 * float(ref["mpjpe"])
    assert abs(got["p_mpjpe"] - float(ref["p_mpjpe"])) < 1e-4 * float(ref["p_mpjpe"])
    assert abs(got["acceleration_error"] - float(ref["acceleration_error"])) < 2e-5 * float(ref["acceleration_error"])
    assert sorted(got["activity_name_sequence"]) == sorted(ref["activity_name_sequence"])
    # two ranks: disjoint shards that cover the set; their summed tables give the same result
    evs = []
    for rank in range(2):
        e = K.Evaluator(action_names=clips.action_names)
        for xb, ls, fac, act, res in K.DeviceClipLoader(clips, batch_size=8, rank=rank, world_size=2):
            idx = [int(np.where((exp["sp_test_x"] == xb[k].cpu().numpy()).all(axis=(1, 2, 3)))[0][0]) for k in range(xb.shape[0])]
            assert idx == list(range(rank, 4, 2))
            e.update(pred[idx].cuda(), ls, fac, act, res)
        evs.append(e)
    evs[0].sums += evs[1].sums
    both = evs[0].result()
    assert abs(both["mpjpe"] - got["mpjpe"]) < 1e-9 * got["mpjpe"] and both["activity_name_sequence"] == got["activity_name_sequence"]


def test_gather_out_of_range_index_yields_zero_clip():
    import ctypes as C
    from kasportsformer_amd import _lib
    lib = _lib.load()
    xa = torch.randn(3, 9, 17, 3, device="cuda")
    idx = torch.tensor([2, 7, -1, 0], device="cuda")
    out = torch.full((4, 9, 17, 3), 5.0, device="cuda")
    _lib.check(lib.kasf_gather_clips(xa.data_ptr(), None, idx.data_ptr(), None, 3, 4, 9, out.data_ptr(), None,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert torch.equal(out[0], xa[2]) and torch.equal(out[3], xa[0]) and float(out[1].abs().max()) == 0 and float(out[2].abs().max()) == 0


def test_train_one_epoch_over_device_loader_matches_oracle_loop():
    """Loader -> model -> loss -> fused optimizer through K.train_one_epoch against the reference's loop restated with the oracle (sp:201-243)."""
    import kasportsformer_amd as K
    from gpu_util import make_pair
    clips = K.pack_clip_directory(os.path.join(CLIPS, "SPgt-9", "train"))
    oracle, model = make_pair(1, 9, "fp32")
    opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
    topt = torch.optim.AdamW(oracle.parameters(), lr=5e-4, weight_decay=0.01)
    loader = K.DeviceClipLoader(clips, batch_size=2, shuffle=True, flip=True, seed=5)
    loader.set_epoch(0)
    batches = [(x.cpu(), y.cpu()) for x, y in loader]
    loader.set_epoch(0)                                               # same epoch -> same order and the same flips
    got = K.train_one_epoch(model, loader, opt)
    oracle.train()
    sums, n = [0.0, 0.0, 0.0, 0.0], 0
    for x, y in batches:
        pred = oracle(x)
        topt.zero_grad()
        total, (l1, l2, l3) = O.loss_total(pred, y)
        for k, v in enumerate((total, l1, l2, l3)):
            sums[k] += float(v) * x.shape[0]
        n += x.shape[0]
        total.backward()
        topt.step()
    for k, name in enumerate(("loss_total", "loss_mpjpe", "loss_n_mpjpe", "loss_velocity")):
        assert abs(got[name] - sums[k] / n) < 2e-3 * abs(sums[k] / n), (name, got[name], sums[k] / n)
    worst = max(float((p.detach().cpu() - q.detach()).abs().max()) for p, q in zip(model.parameters(), oracle.parameters()))
    assert worst < 3 * 5e-4                                           # three AdamW steps at lr 5e-4
