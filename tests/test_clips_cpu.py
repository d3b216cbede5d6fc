"""Clip files: the reference's pickle format (fixture written by the reference's own writers, tests/golden/make_golden.py) -> packed
flat file -> arrays.  Host logic only; the on-device gather is covered by tests/test_gpu_clips.py."""
import os
import pickle

import numpy as np
import pytest
import torch

import kasportsformer_amd as K

CLIPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clips")
EXPECTED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clips_expected.npz")


@pytest.mark.parametrize("tag,set_name", [("sp", "SPgt-9"), ("wp", "WPdete-9")])
def test_pack_matches_what_the_reference_dataset_returns(tmp_path, tag, set_name):
    exp = np.load(EXPECTED)
    tr = K.pack_clip_directory(os.path.join(CLIPS, set_name, "train"), out_path=str(tmp_path / "train.kclips"))
    te = K.pack_clip_directory(os.path.join(CLIPS, set_name, "test"), out_path=str(tmp_path / "test.kclips"))
    assert np.array_equal(tr.arrays["x"], exp[f"{tag}_train_x"]) and np.array_equal(tr.arrays["y"], exp[f"{tag}_train_y"])
    assert np.array_equal(te.arrays["x"], exp[f"{tag}_test_x"])
    assert np.array_equal(te.arrays["label_scaled"], exp[f"{tag}_test_label_scaled"].astype(np.float32))
    assert np.array_equal(te.arrays["factor"], exp[f"{tag}_test_factor"].astype(np.float32))
    assert np.array_equal(te.arrays["res"], exp[f"{tag}_test_res"].astype(np.float32))
    assert [te.action_names[i] for i in te.arrays["action_ids"]] == list(exp[f"{tag}_test_action"])
    # the packed file round-trips, memory-mapped and read
    for path, ref in ((tmp_path / "train.kclips", tr), (tmp_path / "test.kclips", te)):
        for mm in (True, False):
            back = K.PackedClips.load(str(path), mmap=mm)
            assert back.split == ref.split and back.action_names == ref.action_names and len(back) == len(ref) and back.n_frames == 9
            for k, v in ref.arrays.items():
                assert back.arrays[k].dtype == v.dtype and np.array_equal(back.arrays[k], v)


def test_bad_files_are_rejected(tmp_path):
    with pytest.raises(FileNotFoundError):
        K.pack_clip_directory(str(tmp_path))
    p = tmp_path / "x.kclips"
    p.write_bytes(b"not a clip file at all")
    with pytest.raises(ValueError):
        K.PackedClips.load(str(p))
    tr = K.pack_clip_directory(os.path.join(CLIPS, "SPgt-9", "train"), out_path=str(p))
    data = p.read_bytes()
    p.write_bytes(data[:len(data) // 2])
    with pytest.raises(ValueError):
        K.PackedClips.load(str(p))
    evil = tmp_path / "00000000.pkl"                 # clip files may only construct numpy objects
    evil.write_bytes(pickle.dumps({"data_input": os.path.join}))
    with pytest.raises(pickle.UnpicklingError):
        K.read_clip_file(str(evil))
    with pytest.raises(ValueError):
        K.PackedClips("train", {"x": np.zeros((2, 9, 16, 3), np.float32), "y": np.zeros((2, 9, 16, 3), np.float32)})


def test_shard_plan_follows_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    n, world = 11, 4
    seen = []
    for rank in range(world):
        got = K.shard_indices(n, epoch_seed=5 + 3, shuffle=True, rank=rank, world_size=world)
        ds = DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=True, seed=5)
        ds.set_epoch(3)
        assert got.tolist() == list(ds)
        seen += got.tolist()
    assert sorted(set(seen)) == list(range(n)) and len(seen) == 12           # padded by wrapping
    ev = [K.shard_indices(n, 0, False, r, world, pad=False).tolist() for r in range(world)]
    assert sorted(sum(ev, [])) == list(range(n))                             # evaluation: every clip exactly once
