"""Offline clip slicing against fixtures produced by the reference's own readers (tests/golden/make_golden.py
``slicing_fixture``: DataReaderSportsPose / DataReaderWorldPose on miniature source files): clip frame indices, normalised inputs and
labels, test-split extras -- all bit for bit, including which random numbers ``resample`` consumes."""
import os

import numpy as np
import pytest

import kasportsformer_amd as K
from kasportsformer_amd import slicing as S

FX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slicing.npz"))
T = 9


def _source(tag, variant):
    k = f"{tag}_{variant}_src_"
    src = {"train": {}, "test": {}}
    for name in FX.files:
        if name.startswith(k):
            split, field = name[len(k):].split("_", 1)
            src[split][field] = FX[name]
    return src


@pytest.mark.parametrize("tag", ["sp", "wp"])
@pytest.mark.parametrize("variant,seed", [("a", 10), ("b", 11)])
def test_sliced_clips_equal_the_reference_readers(tag, variant, seed):
    src = _source(tag, variant)
    assert ("confidence" in src["train"]) == (variant == "b")
    np.random.seed(1000 + seed)                                         # the reference draws from numpy's global generator
    train, test, ids = K.slice_source(src, tag, n_frames=T)
    k = f"{tag}_{variant}_"
    assert np.array_equal(ids["train"], FX[k + "ids_train"]) and np.array_equal(ids["test"], FX[k + "ids_test"])
    assert np.array_equal(train.arrays["x"], FX[k + "train_data"].astype(np.float32))
    lab = FX[k + "train_label"]
    assert np.array_equal(train.arrays["y"], (lab - lab[..., 0:1, :]).astype(np.float32))          # save_clips_train: root-relative labels
    assert np.array_equal(test.arrays["x"], FX[k + "test_data"].astype(np.float32))
    assert np.array_equal(test.arrays["label"], FX[k + "test_label"])
    assert np.array_equal(test.arrays["label_scaled"], FX[k + "test_label_scaled"].astype(np.float32))
    assert np.array_equal(test.arrays["factor"], FX[k + "test_factor"].astype(np.float32))
    assert np.array_equal(test.arrays["res"], FX[k + "test_test_hw"].astype(np.float32))
    want_actions = [str(a) for a in FX[k + "test_action"][:, 0]]
    assert [test.action_names[i] for i in test.arrays["action_ids"]] == want_actions
    # same clips through a private generator: the global state is left alone
    state = np.random.get_state()[1].copy()
    _, _, ids2 = K.slice_source(src, tag, n_frames=T, rng=np.random.RandomState(1000 + seed))
    assert np.array_equal(ids2["train"], ids["train"]) and np.array_equal(np.random.get_state()[1], state)


def test_slicer_structure_on_hand_made_lists():
    """Properties the two slicers must have whatever the random draws: full windows are contiguous and stay inside one video; SportsPose
    stretches a too-short video once per id and never the last video; WorldPose stretches tails of at least T/2 frames."""
    vids = np.array(list("aaaaaaaaaaaa" "bbbb" "cccccccccccccc" "bbbbb" "dd"))        # 12 a, 4 b, 14 c, 5 b again, 2 d (last)
    rng = np.random.RandomState(0)
    sp = S.split_clips(vids, 9, 3, rng=rng)
    full = [c for c in sp if np.array_equal(c, np.arange(c[0], c[0] + 9))]
    assert [int(c[0]) for c in full] == [0, 3, 16, 19]                                  # a: starts 0, 3 (0+9, 3+9 <= 12); c: 16, 19 (+9 <= 30)
    stretched = [c for c in sp if not np.array_equal(c, np.arange(c[0], c[0] + 9))]
    assert len(stretched) == 1 and stretched[0].min() >= 12 and stretched[0].max() <= 15  # first 'b' run only; the second 'b' run is already "saved"
    assert all(len(set(vids[c])) == 1 for c in sp)
    wp = S.mysplit_clips(vids, 9, 9, rng=np.random.RandomState(0))
    starts = [int(c[0]) for c in wp if np.array_equal(c, np.arange(c[0], c[0] + 9))]
    assert starts == [0, 16]
    tails = [c for c in wp if not np.array_equal(c, np.arange(c[0], c[0] + 9))]
    # tails: a has 3 left (< 4.5: dropped), first b run 4 (< 4.5: dropped), c has 5 left (kept), second b run 5 (kept), d is last (dropped)
    assert len(tails) == 2 and tails[0].min() >= 25 and tails[0].max() <= 29 and tails[1].min() >= 30 and tails[1].max() <= 34
    assert S.split_clips([], 9, 3) == [] and S.mysplit_clips([], 9, 3) == []
    with pytest.raises(ValueError):
        S.mysplit_clips(vids, 9, 10)


def test_resample_branches_equal_the_reference():
    np.random.seed(77)
    calls = ((5, 9, {}), (20, 9, {}), (9, 9, {}), (4, 27, {}), (7, 9, {"randomness": False}), (30, 9, {"replay": True}), (4, 9, {"replay": True}))
    for i, (a, b, kw) in enumerate(calls):
        got = S.resample(a, b, **kw)
        assert np.array_equal(np.asarray(got, dtype=np.int64), FX[f"resample_{i}"]), (i, got, FX[f"resample_{i}"])
        assert len(got) == b and got.min() >= 0 and got.max() < a


def test_sliced_clips_round_trip_through_the_packed_file(tmp_path):
    src = _source("sp", "a")
    train, test, _ = K.slice_source(src, "sp", n_frames=T, rng=np.random.RandomState(1))
    for split in (train, test):
        path = str(tmp_path / f"{split.split}.kasf")
        split.save(path)
        back = K.PackedClips.load(path)
        assert back.split == split.split and back.action_names == split.action_names
        for name, a in split.arrays.items():
            assert np.array_equal(back.arrays[name], a), name
