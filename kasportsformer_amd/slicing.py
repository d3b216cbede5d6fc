"""Offline clip slicing: turns a monolithic per-frame source dictionary into T-frame clips.

Reference: ``DataReaderSportsPose`` / ``DataReaderWorldPose`` (data/reader/sp_reader.py:25-169,205-249, data/reader/wp_reader.py:25-135,
159-199) driven by data/preprocessor/clip_generate_{sp,wp}.py:28-118, which write one pickle per clip.  Here the same clips go straight into
``PackedClips`` (the flat, memory-mappable form ``DeviceClipLoader`` uploads): no per-clip files.

What is reproduced exactly, quirks included (pinned by fixtures generated with the reference's own readers, tests/golden/make_golden.py):
  * pixel -> normalised screen coordinates with the reference's rounding points (float32 divide/multiply, float64 subtract, float32 store);
  * SportsPose slicing (``split_clips``, sp_reader.py:103-125, MotionBERT's): windows of T frames every ``stride`` frames inside a video; a
    video too short for one window is resampled to T frames ONCE per video id -- unless it is the last video of the list, which is dropped;
  * WorldPose slicing (``mysplit_clips``, wp_reader.py:94-109): same windows; the tail of every video (what is left after its last window) is
    resampled to T frames when it is at least T/2 long -- again not for the last video of the list;
  * ``resample`` (sp_reader.py:129-150) draws from numpy's GLOBAL generator exactly like the reference (one ``randint(2, size=T)`` for a
    shorter segment, one ``random(T)`` for a longer one), so ``np.random.seed(s)`` before slicing gives the reference's clips bit for bit;
    pass ``rng=np.random.RandomState(s)`` to keep the global state untouched.
Host-side numpy only: this runs once per dataset, offline, and has no GPU part.
"""
from __future__ import annotations

import numpy as np

from .data import PackedClips

SP_RES = {"outdoors": (1312, 1216), "indoors": (1216, 1936)}      # sp_reader.py:30-33 (res_w, res_h) per camera
WP_RES = (1920, 1080)                                             # wp_reader.py:28


def resample(ori_len: int, target_len: int, replay: bool = False, randomness: bool = True, rng=None) -> np.ndarray:
    """Frame indices in [0, ori_len) that stretch / shrink a segment to ``target_len`` frames (sp_reader.py:129-150)."""
    rng = np.random if rng is None else rng
    if replay:
        if ori_len > target_len:
            st = rng.randint(ori_len - target_len)
            return np.arange(st, st + target_len)
        return np.arange(target_len) % ori_len
    if not randomness:
        return np.linspace(0, ori_len, num=target_len, endpoint=False, dtype=int)
    even = np.linspace(0, ori_len, num=target_len, endpoint=False)
    if ori_len < target_len:
        sel = rng.randint(2, size=even.shape)
        picked = np.sort(sel * np.floor(even) + (1 - sel) * np.ceil(even))
    else:
        picked = rng.random(even.shape) * (even[1] - even[0]) + even
    return np.clip(picked, a_min=0, a_max=ori_len - 1).astype(np.uint32)


def _runs(vid_list):
    """Maximal stretches of equal video id: [(start, end, id)]."""
    v = np.asarray(vid_list)
    if v.size == 0:
        return []
    cut = np.flatnonzero(v[1:] != v[:-1]) + 1
    starts = np.concatenate(([0], cut))
    ends = np.concatenate((cut, [v.size]))
    return [(int(s), int(e), v[s].item() if hasattr(v[s], "item") else v[s]) for s, e in zip(starts, ends)]


def _windows(s, e, n_frames, stride):
    k = 0
    while s + k * stride + n_frames <= e:
        yield s + k * stride
        k += 1


def split_clips(vid_list, n_frames: int, data_stride: int, rng=None) -> list:
    """SportsPose slicing (sp_reader.py:103-125): list of index arrays of length ``n_frames`` into the frame axis."""
    if not 1 <= data_stride:
        raise ValueError("data_stride must be positive")
    runs, out, saved = _runs(vid_list), [], set()
    for r, (s, e, vid) in enumerate(runs):
        for w in _windows(s, e, n_frames, data_stride):
            out.append(np.arange(w, w + n_frames))
            saved.add(vid)
        last = r == len(runs) - 1
        if not last and vid not in saved:          # the video never produced a window: stretch it once (ids are remembered across re-appearances)
            out.append(resample(e - s, n_frames, rng=rng).astype(np.int64) + s)
            saved.add(vid)
    return out


def mysplit_clips(vid_list, n_frames: int, data_stride: int, rng=None) -> list:
    """WorldPose slicing (wp_reader.py:94-109)."""
    if not 1 <= data_stride <= n_frames:
        raise ValueError("mysplit_clips needs 1 <= data_stride <= n_frames (a larger stride walks off the video in the reference too)")
    runs, out = _runs(vid_list), []
    for r, (s, e, _) in enumerate(runs):
        nxt = s
        for w in _windows(s, e, n_frames, data_stride):
            out.append(np.arange(w, w + n_frames))
            nxt = w + data_stride
        last = r == len(runs) - 1
        if not last and nxt < e and (e - nxt) >= n_frames / 2:      # the tail after the last window, if at least half a clip long
            out.append(resample(e - nxt, n_frames, rng=rng).astype(np.int64) + nxt)
    return out


def _normalise_xy(px: np.ndarray, res_w: np.ndarray, res_h: np.ndarray) -> np.ndarray:
    """``a[i] = a[i] / res_w * 2 - [1, res_h / res_w]`` on a float32 array (sp_reader.py:38): float32 scale, float64 shift, float32 store."""
    scaled = (px.astype(np.float32) / res_w[:, None, None].astype(np.float32) * np.float32(2)).astype(np.float32)
    shift = np.stack((np.ones_like(res_w, dtype=np.float64), res_h.astype(np.float64) / res_w.astype(np.float64)), axis=-1)      # [N,2]
    return (scaled.astype(np.float64) - shift[:, None, :]).astype(np.float32)


def _frame_res(split: dict, dataset: str):
    n = len(split["source"])
    if dataset == "sp":
        cams = list(split["camera_name"])
        bad = [i for i, c in enumerate(cams) if c not in SP_RES]
        if bad:
            raise AssertionError("%d data item has an invalid camera name" % bad[0])        # sp_reader.py:35
        wh = np.array([SP_RES[c] for c in cams], dtype=np.int64).reshape(-1, 2)
    else:
        wh = np.tile(np.array(WP_RES, dtype=np.int64), (n, 1))
    return wh[:, 0], wh[:, 1]


def _inputs_and_labels(split: dict, dataset: str, sample_stride: int, read_confidence: bool):
    rw, rh = _frame_res(split, dataset)
    j2 = np.asarray(split["joint_2d"])[::sample_stride, :, :2]
    j3 = np.asarray(split["joint3d_image"])[::sample_stride, :, :3]
    # the reference indexes the camera list by the position in the STRIDED array (sp_reader.py:29-38); sample_stride is 1 in every shipped call
    x = _normalise_xy(j2, rw[:len(j2)], rh[:len(j2)])
    if read_confidence:
        if "confidence" in split:
            conf = np.asarray(split["confidence"])[::sample_stride].astype(np.float32)
            conf = conf[:, :, None] if conf.ndim == 2 else conf
        else:
            conf = np.ones(x.shape[:2] + (1,))            # float64 ones -> the concatenation below is float64, like the reference's (sp_reader.py:55-57)
        x = np.concatenate((x, conf), axis=2)
    lab = j3.astype(np.float32)
    lab[:, :, :2] = _normalise_xy(j3[:, :, :2], rw[:len(j3)], rh[:len(j3)])
    lab[:, :, 2:] = (j3[:, :, 2:].astype(np.float32) / rw[:len(j3), None, None].astype(np.float32) * np.float32(2)).astype(np.float32)
    return x, lab


def slice_source(source: dict, dataset: str, n_frames: int = 27, sample_stride: int = 1, data_stride_train: int | None = None,
                 data_stride_test: int | None = None, read_confidence: bool = True, root_rel: bool = True, rng=None):
    """``get_sliced_data_sp`` / ``get_sliced_data_wp`` + ``save_clips_{train,test}`` in one step.

    ``source``: the unpickled source dictionary, ``{'train': {...}, 'test': {...}}`` with per-frame ``joint_2d`` [N,17,>=2] (pixels),
    ``joint3d_image`` [N,17,3], ``source`` [N] (video id), ``camera_name`` [N] (SportsPose only), optional ``confidence`` [N,17]; the test
    split also carries ``action`` [N], ``2.5d_factor`` [N], ``joints_2.5d_image`` [N,17,3].  ``dataset``: ``'sp'`` or ``'wp'``.
    Strides default to the CLI's (``n_frames // 3`` for training, ``n_frames`` for testing, clip_generate_sp.py:97-99).
    Returns ``(train: PackedClips, test: PackedClips, ids)`` with ``ids = {'train': [...], 'test': [...]}`` the frame indices of every clip.
    The train split is sliced first, then the test split -- the order in which the reference consumes its random numbers."""
    if dataset not in ("sp", "wp"):
        raise ValueError("dataset must be 'sp' or 'wp'")
    st_train = n_frames // 3 if data_stride_train is None else data_stride_train
    st_test = n_frames if data_stride_test is None else data_stride_test
    splitter = split_clips if dataset == "sp" else mysplit_clips
    ids = {}
    for name, stride in (("train", st_train), ("test", st_test)):
        clips = splitter(np.asarray(source[name]["source"])[::sample_stride], n_frames, stride, rng=rng)
        ids[name] = np.stack(clips).astype(np.int64) if clips else np.zeros((0, n_frames), np.int64)
    xtr, ltr = _inputs_and_labels(source["train"], dataset, sample_stride, read_confidence)
    xte, lte = _inputs_and_labels(source["test"], dataset, sample_stride, read_confidence)
    x_train, y_train = xtr[ids["train"]], ltr[ids["train"]]
    if root_rel:                                           # clip_generate_sp.py:39-40
        y_train = y_train - y_train[..., 0:1, :]
    train = PackedClips("train", {"x": x_train.astype(np.float32), "y": y_train.astype(np.float32)})
    te = source["test"]
    actions = np.asarray(te["action"])[ids["test"]]                            # [n, T]
    if len(actions) and not (actions == actions[:, :1]).all():
        bad = int(np.flatnonzero(~(actions == actions[:, :1]).all(axis=1))[0])
        raise AssertionError(f"wait, clip index {bad} contains more than one action ??")          # clip_generate_sp.py:61-62
    per_clip = [str(a) for a in actions[:, 0]] if len(actions) else []
    names = sorted(set(per_clip))
    rw, rh = _frame_res(te, dataset)
    first = ids["test"][:, 0] if len(ids["test"]) else np.zeros(0, np.int64)
    test = PackedClips("test", {
        "x": xte[ids["test"]].astype(np.float32),
        "label": lte[ids["test"]].astype(np.float32),
        "label_scaled": np.asarray(te["joints_2.5d_image"])[ids["test"]].astype(np.float32),
        "factor": np.asarray(te["2.5d_factor"])[ids["test"]].astype(np.float32),
        "res": np.stack((rw[first], rh[first]), axis=-1).astype(np.float32),       # get_hw_sp: the first frame's camera (sp_reader.py:188-192)
        "action_ids": np.array([names.index(a) for a in per_clip], dtype=np.int32)}, names)
    return train, test, ids
