"""Clip storage and loading for the path's callers.

Reference side: one pickle per clip, ``%08d.pkl`` under ``<data_root>/<clip_set_name>/{train,test}/``
(data/preprocessor/clip_generate_sp.py:28-79), read item by item by ``SportsPose3DDataset`` / ``WorldPose3DDataset``
(data/reader/sp_dataset.py:45-92) in 19 DataLoader worker processes, with a coin-flip left/right augmentation per training clip.

Here: the clip directory is packed ONCE into a single flat file of contiguous float32 arrays (``pack_clip_directory``); the file is
memory-mapped, uploaded whole into HBM (11 KB per 27-frame training clip: a million clips are 11 GB of the 288 GB), and every batch
is one gather kernel that also applies the flip (``kasf_gather_clips``).  No worker processes, no per-item unpickling, no host copies
inside an epoch.  Sharding across ranks follows torch's DistributedSampler (pad to a multiple of the world size, stride by rank).
"""
from __future__ import annotations

import ctypes as C
import io
import json
import os
import pickle

import numpy as np
import torch

from . import _lib

MAGIC = b"KASFCLP1"
_ALIGN = 64


class _ClipUnpickler(pickle.Unpickler):
    """Clip files hold dicts of numpy arrays, strings and numbers: nothing outside numpy may be constructed."""

    def find_class(self, module, name):
        if module.split(".")[0] == "numpy":
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"clip file references {module}.{name}: only numpy objects are allowed")


def read_clip_file(path: str) -> dict:
    with open(path, "rb") as f:
        return _ClipUnpickler(io.BytesIO(f.read())).load()


class PackedClips:
    """One split of a clip set as flat arrays.  ``x`` [N,T,17,3]; train: ``y`` [N,T,17,3] (root-relative labels);
    test: ``label_scaled`` [N,T,17,3] (mm), ``factor`` [N,T], ``res`` [N,2] (w,h), ``action_ids`` [N] into ``action_names``."""

    TRAIN_ARRAYS = ("x", "y")
    TEST_ARRAYS = ("x", "label_scaled", "factor", "res", "action_ids")

    def __init__(self, split: str, arrays: dict, action_names=()):
        if split not in ("train", "test"):
            raise ValueError(f"unknown split {split!r}")
        self.split, self.arrays, self.action_names = split, arrays, list(action_names)
        need = self.TRAIN_ARRAYS if split == "train" else self.TEST_ARRAYS
        missing = [k for k in need if k not in arrays]
        if missing:
            raise ValueError(f"packed {split} clips lack {missing}")
        x = arrays["x"]
        if x.ndim != 4 or x.shape[2:] != (17, 3):
            raise ValueError(f"clip inputs must be [N,T,17,3], got {x.shape}")

    def __len__(self):
        return self.arrays["x"].shape[0]

    @property
    def n_frames(self):
        return self.arrays["x"].shape[1]

    # ---- file format: MAGIC, u64 header length, JSON header, then the arrays at 64-byte aligned offsets ----
    def save(self, path: str):
        metas, off = [], 0
        for name, a in self.arrays.items():
            a = np.ascontiguousarray(a)
            metas.append({"name": name, "dtype": a.dtype.str, "shape": list(a.shape), "offset": off, "nbytes": a.nbytes})
            off += (a.nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        header = json.dumps({"split": self.split, "action_names": self.action_names, "arrays": metas}).encode()
        pad = (-(len(MAGIC) + 8 + len(header))) % _ALIGN
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(MAGIC)
            f.write(np.uint64(len(header) + pad).tobytes())
            f.write(header + b" " * pad)
            base = f.tell()
            for m, a in zip(metas, self.arrays.values()):
                f.seek(base + m["offset"])
                f.write(np.ascontiguousarray(a).tobytes())
            f.truncate(base + off)
        os.replace(tmp, path)

    @classmethod
    def load(cls, path: str, mmap: bool = True) -> "PackedClips":
        with open(path, "rb") as f:
            if f.read(len(MAGIC)) != MAGIC:
                raise ValueError(f"{path} is not a packed clip file")
            hlen = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
            header = json.loads(f.read(hlen).decode())
            base = f.tell()
        size = os.path.getsize(path)
        arrays = {}
        for m in header["arrays"]:
            if base + m["offset"] + m["nbytes"] > size:
                raise ValueError(f"{path} is truncated (array {m['name']})")
            if mmap:
                arrays[m["name"]] = np.memmap(path, dtype=np.dtype(m["dtype"]), mode="r", offset=base + m["offset"], shape=tuple(m["shape"]))
            else:
                with open(path, "rb") as f:
                    f.seek(base + m["offset"])
                    arrays[m["name"]] = np.frombuffer(f.read(m["nbytes"]), dtype=np.dtype(m["dtype"])).reshape(m["shape"])
        return cls(header["split"], arrays, header["action_names"])


def pack_clip_directory(clip_dir: str, split: str | None = None, out_path: str | None = None) -> PackedClips:
    """Reads every ``*.pkl`` of ``<data_root>/<clip_set_name>/<split>`` in sorted order (the order of ``_generate_file_list``,
    sp_dataset.py:22-28) and returns / writes the packed form.  float64 fields are stored as float32 (what the model and the metric
    kernel consume); action names are numbered in sorted order so that every rank agrees on the ids."""
    split = split or os.path.basename(os.path.normpath(clip_dir))
    files = sorted(n for n in os.listdir(clip_dir) if n.endswith(".pkl"))
    if not files:
        raise FileNotFoundError(f"no clip files in {clip_dir}")
    clips = [read_clip_file(os.path.join(clip_dir, n)) for n in files]
    x = np.stack([np.asarray(c["data_input"], dtype=np.float32) for c in clips])
    if split == "train":
        arrays = {"x": x, "y": np.stack([np.asarray(c["data_label"], dtype=np.float32) for c in clips])}
        names = []
    else:
        names = sorted({str(c["data_action"]) for c in clips})
        arrays = {"x": x,
                  "label_scaled": np.stack([np.asarray(c["data_label_scaled"], dtype=np.float32) for c in clips]),
                  "factor": np.stack([np.asarray(c["data_factor"], dtype=np.float32) for c in clips]),
                  "res": np.stack([np.asarray(c["data_res"], dtype=np.float32) for c in clips]),
                  "action_ids": np.array([names.index(str(c["data_action"])) for c in clips], dtype=np.int32)}
    packed = PackedClips(split, arrays, names)
    if out_path:
        packed.save(out_path)
    return packed


def shard_indices(n: int, epoch_seed: int, shuffle: bool, rank: int, world_size: int, pad: bool = True) -> torch.Tensor:
    """DistributedSampler's index plan: permutation (if shuffling), padded by wrapping to a multiple of ``world_size``, strided by rank.
    ``pad=False`` (evaluation): no clip is counted twice, shards may differ in length by one."""
    if not pad:
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(epoch_seed)) if shuffle else torch.arange(n)
        return idx[rank::world_size].contiguous()
    if shuffle:
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(epoch_seed))
    else:
        idx = torch.arange(n)
    total = (n + world_size - 1) // world_size * world_size
    if total > n:
        idx = torch.cat((idx, idx[:total - n]))
    return idx[rank:total:world_size].contiguous()


class DeviceClipLoader:
    """Iterates a packed split from HBM.  Train: yields ``(joint_input, joint_label)``; test: yields ``(joint_input, joint_label_scaled,
    joint_factor, joint_action, joint_res)`` -- the tuples of the reference's DataLoaders (train_and_evaluate_sp.py:251-255), on the GPU."""

    def __init__(self, clips: PackedClips, batch_size: int, shuffle: bool | None = None, flip: bool = True, seed: int = 0, rank: int = 0,
                 world_size: int = 1, drop_last: bool = False, device="cuda"):
        self.clips, self.batch_size, self.flip, self.seed = clips, int(batch_size), bool(flip), int(seed)
        self.shuffle = (clips.split == "train") if shuffle is None else bool(shuffle)
        self.rank, self.world_size, self.drop_last, self.epoch = rank, world_size, drop_last, 0
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceClipLoader keeps the clips in GPU memory; there is no CPU path")
        self._lib = _lib.load()
        self.dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(self.device) for k, v in clips.arrays.items()}   # whole split resident in HBM

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def _plan(self):
        return shard_indices(len(self.clips), self.seed + self.epoch, self.shuffle, self.rank, self.world_size, pad=self.clips.split == "train")

    def __len__(self):
        n = self._plan().numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _gather(self, index_dev, flip_dev, B):
        T = self.clips.n_frames
        x = torch.empty(B, T, 17, 3, device=self.device)
        train = self.clips.split == "train"
        y = torch.empty(B, T, 17, 3, device=self.device)
        second = self.dev["y"] if train else self.dev["label_scaled"]
        _lib.check(self._lib.kasf_gather_clips(self.dev["x"].data_ptr(), second.data_ptr(), index_dev.data_ptr(),
                                               flip_dev.data_ptr() if flip_dev is not None else None, len(self.clips), B, T, x.data_ptr(), y.data_ptr(),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return x, y

    def __iter__(self):
        idx = self._plan()
        train = self.clips.split == "train"
        flips = None
        if train and self.flip:       # sp_dataset.py:75-78: each clip flipped with probability 1/2 (input and label together)
            flips = (torch.rand(idx.numel(), generator=torch.Generator().manual_seed((self.seed + self.epoch) * 7919 + 13 + self.rank)) > 0.5).to(torch.uint8)
        idx_dev = idx.to(self.device)
        flips_dev = flips.to(self.device) if flips is not None else None
        n = idx.numel()
        stop = n - n % self.batch_size if self.drop_last else n
        for s in range(0, stop, self.batch_size):
            e = min(s + self.batch_size, stop)
            sel = idx_dev[s:e]
            x, y = self._gather(sel, flips_dev[s:e] if flips_dev is not None else None, e - s)
            if train:
                yield x, y
            else:
                ids = self.clips.arrays["action_ids"][idx[s:e].numpy()]
                yield x, y, self.dev["factor"][sel], [self.clips.action_names[int(i)] for i in ids], self.dev["res"][sel]
