"""Synthetic clips of the benchmark / smoke workloads (SURVEY §8(d) "Synthetic inputs"): no dataset ships with the reference and there is
no network, so every throughput number is measured on these.  Shapes and value ranges follow the reference's clip files
(data/reader/sp_reader.py:25-57: x, y normalised by the frame width, confidence 1.0 for ground-truth 2-D input; labels root-relative,
data/preprocessor/clip_generate_sp.py:39-40)."""
from __future__ import annotations

import torch

ACTIONS = ("soccer", "tennis", "jump", "throw_baseball", "volley")     # 5 SportsPose activity names (data_action values)


def synthetic_clips(B, T, seed=1234, res=(1312, 1216), det_conf=False):
    """Inputs [B,T,17,3] (temporally smoothed x, y in [-1,1] x [-h/w,h/w], confidence 1 or ~U(0,1) for detector input) and root-relative
    labels [B,T,17,3], both CPU float32, from a seeded generator."""
    g = torch.Generator().manual_seed(seed)
    w, h = res
    noise = torch.rand(B, T, 17, 2, generator=g) * 2 - 1
    xy = torch.empty_like(noise)
    xy[:, 0] = noise[:, 0]
    for t in range(1, T):
        xy[:, t] = 0.9 * xy[:, t - 1] + 0.1 * noise[:, t]
    xy[..., 1] *= h / w
    conf = torch.rand(B, T, 17, 1, generator=g) if det_conf else torch.ones(B, T, 17, 1)
    x = torch.cat((xy, conf), dim=-1)
    y = torch.randn(B, T, 17, 3, generator=g) * 0.25
    y = y - y[:, :, :1]
    return x.contiguous(), y.contiguous()


def synthetic_test_extras(y, seed=4321, res_choices=((1312, 1216), (1216, 1936))):
    """Test-split fields of a clip (clip_generate_sp.py:52-79): per-frame ``factor`` ~U(0.8,1.2), ``res`` (w,h), an action name and
    ``label_scaled`` (mm) consistent with ``y`` plus noise, so that the metrics are non-zero."""
    g = torch.Generator().manual_seed(seed)
    B, T = y.shape[:2]
    factor = torch.rand(B, T, generator=g) * 0.4 + 0.8
    pick = torch.randint(0, len(res_choices), (B,), generator=g)
    res = torch.tensor([res_choices[int(i)] for i in pick], dtype=torch.int64)
    actions = [ACTIONS[int(i)] for i in torch.randint(0, len(ACTIONS), (B,), generator=g)]
    w = res[:, 0].float()[:, None, None, None]
    label_scaled = y * w / 2 * factor[:, :, None, None] + torch.randn(B, T, 17, 3, generator=g) * 20.0
    label_scaled = label_scaled - label_scaled[:, :, :1]
    return label_scaled.contiguous(), factor.contiguous(), res, actions
