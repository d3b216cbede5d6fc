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


def _lr_flip(p):
    """utils/utilities.py:128-135 on a CPU tensor [..., 17, C]: first channel negated, left joints [1,2,3,14,15,16] swapped with right [4,5,6,11,12,13]."""
    out = p.clone()
    out[..., 0] = -out[..., 0]
    left, right = [1, 2, 3, 14, 15, 16], [4, 5, 6, 11, 12, 13]
    out[..., left + right, :] = out[..., right + left, :].clone()
    return out


def teacher_labels(x, seed=2024):
    """LEARNABLE labels for training-fidelity runs: a fixed, seeded two-layer map of the 2-D pose instead of independent noise (on noise labels
    MPJPE stays at the label scale whatever the model does, which hides any training-quality gap between arithmetic modes).
    x [B,T,17,3] -> root-relative y [B,T,17,3]: x/y = half the root-relative 2-D pose, depth = W2 . tanh(W1 . [pose, temporal-neighbour mean])
    per frame, so that the spatial AND the temporal mixers have something to learn; symmetrised over the left/right flip like real poses are
    (the evaluation procedure averages a flipped view, train_and_evaluate_sp.py:46-51)."""
    g = torch.Generator().manual_seed(seed)
    w1 = torch.randn(34, 64, generator=g) * (3.0 / 34 ** 0.5)
    w2 = torch.randn(64, 17, generator=g) * (0.3 / 8.0)

    def base(x):
        B, T = x.shape[:2]
        xy = x[..., :2] - x[:, :, :1, :2]
        f = xy.reshape(B, T, 34)
        ctx = f.clone()
        if T > 2:
            ctx[:, 1:-1] = (f[:, :-2] + f[:, 2:]) / 2
        z = torch.tanh((0.5 * f + 0.5 * ctx) @ w1) @ w2
        return torch.cat((0.5 * xy, z.unsqueeze(-1)), dim=-1)

    y = 0.5 * (base(x) + _lr_flip(base(_lr_flip(x))))
    return (y - y[:, :, :1]).contiguous()


def teacher_clips(B, T, seed=1234, res=(1312, 1216), det_conf=False, teacher_seed=2024):
    """``synthetic_clips`` inputs with ``teacher_labels`` labels."""
    x, _ = synthetic_clips(B, T, seed=seed, res=res, det_conf=det_conf)
    return x, teacher_labels(x, teacher_seed)


def synthetic_test_extras(y, seed=4321, res_choices=((1312, 1216), (1216, 1936)), noise_mm=20.0):
    """Test-split fields of a clip (clip_generate_sp.py:52-79): per-frame ``factor`` ~U(0.8,1.2), ``res`` (w,h), an action name and
    ``label_scaled`` (mm) consistent with ``y`` plus noise, so that the metrics are non-zero."""
    g = torch.Generator().manual_seed(seed)
    B, T = y.shape[:2]
    factor = torch.rand(B, T, generator=g) * 0.4 + 0.8
    pick = torch.randint(0, len(res_choices), (B,), generator=g)
    res = torch.tensor([res_choices[int(i)] for i in pick], dtype=torch.int64)
    actions = [ACTIONS[int(i)] for i in torch.randint(0, len(ACTIONS), (B,), generator=g)]
    w = res[:, 0].float()[:, None, None, None]
    label_scaled = y * w / 2 * factor[:, :, None, None] + torch.randn(B, T, 17, 3, generator=g) * noise_mm
    label_scaled = label_scaled - label_scaled[:, :, :1]
    return label_scaled.contiguous(), factor.contiguous(), res, actions
