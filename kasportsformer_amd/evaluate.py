"""Evaluation side of the path on the GPU.

Mirrors, with the same names and argument meaning:
  joint_flip                      utils/utilities.py:128-135
  predict_flip_tta                train_and_evaluate_sp.py:46-55   (two forwards, flip back, average, root zeroing)
  Evaluator / evaluate_one_epoch  train_and_evaluate_sp.py:27-149  (de-normalise, x factor, root-centre, MPJPE / per-joint error /
                                  acceleration error / P-MPJPE per frame, macro-average over actions), utils/error_calc.py:5-48

The reference copies every batch to the host and loops over clips in numpy; here one kernel launch per batch produces the
per-frame metrics and accumulates the per-action sums on the device, and the host reads a [n_actions, 22] fp64 table once at the end.
GPU only: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _poses(x: torch.Tensor, what: str) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError(f"kasportsformer_amd.{what} runs on the GPU only")
    if x.shape[-2:] != (17, 3):
        raise RuntimeError(f"{what}: expected [..., 17, 3] poses, got {tuple(x.shape)}")
    return x.contiguous().float()


def joint_flip(joint_data: torch.Tensor, deep_copy: bool = True) -> torch.Tensor:
    """x negated and left joints [1,2,3,14,15,16] swapped with right [4,5,6,11,12,13].  ``deep_copy=False`` writes the result back
    into ``joint_data`` (the reference's in-place mode) and returns it."""
    src = _poses(joint_data, "joint_flip")
    out = torch.empty_like(src)
    _lib.check(_lib.load().kasf_joint_flip(src.data_ptr(), out.data_ptr(), src.numel() // 51, _stream()))
    if not deep_copy:
        joint_data.copy_(out)
        return joint_data
    return out


def predict_flip_tta(model, joint_input: torch.Tensor, flip: bool = True) -> torch.Tensor:
    """The evaluation-time prediction: flip-TTA average (``flip=True``, configs/*.yaml:53) and root joint set to zero.
    In eval mode BatchNorm uses running statistics, so both views go through ONE forward of the stacked batch."""
    x = _poses(joint_input, "predict_flip_tta")
    lib = _lib.load()
    B = x.shape[0]
    with torch.no_grad():
        if flip:
            both = torch.empty((2 * B,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
            both[:B].copy_(x)
            _lib.check(lib.kasf_joint_flip(x.data_ptr(), both[B:].data_ptr(), x.numel() // 51, _stream()))
            if model.training:                       # batch statistics would couple the two views: keep them separate
                p, pf = model(both[:B]), model(both[B:])
            else:
                pp = model(both)
                p, pf = pp[:B], pp[B:]
            out = torch.empty_like(p)
            _lib.check(lib.kasf_tta_merge(p.data_ptr(), pf.data_ptr(), out.data_ptr(), p.numel() // 51, _stream()))
        else:
            p = model(x)
            out = torch.empty_like(p)
            _lib.check(lib.kasf_tta_merge(p.data_ptr(), None, out.data_ptr(), p.numel() // 51, _stream()))
    return out


def clip_metrics(pred: torch.Tensor, label_scaled: torch.Tensor, factor: torch.Tensor, res: torch.Tensor, action_ids: torch.Tensor | None = None,
                 action_sums: torch.Tensor | None = None):
    """Per-frame metrics of a batch: returns (mpjpe [B,T], jpe [B,T,17], accel [B,T-2], p_mpjpe [B,T]) as device tensors.
    ``action_sums`` ([n_actions, 22] float64, device) is accumulated into when given together with ``action_ids`` ([B] int32)."""
    pred = _poses(pred, "clip_metrics")
    dev = pred.device
    B, T = pred.shape[0], pred.shape[1]
    label_scaled = _poses(label_scaled.to(dev), "clip_metrics")
    factor = factor.to(dev).contiguous().float()
    res = res.to(dev).contiguous().float()
    if label_scaled.shape != pred.shape or factor.shape != (B, T) or res.shape != (B, 2):
        raise RuntimeError("clip_metrics: label_scaled [B,T,17,3], factor [B,T], res [B,2] expected")
    mp = torch.empty(B, T, device=dev)
    pm = torch.empty(B, T, device=dev)
    ac = torch.empty(B, T - 2, device=dev)
    jp = torch.empty(B, T, 17, device=dev)
    a_ptr = s_ptr = None
    n_act = 0
    if action_sums is not None:
        action_ids = action_ids.to(dev).contiguous().to(torch.int32)
        assert action_sums.dtype == torch.float64 and action_sums.is_contiguous() and action_sums.shape[1] == _lib.EVAL_COLS
        a_ptr, s_ptr, n_act = action_ids.data_ptr(), action_sums.data_ptr(), action_sums.shape[0]
    _lib.check(_lib.load().kasf_eval_metrics(pred.data_ptr(), label_scaled.data_ptr(), factor.data_ptr(), res.data_ptr(), a_ptr, B, T, n_act,
                                             mp.data_ptr(), pm.data_ptr(), ac.data_ptr(), jp.data_ptr(), s_ptr, _stream()))
    return mp, jp, ac, pm


class Evaluator:
    """Accumulates the reference's evaluation result over batches without leaving the device."""

    MAX_ACTIONS = 64

    def __init__(self, num_joints: int = 17, device="cuda", action_names=None):
        """``action_names``: fixed id table (e.g. ``PackedClips.action_names``) -- required when several ranks evaluate shards and
        ``reduce()`` sums their tables; without it ids are handed out in first-seen order."""
        if num_joints != 17:
            raise NotImplementedError("17-joint skeleton only")
        self.sums = torch.zeros(self.MAX_ACTIONS, _lib.EVAL_COLS, dtype=torch.float64, device=device)
        self.action_ids = {a: i for i, a in enumerate(action_names or ())}     # name -> id
        if len(self.action_ids) > self.MAX_ACTIONS:
            raise RuntimeError("too many distinct actions")

    def reduce(self, group=None):
        """Sums the per-action tables of all ranks (one 11 KB all-reduce per evaluation)."""
        import torch.distributed as dist
        dist.all_reduce(self.sums, group=group)

    def update(self, predicted_result, joint_label_scaled, joint_factor, joint_action, joint_res):
        """Same five per-batch values the reference's loop handles (``predicted_result`` = output of ``predict_flip_tta``)."""
        for a in joint_action:
            if a not in self.action_ids:
                if len(self.action_ids) == self.MAX_ACTIONS:
                    raise RuntimeError("too many distinct actions")
                self.action_ids[a] = len(self.action_ids)
        ids = torch.tensor([self.action_ids[a] for a in joint_action], dtype=torch.int32)
        res = joint_res if torch.is_tensor(joint_res) else torch.as_tensor(np.asarray(joint_res))
        return clip_metrics(predicted_result, joint_label_scaled, joint_factor, res, ids, self.sums)

    def result(self):
        names = list(self.action_ids)
        s = self.sums[:len(names)].cpu().numpy()            # the only device->host copy of the evaluation
        seen = s[:, 20] > 0
        names, s = [n for n, k in zip(names, seen) if k], s[seen]
        m_act = s[:, 0] / s[:, 20]
        p_act = s[:, 1] / s[:, 20]
        a_act = s[:, 2] / s[:, 21]
        j_act = s[:, 3:20] / s[:, 20:21]
        return {"mpjpe": float(m_act.mean()), "p_mpjpe": float(p_act.mean()), "acceleration_error": float(a_act.mean()),
                "activity_name_sequence": names, "mpjpe_activity": [float(v) for v in m_act], "mpjpe_joint": j_act.mean(axis=0)}


def _broadcast_buffers(model, data_parallel=None):
    import torch.distributed as dist
    if data_parallel is not None:
        return data_parallel.sync_buffers_from_rank0()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in (model._flat_buffers, model._nbt):
            dist.broadcast(t, src=0)


def evaluate_one_epoch(model, test_loader, flip: bool = True, device="cuda", action_names=None, distributed: bool = False, data_parallel=None):
    """evaluate_one_epoch_new (train_and_evaluate_sp.py:27-149) / evaluate_one_epoch (train_and_evaluate_wp.py:25-135): ``test_loader``
    yields (joint_input, joint_label_scaled, joint_factor, joint_action, joint_res) like the reference's test DataLoader.
    ``distributed``: every rank evaluates its shard and the per-action tables are summed.  All ranks must then evaluate the SAME model:
    BatchNorm running statistics are per-rank during training, so rank 0's are broadcast first (``nn.DataParallel`` keeps replica 0's)."""
    was_training = model.training
    model.eval()
    if distributed:
        _broadcast_buffers(model, data_parallel)
    ev = Evaluator(device=device, action_names=action_names)
    for joint_input, joint_label_scaled, joint_factor, joint_action, joint_res in test_loader:
        pred = predict_flip_tta(model, joint_input.to(device), flip)
        ev.update(pred, joint_label_scaled, joint_factor, list(joint_action), joint_res)
    model.train(was_training)
    if distributed:
        ev.reduce()
    return ev.result()
