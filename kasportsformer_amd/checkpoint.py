"""Checkpoint files interchangeable with the reference's training scripts.

File layout (utils/utilities.py:110-118): ``torch.save`` of
``{'epoch': epoch + 1, 'learning_rate': lr, 'optimizer': AdamW.state_dict(), 'model': state_dict, 'min_mpjpe': float, 'wandb_id': str}``.
The reference wraps the model in ``nn.DataParallel`` before saving, so every model key carries a ``module.`` prefix
(train_and_evaluate_sp.py:262-264), and its resume code reads ``checkpoint['lr']`` / ``checkpoint['wandb_run_id']`` although the
writer stores ``'learning_rate'`` / ``'wandb_id'`` (:291-297) -- the reader here accepts both spellings.

The optimiser entry is a genuine ``torch.optim.AdamW`` state_dict (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` in
``model.parameters()`` order, no entry for parameters that never received a gradient), built from / scattered into
``FusedAdamW``'s flat moment arrays, so either side can resume the other's run.
"""
from __future__ import annotations

import os

import torch

PREFIX = "module."


def strip_module_prefix(state_dict: dict) -> dict:
    """Keys of a DataParallel-saved state_dict without their ``module.`` prefix (no-op for plain keys)."""
    return {(k[len(PREFIX):] if k.startswith(PREFIX) else k): v for k, v in state_dict.items()}


def adamw_state_dict(optimizer) -> dict:
    """``FusedAdamW`` -> ``torch.optim.AdamW.state_dict()`` layout."""
    model, g = optimizer.model, optimizer.param_groups[0]
    live = {id(p): (off, numel, shape) for p, off, numel, shape in model._live}
    state, count = {}, 0
    for n, p in enumerate(model.parameters()):
        count = n + 1
        if id(p) in live and optimizer.step_index > 0:
            off, numel, shape = live[id(p)]
            state[n] = {"step": torch.tensor(float(optimizer.step_index)),
                        "exp_avg": optimizer.exp_avg[off:off + numel].view(shape).detach().cpu().clone(),
                        "exp_avg_sq": optimizer.exp_avg_sq[off:off + numel].view(shape).detach().cpu().clone()}
    group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(count))}
    return {"state": state, "param_groups": [group]}


def load_adamw_state_dict(optimizer, sd: dict):
    """``torch.optim.AdamW.state_dict()`` (or ``FusedAdamW.state_dict()``) -> ``FusedAdamW``."""
    if "exp_avg" in sd:                                  # FusedAdamW's own flat form
        return optimizer.load_state_dict(sd)
    model = optimizer.model
    params = list(model.parameters())
    group = sd["param_groups"][0]
    if len(group["params"]) != len(params):
        raise ValueError(f"optimizer state covers {len(group['params'])} parameters, the model has {len(params)}")
    live = {id(p): (off, numel, shape) for p, off, numel, shape in model._live}
    optimizer.exp_avg.zero_()
    optimizer.exp_avg_sq.zero_()
    steps = set()
    for slot, pid in enumerate(group["params"]):
        st = sd["state"].get(pid)
        if st is None:
            continue
        p = params[slot]
        if id(p) not in live:
            raise ValueError("optimizer state present for a parameter that never receives a gradient")
        off, numel, shape = live[id(p)]
        if tuple(st["exp_avg"].shape) != tuple(shape):
            raise ValueError(f"optimizer state shape {tuple(st['exp_avg'].shape)} does not match parameter {tuple(shape)}")
        optimizer.exp_avg[off:off + numel].copy_(st["exp_avg"].reshape(-1))
        optimizer.exp_avg_sq[off:off + numel].copy_(st["exp_avg_sq"].reshape(-1))
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): one fused update cannot resume that")
    optimizer.step_index = steps.pop() if steps else 0
    pg = optimizer.param_groups[0]
    pg.update(lr=float(group["lr"]), betas=tuple(group["betas"]), eps=float(group["eps"]), weight_decay=float(group["weight_decay"]))


def _load_file(path):
    """``torch.load(weights_only=True)`` that also accepts what the reference's loop pickles next to the tensors: ``min_mpjpe`` is the
    ``np.float64`` that ``np.mean`` returned (train_and_evaluate_sp.py:350-351), i.e. ``numpy.core.multiarray.scalar`` + a dtype object."""
    import numpy as np
    allow = [np.dtype, np.float64, np.float32, np.int64, type(np.dtype(np.float64)), type(np.dtype(np.float32)), type(np.dtype(np.int64))]
    for mod in ("numpy._core.multiarray", "numpy.core.multiarray"):
        try:
            allow.append(getattr(__import__(mod, fromlist=["scalar"]), "scalar"))
        except (ImportError, AttributeError):
            pass
    with torch.serialization.safe_globals(allow):
        return torch.load(path, map_location="cpu", weights_only=True)


def checkpoint_save(checkpoint_path, epoch, lr, optimizer, model, min_mpjpe, wandb_id, module_prefix: bool = True, data_parallel=None):
    """utils/utilities.py:110-118, same argument order.  ``module_prefix=True`` writes the keys the reference's DataParallel-wrapped
    ``load_state_dict(strict=True)`` expects.  With ``data_parallel`` (every rank must call): rank 0's BatchNorm buffers are broadcast first
    -- ``nn.DataParallel`` keeps replica 0's running statistics (train_and_evaluate_sp.py:262-264) -- and only rank 0 writes the file."""
    if data_parallel is not None:
        data_parallel.sync_buffers_from_rank0()
        if data_parallel.rank != 0:
            return
    msd = {((PREFIX + k) if module_prefix else k): v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tmp = str(checkpoint_path) + ".tmp"
    torch.save({"epoch": epoch + 1, "learning_rate": lr, "optimizer": adamw_state_dict(optimizer) if optimizer is not None else None, "model": msd,
                "min_mpjpe": float(min_mpjpe), "wandb_id": wandb_id}, tmp)
    os.replace(tmp, checkpoint_path)


def checkpoint_load(checkpoint_path, model, optimizer=None, resume: bool = False) -> dict:
    """train_and_evaluate_sp.py:171-176 (evaluation) and :285-301 (training, ``resume``).  Loads the weights ``strict=True`` whichever way
    the keys are prefixed; with ``resume`` also restores the optimiser and returns ``epoch`` / ``lr`` / ``min_mpjpe`` / ``wandb_run_id``."""
    if not os.path.exists(checkpoint_path):
        raise Exception("checkpoint path is wrong, check your configuration")          # sp:300-301
    ck = _load_file(checkpoint_path)
    model.load_state_dict(strip_module_prefix(ck["model"]), strict=True)
    info = {"epoch": 0, "lr": None, "min_mpjpe": float("inf"), "wandb_run_id": None}
    if resume:
        info["lr"] = ck["lr"] if "lr" in ck else ck.get("learning_rate")
        info["epoch"] = ck["epoch"]
        info["min_mpjpe"] = ck["min_mpjpe"]
        info["wandb_run_id"] = ck.get("wandb_run_id", ck.get("wandb_id"))
        if optimizer is not None and ck.get("optimizer") is not None:
            load_adamw_state_dict(optimizer, ck["optimizer"])
    return info
