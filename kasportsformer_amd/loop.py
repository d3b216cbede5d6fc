"""The per-epoch training loop that drives the path (train_and_evaluate_sp.py:201-243 / train_and_evaluate_wp.py:187-229).

Same order of operations -- forward, zero_grad, 3-term loss, backward, step -- with the four ``.item()`` host synchronisations per step
of the reference replaced by a device-side running sum: the batch-size-weighted averages (what ``AverageMetering`` reports) are read
once per epoch.
"""
from __future__ import annotations

import torch

from .functional import loss3


def train_one_epoch(model, train_loader, optimizer, data_parallel=None, lambda_n_mpjpe: float = 0.5, lambda_mpjpe_velocity: float = 20.0,
                    device="cuda") -> dict:
    """One pass over ``train_loader`` (batches ``(x, y)`` like the reference's DataLoader or ``DeviceClipLoader``).  Returns the epoch
    averages ``{'loss_total', 'loss_mpjpe', 'loss_n_mpjpe', 'loss_velocity'}`` weighted by batch size (utils/utilities.py:95-108)."""
    model.train()
    sums = torch.zeros(4, dtype=torch.float64, device=device)
    count = 0
    for x, y in train_loader:
        x, y = x.to(device), y.to(device)
        predict_result = model(x)
        optimizer.zero_grad()
        loss_total, parts = loss3(predict_result, y, lambda_n_mpjpe, lambda_mpjpe_velocity)    # parts = [total, mpjpe, n_mpjpe, velocity]
        sums += parts.double() * x.shape[0]
        count += x.shape[0]
        loss_total.backward()
        if data_parallel is not None:
            data_parallel.finish_gradients(optimizer)
        optimizer.step()
    avg = (sums / max(count, 1)).cpu().tolist()               # the only host synchronisation of the epoch
    return {"loss_total": avg[0], "loss_mpjpe": avg[1], "loss_n_mpjpe": avg[2], "loss_velocity": avg[3]}
