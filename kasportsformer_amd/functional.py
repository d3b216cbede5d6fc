"""Fused training loss of the path (utils/loss_calc.py:6-27 combined as in train_and_evaluate_sp.py:212-222)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class _Loss3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, lambda_n, lambda_v):
        if not pred.is_cuda:
            raise RuntimeError("kasportsformer_amd.loss3 runs on the GPU only")
        pred, target = pred.contiguous().float(), target.contiguous().float()
        B, T = pred.shape[0], pred.shape[1]
        dpred = torch.empty_like(pred)
        scratch = torch.empty(4 + 4 * B, dtype=torch.float32, device=pred.device)      # [0:4] the result, the rest per-clip sums (kasf.h)
        lib = _lib.load()
        _lib.check(lib.kasf_loss3(pred.data_ptr(), target.data_ptr(), dpred.data_ptr(), scratch.data_ptr(), scratch.numel(), B, T, float(lambda_n), float(lambda_v),
                                  1.0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        losses = scratch[:4]
        ctx.save_for_backward(dpred)
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), losses

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        (dpred,) = ctx.saved_tensors
        return dpred * g_total, None, None, None


def loss3(pred: torch.Tensor, target: torch.Tensor, lambda_n_mpjpe: float = 0.5, lambda_velocity: float = 20.0):
    """Returns (total, parts) with parts = [total, mpjpe, n_mpjpe, velocity] (device tensor, no host sync).
    total = mpjpe + lambda_n * n_mpjpe + lambda_v * velocity (configs/*.yaml:30-31)."""
    return _Loss3.apply(pred, target, lambda_n_mpjpe, lambda_velocity)
