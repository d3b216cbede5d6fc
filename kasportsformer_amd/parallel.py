"""Single-node data parallelism: one process per GPU, replicated weights, RCCL all-reduce over xGMI.

Replaces the reference's ``nn.DataParallel`` (train_and_evaluate_sp.py:262-263).  Clips are independent
except for BatchNorm batch statistics, which the reference already computes per replica; the only
exchange step is the gradient all-reduce.  Gradients live in ONE flat fp32 array laid out layer by layer,
so each backward stage finalises a contiguous range that is all-reduced (sum) asynchronously while the
next layer's backward kernels run; 1/world_size is folded into the optimizer step.

Set ``GPU_MAX_HW_QUEUES=8`` in the environment of every rank (before the process loads the HIP runtime): the engine keeps three streams
busy, RCCL brings its own, and with the default of 4 hardware queues per process they end up sharing queues -- measured -6 % on the step
with a process group merely initialised, gone with 8 queues.
"""
from __future__ import annotations

import os
import warnings

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, model, process_group=None, overlap=True, stages_per_bucket=7, optimizer=None, grad_dtype="fp32"):
        """``stages_per_bucket`` backward stages (layers) share one all-reduce: 4 buckets of ~30 MB instead of 27 of 4.5 MB keep the
        per-bucket host and launch overhead off the step (xGMI ring all-reduce of 117 MB is ~1 ms; the point of the buckets is overlap).

        ``grad_dtype="bf16"`` (optional): every bucket is cast to bf16, all-reduced (58.7 MB per step instead of 117.4 MB on the wire) and
        widened back into the flat fp32 gradient; master weights, moments and the optimizer step stay fp32.  The sum over ranks is then formed in bf16 by the
        collective (every addition rounds to 8 significant bits): at the per-rank batch of configs[2] (32 clips, a 14 ms step) it halves what the all-reduce has
        to hide; the default stays fp32."""
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError(f"grad_dtype {grad_dtype!r}: 'fp32' or 'bf16'")
        self.grad_dtype = grad_dtype
        self._wire = {}                  # bucket index within the step -> persistent bf16 staging buffer (grown when a later step's bucket is larger)
        self._bucket = 0                 # buckets reduced so far in this step (reset by finish_gradients)
        if not dist.is_initialized():
            raise RuntimeError("init torch.distributed first (backend 'nccl' = RCCL on ROCm; 'gloo' for CPU rehearsal)")
        if model._flat.is_cuda and os.environ.get("GPU_MAX_HW_QUEUES") is None:
            warnings.warn("kasportsformer_amd.DataParallel: GPU_MAX_HW_QUEUES is not set; with RCCL's streams the three branch streams of the engine "
                          "share hardware queues (about -6 % throughput).  Export GPU_MAX_HW_QUEUES=8 before starting the ranks.")
        self.model, self.group, self.overlap = model, process_group, overlap
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self._pending = []
        self.optimizer = None
        self._scale_in_place = False
        if optimizer is not None:
            self.attach_optimizer(optimizer)
        self.sync_from_rank0()
        model.grad_stage_hook = self._on_stage if overlap else None
        model.grad_stage_group = stages_per_bucket

    def attach_optimizer(self, optimizer):
        """The all-reduce is a SUM; the mean over ranks is taken once per step, where it is cheapest for the optimizer at hand: ``FusedAdamW`` folds
        ``grad_scale = 1 / world_size`` into its update; for any other optimizer (INTEGRATION.md path A: a stock ``torch.optim.AdamW(model.parameters())``,
        whose per-parameter ``.grad`` tensors are views of the flat gradient) ``finish_gradients`` scales the finished flat gradient in place, one launch."""
        self._scale_in_place = not hasattr(optimizer, "grad_scale")
        if self._scale_in_place:
            if not self.model.attach_param_grads:
                raise RuntimeError("a torch.optim optimizer reads p.grad: leave model.attach_param_grads = True (the per-parameter views of the flat gradient)")
        else:
            optimizer.grad_scale = 1.0 / self.world
        self.optimizer = optimizer

    def sync_buffers_from_rank0(self):
        """BatchNorm running statistics / counters of rank 0 to every rank: what evaluation and checkpoints must see (``nn.DataParallel``
        only ever keeps replica 0's buffer updates, train_and_evaluate_sp.py:262-264).  9.4 KB."""
        m = self.model
        for t in (m._flat_buffers, m._nbt):
            dist.broadcast(t, src=0, group=self.group)

    def sync_from_rank0(self):
        """Broadcast parameters and BatchNorm buffers (DataParallel keeps replica 0's buffers)."""
        m = self.model
        for t in (m._flat, m._flat_buffers, m._nbt):
            dist.broadcast(t, src=0, group=self.group)
        m.mark_weights_dirty()

    def _reduce(self, grad_slice, async_op):
        """One bucket: returns (work, staging buffer or None).  bf16: cast on the current stream (the bucket's gradients are final there), reduce the copy."""
        if self.grad_dtype == "fp32":
            return dist.all_reduce(grad_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op), None
        # keyed by the bucket's ORDER in the step, never by its address: the flat gradient is a fresh allocation every step (model.py: _launch_backward), so an
        # address key grew a new 58 MB set whenever the caching allocator moved it (a different batch size, an evaluation pass in between) and never freed one
        key, n = self._bucket, grad_slice.numel()
        self._bucket += 1
        buf = self._wire.get(key)
        if buf is None or buf.numel() < n or buf.device != grad_slice.device:
            buf = self._wire[key] = torch.empty(n, dtype=torch.bfloat16, device=grad_slice.device)
        buf = buf[:n]
        buf.copy_(grad_slice)
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op), buf

    def _on_stage(self, stage, grad_slice):
        work, buf = self._reduce(grad_slice, True)
        self._pending.append((work, buf, grad_slice))

    def finish_gradients(self, optimizer=None):
        """Call after loss.backward(): waits for the bucketed all-reduces (or runs one when overlap is off).  Pass (or attach) the optimizer so
        that its ``grad_scale`` is 1/world_size; stepping with an unscaled optimizer at world_size > 1 raises instead of silently training at
        world_size times the gradient."""
        m = self.model
        opt = optimizer if optimizer is not None else self.optimizer
        if opt is not None and self.optimizer is not opt:
            self.attach_optimizer(opt)
        if self.world > 1 and self.optimizer is None:
            raise RuntimeError("DataParallel.finish_gradients: no optimizer attached (DataParallel(model, optimizer=opt) or finish_gradients(opt)): "
                               "the all-reduced gradient is a sum over ranks and nobody would divide it by world_size")
        if self.overlap:
            for work, buf, grad_slice in self._pending:
                work.wait()
                if buf is not None:
                    grad_slice.copy_(buf)                  # widen the reduced bf16 bucket back into the flat fp32 gradient
            self._pending = []
        elif m.flat_grad is not None:
            self._bucket = 0
            _, buf = self._reduce(m.flat_grad[:m.n_live], False)
            if buf is not None:
                m.flat_grad[:m.n_live].copy_(buf)
        self._bucket = 0
        if self._scale_in_place and self.world > 1 and m.flat_grad is not None:
            m.flat_grad[:m.n_live].mul_(1.0 / self.world)         # p.grad of every live parameter is a view of this array

    def __call__(self, x, return_rep=False):
        return self.model(x, return_rep)
