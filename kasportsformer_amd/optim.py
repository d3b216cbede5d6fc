"""AdamW over the model's flat parameter array (one kernel instead of 2,611 small updates).

Semantics are torch.optim.AdamW's (train_and_evaluate_sp.py:270-272: lr 5e-4, wd 0.01, betas (0.9, 0.999),
eps 1e-8); parameters that never receive a gradient (the 208 norm1_limb tensors of the non-bone blocks,
KASportsFormer.py:73) are skipped exactly like torch skips ``grad is None``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class FusedAdamW:
    def __init__(self, model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.model = model
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.step_index = 0
        self.exp_avg = torch.zeros(model.n_live, dtype=torch.float32, device=model._flat.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.grad_scale = 1.0            # DataParallel sets 1/world_size (all-reduce is a sum)

    def zero_grad(self, set_to_none=True):
        self.model.flat_grad = None
        if self.model.attach_param_grads:
            for p in self.model.parameters():
                p.grad = None

    @torch.no_grad()
    def step(self):
        m = self.model
        if m.flat_grad is None:
            raise RuntimeError("FusedAdamW.step() called before backward")
        g = self.param_groups[0]
        self.step_index += 1
        _lib.check(_lib.load().kasf_adamw_step(m._flat.data_ptr(), m.flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                               m.n_live, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_index,
                                               self.grad_scale, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        if m._const_index is not None:   # use_layer_scale=False: the layer-scale slices are the constant 1, not parameters
            m._flat.index_fill_(0, m._const_index, 1.0)
        m.mark_weights_dirty()

    def state_dict(self):
        return {"step": self.step_index, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "param_groups": self.param_groups}

    def load_state_dict(self, sd):
        self.step_index = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups = sd["param_groups"]
