"""Learning-rate schedule of the reference loop (train_and_evaluate_sp.py:273,325-329,393-397) driving the fused optimizer's param_groups."""
import random

import torch

import kasportsformer_amd as K


class _Opt:                      # what FusedAdamW exposes to a scheduler
    def __init__(self, lr):
        self.param_groups = [dict(lr=lr)]


def test_plateau_scheduler_matches_torch():
    rng = random.Random(3)
    for trial in range(5):
        w = torch.nn.Parameter(torch.zeros(1))
        topt = torch.optim.AdamW([w], lr=5e-4)
        ref = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, factor=0.9, patience=2)           # sp:273
        mine_opt = _Opt(5e-4)
        mine = K.ReduceLROnPlateau(mine_opt, factor=0.9, patience=2)
        metric = 100.0
        for epoch in range(120):
            metric = metric * (1.0 - 0.01 * rng.random()) if rng.random() < 0.4 else metric * (1.0 + 0.02 * rng.random())
            ref.step(metric)
            mine.step(metric)
            assert abs(mine_opt.param_groups[0]["lr"] - topt.param_groups[0]["lr"]) < 1e-15, (trial, epoch)
        assert topt.param_groups[0]["lr"] < 5e-4                                                  # the schedule did fire


def test_warmup_follows_reference_formula():
    lr, wu = 5e-4, 10                                                                              # configs/*.yaml:21,25-26
    assert K.warmup_lr(0, lr, wu) == lr / 100 and abs(K.warmup_lr(wu, lr, wu) - lr) < 1e-18
    opt = _Opt(lr)
    seen = [K.apply_warmup(opt, e, lr, wu) for e in range(13)]
    assert all(a < b for a, b in zip(seen[:10], seen[1:11])) and seen[10] == seen[11] == seen[12]   # epoch <= warmup_epoches ramps, then holds
    assert abs(seen[5] - (lr / 100 + (lr - lr / 100) * 0.5)) < 1e-18
