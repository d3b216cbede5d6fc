"""Learning-rate schedule of the reference training loop, for the fused optimizer.

  warmup_lr            train_and_evaluate_sp.py:325-329   linear ramp from lr/100 to lr over ``warmup_epoches`` epochs (inclusive bound)
  ReduceLROnPlateau    train_and_evaluate_sp.py:273,393-397   optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.9, patience=2) stepped on the
                       validation MPJPE after the warm-up epochs; torch's class insists on a torch.optim.Optimizer, FusedAdamW only has ``param_groups``,
                       so its defaults (mode 'min', threshold 1e-4 'rel', cooldown 0, min_lr 0, eps 1e-8) are restated here and tested against it.
"""
from __future__ import annotations


def warmup_lr(epoch: int, learning_rate: float, warmup_epoches: int) -> float:
    """Learning rate of ``epoch`` (0-based) while ``epoch <= warmup_epoches``."""
    start = learning_rate / 100
    return start + (learning_rate - start) * (epoch / warmup_epoches)


def apply_warmup(optimizer, epoch: int, learning_rate: float, warmup_epoches: int, warmup: bool = True) -> float:
    """The per-epoch prologue of the reference loop: sets and returns the learning rate in effect."""
    if warmup and epoch <= warmup_epoches:
        for g in optimizer.param_groups:
            g["lr"] = warmup_lr(epoch, learning_rate, warmup_epoches)
    return optimizer.param_groups[0]["lr"]


class ReduceLROnPlateau:
    def __init__(self, optimizer, factor: float = 0.1, patience: int = 10, threshold: float = 1e-4, cooldown: int = 0, min_lr: float = 0.0, eps: float = 1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.optimizer, self.factor, self.patience, self.threshold = optimizer, factor, patience, threshold
        self.cooldown, self.min_lr, self.eps = cooldown, min_lr, eps
        self.best, self.num_bad_epochs, self.cooldown_counter, self.last_epoch = float("inf"), 0, 0, 0

    def step(self, metrics: float):
        current = float(metrics)
        self.last_epoch += 1
        if current < self.best * (1.0 - self.threshold):
            self.best, self.num_bad_epochs = current, 0
        else:
            self.num_bad_epochs += 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad_epochs = 0
        if self.num_bad_epochs > self.patience:
            for g in self.optimizer.param_groups:
                new_lr = max(float(g["lr"]) * self.factor, self.min_lr)
                if float(g["lr"]) - new_lr > self.eps:
                    g["lr"] = new_lr
            self.cooldown_counter = self.cooldown
            self.num_bad_epochs = 0

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)
