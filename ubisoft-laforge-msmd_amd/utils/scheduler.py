"""Learning-rate warm-up (drop-in surface of reference utils/scheduler.py:8-67); host-side arithmetic only."""
from torch.optim.lr_scheduler import ReduceLROnPlateau, _LRScheduler


class GradualWarmupScheduler(_LRScheduler):
    """lr ramps from base_lr*(0 if multiplier == 1 else 1) to base_lr*multiplier over `total_epoch` scheduler
    steps, then `after_scheduler` (optional) takes over.  Same constructor and methods as the reference."""

    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler=None):
        if multiplier < 1.0:
            raise ValueError("multiplier should be greater thant or equal to 1.")
        self.multiplier = multiplier
        self.total_epoch = total_epoch
        self.after_scheduler = after_scheduler
        self.finished = False
        super().__init__(optimizer)

    def _warm(self, epoch):
        if self.multiplier == 1.0:
            return [b * (float(epoch) / self.total_epoch) for b in self.base_lrs]
        return [b * ((self.multiplier - 1.0) * epoch / self.total_epoch + 1.0) for b in self.base_lrs]

    def get_lr(self):
        if self.last_epoch <= self.total_epoch:
            return self._warm(self.last_epoch)
        if self.after_scheduler is None:
            return [b * self.multiplier for b in self.base_lrs]
        if not self.finished:
            self.after_scheduler.base_lrs = [b * self.multiplier for b in self.base_lrs]
            self.finished = True
        return self.after_scheduler.get_last_lr()

    def step_ReduceLROnPlateau(self, metrics, epoch=None):
        given = epoch
        epoch = self.last_epoch + 1 if epoch is None else epoch
        self.last_epoch = epoch if epoch != 0 else 1
        if self.last_epoch <= self.total_epoch:
            for group, lr in zip(self.optimizer.param_groups, [b * ((self.multiplier - 1.0) * self.last_epoch /
                                                                   self.total_epoch + 1.0) for b in self.base_lrs]):
                group["lr"] = lr
        else:
            self.after_scheduler.step(metrics, None if given is None else given - self.total_epoch)

    def step(self, epoch=None, metrics=None):
        if isinstance(self.after_scheduler, ReduceLROnPlateau):
            return self.step_ReduceLROnPlateau(metrics, epoch)
        if self.finished and self.after_scheduler:
            self.after_scheduler.step(None if epoch is None else epoch - self.total_epoch)
            self._last_lr = self.after_scheduler.get_last_lr()
            return None
        return super().step(epoch)


class LrSchedule:
    """The reference's learning-rate schedules (training_script.py:571-581, stepping rule l.222-224) on a 1-element
    host-side optimizer: `lr` is the value the reference's optimizer would use at the current iteration.
    'Warmup' = GradualWarmupScheduler(opt, 1, warm_iter); 'WarmupThenDecay' puts CosineAnnealingLR(opt, cos_max_iter -
    warm_iter, lr * min_lr_ratio) behind it and stops stepping at cos_max_iter; anything else = constant lr."""

    def __init__(self, args):
        import torch
        self.kind = getattr(args, "scheduler", "Warmup")
        self.base = float(args.lr)
        self.warm = int(getattr(args, "warm_iter", 0) or 0)
        self.cos_max = int(getattr(args, "cos_max_iter", 0) or 0)
        self.opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=self.base)
        self.sched = None
        self.calls = 0
        if self.warm > 0 and self.kind in ("Warmup", "WarmupThenDecay"):
            after = None
            if self.kind == "WarmupThenDecay":
                after = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, self.cos_max - self.warm,
                                                                   self.base * float(args.min_lr_ratio))
            self.sched = GradualWarmupScheduler(self.opt, 1, self.warm, after)

    @property
    def lr(self):
        return float(self.opt.param_groups[0]["lr"]) if self.sched is not None else self.base

    def step(self, it):
        """Call once per iteration AFTER the optimizer step, with the iteration index (reference l.222-224)."""
        if self.sched is not None and (self.kind != "WarmupThenDecay" or it < self.cos_max):
            self.opt.step()   # no-op on the gradient-less dummy parameter; keeps torch's call-order check quiet
            self.sched.step()
            self.calls += 1

    def replay(self, n_calls):
        for _ in range(int(n_calls)):
            self.opt.step()
            self.sched.step()
        self.calls = int(n_calls)
