"""Noam learning-rate schedule around an optimiser, with the interface of the reference's
ScheduledOptim (lib/transformer/Optim.py:6-39): `step_and_update_lr()`, `zero_grad()`, and the attributes
`_optimizer`, `n_warmup_steps`, `n_current_steps`, `init_lr`.

    lr(n) = d_model^-0.5 * min(n^-0.5, n * n_warmup_steps^-1.5)

where n counts calls of step_and_update_lr() and is incremented BEFORE the rate is evaluated, so the first
update uses lr(1).  The rate replaces whatever `lr` the optimiser was built with.
"""
import math


def noam_lr(step, d_model, n_warmup_steps):
    """Rate of update number `step` (1-based)."""
    warm = step * n_warmup_steps ** -1.5
    decay = 1.0 / math.sqrt(step)
    return min(decay, warm) / math.sqrt(d_model)


class ScheduledOptim:
    def __init__(self, optimizer, d_model, n_warmup_steps):
        self._optimizer = optimizer
        self._d_model = d_model
        self.n_warmup_steps = n_warmup_steps
        self.n_current_steps = 0
        self.init_lr = d_model ** -0.5

    # reference-named helpers, kept for callers that poke at them
    def _get_lr_scale(self):
        return noam_lr(self.n_current_steps, 1.0, self.n_warmup_steps)

    def _update_learning_rate(self):
        self.n_current_steps += 1
        rate = noam_lr(self.n_current_steps, self._d_model, self.n_warmup_steps)
        for group in self._optimizer.param_groups:
            group["lr"] = rate
        return rate

    def step_and_update_lr(self):
        self._update_learning_rate()
        self._optimizer.step()

    def zero_grad(self):
        self._optimizer.zero_grad()
