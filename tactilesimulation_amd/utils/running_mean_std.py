"""Streaming mean / variance of observations, kept on the device of the batch — what the reference's PPO / GD loops use to normalise
observations when `obs_rms` is on (its utils/running_mean_std.py; same interface: update / normalize / mean / var / count / to).

State is (count, mean, M2) with M2 the running sum of squared deviations, merged batch-wise by Chan's parallel rule; `var` is derived from
it.  float32 state and the 1e-5 inside the square root of normalize() follow the reference, so results agree with it to float32 rounding
(golden vectors: tests/golden/policy_and_utils.npz, tests/test_policy_and_utils.py).  A batched environment feeds it all B observations of
a step at once.
"""
import torch


class RunningMeanStd:
    def __init__(self, epsilon=1e-4, shape=(), device="cuda:0"):
        self._count = float(epsilon)                                  # pseudo-count of the prior (mean 0, variance 1)
        self.mean = torch.zeros(shape, dtype=torch.float32, device=device)
        self._m2 = torch.full(tuple(shape), self._count, dtype=torch.float32, device=device)     # variance 1 x count

    @property
    def count(self):
        return self._count

    @count.setter
    def count(self, c):
        """Assigning count keeps `var` (as with the reference's three plain attributes: code that restores mean / var / count from a
        checkpoint, or the reference's own to(), assigns them in any order)."""
        c = float(c)
        self._m2 = self._m2 * (c / self._count)
        self._count = c

    @property
    def var(self):
        return self._m2 / self.count

    @var.setter
    def var(self, v):
        self._m2 = torch.as_tensor(v, dtype=torch.float32, device=self.mean.device) * self.count

    def to(self, device):
        other = RunningMeanStd(self.count, tuple(self.mean.shape), device)
        other.mean, other._m2 = self.mean.to(device).clone(), self._m2.to(device).clone()
        return other

    @torch.no_grad()
    def update(self, batch):
        """batch [n, *shape]: n new samples."""
        n = batch.shape[0]
        mu = batch.mean(dim=0)
        self.update_from_moments(mu, ((batch - mu) ** 2).mean(dim=0), n)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        total = self.count + batch_count
        shift = batch_mean - self.mean
        self._m2 = self._m2 + batch_var * batch_count + shift * shift * (self.count * batch_count / total)
        self.mean = self.mean + shift * (batch_count / total)
        self._count = float(total)

    def normalize(self, x, un_norm=False):
        scale = (self.var + 1e-5).sqrt()
        return x * scale + self.mean if un_norm else (x - self.mean) / scale
