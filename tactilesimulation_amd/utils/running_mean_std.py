"""Running mean / variance of a data stream (the reference's utils/running_mean_std.py: observation normalisation of the
PPO / GD loops when `obs_rms` is on), kept on the device of the batch.  Same update rule (parallel-variance merge of batch
moments), same float32 state, same normalize() epsilon; golden vectors in tests/golden/policy_and_utils.npz."""
import torch


class RunningMeanStd:
    def __init__(self, epsilon=1e-4, shape=(), device="cuda:0"):
        self.mean = torch.zeros(shape, dtype=torch.float32, device=device)
        self.var = torch.ones(shape, dtype=torch.float32, device=device)
        self.count = epsilon

    def to(self, device):
        r = RunningMeanStd(device=device)
        r.mean, r.var, r.count = self.mean.to(device).clone(), self.var.to(device).clone(), self.count
        return r

    @torch.no_grad()
    def update(self, arr):
        """arr [n, *shape]: one batch of samples (for a batched environment: the observations of all environments)."""
        self.update_from_moments(arr.mean(dim=0), arr.var(dim=0, unbiased=False), arr.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        m2 = self.var * self.count + batch_var * batch_count + delta.square() * self.count * batch_count / tot
        self.mean = self.mean + delta * batch_count / tot
        self.var = m2 / tot
        self.count = tot

    def normalize(self, arr, un_norm=False):
        s = torch.sqrt(self.var + 1e-5)
        return arr * s + self.mean if un_norm else (arr - self.mean) / s
