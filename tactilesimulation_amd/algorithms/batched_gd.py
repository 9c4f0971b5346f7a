"""Batched analytic-gradient policy optimisation (BPTT through the simulator) — the MI355X counterpart of the reference's
algorithms/gd.py:145-264: there, `num_episodes` episodes run serially through one environment per epoch; here they
run as one batch of environments on each GPU, and ranks (one per GPU) exchange nothing but the flat policy gradient
(tactilesimulation_amd.dist.allreduce_policy_grad_, RCCL over xGMI with the nccl backend).
"""
import torch

from ..dist import allreduce_policy_grad_


class _LinearGemmBiasGrad(torch.autograd.Function):
    """torch.nn.functional.linear with the bias gradient computed as a GEMM (ones[1, B] @ g) instead of autograd's column
    reduction g.sum(0).  Same values; the reason is the HIP-graph path: for B >= 2048 rows torch's reduction splits the
    column sum over several blocks, which synchronise through a semaphore buffer zeroed by a hipMemsetAsync node, and on
    this ROCm stack memset nodes are not reliably ordered when a captured graph is REPLAYED (the same defect as
    profiles/r02_graphed_rollout_fix.md found in the simulator's own zero-fills).  Replays then returned bias gradients of
    the 64-wide layers that were 40-140 % off, silently: the loss and every other gradient were right
    (tools/graph_mlp_probe.py reproduces it without the simulator; profiles/r02_graph_bias_grad.md)."""

    @staticmethod
    def forward(ctx, x, weight, bias, ones):
        ctx.save_for_backward(x, weight, ones)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight, ones = ctx.saved_tensors
        return g @ weight, g.t() @ x, (ones @ g).reshape(-1), None


class Actor(torch.nn.Module):
    """Deterministic part of utils/model.py:123-151 DiagGaussianActor for gd_tactile.yaml: 393 -> 64 -> 64 -> 3, ELU,
    plus the (unused in deterministic mode) log-std vector: 29 574 parameters, the all-reduce payload of SURVEY.md §8e."""

    def __init__(self, obs_dim=393, act_dim=3, hidden=(64, 64), dtype=torch.float32):
        super().__init__()
        dims = (obs_dim,) + tuple(hidden)
        layers = []
        for a, b in zip(dims[:-1], dims[1:]):
            layers += [torch.nn.Linear(a, b), torch.nn.ELU()]
        layers.append(torch.nn.Linear(dims[-1], act_dim))
        self.mu_net = torch.nn.Sequential(*layers)
        self.logstd = torch.nn.Parameter(torch.full((act_dim,), -1.0))
        self.to(dtype)
        self._ones = None

    def forward(self, obs):
        if self._ones is None or self._ones.shape[1] != obs.shape[0] or self._ones.dtype != obs.dtype or self._ones.device != obs.device:
            self._ones = torch.ones(1, obs.shape[0], device=obs.device, dtype=obs.dtype)
        x = obs
        for m in self.mu_net:
            x = _LinearGemmBiasGrad.apply(x, m.weight, m.bias, self._ones) if isinstance(m, torch.nn.Linear) else m(x)
        return x


def rollout_loss(env, actor, horizon, q0=None, goal=None, disturbances=None):
    """-sum of rewards of all environments over one episode (un-normalised; see train_epoch)."""
    obs = env.reset(q0, goal)
    total = obs.new_zeros(())
    for t in range(horizon):
        u = actor(obs)
        obs, rew, _ = env.step(u, None if disturbances is None else disturbances[t])
        total = total - rew.sum()
    return total


class GraphedRollout:
    """rollout_loss + backward of one episode captured in ONE HIP graph (torch.cuda.CUDAGraph): the closed loop issues ~40
    small launches per env-step (policy MLP, observation / reward formulas, autograd, one simulator launch each way), which
    an eager-mode host cannot feed fast enough; a replay costs one launch.  q0 [B, 7], goal [B, 3], disturbances
    [T, B, 2] are STATIC device tensors: write new episode data into them (copy_) before replay().  After replay(),
    `loss` holds -sum of rewards and the actor's .grad the un-normalised policy gradient (then allreduce / clip / step as
    in train_epoch).  The simulator's host-side tape counters are advanced at capture time and end where they started
    (forward pushes, backward pops), so every replay is a whole episode."""

    def __init__(self, env, actor, horizon, q0, goal, disturbances, warmup=2):
        self.env, self.actor, self.horizon = env, actor, horizon
        self.q0, self.goal, self.dist = q0, goal, disturbances
        side = torch.cuda.Stream(env.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                         # eager warm-up on a side stream (allocator, lazy init)
            for _ in range(warmup):
                for p in actor.parameters():
                    p.grad = None
                rollout_loss(env, actor, horizon, q0=q0, goal=goal, disturbances=disturbances).backward()
        torch.cuda.current_stream().wait_stream(side)
        for p in actor.parameters():
            p.grad = None
        # Capture on the SAME side stream the warm-up ran on: the parameters' AccumulateGrad nodes were created there, and a
        # capture on another stream makes autograd fork the gradient accumulation onto the warm-up stream inside the graph —
        # replays with new episode data then read gradient buffers before that branch has written them (garbage policy
        # gradients of 1e16 ... 1e24 at B = 4096, profiles/r02_graphed_rollout_fix.md; replays of unchanged data hid it).
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.loss = rollout_loss(env, actor, horizon, q0=q0, goal=goal, disturbances=disturbances)
            self.loss.backward()

    def replay(self):
        self.graph.replay()
        return self.loss


def train_epoch_graphed(gr, optimizer, global_episodes, grad_clip=1.0):
    """train_epoch with the roll-out and its backward replayed from a HIP graph (GraphedRollout)."""
    loss = gr.replay()                                        # static tensor: clone it to keep a value across epochs
    params = list(gr.actor.parameters())
    allreduce_policy_grad_(params, global_episodes)
    if grad_clip:
        torch.nn.utils.clip_grad_norm_(params, grad_clip)
    optimizer.step()
    return loss


def train_epoch(env, actor, optimizer, horizon, global_episodes, grad_clip=1.0, **rollout_kw):
    """One optimiser step on `global_episodes` episodes (= sum over ranks of env.B): local BPTT, ONE all-reduce of the
    flat gradient, normalisation by the global episode count, then clip-by-global-norm and Adam (gd.py:157-164,258)."""
    optimizer.zero_grad(set_to_none=True)
    loss = rollout_loss(env, actor, horizon, **rollout_kw)
    loss.backward()
    allreduce_policy_grad_(list(actor.parameters()), global_episodes)
    if grad_clip:
        torch.nn.utils.clip_grad_norm_(actor.parameters(), grad_clip)
    optimizer.step()
    return float(loss.detach()) / env.B
