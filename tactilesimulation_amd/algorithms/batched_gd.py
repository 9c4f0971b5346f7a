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


class _LinearDeferredWeightGrad(torch.autograd.Function):
    """linear() whose backward only propagates to its input (g @ W) and parks (x, g) in `sink`; the weight and bias gradients
    of ALL env-steps of the episode are then formed at once by Actor.assemble_grads().  Per env-step the weight gradient is a
    [out x B] @ [B x in] GEMM — 64 x 393 outputs with K = B = 4096: 14 workgroups looping over K, 40-63 us each for 0.2 GFLOP, three
    of them per step, 27 % of the closed loop's GPU time (profiles/r02_closed_loop_kernels.md).  Over the episode the same sum
    is one batched GEMM with T x 14 workgroups."""

    @staticmethod
    def forward(ctx, x, weight, bias, sink):
        ctx.save_for_backward(x, weight)
        ctx.sink = sink
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        ctx.sink.append((x, g))
        return (g @ weight) if ctx.needs_input_grad[0] else None, None, None, None


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
        self.defer_weight_grads = False         # GraphedRollout / train_epoch(defer=True) switch this on
        self._sinks = [[] for m in self.mu_net if isinstance(m, torch.nn.Linear)]
        self._armed = False                     # deferred mode: begin_episode() was called and its gradients were not assembled yet

    def forward(self, obs):
        if self._ones is None or self._ones.shape[1] != obs.shape[0] or self._ones.dtype != obs.dtype or self._ones.device != obs.device:
            self._ones = torch.ones(1, obs.shape[0], device=obs.device, dtype=obs.dtype)
        if self.defer_weight_grads and torch.is_grad_enabled() and not self._armed:
            raise RuntimeError("Actor is in deferred-weight-gradient mode (GraphedRollout / defer_weight_grads = True): call begin_episode() "
                               "before the roll-out and assemble_grads() after backward(), or set defer_weight_grads = False — a plain "
                               "backward() would leave weight.grad / bias.grad empty")
        x, k = obs, 0
        for m in self.mu_net:
            if not isinstance(m, torch.nn.Linear):
                x = m(x)
            elif self.defer_weight_grads:
                x = _LinearDeferredWeightGrad.apply(x, m.weight, m.bias, self._sinks[k]); k += 1
            else:
                x = _LinearGemmBiasGrad.apply(x, m.weight, m.bias, self._ones)
        return x

    # the reference's checkpoint layout (utils/model.py:123-151 DiagGaussianActor: feature_net.body.{0,2}, mean_net, logstd)
    _REF_KEYS = (("feature_net.body.0", "mu_net.0"), ("feature_net.body.2", "mu_net.2"), ("mean_net", "mu_net.4"))

    def load_reference_state_dict(self, sd):
        """Load a DiagGaussianActor state_dict saved by the reference (algorithms/gd.py:189-194) — 393 -> 64 -> 64 -> 3 only."""
        own = {"logstd": torch.as_tensor(sd["logstd"])}
        for ref, mine in self._REF_KEYS:
            for part in ("weight", "bias"):
                own["%s.%s" % (mine, part)] = torch.as_tensor(sd["%s.%s" % (ref, part)])
        self.load_state_dict(own)

    def reference_state_dict(self):
        """The parameters under the reference's names (for its `load` / evaluation scripts)."""
        sd, out = self.state_dict(), {"logstd": self.logstd.detach().clone()}
        for ref, mine in self._REF_KEYS:
            for part in ("weight", "bias"):
                out["%s.%s" % (ref, part)] = sd["%s.%s" % (mine, part)].detach().clone()
        return out

    def begin_episode(self):
        """Deferred mode: forget the (input, output-gradient) pairs of earlier backward passes."""
        for s in self._sinks:
            s.clear()
        self._armed = True

    def assemble_grads(self, keep=False, pairs=None):
        """Deferred mode, after the episode's backward: weight.grad = sum_t g_t^T x_t as ONE batched GEMM per layer and
        bias.grad = sum_t sum_b g_t (assigned, not accumulated).  Runs eagerly, outside any graph.
        pairs: per-layer lists of (x, g) to assemble from INSTEAD of the actor's own sinks — a GraphedRollout passes the static
        buffers of its captured graph (rewritten by every replay), which it keeps apart from the sinks so that eager backward passes
        through the same actor can neither add to them nor clear them.  keep=True leaves the actor's sinks as they are."""
        linears = [m for m in self.mu_net if isinstance(m, torch.nn.Linear)]
        own = pairs is None
        for m, sink in zip(linears, self._sinks if own else pairs):
            if not sink:
                m.weight.grad = m.bias.grad = None
                continue
            X, G = torch.stack([x for x, _ in sink]), torch.stack([g for _, g in sink])        # [T, B, in], [T, B, out]
            m.weight.grad = torch.bmm(G.transpose(1, 2), X).sum(0)
            m.bias.grad = G.sum((0, 1))
            if own and not keep:
                sink.clear()
        if own and not keep:
            self._armed = False


class CNNActor(torch.nn.Module):
    """The reference's CNN policy for the `tactile_map` observation — TactilePushEnv's DEFAULT observation_type (envs/tactile_push_env.py:21,37-40)
    — utils/model.py:37-98 CNN + CNNActor: Conv2d(k, stride) + activation per layer, Flatten, Linear -> hidden, activation, Linear -> action_dim, and
    the log-std vector.  Same constructor arguments and the same state_dict names (feature_net.body.N, mean_net, logstd), so a checkpoint of the
    reference loads with load_state_dict (tests/test_policy_and_utils.py checks parameters -> outputs against vectors recorded from the reference's
    class).  forward() returns the MEAN action (what DiagGaussian.mode() / act(deterministic=True) gives; the batched GD loop is deterministic).
    Takes the tactile image [B, C, rows, cols] or BatchedTactilePushEnv's tactile_map observation tuple (image, goal-in-gripper-frame); the reference's
    CNNActor looks at the image only, and so does this one unless state_dim > 0 (then the state part is concatenated to the CNN features in front
    of mean_net — an extension, off by default).  Runs as plain torch modules (MIOpen convolutions): the closed loop with this policy is the
    per-env-step graph (GraphedRollout), not the fused episode launch, whose in-kernel policy is the 393-64-64-3 MLP."""

    _ACT = {"tanh": torch.nn.Tanh, "relu": torch.nn.ReLU, "elu": torch.nn.ELU, "identity": torch.nn.Identity}

    def __init__(self, obs_shape, action_dim, cfg_network, state_dim=0, dtype=torch.float32):
        super().__init__()
        assert len(obs_shape) == 3                      # (feature_channels, rows, cols)
        cfg = cfg_network["actor_cnn"]
        act = self._ACT[cfg["activation"].lower()]
        in_ch, rows, cols = obs_shape
        mods = []
        for k, n, st in zip(cfg["kernel_sizes"], cfg["layer_sizes"], cfg["stride_sizes"]):
            mods += [torch.nn.Conv2d(in_ch, n, k, stride=st), act()]
            if cfg.get("layernorm", False):
                mods.append(torch.nn.LayerNorm(n))
            in_ch, rows, cols = n, (rows - k) // st + 1, (cols - k) // st + 1
        mods += [torch.nn.Flatten(), torch.nn.Linear(rows * cols * in_ch, cfg["hidden_size"]), act()]
        self.feature_net = torch.nn.Module()
        self.feature_net.body = torch.nn.Sequential(*mods)
        self.feature_net.out_features = cfg["hidden_size"]
        self.state_dim = int(state_dim)
        self.mean_net = torch.nn.Linear(cfg["hidden_size"] + self.state_dim, action_dim)
        self.logstd = torch.nn.Parameter(torch.ones(action_dim) * cfg_network.get("actor_logstd_init", -1.0))
        self.to(dtype)

    # as_gemm = True runs each convolution as im2col (F.unfold) + ONE matmul — the same numbers up to the summation order (tests/test_policy_and_utils.py:
    # outputs and gradients 1e-12 in fp64 against torch's Conv2d), the same parameters.  OFF by default: it is correct eagerly and in small captured graphs
    # (tests/test_gpu_batched_env.py), but capturing the B = 4096 x 100-step closed loop with it ends in a segmentation fault inside
    # torch.cuda.CUDAGraph.capture_end on this stack (ROCm 7.0.2 runtime, torch 2.10; round 6) — the per-step graph is the only fast path for this policy.
    as_gemm = False

    def _body(self, x):
        if not self.as_gemm:
            return self.feature_net.body(x)
        for m in self.feature_net.body:
            if isinstance(m, torch.nn.Conv2d):
                B, _, H, Wd = x.shape
                k, st = m.kernel_size, m.stride
                cols = torch.nn.functional.unfold(x, k, stride=st)                                   # [B, Cin k k, positions]
                x = torch.matmul(m.weight.reshape(m.out_channels, -1), cols) + m.bias.reshape(1, -1, 1)
                x = x.reshape(B, m.out_channels, (H - k[0]) // st[0] + 1, (Wd - k[1]) // st[1] + 1)
            else:
                x = m(x)
        return x

    def forward(self, obs):
        img, state = (obs if isinstance(obs, (tuple, list)) else (obs, None))
        f = self._body(img.contiguous())
        if self.state_dim:
            f = torch.cat([f, state], dim=1)
        return self.mean_net(f)


def rollout_loss(env, actor, horizon, q0=None, goal=None, disturbances=None):
    """-sum of rewards of all environments over one episode (un-normalised; see train_epoch)."""
    obs = env.reset(q0, goal)
    acc = None                                   # per-environment return: one add per env-step, no per-step reduction
    for t in range(horizon):
        u = actor(obs)
        obs, rew, _ = env.step(u, None if disturbances is None else disturbances[t])
        acc = rew if acc is None else acc + rew
    return -acc.sum()


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
        sim = getattr(env, "sim", None)
        if sim is not None:          # host-side batch state is baked into captured launches: fine for BDF1, not for BDF2's history flag
            from ..model import blob as _blob
            if int(sim.model.I[_blob.TSIM_IH_INTEGRATOR]) != 1:
                raise RuntimeError("GraphedRollout: BDF2 models cannot be captured (the integrator's history flag is host state baked into "
                                   "the captured launch)")
        self.deferred = hasattr(actor, "assemble_grads")      # the weight gradients of all env-steps as one GEMM after the replay
        if self.deferred:
            was_deferred = actor.defer_weight_grads
            actor.defer_weight_grads = True
        side = torch.cuda.Stream(env.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                         # eager warm-up on a side stream (allocator, lazy init)
            for _ in range(warmup):
                for p in actor.parameters():
                    p.grad = None
                if self.deferred:
                    actor.begin_episode()
                rollout_loss(env, actor, horizon, q0=q0, goal=goal, disturbances=disturbances).backward()
                if self.deferred:
                    actor.assemble_grads()
        torch.cuda.current_stream().wait_stream(side)
        for p in actor.parameters():
            p.grad = None
        if self.deferred:
            actor.begin_episode()
        # Capture on the SAME side stream the warm-up ran on: the parameters' AccumulateGrad nodes were created there, and a
        # capture on another stream makes autograd fork the gradient accumulation onto the warm-up stream inside the graph —
        # replays with new episode data then read gradient buffers before that branch has written them (garbage policy
        # gradients of 1e16 ... 1e24 at B = 4096, profiles/r02_graphed_rollout_fix.md; replays of unchanged data hid it).
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.loss = rollout_loss(env, actor, horizon, q0=q0, goal=goal, disturbances=disturbances)
            self.loss.backward()
        if self.deferred:
            # The (x, g) pairs the captured backward parked are the graph's static buffers: take them out of the actor's sinks and hand
            # the actor back in the mode it came in.  Left in the sinks they were shared with every later eager pass through the actor:
            # an eager backward appended its own pairs (the next replay then summed them in: exactly twice the gradient in a CPU
            # repro) and an eager begin_episode() cleared the graph's (the next replay set every .grad to None and optimizer.step()
            # silently did nothing) — ADVICE r02.
            self._pairs = [list(s) for s in actor._sinks]
            for s in actor._sinks:
                s.clear()
            actor._armed = False
            actor.defer_weight_grads = was_deferred

    def replay(self):
        self.graph.replay()
        if self.deferred:
            self.actor.assemble_grads(pairs=self._pairs)
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
    deferred = getattr(actor, "defer_weight_grads", False)
    if deferred:
        actor.begin_episode()
    loss = rollout_loss(env, actor, horizon, **rollout_kw)
    loss.backward()
    if deferred:
        actor.assemble_grads()
    allreduce_policy_grad_(list(actor.parameters()), global_episodes)
    if grad_clip:
        torch.nn.utils.clip_grad_norm_(actor.parameters(), grad_clip)
    optimizer.step()
    return float(loss.detach()) / env.B
