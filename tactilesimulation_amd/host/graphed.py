"""GraphedEpisode — one open-loop episode of a batch (reset, the episode launch forward, the episode launch backward, an optional reduction of
the result) captured in ONE HIP graph and replayed.

An episode of the batched surface is three library calls (tsim_reset, tsim_rollout, tsim_backward_episode: ~12 kernel dispatches with the
tactile pass, the LPT ordering and the state / adjoint resets) plus whatever torch does with the gradient: ~0.3 ms of host work when issued eagerly
from python.  At BASELINE's headline batch that hides behind 3.9 ms of kernels (bench.py --graph: 20.95 M env-steps/s against 20.97 M eager,
profiles/r05_graphed_episode.md); it is what a small batch or a short episode lasts.  A replay is one launch.  Reference counterpart: the body of `for episode in ...` in
algorithms/gd.py:224-259 for an open-loop action table (the closed loop has its own capture, algorithms/batched_gd.GraphedRollout).

The inputs are STATIC device tensors — q0 [B, nr], u [T, B, nu], the loss seeds [T, B, .] — : write new episode data into them (copy_) before
replay().  Host-side batch state (tape length, the LPT order's length) is baked into the captured launches and ends where it started (forward
pushes the tape, backward pops it), so every replay is a whole episode; BDF2 models cannot be captured (their history flag is host state)."""
import torch

from ..model import blob as _blob


class GraphedEpisode:
    def __init__(self, sim, q0, u, num_steps, seeds=None, tactile_mask=None, post=None, warmup=1):
        """seeds: (df_dq, df_dvar, df_dtactile) [T, B, .] for forward + adjoint, None for forward-only; post(rollout_dict, df_du) -> anything:
        captured behind the launches (e.g. the reduction of df_du into a gradient buffer)."""
        if int(sim.model.I[_blob.TSIM_IH_INTEGRATOR]) != 1:
            raise RuntimeError("GraphedEpisode: BDF2 models cannot be captured (the integrator's history flag is host state baked into the captured launch)")
        self.sim, self.q0, self.u, self.S, self.seeds, self.mask, self.post = sim, q0, u, int(num_steps), seeds, tactile_mask, post
        dev = sim.device
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                          # eager warm-up: allocations (per-frame pose records), the LPT order of this episode length
            for _ in range(max(1, warmup)):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.rollout, self.df_du, self.extra = self._body()

    def _body(self):
        sim = self.sim
        sim.reset(self.q0, None, backward_flag=self.seeds is not None)
        ro = sim.rollout(self.u, self.S, tactile_mask=self.mask) if self.mask is not None else sim.rollout(self.u, self.S)
        du = None
        if self.seeds is not None:
            du = sim.backward_episode(int(self.u.shape[0]), self.S, *self.seeds, **({"tactile_mask": self.mask} if self.mask is not None else {}))
        return ro, du, (self.post(ro, du) if self.post is not None else None)

    def replay(self):
        self.graph.replay()
        return self.rollout, self.df_du, self.extra
