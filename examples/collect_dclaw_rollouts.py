"""Forward-only roll-out collection on the batched D'Claw environment — BASELINE configs[3]'s usage (the reference trains TactileRotation-v1
with PPO over SubprocVecEnv, one simulator per process; here one batch per GPU): a policy acts on all environments at once, finished
environments are reset individually (with a new variant of the randomised model), and (obs, action, reward, done) batches come out.
The policy here is a random linear map — the point is the collector and its rate, not the learning.

    python examples/collect_dclaw_rollouts.py --batch 2048 --steps 200
    python -m torch.distributed.run --nproc-per-node 8 examples/collect_dclaw_rollouts.py       # 16 384 environments, no collective
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.envs.dclaw_rotate import BatchedDClawRotateEnv, GraphedCollector      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048, help="environments per GPU (16 384 / 8 in BASELINE configs[3])")
    ap.add_argument("--steps", type=int, default=200, help="env-steps to collect per environment (the env's episode limit)")
    ap.add_argument("--variants", type=int, default=0, help="pool of K host-compiled randomised models instead of continuous draws (rounds 3-4); 0: off")
    ap.add_argument("--no-randomize", action="store_true", help="no domain randomisation (default: every environment draws its own damping, cap radius, "
                    "end-effector offset and cap location on the device at each reset, envs/dclaw_rotate_env.py:164-184)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--eager", action="store_true", help="plain python loop instead of one HIP graph per collection step")
    ap.add_argument("--eval-budget", type=int, default=128, help="residual evaluations one sub-step may take before it is flagged and left "
                    "(include/tsim.h tsim_set_solver_options; 0 = none: with randomised variants a creeping sub-step then stalls the batch)")
    a = ap.parse_args()
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    env = BatchedDClawRotateEnv(a.batch, device="cuda:%d" % local, dtype=dt, seed=a.seed + rank, variants=a.variants, randomize=not a.no_randomize and not a.variants)
    env.sim.set_solver_options(cross_kinks=True, eval_budget=a.eval_budget)
    torch.manual_seed(a.seed)
    W = torch.randn(env.obs_dim, env.act_dim, device=env.device, dtype=dt) * 0.02
    policy = lambda obs: torch.tanh(obs @ W) + 0.3 * torch.randn(a.batch, env.act_dim, device=env.device, dtype=dt)
    zero = lambda: torch.zeros((), device=env.device, dtype=torch.long)
    episodes, successes, nonconv = zero(), zero(), zero()
    ret = torch.zeros(a.batch, device=env.device, dtype=dt)
    if a.eager:
        obs = env.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(a.steps):
            obs, r, done, info = env.step(policy(obs))
            ret += r
            nonconv += (info["status"] != 0).sum(); episodes += done.sum(); successes += (done & info["success"]).sum()
            obs = env.reset(done)                                          # only the finished environments start over; no host round trip
    else:
        col = GraphedCollector(env, policy)                                # one HIP-graph replay per collection step
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(a.steps):
            obs, u, r, done, nxt = col.step()                              # static tensors: a learner would copy them into its roll-out storage
            ret += r
            nonconv += (col.status != 0).sum(); episodes += done.sum(); successes += (done & col.success).sum()
    torch.cuda.synchronize(); dt_ = time.perf_counter() - t0
    print("rank %d: %d environments x %d env-steps in %.2f s = %.2f M env-steps/s; %d episodes ended (%d by success), mean return so far %.2f, "
          "non-converged env-steps %d" % (rank, a.batch, a.steps, dt_, a.batch * a.steps / dt_ / 1e6, int(episodes), int(successes), float(ret.mean()), int(nonconv)))


if __name__ == "__main__":
    main()
