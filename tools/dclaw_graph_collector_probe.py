"""Probe: the D'Claw collector's per-step body (policy, env.step, per-environment reset) captured in ONE HIP graph and replayed, against the eager
loop: env-steps/s and equality of the collected observations / rewards (same seeds)."""
import os, sys, time, json
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT)
from tactilesimulation_amd.envs.dclaw_rotate import BatchedDClawRotateEnv
B, STEPS = 2048, 100
dt = torch.float32
def make():
    env = BatchedDClawRotateEnv(B, dtype=dt, seed=0, variants=0)
    env._gen = None                                              # default generator: capturable
    torch.manual_seed(0)
    W = torch.randn(env.obs_dim, env.act_dim, device=env.device, dtype=dt) * 0.02
    return env, W
def body(env, W, obs):
    u = torch.tanh(obs @ W) + 0.3 * torch.randn(B, env.act_dim, device=env.device, dtype=dt)
    o, r, done, info = env.step(u)
    o = env.reset(done)
    return o, r, done
res = {}
for mode in ("eager", "graph"):
    env, W = make()
    torch.manual_seed(1)
    obs = env.reset().clone()
    rsum = torch.zeros(B, device=env.device, dtype=dt)
    if mode == "graph":
        static_obs = obs.clone()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2): body(env, W, static_obs)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            o, r, d = body(env, W, static_obs)
            static_obs.copy_(o)
        torch.cuda.synchronize()
        torch.manual_seed(1); static_obs.copy_(env.reset())           # a fresh batch of episodes
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(STEPS):
        if mode == "eager":
            obs, r, d = body(env, W, obs)
        else:
            g.replay()
        rsum += r
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    res[mode] = {"env_steps_per_s": B * STEPS / el, "ms_per_step": el / STEPS * 1e3, "return_mean": float(rsum.mean()), "finite": bool(torch.isfinite(rsum).all())}
    print(mode, json.dumps(res[mode]), flush=True)
