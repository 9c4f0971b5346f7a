"""Simulator gradients in the regime a TRAINED policy drives the environments into (pad pressed on the box, sliding), not the random
open-loop actions of the bench workload: train the fused closed loop for N epochs, take the 6 actuator inputs per env-step the policy
produced in one more episode, replay them OPEN LOOP through tsim_rollout / tsim_backward_episode (fp32 and fp64 kernels) with the
reward's own partials as seeds, and compare trajectories and dL/du with the fp64 oracle (literal solver) on a subset: the environments
with the largest losses plus a random sample.   python tools/trained_regime_grad_check.py [epochs]"""
import os, sys, json, math, threading
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
from train_tactile_push_gd_batched import draw_episode
from tactilesimulation_amd.envs.tactile_push import BatchedTactilePushEnv
from tactilesimulation_amd.envs.push_closed_loop import FusedPushEpisode, train_epoch_fused
from tactilesimulation_amd.algorithms.batched_gd import Actor
from tactilesimulation_amd.host.batch import BatchSim
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.workloads import asset
from oracle.oracle import OracleSim


def run(E=80, verbose=True):
    B, T, S, dev, dt = 4096, 100, 5, "cuda:0", torch.float32
    model = load_model(asset("pusher"))
    env = BatchedTactilePushEnv(model, B, device=dev, dtype=dt, gradient=True, seed=0, tape_steps=T)
    torch.manual_seed(0)
    actor = Actor(dtype=dt).to(dev)
    opt = torch.optim.Adam(actor.parameters(), lr=5e-3, betas=(0.7, 0.95))
    rng = np.random.default_rng(0)
    ep = FusedPushEpisode(env, actor, T)
    for epoch in range(E):
        for g in opt.param_groups: g["lr"] = (1e-5 - 5e-3) * float(epoch / 300) + 5e-3
        q0, goal, dist = draw_episode(rng, B, T, dev, dt, 1)
        loss = train_epoch_fused(ep, opt, q0, goal, dist, B)
    print("trained %d epochs, loss per episode %.1f" % (E, float(loss) / B))
    q0, goal, dist = draw_episode(rng, B, T, dev, dt, 1)
    ep.rollout(q0, goal, dist)
    u6 = torch.zeros(T, B, 6, device=dev, dtype=torch.float64)
    u6[:, :, 0:3] = torch.tanh(ep.u.double()); u6[:, :, 3:5] = dist.double()
    wq, wv = ep.df_dq.double().clone(), ep.df_dvar.double().clone()
    # per-environment loss, to pick the hard ones
    g_ = ep.goal.unsqueeze(0); dp = ep.q[:, :, 3:5] - g_[:, :, 0:2]; dr = ep.q[:, :, 6] - g_[:, :, 2]; dtt = ep.var[:, :, 0:3] - ep.var[:, :, 3:6]
    le = ((dp ** 2).sum(2) * 100.0 + dr ** 2 * (0.1 * (36.0 / math.pi) ** 2) + (dtt ** 2).sum(2) * 2500.0).sum(0)
    hard = torch.argsort(le, descending=True)[:16].cpu().numpy()
    idx = np.unique(np.concatenate([hard, np.random.default_rng(5).choice(B, 32, replace=False)]))
    q_fused = ep.q.double().cpu().numpy()
    res = {}
    for name, kd in (("f32", torch.float32), ("f64", torch.float64)):
        sim = BatchSim(model, B, dtype=kd, tape_capacity=T * S)
        sim.reset(q0.to(kd), None, backward_flag=True)
        ro = sim.rollout(u6.to(kd), S, want_qd=True)
        sig = sim.branch_signature().cpu().numpy()
        du = sim.backward_episode(T, S, wq.to(kd), wv.to(kd), None)
        res[name] = dict(q=ro["q"].double().cpu().numpy(), tac=ro["tactile"].double().cpu().numpy(), du=du.double().cpu().numpy(), sig=sig, bad=int((ro["status"] != 0).sum()))
        print(name, "kernels: flagged environments", res[name]["bad"], " max |q - q_fused| %.2e" % np.abs(res[name]["q"] - q_fused).max())
        del sim
    # oracle on the subset
    n = len(idx); q0n, u6n, wqn, wvn = q0.double().cpu().numpy(), u6.cpu().numpy(), wq.cpu().numpy(), wv.cpu().numpy()
    O = {"q": np.zeros((T, n, 7)), "tac": np.zeros((T, n, 390)), "du": np.zeros((T, n, 6)), "sig": np.zeros((T * S, n, 2), dtype=np.int64), "bad": np.zeros(n, dtype=int)}
    def work(i, nthr):
        o = OracleSim(model, solver="literal")
        for j in range(i, n, nthr):
            e = idx[j]; o.reset(q0n[e], record=True)
            for t in range(T):
                bad, sg = o.forward_sig(u6n[t, e], S); O["bad"][j] += bad; O["sig"][t * S:(t + 1) * S, j] = sg
                O["q"][t, j] = o.state()[0]; O["tac"][t, j] = o.outputs()[1]
            for t in reversed(range(T)):
                dq = np.zeros((S, 7)); dq[-1] = wqn[t, e]; dv = np.zeros((S, 6)); dv[-1] = wvn[t, e]
                O["du"][t, j] = o.backward_steps(S, dq, dv, np.zeros((S, 390))).sum(0)
    nthr = max(1, min(len(os.sched_getaffinity(0)), 32, n))
    th = [threading.Thread(target=work, args=(i, nthr)) for i in range(nthr)]; [t.start() for t in th]; [t.join() for t in th]
    print("oracle: non-converged sub-steps on the subset:", int(O["bad"].sum()))
    out = {"epochs": E, "subset": len(idx), "hard": len(hard)}
    for name in ("f32", "f64"):
        r = res[name]
        dq = np.abs(r["q"][:, idx] - O["q"]).max(axis=(0, 2))
        same = (r["sig"][:, idx] == O["sig"]).all(axis=(0, 2))
        eg = np.abs(r["du"][:, idx] - O["du"]).max(axis=(0, 2)) / np.abs(O["du"]).max(axis=(0, 2))
        ishard = np.isin(idx, hard)
        out[name] = dict(q_err_max=float(dq.max()), q_err_median=float(np.median(dq)), branch_agree=int(same.sum()), grad_err_median=float(np.median(eg)),
                         grad_err_max_agreeing=float(eg[same].max()) if same.any() else None, grad_err_max_all=float(eg.max()),
                         grad_err_within_1e4=int((eg < 1e-4).sum()), grad_err_second_largest=float(np.sort(eg)[-2]),
                         grad_err_max_hard_agreeing=float(eg[same & ishard].max()) if (same & ishard).any() else None, hard_agree=int((same & ishard).sum()),
                         dLdu_scale_median=float(np.median(np.abs(O["du"]).max(axis=(0, 2)))), dLdu_scale_max=float(np.abs(O["du"]).max()))
        print(name, json.dumps(out[name]))
    return out


if __name__ == "__main__":
    out = run(int(sys.argv[1]) if len(sys.argv) > 1 else 80)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "trained_regime_grad_check.json"), "w"), indent=1)
