"""Which modelling [CHOICE] makes the grasp closing of the rounds-1-3 TactileInsertion stand-in (workloads.insertion_workload) hard for the XML's
Newton loop?  CPU oracle, 192 environments x 14 env-steps x 5 sub-steps per variant: the density of the bodies that state none (guide / finger
meshes: [CHOICE] 1.0), the pads' contact lattice (cylinder caps: angle_res x radius_res, XML: 8 x 4 -> 66 points), and — for scale — the same
model on the reference's own episode (insertion_attempt_workload: settled grasp, 45 sub-steps).  VERDICT r03 "next" 2(b)."""
import copy, json, os, sys, time
import multiprocessing as mp
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tactilesimulation_amd.model import compiler as mc
from tactilesimulation_amd import workloads as W
from oracle.oracle import OracleSim

BASE = mc.load_model(W.asset("tactile_insertion"))
N = int(os.environ.get("SWEEP_ENVS", "192"))


def variant(density=None, pad_res=None):
    spec = copy.deepcopy(BASE.spec)
    for J in spec["joints"]:
        b = J["body"]
        if density is not None and b["name"] in ("gripper_left_guide", "finger_left", "gripper_right_guide", "finger_right"):
            b["density"] = float(density)
        if pad_res is not None and b["name"].startswith("tactile_pad_"):
            b["contact_res"] = list(pad_res)
    return mc.compile_spec(spec)


def run(args):
    (density, pad_res, workload), rng = args
    m = variant(density, pad_res)
    o = OracleSim(m)
    if workload == "grasp":
        q0, u = W.insertion_workload(N, 14, seed=7); S = 5
    else:
        q0, u = W.insertion_attempt_workload(N, seed=7); S = 1
    per, bad = [], 0
    for e in rng:
        o.reset(q0[e])
        for t in range(u.shape[1]):
            for s in range(S):
                s0 = o.stats(); bad += o.forward(u[e, t], 1) != 0; s1 = o.stats()
                per.append((s1["evals"] - s1["newton_iters"]) - (s0["evals"] - s0["newton_iters"]))
    return per, bad


if __name__ == "__main__":
    cases = [(None, None, "grasp"), (10.0, None, "grasp"), (100.0, None, "grasp"), (1000.0, None, "grasp"),
             (None, (8, 2), "grasp"), (None, (4, 2), "grasp"), (None, (16, 4), "grasp"), (None, (8, 8), "grasp"), (1000.0, (4, 2), "grasp"),
             (None, None, "attempt"), (1000.0, None, "attempt")]
    rows = []
    with mp.Pool(8) as pool:
        for c in cases:
            t0 = time.time()
            res = pool.map(run, [(c, r) for r in np.array_split(np.arange(N), 16)])
            per = np.concatenate([np.array(p) for p, _ in res]); bad = sum(b for _, b in res)
            rows.append({"mesh_density": c[0] or 1.0, "pad_lattice": list(c[1] or (8, 4)), "points_per_pad": 2 * (1 + (c[1] or (8, 4))[0] * (c[1] or (8, 4))[1]), "workload": c[2],
                         "substeps": int(per.size), "evals_mean": float(per.mean()), "evals_p99": float(np.percentile(per, 99)), "evals_max": int(per.max()),
                         "substeps_over_40_evals": int((per > 40).sum()), "substeps_at_max_iter": int(bad), "seconds": time.time() - t0})
            print(json.dumps(rows[-1]), flush=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "insertion_choice_sweep.json"), "w"), indent=1)
