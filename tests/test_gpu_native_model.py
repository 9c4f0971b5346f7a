"""A batch created through the model loader of the C ABI (include/tsim_model.h: tsim_model_load + tsim_batch_create_from_model — what a C host
calls in place of redmax_py.Simulation(model_path), envs/redmax_torch_env.py:33) simulates what the batch of the Python-compiled model does:
the same bits where the two blobs are the same bits (the repository's own small models), and the fp64 oracle's trajectory."""
import os

import numpy as np
import pytest
import torch

from tactilesimulation_amd.model.compiler import load_model

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"slider_push": (np.zeros(4), lambda r: np.array([r.uniform(0.2, 1.0)])),
         "pad_press": (np.array([0.0, 0.0, -1e-3, 0.0, 0.0, 0.0]), lambda r: np.zeros(0)),
         "joint_laws": (np.array([0.02, 0.0, 0.3]), lambda r: np.array([r.uniform(-1, 1), r.uniform(-1, 1), r.uniform(-1.5, 1.5)])),
         "sphere_rest": (np.array([0.0, 0.0, -1.3e-4]), lambda r: np.array([r.uniform(-0.2, 0.9), r.uniform(-0.4, 0.4), r.uniform(-0.3, 0.5)]))}


def _native_batch(nm, py, B_, dtype, tape):
    """a BatchSim whose handle comes from tsim_batch_create_from_model"""
    from tactilesimulation_amd.host import capi
    from tactilesimulation_amd.host.batch import BatchSim
    sim = BatchSim(py, B_, dtype=dtype, tape_capacity=tape)
    capi.lib().tsim_batch_destroy(sim._h)
    sim._h = nm.create_batch_handle(B_, tape, capi.TSIM_F32 if dtype == torch.float32 else capi.TSIM_F64, torch.cuda.current_device())
    L = capi.lib()
    assert (L.tsim_ndof_r(sim._h), L.tsim_ndof_u(sim._h), L.tsim_ndof_var(sim._h), L.tsim_ndof_tactile(sim._h)) == (py.ndof_r, py.ndof_u, py.ndof_var, py.ndof_tactile)
    assert L.tsim_timestep(sim._h) == py.h
    return sim


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_natively_loaded_model_simulates_like_the_python_compiled_one(name, dtype):
    from tactilesimulation_amd.host.batch import BatchSim
    from tactilesimulation_amd.host.native_model import NativeModel
    path = os.path.join(HERE, "models", name + ".xml")
    nm, py = NativeModel(path), load_model(path)
    I, F = nm.blob()
    assert np.array_equal(I, py.I) and F.tobytes() == py.F.tobytes()      # (no meshes, axis-aligned frames: the two compilers agree to the bit)
    q0c, us = CASES[name]
    B_, T, S = 8, 8, 4
    rng = np.random.default_rng(3)
    q0 = torch.tensor(np.tile(q0c, (B_, 1)) + 1e-3 * rng.normal(size=(B_, q0c.size)), device="cuda", dtype=dtype)
    u = torch.tensor(np.stack([[us(rng) for _ in range(T)] for _ in range(B_)]).reshape(B_, T, py.ndof_u), device="cuda", dtype=dtype)
    a, b = BatchSim(py, B_, dtype=dtype, tape_capacity=T * S), _native_batch(nm, py, B_, dtype, T * S)
    wq = torch.tensor(rng.normal(size=(B_, py.ndof_r)), device="cuda", dtype=dtype)
    for sim in (a, b):
        sim.reset(q0, None, backward_flag=True)
    for t in range(T):
        oa, ob = a.step(u[:, t], S, want_qd=True), b.step(u[:, t], S, want_qd=True)
        for k in oa:
            assert torch.equal(oa[k], ob[k]), (name, t, k)
    ga, gb = a.backward_steps(T * S, df_dq=wq), b.backward_steps(T * S, df_dq=wq)      # (seed on the last sub-step, as StepSimFunction.backward)
    assert torch.equal(ga, gb) and torch.isfinite(ga).all()
    assert py.ndof_u == 0 or ga.abs().max() > 0


def test_natively_loaded_model_follows_the_oracle():
    """... and the oracle (fp64, one environment at a time) on the blob the native loader made"""
    import copy
    from oracle.oracle import OracleSim
    from tactilesimulation_amd.host.native_model import NativeModel
    path = os.path.join(HERE, "models", "slider_push.xml")
    nm, py = NativeModel(path), load_model(path)
    m = copy.copy(py)
    m.I, m.F = nm.blob()
    B_, T, S = 4, 10, 4
    rng = np.random.default_rng(5)
    q0 = 1e-3 * rng.normal(size=(B_, py.ndof_r))
    u = rng.uniform(0.2, 1.0, size=(B_, T, py.ndof_u))
    sim = _native_batch(nm, py, B_, torch.float64, T * S)
    sim.reset(torch.tensor(q0, device="cuda"), None, backward_flag=False)
    o = OracleSim(m)
    traj = [sim.step(torch.tensor(u[:, t], device="cuda"), S)["q"].cpu().numpy() for t in range(T)]
    for e in range(B_):
        o.reset(q0[e])
        for t in range(T):
            assert o.forward(u[e, t], S) == 0
            assert np.allclose(traj[t][e], o.state()[0], rtol=0, atol=1e-9), (e, t)


@pytest.mark.parametrize("tables", [False, True])
def test_plain_c_host_runs_the_same_trajectory(tmp_path, tables):
    """examples/c_host/step_from_xml.c — a C program with nothing but include/tsim.h, include/tsim_model.h and the HIP runtime (no Python,
    no torch in the process) — loads the XML, steps B environments and differentiates; its printed fp64 numbers are the Python host's, digit for digit.
    tables: with one contact stiffness per environment, the column found through tsim_model_table_offset (domain randomisation from C)."""
    import shutil
    import subprocess
    from tactilesimulation_amd.host import capi
    from tactilesimulation_amd.host.batch import BatchSim
    root = os.path.dirname(HERE)
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no C compiler / HIP headers on this machine")
    exe = str(tmp_path / "step_from_xml")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.run([cc, "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "c_host", "step_from_xml.c"), "-o", exe, "-L" + libdir, "-ltsim_hip", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, text=True)
    xml = os.path.join(HERE, "models", "slider_push.xml")
    B_, T, S, uval = 8, 6, 5, 0.6
    r = subprocess.run([exe, xml, str(B_), str(T), str(uval)] + (["pad:puck"] if tables else []), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    py = load_model(xml)
    assert lines[0].startswith("model ") and "ndof_r %d ndof_u %d ndof_var %d ndof_tactile %d" % (py.ndof_r, py.ndof_u, py.ndof_var, py.ndof_tactile) in lines[0] and "generic" in lines[0]
    sim = BatchSim(py, B_, dtype=torch.float64, tape_capacity=T * S)
    if tables:
        col = py.table_offset("pair", ("pad", "puck"), "kn")
        assert lines[1].startswith("per-environment tables: column %d " % col)
        lines = [lines[0]] + lines[2:]
        tab = sim.base_tables()
        tab[:, col] = torch.tensor([py.F[col] * (1.0 + 0.05 * e) for e in range(B_)], device="cuda", dtype=torch.float64)
        sim.set_env_tables(tab)
    sim.reset(torch.zeros(B_, py.ndof_r, device="cuda", dtype=torch.float64), None, backward_flag=True)
    u = torch.tensor([[uval * (1.0 + 0.1 * e)] * py.ndof_u for e in range(B_)], device="cuda", dtype=torch.float64)
    traj = []
    for t in range(T):
        q = sim.step(u, S)["q"].cpu().numpy()
        traj.append(q)
        a, b = lines[1 + t].split(" | ")
        assert a.split()[:3] == ["step", str(t), "q[0]"] and b.split()[0] == "q[%d]" % (B_ - 1)
        assert [float(x) for x in a.split()[3:]] == q[0].tolist() and [float(x) for x in b.split()[1:]] == q[-1].tolist(), t
    assert lines[1 + T] == "non-converged environments in the last step: 0"
    du = sim.backward_steps(T * S, df_dq=torch.ones(B_, py.ndof_r, device="cuda", dtype=torch.float64)).cpu().numpy()
    got = [float(x) for x in lines[2 + T].split()[1:]]
    assert got == du[0].reshape(-1).tolist() and max(abs(g) for g in got) > 0
    assert np.abs(traj[-1][0]).max() > 1e-4      # (the pad did push the puck)
