"""The model loader behind the C ABI (include/tsim_model.h, csrc/tsim_model.cpp) against the Python model compiler (model/compiler.py).

What a C / C++ host calls in place of `redmax_py.Simulation(model_path)` (envs/redmax_torch_env.py:33) and the update_* family.
The two compilers must produce the same ints and the same reals to round-off of the host's double arithmetic (the Python one goes through
numpy / BLAS products, the native one through plain double arithmetic: last-bit differences, bounded here at 1e-12 relative / 1e-15 absolute).
No GPU: the loader is host code.  The models of the reference are read where /root/reference exists (this container); the synthetic
models written here cover every body / joint / contact / sensor kind of the format everywhere."""
import os

import numpy as np
import pytest

REF = "/root/reference"
REF_XMLS = ["envs/assets/pusher/pusher.xml", "envs/assets/tactile_insertion/tactile_insertion.xml", "envs/assets/stable_grasp/stable_grasp.xml",
            "envs/assets/dclaw_rotate/dclaw_position_control.xml", "envs/assets/dclaw_rotate/dclaw_torque_control.xml", "assets/tactile_pad/tactile_pad.xml"]


def _native(path):
    from tactilesimulation_amd.host.native_model import NativeModel
    return NativeModel(path)


def _python(path):
    from tactilesimulation_amd.model.compiler import parse_xml, compile_spec
    return compile_spec(parse_xml(path))


def _same_blob(I, F, py):
    assert I.dtype == np.int32 and F.dtype == np.float64
    assert np.array_equal(I, py.I), np.nonzero(I != py.I)[0][:10]
    assert F.shape == py.F.shape
    bad = np.nonzero(~np.isclose(F, py.F, rtol=1e-12, atol=1e-15))[0]
    assert len(bad) == 0, (bad[:10], F[bad[:10]], py.F[bad[:10]])


def _same_lookups(nm, py):
    from tactilesimulation_amd.host.native_model import PAIR_FIELDS
    for s in py.meta["sensor_names"]:
        assert nm.image_pos(s) == [tuple(t) for t in py.meta["image_pos"][s]]
        for f in ("kn", "kt", "mu", "damping"):
            assert nm.table_offset("sensor", s, field=f) == py.table_offset("sensor", s, f)
    for k0, k1 in py.meta["pair_keys"]:
        for f in PAIR_FIELDS:
            assert nm.table_offset("pair", k0, k1, f) == py.table_offset("pair", (k0, k1), f)
    for j, (d0, nd) in py.meta["dof_of_joint"].items():
        for k in range(nd):
            assert nm.table_offset("dof", j, field=k) == py.table_offset("dof", (j, k), "damping")


@pytest.mark.parametrize("rel", REF_XMLS)
def test_reference_models_compile_to_the_same_blob(rel):
    path = os.path.join(REF, rel)
    if not os.path.exists(path):
        pytest.skip("reference assets not on this machine")
    nm, py = _native(path), _python(path)
    I, F = nm.blob()
    _same_blob(I, F, py)
    _same_lookups(nm, py)


def test_reference_models_match_the_committed_assets():
    """... and therefore the blobs every other test of this repository simulates (tactilesimulation_amd/assets/*.npz, compiled by the Python compiler)"""
    from tactilesimulation_amd import workloads as W
    from tactilesimulation_amd.model.compiler import load_model
    n = 0
    for rel in REF_XMLS:
        name = os.path.splitext(os.path.basename(rel))[0]
        path = os.path.join(REF, rel)
        if not os.path.exists(path) or not os.path.exists(W.asset(name)):
            continue
        I, F = _native(path).blob()
        _same_blob(I, F, load_model(W.asset(name)))
        n += 1
    if n == 0:
        pytest.skip("reference assets not on this machine")
    assert n >= 5


def test_native_pusher_is_the_compiled_in_asset_at_float_resolution():
    """pusher.xml loaded natively differs from assets/pusher.npz (what the fully static kernels were generated from) in the last bits of a few
    mesh-derived doubles; as floats — all an fp32 kernel sees of a model — the float records are the same bits, which is the host's criterion for
    keeping an fp32 batch on the static:pusher instantiation (csrc/tsim_hip.hip blob_equals_static; tests/test_gpu_static_model.py)"""
    from tactilesimulation_amd import workloads as W
    from tactilesimulation_amd.model import blob as B
    from tactilesimulation_amd.model.compiler import load_model
    path = os.path.join(REF, "envs/assets/pusher/pusher.xml")
    if not os.path.exists(path):
        pytest.skip("reference assets not on this machine")
    I, F = _native(path).blob()
    asset = load_model(W.PUSHER_BLOB)
    nfrec = int(asset.I[B.TSIM_IH_FOFF_CPT])
    assert np.array_equal(I, asset.I)
    assert F[:nfrec].astype(np.float32).tobytes() == asset.F[:nfrec].astype(np.float32).tobytes()
    assert 0 < (F[:nfrec] != asset.F[:nfrec]).sum() <= 16      # (if this ever becomes 0 the two compilers agree to the bit: fine, relax the bound)


def test_the_repositorys_own_small_models_compile_to_the_same_bits():
    """tests/models/*.xml (no meshes, axis-aligned frames): not one bit between the two compilers — the GPU test of the loader
    (tests/test_gpu_native_model.py) relies on it to compare trajectories bit for bit"""
    import glob
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "*.xml")))
    assert len(paths) >= 8
    for path in paths:
        nm, py = _native(path), _python(path)
        I, F = nm.blob()
        assert np.array_equal(I, py.I) and F.tobytes() == py.F.tobytes(), path
        _same_lookups(nm, py)


# --------------------------------------------------------------------------------------------------------------- synthetic models
CUBE_OBJ = """\
# unit-ish box 0.04 x 0.02 x 0.06 with quads, one face given by negative indices, texture / normal references
v -0.02 -0.01 -0.03
v  0.02 -0.01 -0.03
v  0.02  0.01 -0.03
v -0.02  0.01 -0.03
v -0.02 -0.01  0.03
v  0.02 -0.01  0.03
v  0.02  0.01  0.03
v -0.02  0.01  0.03
vn 0 0 1
f 4/1/1 3/1/1 2/1/1 1/1/1
f 5 6 7 8
f 1//1 2//1 6//1 5//1
f 2 3 7 6
f 3 4 8 7
f -5 -8 -4 -1
"""

PAD_POINTS = "5 points\n" + "\n".join("%g %g %g" % p for p in [(0.0, 0.0, 0.002), (0.004, 0.004, 0.002), (-0.004, 0.004, 0.002), (0.004, -0.004, 0.002), (-0.004, -0.004, 0.002)]) + "\n"

TAXELS = "4\n" + "\n".join('"%g %g 0.002" "%d %d" "0 0 1" "1 0 0" "0 1 0"' % (x, y, r, c) for (x, y, r, c) in
                           [(-0.003, -0.003, 0, 0), (0.003, -0.003, 0, 1), (-0.003, 0.003, 1, 0), (0.003, 0.003, 2, 1)]) + "\n"

MODEL_A = """\
<?xml version="1.0" encoding="utf-8"?>
<!-- a pad on a planar + revolute carriage pushing a puck and a ball; every primitive kind, a mesh, fixed-joint merging -->
<redmax model='synthetic &amp; small'>
    <option integrator="BDF1" timestep="4e-3" unit="m-kg" gravity="0. 0. -9.8"/>
    <solver_option tol="1e-9" max_iter="60" max_ls="12"/>
    <ground pos="0 0 0" normal="0 0.01 1"/>
    <default>
        <joint lim_stiffness="25" damping="1.5"/>
        <general_primitive_contact kn="4e3" kt="4." mu="1.2" damping="50"/>
        <ground_contact kn="2e3" kt="2" mu="0.7" damping="0.4"/>
        <tactile kn="90" kt="7." mu="0.9" damping="9"/>
        <motor P="8." D="0.2" ctrl_range="-2 2" ctrl="force"/>
    </default>
    <robot>
        <link name="carriage">
            <joint name = "carriage_xy" type = "planar" axis0="1 0 0" axis1="0 2 0" pos = "0.01 0 0.1" quat = "1 0 0 0"/>
            <body name = "carriage_body" type = "mesh" filename="box.obj" pos = "0 0 0.01" quat = "0.924 0 0 0.383" density = "800" transform_type="OBJ_TO_JOINT"/>
            <link name="yaw">
                <joint name="yaw" type="revolute" axis="0 0 3" pos="0 0 -0.02" quat="1 0 0 0" lim="-1.2 1.2" damping="0.5"/>
                <body name="yaw_body" type="cuboid" size="0.01 0.012 0.014" pos="0 0.001 0" quat="1 0 0 0" density="50"/>
                <link name="bracket">
                    <joint name="bracket_fixed" type="fixed" pos="0 0 -0.03" quat="0.707 0 0.707 0"/>
                    <body name="bracket_body" type="mesh" filename="box.obj" pos="0.01 0 0.05" quat="1 0 0 0" transform_type="OBJ_TO_WORLD"/>
                    <link name="pad">
                        <joint name="pad_fixed" type="fixed" pos="0.002 0 0.01" quat="1 0 0 0"/>
                        <body name="pad" type="cylinder" density="2" radius="0.015" length="0.004" pos="0 0 0" quat="1 0 0 0" general_contact_angle_resolution="6" general_contact_radius_resolution="3"/>
                    </link>
                    <link name="slider">
                        <joint name="slider" type="prismatic" axis="1 0 0" pos="0 0.02 0" quat="1 0 0 0" lim="-0.01 0.02" lim_stiffness="40"/>
                        <body name="slider_body" type="abstract" mass="0.02" inertia="1e-6 2e-6 3e-6" pos="0 0 0.001" quat="0.966 0.259 0 0">
                            <collision contacts="pad_points.txt" pos="0 0 0.001" quat="1 0 0 0"/>
                        </body>
                    </link>
                </link>
            </link>
        </link>
    </robot>
    <robot>
        <link name="puck">
            <joint name="puck" type="free3d-euler" pos="0.06 0 0.0251" quat="1 0 0 0" damping="0.01"/>
            <body name="puck" type="cuboid" size="0.05 0.04 0.05" pos="0 0 0" quat="1 0 0 0" density="500" general_contact_resolution="3 2 3"/>
        </link>
        <link name="ball">
            <joint name="ball" type="free3d-exp" pos="-0.05 0.03 0.0152" quat="1 0 0 0"/>
            <body name="ball" type="sphere" radius="0.015" pos="0 0 0" quat="1 0 0 0" density="300"/>
        </link>
        <link name="post">
            <joint name="post" type="translational" pos="0.1 0.1 0.03" quat="1 0 0 0" damping="3"/>
            <body name="post" type="cylinder" radius="0.01" length="0.06" pos="0 0 0" quat="1 0 0 0" density="700"/>
        </link>
    </robot>
    <contact>
        <ground_contact body="puck"/>
        <ground_contact body="ball" kn="3e3" mu="0.5"/>
        <ground_contact body="post" damping="0.2"/>
        <general_primitive_contact general_body="pad" primitive_body="puck" kn="1e2" kt="8." mu="1." damping="10"/>
        <general_primitive_contact general_body="pad" primitive_body="ball"/>
        <general_primitive_contact general_body="slider_body" primitive_body="post" mu="0.3"/>
    </contact>
    <actuator>
        <motor joint="carriage_xy" ctrl="force" ctrl_range="-1 1"/>
        <motor joint="yaw"/>
        <motor joint="slider" ctrl="position" P="20" D="0.5" ctrl_range="-0.01 0.02"/>
        <motor joint="puck" ctrl="force" ctrl_range="-0.1 0.1"/>
    </actuator>
    <sensor>
        <tactile body="pad" name="pad_array" type="rect_array" rect_pos0="0.006 0.005 0.002" rect_pos1="-0.006 -0.005 0.002" axis0="-1 0 0" axis1="0 -2 0" resolution="5 4" kn="120"/>
        <tactile body="slider_body" name="slider_skin" type="abstract" spec="skin_taxels.txt" pos="0 0 0.0005" quat="1 0 0 0" mu="1.1"/>
    </sensor>
    <variable>
        <endeffector joint="pad_fixed" pos="-0.005 0 0"/>
        <endeffector joint="puck" pos="-0.025 0 0" name="puck_face"/>
    </variable>
    <virtual>
        <cuboid name="goal" pos="1 0 0.025" quat="1 0 0 0" size="0.05 0.05 0.05"/>
    </virtual>
</redmax>
"""

MODEL_B = """\
<redmax model="bare">
    <robot>
        <link name="a">
            <joint name="a" type="revolute" axis="0 1 0" pos="0 0 0.2"/>
            <body name="a" type="cuboid" size="0.02 0.02 0.1" pos="0 0 -0.05"/>
            <link name="b">
                <joint name="b" type="revolute" axis="0 1 0" pos="0 0 -0.1" quat="0.9 0.1 0 0"/>
                <body name="b" type="sphere" radius="0.02" pos="0 0 -0.05" density="10"/>
            </link>
        </link>
    </robot>
</redmax>
"""


@pytest.fixture()
def model_dir(tmp_path):
    for name, text in (("a.xml", MODEL_A), ("b.xml", MODEL_B), ("box.obj", CUBE_OBJ), ("pad_points.txt", PAD_POINTS), ("skin_taxels.txt", TAXELS)):
        (tmp_path / name).write_text(text)
    return tmp_path


def test_synthetic_models_compile_to_the_same_blob(model_dir):
    from tactilesimulation_amd.model import blob as B
    for name in ("a.xml", "b.xml"):
        path = str(model_dir / name)
        nm, py = _native(path), _python(path)
        I, F = nm.blob()
        _same_blob(I, F, py)
        _same_lookups(nm, py)
    py = _python(str(model_dir / "a.xml"))
    # the model is what it was written to be: 2 + 1 + 1 + (3 + 1 + 1 + 1) + (3 + 3) + 3 dofs, fixed joints merged, both sensor kinds
    assert py.ndof_r == 19 and py.ndof_u == 2 + 1 + 1 + 6 and py.I[B.TSIM_IH_NSENSOR] == 2 and py.I[B.TSIM_IH_NPAIR] == 6 and py.ndof_var == 6
    assert py.meta["sensor_taxels"][1][2:] == (3, 2)      # abstract sensor: rows / cols from the largest image position


def test_mesh_mass_properties_known_answer(model_dir):
    """the OBJ is a 0.04 x 0.02 x 0.06 box: volume, centre and inertia of the first link (density 800, rotated 45 degrees about z, lifted 0.01) by hand"""
    from tactilesimulation_amd.model import blob as B
    I, F = _native(str(model_dir / "a.xml")).blob()
    lf = F[I[B.TSIM_IH_FOFF_LINK]:][:B.TSIM_LF_SIZE]
    m = 800 * 0.04 * 0.02 * 0.06
    assert abs(lf[B.TSIM_LF_MASS] - m) < 1e-15
    assert np.allclose(lf[B.TSIM_LF_COM:B.TSIM_LF_COM + 3], [0, 0, 0.01], atol=1e-15)
    ixx, iyy, izz = m / 12 * (0.02 ** 2 + 0.06 ** 2), m / 12 * (0.04 ** 2 + 0.06 ** 2), m / 12 * (0.04 ** 2 + 0.02 ** 2)
    q = np.array([0.924, 0, 0, 0.383]); q /= np.linalg.norm(q)
    c, s = 1 - 2 * q[3] ** 2, 2 * q[0] * q[3]
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    Ic = R @ np.diag([ixx, iyy, izz]) @ R.T
    got = lf[B.TSIM_LF_INERTIA:B.TSIM_LF_INERTIA + 6]
    assert np.allclose(got, [Ic[0, 0], Ic[1, 1], Ic[2, 2], Ic[0, 1], Ic[0, 2], Ic[1, 2]], rtol=1e-12, atol=1e-18)


UPDATES = [("joint_damping", "yaw", [0.75], None), ("joint_location", "slider", [0.001, 0.021, -0.002], None), ("body_density", "puck", [650.0], None),
           ("body_size", "puck", [0.045, 0.05, 0.04], None), ("body_size", "ball", [0.012], None), ("body_size", "post", [0.05, 0.008], None),
           ("endeffector_position", "puck_face", [-0.02, 0.001, 0.0], None), ("contact_parameters", "pad", [150.0, float("nan"), 0.8, float("nan")], "puck"),
           ("tactile_parameters", "pad", [float("nan"), 6.0, float("nan"), 12.0], None), ("tactile_parameters", "slider_skin", [70.0, 5.0, 1.0, 8.0], None),
           ("virtual_object", "goal", [0.5, 0.1, 0.025, 1, 0, 0, 0], None)]


def test_updates_recompile_like_the_python_spec_edits(model_dir):
    """the update_* family (envs/dclaw_rotate_env.py:173-178, envs/stable_grasp_env.py:122, envs/tactile_insertion_env.py:254-279): edits accumulate"""
    from tactilesimulation_amd.model.compiler import parse_xml, compile_spec, edit_spec
    path = str(model_dir / "a.xml")
    nm, spec = _native(path), parse_xml(path)
    before = nm.blob()
    for what, name, vals, name2 in UPDATES:
        nm.update(what, name, vals, name2)
        if what in ("contact_parameters", "tactile_parameters"):
            kw = {k: (None if np.isnan(v) else v) for k, v in zip(("kn", "kt", "mu", "damping"), vals)}
            edit_spec(spec, what, (name, name2) if name2 else name, **kw)
        else:
            edit_spec(spec, what, name, vals[0] if what in ("joint_damping", "body_density") else vals)
        I, F = nm.blob()
        _same_blob(I, F, compile_spec(spec))
    assert not np.array_equal(before[1], nm.blob()[1])


def test_a_failed_update_leaves_the_model_unchanged(model_dir):
    nm = _native(str(model_dir / "a.xml"))
    I0, F0 = nm.blob()
    for what, name, vals, name2, msg in (("joint_damping", "no_such_joint", [1.0], None, "unknown joint"), ("body_size", "slider_body", [1.0], None, "abstract"),
                                         ("contact_parameters", "pad", [1, 1, 1, 1], "post", "no general_primitive_contact"), ("body_size", "puck", [1.0], None, "3 values"),
                                         ("endeffector_position", "nowhere", [0, 0, 0], None, "unknown endeffector"), ("tactile_parameters", "puck", [1, 1, 1, 1], None, "no tactile sensor")):
        with pytest.raises(RuntimeError, match=msg):
            nm.update(what, name, vals, name2)
    I1, F1 = nm.blob()
    assert np.array_equal(I0, I1) and np.array_equal(F0, F1)
    with pytest.raises(KeyError):
        nm.table_offset("pair", "pad", "post", "kn")
    with pytest.raises(RuntimeError, match="unknown tactile sensor"):
        nm.image_pos("nope")


def test_blob_file_round_trip(model_dir, tmp_path):
    nm = _native(str(model_dir / "a.xml"))
    out = str(tmp_path / "a.tsimblob")
    nm.save_blob(out)
    again = _native(out)
    I0, F0 = nm.blob()
    I1, F1 = again.blob()
    assert np.array_equal(I0, I1) and F0.tobytes() == F1.tobytes()
    assert os.path.getsize(out) == 16 + 4 * len(I0) + 8 * len(F0)
    with pytest.raises(RuntimeError, match="no description"):
        again.update("joint_damping", "yaw", [1.0])
    raw = open(out, "rb").read()
    (tmp_path / "short.tsimblob").write_bytes(raw[:len(raw) // 2])
    with pytest.raises(RuntimeError, match="truncated"):
        _native(str(tmp_path / "short.tsimblob"))
    (tmp_path / "magic.tsimblob").write_bytes(b"\0\0\0\0" + raw[4:])
    with pytest.raises(RuntimeError, match="magic"):
        _native(str(tmp_path / "magic.tsimblob"))


@pytest.mark.parametrize("text,msg", [
    ("<mujoco/>", "not a redmax model"),
    ("<redmax><robot><link name='l'><joint name='j' type='revolute'/></link></robot></redmax>", "needs one <joint> and one <body>"),
    ("<redmax><robot><link name='l'><joint name='j' type='hinge'/><body name='b' type='sphere' radius='1'/></link></robot></redmax>", "joint type 'hinge'"),
    ("<redmax><robot><link name='l'><joint name='j' type='revolute'/><body name='b' type='capsule'/></link></robot></redmax>", "body type 'capsule'"),
    ("<redmax><robot><link name='l'><joint name='j' type='revolute' pos='0 0'/><body name='b' type='sphere' radius='1'/></link></robot></redmax>", "expected 3 numbers"),
    ("<redmax><option unit='cm-g'/></redmax>", "m-kg"),
    ("<redmax><robot><link name='l'><joint name='j' type='revolute'/><body name='b' type='sphere' radius='1'/></link></robot><contact><ground_contact body='b'/></contact></redmax>", "without <ground>"),
    ("<redmax><robot><link name='l'><joint name='j' type='revolute'/><body name='b' type='mesh' filename='missing.obj'/></link></robot></redmax>", "cannot open mesh"),
    ("<redmax><robot><link name='l'>", "XML"),
    ("<redmax><option integrator='RK4'/></redmax>", "integrator 'RK4'"),
    ("<redmax>" + "<a>" * 5000 + "</a>" * 5000 + "</redmax>", "nested deeper than 256"),
])
def test_bad_models_fail_with_a_reason(tmp_path, text, msg):
    p = tmp_path / "bad.xml"
    p.write_text(text)
    with pytest.raises(RuntimeError, match=msg):
        _native(str(p))


def test_missing_file():
    with pytest.raises(RuntimeError, match="cannot open"):
        _native("/nonexistent/model.xml")


FUZZ = r"""
import os, random, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import test_native_model_loader as T
from tactilesimulation_amd.host.native_model import NativeModel
d = sys.argv[2]
rng = random.Random(11)
ok = err = 0
for it in range(int(sys.argv[3])):
    s = list(T.MODEL_A)
    for _ in range(rng.randint(1, 4)):
        k = rng.randrange(len(s)); op = rng.randrange(5)
        if op == 0: del s[k:k + rng.randint(1, 30)]
        elif op == 1: s[k] = rng.choice('<>/"\'= &;-!?0aZ\n')
        elif op == 2: s.insert(k, rng.choice(['<', '>', '"', '</link>', '<link>', '<!--', '-->', '&amp', '1e999', 'nan', '-', '99999999', '<joint/>', '<body type="mesh"/>']))
        elif op == 3: s = s[:k]
        else:
            j = rng.randrange(len(s)); s[k], s[j] = s[j], s[k]
    p = os.path.join(d, "m.xml"); open(p, "w").write("".join(s))
    try:
        I, F = NativeModel(p).blob(); ok += 1
        assert I[29] == len(I) and I[30] == len(F)
    except RuntimeError:
        err += 1
print("ok", ok, "err", err)
"""


def test_mangled_models_are_refused_or_loaded_never_a_crash(model_dir):
    """the loader is C++ inside the product library: whatever the file holds — truncated, unbalanced, numbers out of range — the call returns
    (an error with a reason, or a model); 500 random manglings of the synthetic model in a child process, whose exit code is the check"""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, "-c", FUZZ, root, str(model_dir), "500"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    ok, err = (int(x) for x in r.stdout.split()[1::2])
    assert ok + err == 500 and err > 300 and ok > 5


def test_the_c_host_example_compiles_against_the_headers(tmp_path):
    """examples/c_host/step_from_xml.c is C99 against include/*.h: every declaration it uses parses as C and links against the built library
    (it RUNS on the GPU box: tests/test_gpu_native_model.py)"""
    import shutil
    import subprocess
    from tactilesimulation_amd.host import capi
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h") or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("no C compiler / HIP runtime on this machine")
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    libdir = os.path.dirname(capi.LIB_PATH)
    r = subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                        os.path.join(root, "examples", "c_host", "step_from_xml.c"), "-o", str(tmp_path / "step_from_xml"), "-L" + libdir, "-ltsim_hip",
                        "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_null_and_bad_arguments_are_errors_not_crashes():
    """every entry point of include/tsim_model.h with NULL / out-of-range arguments: an error code (or -1) and a reason in tsim_last_error"""
    import ctypes as C
    from tactilesimulation_amd.host import capi
    L = capi.lib()
    h, out = C.c_void_p(), C.c_void_p()
    assert L.tsim_model_load(None, C.byref(h)) == 1 and L.tsim_model_load(b"/nonexistent.xml", None) == 1
    assert L.tsim_model_blob(None, None, None, None, None) == 1 and b"null model" in L.tsim_last_error()
    assert L.tsim_model_save_blob(None, b"/tmp/never_written.tsimblob") == 1 and L.tsim_model_load_blob(b"/nonexistent", C.byref(h)) == 1
    assert L.tsim_model_image_pos(None, b"a", None, 0) == -1 and L.tsim_model_table_offset(None, 0, b"a", b"b", 0) == -1
    assert L.tsim_model_update(None, 0, b"a", None, None, 0) == 1 and L.tsim_batch_create_from_model(None, 1, 1, 0, 0, C.byref(out)) == 1
    L.tsim_model_free(None)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "pendulum.xml")
    assert L.tsim_model_load(path.encode(), C.byref(h)) == 0
    try:
        assert L.tsim_model_update(h, 99, b"a", None, None, 0) == 1 and b"unknown kind" in L.tsim_last_error()
        assert L.tsim_model_update(h, 0, b"a", None, None, 1) == 1                       # a count without values
        assert L.tsim_model_table_offset(h, 7, b"a", None, 0) == -1 and L.tsim_model_table_offset(h, 0, b"a", None, 0) == -1
        assert L.tsim_model_image_pos(h, b"none", None, 5) == -1 and b"unknown tactile sensor" in L.tsim_last_error()
        assert L.tsim_batch_create_from_model(h, 0, 1, 0, 0, C.byref(out)) == 1 and not out.value      # (refused before any GPU call)
        assert L.tsim_model_blob(h, None, None, None, None) == 0                        # every out-pointer is optional
    finally:
        L.tsim_model_free(h)


def _random_model(rng, max_dof=28, files_dir=None):
    """a random redmax XML: a forest of links with every joint / body kind, ground + general-primitive contacts, motors, rect_array sensors, end-effectors;
    files_dir: abstract bodies may carry contact-point files and sensors abstract taxel files, written there (the D'Claw vocabulary:
    envs/assets/dclaw_rotate/dclaw_position_control.xml:18-21,133-148)"""
    f = lambda lo, hi: "%.6g" % rng.uniform(lo, hi)
    vec = lambda lo, hi, n=3: " ".join(f(lo, hi) for _ in range(n))

    def quat():
        q = rng.normal(size=4)
        return " ".join("%.4f" % x for x in q / np.linalg.norm(q)) if rng.uniform() < 0.7 else "1 0 0 0"
    joints, bodies, budget = [], [], [max_dof]
    ndof = {"fixed": 0, "revolute": 1, "prismatic": 1, "planar": 2, "translational": 3, "free3d-euler": 6, "free3d-exp": 6}

    def link(depth):
        kinds = [k for k in ndof if ndof[k] <= budget[0] and (depth == 0 or not k.startswith("free3d"))]
        jt = kinds[rng.integers(len(kinds))]
        budget[0] -= ndof[jt]
        name = "j%d" % len(joints)
        joints.append((name, jt))
        ja = 'name="%s" type="%s" pos="%s" quat="%s"' % (name, jt, vec(-0.1, 0.1), quat())
        if jt in ("revolute", "prismatic"):
            ja += ' axis="%s"' % vec(-1, 1)
            if rng.uniform() < 0.5:
                ja += ' lim="%s %s" lim_stiffness="%s"' % (f(-1, 0), f(0, 1), f(0, 50))
        if jt == "planar":
            ja += ' axis0="%s" axis1="%s"' % (vec(-1, 1), vec(-1, 1))
        if rng.uniform() < 0.5:
            ja += ' damping="%s"' % f(0, 3)
        bt = ["cuboid", "sphere", "cylinder", "abstract"][rng.integers(4)]
        bname = "b%d" % len(bodies)
        bodies.append((bname, bt))
        ba = 'name="%s" type="%s" pos="%s" quat="%s"' % (bname, bt, vec(-0.05, 0.05), quat())
        if bt == "cuboid":
            ba += ' size="%s" density="%s" general_contact_resolution="%d %d %d"' % (vec(0.01, 0.1), f(1, 1000), rng.integers(2, 5), rng.integers(2, 4), rng.integers(2, 4))
        elif bt == "sphere":
            ba += ' radius="%s" density="%s"' % (f(0.01, 0.05), f(1, 1000))
        elif bt == "cylinder":
            ba += ' radius="%s" length="%s" general_contact_angle_resolution="%d" general_contact_radius_resolution="%d"' % (f(0.01, 0.05), f(0.01, 0.1), rng.integers(3, 9), rng.integers(1, 4))
        else:
            ba += ' mass="%s" inertia="%s"' % (f(0.01, 1), vec(1e-5, 1e-3))
        inner = ""
        if bt == "abstract" and files_dir is not None and rng.uniform() < 0.6:
            pts = rng.uniform(-0.03, 0.03, size=(rng.integers(1, 9), 3))
            with open(os.path.join(files_dir, "pts_%s.txt" % bname), "w") as fh:
                fh.write("%d\n" % len(pts) + "".join("%.6g %.6g %.6g\n" % tuple(x) for x in pts))
            inner = '<collision contacts="pts_%s.txt" pos="%s" quat="%s"/>' % (bname, vec(-0.01, 0.01), quat())
            bodies[-1] = (bname, "abstract+points")
        kids = "".join(link(depth + 1) for _ in range(rng.integers(0, 3))) if depth < 3 and budget[0] > 0 else ""
        return '<link name="l_%s"><joint %s/><body %s>%s</body>%s</link>' % (name, ja, ba, inner, kids) if inner else '<link name="l_%s"><joint %s/><body %s/>%s</link>' % (name, ja, ba, kids)
    robots = "".join("<robot>%s</robot>" % link(0) for _ in range(rng.integers(1, 4)))
    general = [b for b, t in bodies if t in ("cuboid", "cylinder", "abstract+points")]
    prim = [b for b, t in bodies if t in ("cuboid", "sphere", "cylinder")]
    contacts = ""
    for b, t in bodies:
        if t != "abstract" and rng.uniform() < 0.4:      # ("abstract+points" bodies included)
            contacts += '<ground_contact body="%s" kn="%s" mu="%s"/>' % (b, f(1e2, 1e4), f(0, 1))
    pairs = set()
    for _ in range(rng.integers(0, 4)):
        if general and prim:
            g, p_ = general[rng.integers(len(general))], prim[rng.integers(len(prim))]
            if g != p_ and (g, p_) not in pairs:
                pairs.add((g, p_))
                contacts += '<general_primitive_contact general_body="%s" primitive_body="%s" kt="%s" damping="%s"/>' % (g, p_, f(0, 10), f(0, 100))
    motors = "".join('<motor joint="%s" ctrl="%s" ctrl_range="%s %s"%s/>' % (n, ["force", "position"][rng.integers(2)], f(-3, 0), f(0, 3), ' P="%s" D="%s"' % (f(0, 50), f(0, 1)) if rng.uniform() < 0.5 else "")
                     for n, t in joints if ndof[t] > 0 and rng.uniform() < 0.5)
    def sensor(g):
        if files_dir is not None and rng.uniform() < 0.5:
            n = int(rng.integers(1, 10))
            rows = []
            for k in range(n):
                a0 = rng.normal(size=3); a0 /= np.linalg.norm(a0)
                a1 = np.cross(a0, rng.normal(size=3)); a1 /= np.linalg.norm(a1)
                rows.append('"%s" "%d %d" "%s" "%s" "%s"' % (vec(-0.01, 0.01), k // 3, k % 3, " ".join("%.6g" % x for x in np.cross(a1, a0)), " ".join("%.6g" % x for x in a0), " ".join("%.6g" % x for x in a1)))
            with open(os.path.join(files_dir, "tax_%s.txt" % g), "w") as fh:
                fh.write("%d\n" % n + "\n".join(rows) + "\n")
            return '<tactile body="%s" name="s_%s" type="abstract" spec="tax_%s.txt" pos="%s" quat="%s" kt="%s"/>' % (g, g, g, vec(-0.005, 0.005), quat(), f(0, 10))
        return ('<tactile body="%s" name="s_%s" type="rect_array" rect_pos0="%s" rect_pos1="%s" axis0="%s" axis1="%s" resolution="%d %d" kn="%s"/>'
                % (g, g, vec(-0.01, 0.01), vec(-0.01, 0.01), vec(-1, 1), vec(-1, 1), rng.integers(1, 6), rng.integers(1, 6), f(10, 200)))
    sensors = "".join(sensor(g) for g in sorted({g for g, _ in pairs}))
    ee = "".join('<endeffector joint="%s" pos="%s"/>' % (n, vec(-0.05, 0.05)) for n, _ in joints if rng.uniform() < 0.3)
    return ('<redmax model="random"><option integrator="%s" timestep="%s" gravity="%s"/><solver_option tol="1e-9" max_iter="%d" max_ls="%d"/>'
            '<ground pos="%s" normal="%s"/><default><joint lim_stiffness="%s" damping="%s"/><motor P="%s" D="%s" ctrl_range="-1.5 1.5"/></default>%s<contact>%s</contact>'
            '<actuator>%s</actuator><sensor>%s</sensor><variable>%s</variable></redmax>') % (
        ["BDF1", "BDF2"][rng.integers(2)], f(1e-3, 1e-2), vec(-10, 10), rng.integers(10, 100), rng.integers(5, 20), vec(-0.1, 0.1), "%s %s 1" % (f(-0.2, 0.2), f(-0.2, 0.2)),
        f(0, 20), f(0, 2), f(0, 10), f(0, 1), robots, contacts, motors, sensors, ee)


def test_random_models_compile_to_the_same_blob(tmp_path):
    """300 random kinematic forests (every joint and primitive body kind, random frames, contacts, motors, sensors): the two compilers agree on
    every int and every real, and refuse the same models"""
    rng = np.random.default_rng(2026)
    p = str(tmp_path / "r.xml")
    compiled = refused = 0
    for it in range(300):
        open(p, "w").write(_random_model(rng))
        try:
            py = _python(p)
        except Exception:
            py = None
        try:
            nm = _native(p)
        except RuntimeError:
            nm = None
        assert (py is None) == (nm is None), (it, open(p).read())
        if py is None:
            refused += 1
            continue
        I, F = nm.blob()
        try:
            _same_blob(I, F, py)
            _same_lookups(nm, py)
        except AssertionError:
            print(open(p).read())
            raise
        compiled += 1
    assert compiled >= 250, (compiled, refused)
    # ... and 150 more whose abstract bodies carry contact-point files and whose sensors may be abstract taxel files
    files = 0
    for it in range(150):
        text = _random_model(rng, files_dir=str(tmp_path))
        files += "contacts=" in text or 'type="abstract" spec=' in text
        open(p, "w").write(text)
        nm, py = _native(p), _python(p)
        I, F = nm.blob()
        _same_blob(I, F, py)
        _same_lookups(nm, py)
    assert files >= 60, files
