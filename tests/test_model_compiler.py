"""Model compiler: blob layout, merged links, mass properties, point sampling, taxel grid, spec edits."""
import os

import numpy as np
import pytest

import tactilesimulation_amd.model.blob as B
from tactilesimulation_amd.model import compiler as mc
from tactilesimulation_amd.model.geometry import mesh_props, cuboid_surface_lattice, cylinder_cap_points, quat_to_R
from tactilesimulation_amd.workloads import asset

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def test_pusher_blob_dimensions(pusher_model):
    m = pusher_model
    # ndof_r 7, ndof_u 6, ndof_var 6, ndof_tactile 390 (13 x 10 x 3)
    assert (m.ndof_r, m.ndof_u, m.ndof_var, m.ndof_tactile) == (7, 6, 6, 390)
    assert m.n_links == 4 and abs(m.h - 5e-3) < 1e-15      # 7 bodies, fixed joints merged into 4 links
    assert m.I[B.TSIM_IH_NCPT] == 8 + 2 * (8 * 4 + 1)       # box corners + two pad caps
    masses = m.meta["link_mass"]
    assert abs(masses[3] - 600 * 0.05 ** 3) < 1e-12
    # gripper link = wsg50_base (1000 kg/m^3 x 4.910e-4 m^3) + light default-density parts
    assert abs(masses[1] - 0.4910557) < 2e-4
    assert m.meta["image_pos"]["tactile_pad_left"][:11] == [(0, j) for j in range(10)] + [(1, 0)]


def test_blob_roundtrip_and_all_reference_models_load():
    for name, dims in {"pusher": (7, 6, 6, 390), "tactile_pad": (9, 3, 0, 120000), "tactile_insertion": (12, 6, 0, 780),
                       "dclaw_position_control": (10, 9, 12, 2718), "stable_grasp": (12, 6, 0, 780)}.items():
        m = mc.load_model(asset(name))
        assert (m.ndof_r, m.ndof_u, m.ndof_var, m.ndof_tactile) == dims
        assert m.I[B.TSIM_IH_MAGIC] == B.TSIM_MAGIC and len(m.I) == m.I[B.TSIM_IH_NI] and len(m.F) == m.I[B.TSIM_IH_NF]


def test_taxel_grid_geometry(pusher_model):
    m = pusher_model
    nt = m.I[B.TSIM_IH_NTAXEL]
    T = m.F[m.I[B.TSIM_IH_FOFF_TAXEL]:m.I[B.TSIM_IH_FOFF_TAXEL] + 12 * nt].reshape(12, nt).T
    pos = T[:, :3].reshape(13, 10, 3)
    # 1.5 mm pitch in both directions (18 mm / 12, 13.5 mm / 9), planar
    assert np.allclose(np.linalg.norm(pos[1:] - pos[:-1], axis=-1), 1.5e-3, atol=1e-12)
    assert np.allclose(np.linalg.norm(pos[:, 1:] - pos[:, :-1], axis=-1), 1.5e-3, atol=1e-12)
    a0, a1, n = T[:, 3:6], T[:, 6:9], T[:, 9:12]
    assert np.allclose(np.cross(a1, a0), n, atol=1e-12) and np.allclose(np.linalg.norm(n, axis=1), 1.0)


def test_sampling_and_mass_helpers():
    assert len(cuboid_surface_lattice([1, 1, 1], [2, 2, 2])) == 8
    assert len(cuboid_surface_lattice([1, 1, 1], [5, 5, 2])) == 50
    assert len(cuboid_surface_lattice([1, 1, 1], [20, 20, 20])) == 20 ** 3 - 18 ** 3
    assert len(cylinder_cap_points(0.018, 0.003, 8, 4)) == 66
    # unit cube mesh: volume 1, com at centre, inertia 1/6
    V = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], dtype=float)
    Fc = np.array([[0, 2, 3], [0, 3, 1], [4, 5, 7], [4, 7, 6], [0, 1, 5], [0, 5, 4], [2, 6, 7], [2, 7, 3], [0, 4, 6], [0, 6, 2], [1, 3, 7], [1, 7, 5]])
    mp = mesh_props(V, Fc)
    assert abs(mp.m - 1) < 1e-12 and np.allclose(mp.c, 0.5) and np.allclose(mp.Ic, np.eye(3) / 6, atol=1e-12)
    assert np.allclose(quat_to_R([0.7071068, 0, 0.7071068, 0]) @ [0, 0, 1], [1, 0, 0], atol=1e-6)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_mesh_volumes_known_answers():
    """signed volumes of the referenced OBJ meshes."""
    from tactilesimulation_amd.model.geometry import load_obj
    for f, vol in (("wsg50_base.obj", 4.910e-4), ("guide_left.obj", 9.129e-6), ("gelslim_left.obj", 2.374e-5)):
        V, Fc = load_obj(os.path.join(REF, "envs/assets/pusher", f))
        assert abs(mesh_props(V, Fc).m - vol) < 2e-3 * vol


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_fixture_blobs_are_current():
    """tactilesimulation_amd/assets/*.npz equal a fresh compile of the reference XMLs (tools/make_model_fixtures.py)."""
    fresh = mc.load_model(os.path.join(REF, "envs/assets/pusher/pusher.xml"))
    stored = mc.load_model(asset("pusher"))
    assert np.array_equal(fresh.I, stored.I) and np.allclose(fresh.F, stored.F, rtol=0, atol=1e-15)


def test_spec_edits_recompile(pusher_model):
    spec = mc.compile_spec(pusher_model.spec).spec
    mc.edit_spec(spec, "contact_parameters", ("tactile_pad_left", "box"), kn=123.0, mu=0.25)
    mc.edit_spec(spec, "tactile_parameters", "tactile_pad_left", kt=3.0)
    mc.edit_spec(spec, "body_density", "box", 1200.0)
    mc.edit_spec(spec, "joint_damping", "box", 0.7)
    m2 = mc.compile_spec(spec)
    assert np.array_equal(m2.I, pusher_model.I)              # topology untouched -> tsim_update_model accepts it
    pf = m2.F[m2.I[B.TSIM_IH_FOFF_PAIR] + B.TSIM_PF_SIZE:]
    assert pf[B.TSIM_PF_KN] == 123.0 and pf[B.TSIM_PF_MU] == 0.25
    assert m2.F[m2.I[B.TSIM_IH_FOFF_SENSOR] + B.TSIM_SF_KT] == 3.0
    assert abs(m2.meta["link_mass"][3] - 1200 * 0.05 ** 3) < 1e-12
    assert m2.F[m2.I[B.TSIM_IH_FOFF_DOF] + 6 * B.TSIM_DF_SIZE + B.TSIM_DF_DAMPING] == 0.7
    with pytest.raises(KeyError):
        mc.edit_spec(spec, "body_density", "no_such_body", 1.0)


# ------------------------------------------------------------------------------------------------------------------------
# A SECOND, independent reading of the reference's XMLs (plain ElementTree, counting rules written out here, nothing shared with
# model/compiler.py): the oracle and the kernels consume the same compiled blob, so a parser error would be invisible to parity
# tests.  Checked per asset: dof / motor / variable / taxel / contact-point / link counts and the total mass of the closed-form
# bodies against what the compiled blob says.
_NDOF = {"fixed": 0, "revolute": 1, "prismatic": 1, "planar": 2, "translational": 3, "free3d-euler": 6, "free3d-exp": 6}
_NLINKS = {"fixed": 0, "revolute": 1, "prismatic": 1, "planar": 1, "translational": 1, "free3d-euler": 4, "free3d-exp": 2}
_XMLS = {"pusher": "envs/assets/pusher/pusher.xml", "dclaw_position_control": "envs/assets/dclaw_rotate/dclaw_position_control.xml",
         "tactile_insertion": "envs/assets/tactile_insertion/tactile_insertion.xml", "stable_grasp": "envs/assets/stable_grasp/stable_grasp.xml",
         "tactile_pad": "assets/tactile_pad/tactile_pad.xml"}


def _xml_statistics(path):
    import math
    import xml.etree.ElementTree as ET
    root = ET.parse(path).getroot()
    d = os.path.dirname(path)
    fl = lambda s: [float(x) for x in s.split()]
    joints = {j.get("name"): j.get("type") for r in root.iter("robot") for j in r.iter("joint")}        # not the <default> entries
    bodies = {b.get("name"): b for r in root.iter("robot") for b in r.iter("body")}

    def first_count(fn):
        with open(os.path.join(d, fn)) as f:
            return int(f.readline().split()[0])

    def points_of(b):
        t = b.get("type")
        if t == "cuboid":
            nx, ny, nz = (int(x) for x in b.get("general_contact_resolution", "2 2 2").split())
            return nx * ny * nz - max(nx - 2, 0) * max(ny - 2, 0) * max(nz - 2, 0)          # surface lattice
        if t == "cylinder":
            return 2 * (1 + int(b.get("general_contact_angle_resolution", "8")) * int(b.get("general_contact_radius_resolution", "2")))
        if t == "sphere":
            return 1                                                                      # [CHOICE] lowest point on the ground
        if t == "abstract":
            c = b.find("collision")
            return first_count(c.get("contacts"))
        raise AssertionError("sampled body of type %s" % t)
    st = {"ndof_r": sum(_NDOF[t] for t in joints.values()), "n_links": sum(_NLINKS[t] for t in joints.values()),
          "ndof_u": sum(_NDOF[joints[m.get("joint")]] for m in root.iter("motor") if m.get("joint")),
          "ndof_var": 3 * len(list(root.iter("endeffector")))}
    ntax = 0
    for s in root.iter("tactile"):
        if s.get("type") == "rect_array":
            r, c = (int(x) for x in s.get("resolution").split())
            ntax += r * c
        elif s.get("type") == "abstract":
            ntax += first_count(s.get("spec"))
    st["ndof_tactile"] = 3 * ntax
    contact = root.find("contact")
    ncpt = 0
    for c in (contact if contact is not None else []):
        ncpt += points_of(bodies[c.get("body") if c.tag == "ground_contact" else c.get("general_body")])
    st["ncpt"] = ncpt
    mass, has_mesh = 0.0, False

    def body_mass(b):
        t, rho = b.get("type"), float(b.get("density", "1"))                              # [CHOICE] default density 1
        if t == "cuboid":
            sx, sy, sz = fl(b.get("size")); return rho * sx * sy * sz
        if t == "sphere":
            return rho * 4.0 / 3.0 * math.pi * float(b.get("radius")) ** 3
        if t == "cylinder":
            return rho * math.pi * float(b.get("radius")) ** 2 * float(b.get("length"))
        if t == "abstract":
            return float(b.get("mass"))
        return None                                                                       # mesh

    def walk(link, moving):                        # bodies fixed to the world (every joint up to the root is `fixed`) carry no dynamics
        nonlocal mass, has_mesh
        j = link.find("joint")
        moving = moving or (j is not None and j.get("type") != "fixed")
        b = link.find("body")
        if b is not None and moving:
            bm = body_mass(b)
            if bm is None:
                has_mesh = True
            else:
                mass += bm
        for ch in link.findall("link"):
            walk(ch, moving)
    for r in root.iter("robot"):
        for l in r.findall("link"):
            walk(l, False)
    st["closed_form_mass"], st["has_mesh"] = mass, has_mesh
    return st


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("name", list(_XMLS))
def test_compiled_blob_agrees_with_an_independent_reading_of_the_xml(name):
    st = _xml_statistics(os.path.join(REF, _XMLS[name]))
    m = mc.load_model(asset(name))
    assert (m.ndof_r, m.ndof_u, m.ndof_var, m.ndof_tactile) == (st["ndof_r"], st["ndof_u"], st["ndof_var"], st["ndof_tactile"])
    assert m.n_links == st["n_links"], (m.n_links, st["n_links"])
    assert int(m.I[B.TSIM_IH_NCPT]) == st["ncpt"], (int(m.I[B.TSIM_IH_NCPT]), st["ncpt"])
    total = float(sum(m.meta["link_mass"]))                       # links 1..nl: the bodies that move
    if not st["has_mesh"]:
        assert abs(total - st["closed_form_mass"]) < 1e-9 * max(total, 1.0), (total, st["closed_form_mass"])
    else:
        assert total > st["closed_form_mass"]                     # the meshes add theirs (known volumes: test above)
