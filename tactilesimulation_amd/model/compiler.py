"""Model compiler: redmax XML (+ OBJ / contact / taxel files) -> ModelSpec -> flat blob.

Replaces the XML loader inside the reference's (absent) DiffRedMax `redmax_py.Simulation(model_path)`
(call site: envs/redmax_torch_env.py:33). Two stages:

  parse_xml(path)  -> ModelSpec   JSON-able description; meshes are reduced to their unit-density
                                  mass properties so that a spec is self-contained and can be
                                  shipped without the OBJ files.
  compile_spec(spec) -> CompiledModel  int32 I[] + float64 F[] laid out per include/tsim_blob.h.

The `update_*` calls of the reference binding edit the spec and recompile.

Modelling choices where the reference source is unavailable are marked [CHOICE] and listed in
DESIGN.md §Model.
"""
import copy
import json
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import blob as B
from .geometry import (Pose, MassProps, cuboid_props, sphere_props, cylinder_props, load_obj,
                       mesh_props, cuboid_surface_lattice, cylinder_cap_points, quat_to_R)

DEFAULT_DENSITY = 1.0   # [CHOICE] bodies that state no density (pusher.xml:24,27)


def _floats(s, n=None):
    v = [float(t) for t in s.replace(",", " ").split()]
    if n is not None and len(v) != n:
        raise ValueError("expected %d numbers, got %r" % (n, s))
    return v


# =============================================================================== XML -> spec
def parse_xml(path):
    path = os.path.abspath(path)
    base = os.path.dirname(path)
    root = ET.parse(path).getroot()
    if root.tag != "redmax":
        raise ValueError("not a redmax model: %s" % path)

    opt = root.find("option")
    unit = opt.get("unit", "m-kg") if opt is not None else "m-kg"
    if unit != "m-kg":
        raise NotImplementedError("only unit='m-kg' models are supported (all reference XMLs use it)")
    sol = root.find("solver_option")
    spec = {
        "model": root.get("model", ""),
        "options": {
            "integrator": (opt.get("integrator", "BDF1") if opt is not None else "BDF1"),
            "h": float(opt.get("timestep", "1e-2")) if opt is not None else 1e-2,
            "gravity": _floats(opt.get("gravity", "0 0 -9.8"), 3) if opt is not None else [0, 0, -9.8],
            "tol": float(sol.get("tol", "1e-9")) if sol is not None else 1e-9,
            "max_iter": int(sol.get("max_iter", "50")) if sol is not None else 50,
            "max_ls": int(sol.get("max_ls", "20")) if sol is not None else 20,
        },
        "ground": None, "joints": [], "contacts": [], "motors": [], "sensors": [],
        "endeffectors": [], "virtual": {},
    }
    g = root.find("ground")
    if g is not None:
        spec["ground"] = {"pos": _floats(g.get("pos", "0 0 0"), 3), "normal": _floats(g.get("normal", "0 0 1"), 3)}

    dflt = root.find("default")
    d_joint = {"lim_stiffness": 0.0, "damping": 0.0}
    d_gpc = {"kn": 1e3, "kt": 1.0, "mu": 1.0, "damping": 0.0}
    d_gc = {"kn": 1e3, "kt": 1.0, "mu": 1.0, "damping": 0.0}
    d_tac = {"kn": 1e2, "kt": 1.0, "mu": 1.0, "damping": 0.0}
    d_motor = {"P": 0.0, "D": 0.0, "ctrl_range": [-1.0, 1.0], "ctrl": "force"}
    if dflt is not None:
        for tag, dst in (("joint", d_joint), ("general_primitive_contact", d_gpc), ("ground_contact", d_gc),
                         ("tactile", d_tac)):
            e = dflt.find(tag)
            if e is not None:
                for k in dst:
                    if e.get(k) is not None:
                        dst[k] = float(e.get(k))
        e = dflt.find("motor")
        if e is not None:
            for k in ("P", "D"):
                if e.get(k) is not None:
                    d_motor[k] = float(e.get(k))
            if e.get("ctrl_range") is not None:
                d_motor["ctrl_range"] = _floats(e.get("ctrl_range"), 2)
            if e.get("ctrl") is not None:
                d_motor["ctrl"] = e.get("ctrl")

    # ---- kinematic tree, depth first, one joint + one body per <link>
    def world_pose_of(jidx):
        T = Pose()
        chain = []
        while jidx >= 0:
            chain.append(jidx)
            jidx = spec["joints"][jidx]["parent"]
        for j in reversed(chain):
            J = spec["joints"][j]
            T = T * Pose.from_pos_quat(J["pos"], J["quat"])
        return T   # at q = 0 every Q(q) is the identity

    def parse_link(link, parent):
        je, be = link.find("joint"), link.find("body")
        if je is None or be is None:
            raise ValueError("link %r needs one <joint> and one <body>" % link.get("name"))
        jtype = je.get("type")
        if jtype not in B.JOINT_NDOF:
            raise NotImplementedError("joint type %r" % jtype)
        J = {"name": je.get("name"), "type": jtype, "parent": parent,
             "pos": _floats(je.get("pos", "0 0 0"), 3), "quat": _floats(je.get("quat", "1 0 0 0"), 4),
             "damping": float(je.get("damping", d_joint["damping"])),
             "lim_stiffness": float(je.get("lim_stiffness", d_joint["lim_stiffness"])),
             "lim": _floats(je.get("lim"), 2) if (je.get("lim") is not None and jtype in ("revolute", "prismatic")) else None}
        if jtype in ("revolute", "prismatic"):
            J["axes"] = [_floats(je.get("axis", "0 0 1"), 3)]
        elif jtype == "planar":
            J["axes"] = [_floats(je.get("axis0", "1 0 0"), 3), _floats(je.get("axis1", "0 1 0"), 3)]
        else:
            J["axes"] = []
        jidx = len(spec["joints"])
        spec["joints"].append(J)
        J["body"] = parse_body(be, jidx)
        for child in link.findall("link"):
            parse_link(child, jidx)

    def parse_body(be, jidx):
        btype = be.get("type")
        Bd = {"name": be.get("name"), "type": btype, "pos": _floats(be.get("pos", "0 0 0"), 3),
              "quat": _floats(be.get("quat", "1 0 0 0"), 4),
              "density": float(be.get("density", DEFAULT_DENSITY)), "contacts": None}
        if btype == "cuboid":
            Bd["size"] = _floats(be.get("size"), 3)
            res = be.get("general_contact_resolution")
            Bd["contact_res"] = [int(t) for t in res.split()] if res else [2, 2, 2]
        elif btype == "sphere":
            Bd["radius"] = float(be.get("radius"))
        elif btype == "cylinder":
            Bd["radius"] = float(be.get("radius"))
            Bd["length"] = float(be.get("length"))
            Bd["contact_res"] = [int(be.get("general_contact_angle_resolution", "8")),
                                 int(be.get("general_contact_radius_resolution", "2"))]
        elif btype == "mesh":
            V, Fc = load_obj(os.path.join(base, be.get("filename")))
            T = Pose.from_pos_quat(Bd["pos"], Bd["quat"])
            ttype = be.get("transform_type", "OBJ_TO_JOINT")
            if ttype == "OBJ_TO_WORLD":       # vertices are placed in the world at q = 0
                T = world_pose_of(jidx).inv() * T
            elif ttype != "OBJ_TO_JOINT":
                raise NotImplementedError("transform_type %r" % ttype)
            mp = mesh_props(T.apply(V), Fc)   # unit density, joint frame
            Bd["mesh_unit"] = {"vol": mp.m, "com": mp.c.tolist(), "Ic": mp.Ic.tolist()}
            Bd["mesh_file"] = be.get("filename")
        elif btype == "abstract":
            Bd["mass"] = float(be.get("mass"))
            Bd["inertia"] = _floats(be.get("inertia"), 3)
            ce = be.find("collision")
            if ce is not None and ce.get("contacts"):
                pts = _read_points(os.path.join(base, ce.get("contacts")))
                # <collision pos quat> places the point file's frame in the BODY frame (dclaw_position_control.xml:18-21:
                # body frame o collision frame == joint frame, the files hold joint-frame coordinates); stored in the
                # JOINT frame
                Tc = Pose.from_pos_quat(Bd["pos"], Bd["quat"]) * \
                    Pose.from_pos_quat(_floats(ce.get("pos", "0 0 0"), 3), _floats(ce.get("quat", "1 0 0 0"), 4))
                Bd["contacts"] = Tc.apply(pts).tolist()
        else:
            raise NotImplementedError("body type %r" % btype)
        return Bd

    for robot in root.findall("robot"):
        for link in robot.findall("link"):
            parse_link(link, -1)

    ce = root.find("contact")
    if ce is not None:
        for e in ce:
            if e.tag == "ground_contact":
                c = {"type": "ground", "body": e.get("body")}
                src = d_gc
            elif e.tag == "general_primitive_contact":
                c = {"type": "general_primitive", "general_body": e.get("general_body"),
                     "primitive_body": e.get("primitive_body")}
                src = d_gpc
            else:
                raise NotImplementedError("contact type %r" % e.tag)
            for k in ("kn", "kt", "mu", "damping"):
                c[k] = float(e.get(k, src[k]))
            spec["contacts"].append(c)

    ae = root.find("actuator")
    if ae is not None:
        for e in ae.findall("motor"):
            rng = _floats(e.get("ctrl_range"), 2) if e.get("ctrl_range") else list(d_motor["ctrl_range"])
            spec["motors"].append({"joint": e.get("joint"), "ctrl": e.get("ctrl", d_motor["ctrl"]),
                                   "ctrl_range": rng, "P": float(e.get("P", d_motor["P"])),
                                   "D": float(e.get("D", d_motor["D"]))})

    se = root.find("sensor")
    if se is not None:
        for e in se.findall("tactile"):
            s = {"body": e.get("body"), "name": e.get("name"), "type": e.get("type"),
                 "kn": d_tac["kn"], "kt": d_tac["kt"], "mu": d_tac["mu"], "damping": d_tac["damping"]}
            for k in ("kn", "kt", "mu", "damping"):
                if e.get(k) is not None:
                    s[k] = float(e.get(k))
            if s["type"] == "rect_array":
                s["rect_pos0"] = _floats(e.get("rect_pos0"), 3)
                s["rect_pos1"] = _floats(e.get("rect_pos1"), 3)
                s["axis0"] = _floats(e.get("axis0"), 3)
                s["axis1"] = _floats(e.get("axis1"), 3)
                s["resolution"] = [int(t) for t in e.get("resolution").split()]
            elif s["type"] == "abstract":
                s["pos"] = _floats(e.get("pos", "0 0 0"), 3)
                s["quat"] = _floats(e.get("quat", "1 0 0 0"), 4)
                s["taxels"] = _read_taxel_spec(os.path.join(base, e.get("spec")))
            else:
                raise NotImplementedError("tactile type %r" % s["type"])
            spec["sensors"].append(s)

    ve = root.find("variable")
    if ve is not None:
        for i, e in enumerate(ve.findall("endeffector")):
            spec["endeffectors"].append({"joint": e.get("joint"), "pos": _floats(e.get("pos", "0 0 0"), 3),
                                         "name": e.get("name", "endeffector_%d" % i)})
    vv = root.find("virtual")
    if vv is not None:
        for e in vv:
            spec["virtual"][e.get("name")] = {"pos": _floats(e.get("pos", "0 0 0"), 3),
                                              "quat": _floats(e.get("quat", "1 0 0 0"), 4)}
    return spec


def _read_points(path):
    with open(path) as f:
        n = int(f.readline().split()[0])
        pts = [[float(t) for t in f.readline().split()[:3]] for _ in range(n)]
    return np.asarray(pts, dtype=np.float64)


def _read_taxel_spec(path):
    """count, then per line five quoted fields: "pos" "img_r img_c" "normal" "axis0" "axis1"
    (format of envs/assets/dclaw_rotate/tactile/dclaw_fingertip_tactile.txt)."""
    out = []
    with open(path) as f:
        n = int(f.readline().split()[0])
        for _ in range(n):
            fields = [t for t in f.readline().split('"') if t.strip()]
            pos, img, nrm, a0, a1 = [[float(x) for x in t.split()] for t in fields[:5]]
            out.append({"pos": pos, "img": [int(img[0]), int(img[1])], "normal": nrm, "axis0": a0, "axis1": a1})
    return out


# =============================================================================== spec -> blob
class CompiledModel:
    def __init__(self, spec, I, F, meta):
        self.spec, self.I, self.F, self.meta = spec, I, F, meta

    def table_offset(self, kind, key, field):
        """Index into F[] of one numeric parameter, for per-environment tables (BatchSim.set_env_tables).
        kind 'pair': key = (general_body | 'ground', primitive_body | body), field in kn kt mu damping shape0..3;
        kind 'sensor': key = sensor name, field in kn kt mu damping; kind 'dof': key = (joint name, k), field damping."""
        I = self.I
        if kind == "pair":
            idx = [tuple(k) for k in self.meta["pair_keys"]].index(tuple(key))
            f = {"kn": B.TSIM_PF_KN, "kt": B.TSIM_PF_KT, "mu": B.TSIM_PF_MU, "damping": B.TSIM_PF_KD}
            f.update({"shape%d" % i: B.TSIM_PF_SHAPE + i for i in range(4)})
            return int(I[B.TSIM_IH_FOFF_PAIR]) + idx * B.TSIM_PF_SIZE + f[field]
        if kind == "sensor":
            idx = self.meta["sensor_names"].index(key)
            f = {"kn": B.TSIM_SF_KN, "kt": B.TSIM_SF_KT, "mu": B.TSIM_SF_MU, "damping": B.TSIM_SF_KD}
            return int(I[B.TSIM_IH_FOFF_SENSOR]) + idx * B.TSIM_SF_SIZE + f[field]
        if kind == "dof":
            d0, nd = self.meta["dof_of_joint"][key[0]]
            return int(I[B.TSIM_IH_FOFF_DOF]) + (d0 + key[1]) * B.TSIM_DF_SIZE + {"damping": B.TSIM_DF_DAMPING}[field]
        raise KeyError(kind)

    ndof_r = property(lambda s: int(s.I[B.TSIM_IH_NR]))
    ndof_u = property(lambda s: int(s.I[B.TSIM_IH_NU]))
    ndof_var = property(lambda s: 3 * int(s.I[B.TSIM_IH_NVAR]))
    ndof_tactile = property(lambda s: 3 * int(s.I[B.TSIM_IH_NTAXEL]))
    h = property(lambda s: float(s.F[B.TSIM_FH_H]))
    n_links = property(lambda s: int(s.I[B.TSIM_IH_NL]))

    def save(self, path):
        np.savez_compressed(path, I=self.I, F=self.F, spec=np.frombuffer(json.dumps(self.spec).encode(), dtype=np.uint8))

    @staticmethod
    def load(path):
        z = np.load(path)
        spec = json.loads(bytes(z["spec"]).decode())
        cm = compile_spec(spec)
        # The stored arrays ARE the model (one ulp of numpy / BLAS drift in a recompile must not change what the kernels and the
        # oracle simulate); the recompile only guards against compiler-version skew.
        if not (np.array_equal(cm.I, z["I"]) and cm.F.shape == z["F"].shape and np.allclose(cm.F, z["F"], rtol=1e-12, atol=1e-300)):
            raise RuntimeError("model blob %s does not match its embedded spec (compiler version skew)" % path)
        cm.I, cm.F = z["I"].copy(), z["F"].copy()
        return cm


def load_model(path):
    """XML or precompiled .npz."""
    if path.endswith(".npz"):
        return CompiledModel.load(path)
    return compile_spec(parse_xml(path))


def _body_props(Bd):
    """Mass properties of one body in its JOINT frame."""
    t = Bd["type"]
    if t == "mesh":
        mu = Bd["mesh_unit"]
        return MassProps(mu["vol"], mu["com"], mu["Ic"]).scaled(Bd["density"])
    Tb = Pose.from_pos_quat(Bd["pos"], Bd["quat"])
    if t == "cuboid":
        mp = cuboid_props(Bd["size"], Bd["density"])
    elif t == "sphere":
        mp = sphere_props(Bd["radius"], Bd["density"])
    elif t == "cylinder":
        mp = cylinder_props(Bd["radius"], Bd["length"], Bd["density"])
    elif t == "abstract":
        mp = MassProps(Bd["mass"], np.zeros(3), np.diag(Bd["inertia"]))
    else:
        raise NotImplementedError(t)
    return mp.transformed(Tb)


def _body_contact_points(Bd):
    """Sampled surface points of a general (non-primitive) contact body, JOINT frame."""
    t = Bd["type"]
    Tb = Pose.from_pos_quat(Bd["pos"], Bd["quat"])
    if t == "cuboid":
        return Tb.apply(cuboid_surface_lattice(Bd["size"], Bd["contact_res"]))
    if t == "cylinder":
        return Tb.apply(cylinder_cap_points(Bd["radius"], Bd["length"], *Bd["contact_res"]))
    if t == "abstract" and Bd["contacts"] is not None:
        return np.asarray(Bd["contacts"], dtype=np.float64)
    if t == "sphere":
        return None
    raise NotImplementedError("contact points for body type %r (%s)" % (t, Bd["name"]))


def _primitive_of(Bd):
    t = Bd["type"]
    if t == "cuboid":
        return B.TSIM_P_CUBOID, [0.5 * s for s in Bd["size"]] + [0.0]
    if t == "sphere":
        return B.TSIM_P_SPHERE, [Bd["radius"], 0.0, 0.0, 0.0]
    if t == "cylinder":
        return B.TSIM_P_CYLINDER, [Bd["radius"], 0.5 * Bd["length"], 0.0, 0.0]
    raise NotImplementedError("body type %r cannot be a contact primitive (%s)" % (t, Bd["name"]))


def compile_spec(spec):
    spec = copy.deepcopy(spec)
    joints = spec["joints"]
    nj = len(joints)

    # ---- links: merge fixed joints into their first non-fixed ancestor (or the world)
    link_of_joint = [0] * nj          # link index that joint j's frame is rigidly attached to
    T_link_joint = [None] * nj        # pose of joint j's frame in that link's frame
    links = [{"parent": -1, "joint": -1, "mp": MassProps()}]      # link 0 = world
    dof0 = 0
    for j, J in enumerate(joints):
        par = J["parent"]
        E_pj0 = Pose.from_pos_quat(J["pos"], J["quat"])
        plink = link_of_joint[par] if par >= 0 else 0
        T_pl = (T_link_joint[par] if par >= 0 else Pose()) * E_pj0   # joint-0 frame in the parent LINK frame
        nd = B.JOINT_NDOF[J["type"]]
        jdesc = {"damping": J["damping"], "lim": J["lim"], "lim_stiffness": J["lim_stiffness"]}
        if nd == 0:
            link_of_joint[j] = plink
            T_link_joint[j] = T_pl
        elif J["type"] == "free3d-euler":
            # [CHOICE] free3d-euler = translation (parent frame) followed by intrinsic X-Y-Z Euler rotations, realised as
            # a translational joint and three revolute joints with massless intermediate links: q = (x, y, z, a, b, c),
            # R = Rx(a) Ry(b) Rz(c). No new kernel joint type is needed; for the yaw-only motions the reference's
            # insertion env commands (tactile_insertion_env.py:181-196) Euler and rotation-vector coordinates coincide.
            chain = [("translational", [], 3, T_pl), ("revolute", [[1.0, 0, 0]], 1, Pose()),
                     ("revolute", [[0, 1.0, 0]], 1, Pose()), ("revolute", [[0, 0, 1.0]], 1, Pose())]
            for jt_, axes_, n_, E_ in chain:
                links.append(dict(jdesc, parent=plink, joint=j, E_pj0=E_, dof0=dof0, ndof=n_, mp=MassProps(),
                                  jtype=jt_, axes=axes_))
                plink = len(links) - 1
                dof0 += n_
            link_of_joint[j] = plink
            T_link_joint[j] = Pose()
        elif J["type"] == "free3d-exp":
            # translation (parent frame) followed by a rotation-vector (exponential-coordinate) spherical joint,
            # q = (x, y, z, theta) as examples/RollingBallExp/test_sim_speed.py:54 describes; massless link in between
            for jt_, n_, E_ in (("translational", 3, T_pl), ("spherical-exp", 3, Pose())):
                links.append(dict(jdesc, parent=plink, joint=j, E_pj0=E_, dof0=dof0, ndof=n_, mp=MassProps(),
                                  jtype=jt_, axes=[]))
                plink = len(links) - 1
                dof0 += n_
            link_of_joint[j] = plink
            T_link_joint[j] = Pose()
        else:
            link_of_joint[j] = len(links)
            T_link_joint[j] = Pose()
            links.append(dict(jdesc, parent=plink, joint=j, E_pj0=T_pl, dof0=dof0, ndof=nd, mp=MassProps(),
                              jtype=J["type"], axes=J["axes"]))
            dof0 += nd
        L = links[link_of_joint[j]]
        L["mp"] = L["mp"] + _body_props(J["body"]).transformed(T_link_joint[j])
    nl = len(links) - 1
    nr = dof0
    if nr > 30:
        raise NotImplementedError("more than 30 reduced dofs")
    for i in range(1, nl + 1):
        L = links[i]
        m = ((1 << L["ndof"]) - 1) << L["dof0"]
        L["ancmask"] = m | (links[L["parent"]]["ancmask"] if L["parent"] > 0 else 0)
    links[0]["ancmask"] = 0

    jidx_by_name = {J["name"]: j for j, J in enumerate(joints)}
    body_joint = {J["body"]["name"]: j for j, J in enumerate(joints)}

    def body_link_pose(name):
        """(link, pose of the BODY frame in the link frame, body dict)"""
        if name not in body_joint:
            raise KeyError("unknown body %r" % name)
        j = body_joint[name]
        Bd = joints[j]["body"]
        return link_of_joint[j], T_link_joint[j] * Pose.from_pos_quat(Bd["pos"], Bd["quat"]), Bd

    def body_link_jointpose(name):
        j = body_joint[name]
        return link_of_joint[j], T_link_joint[j], joints[j]["body"]

    # ---- contact pairs + points
    pairs, cpts = [], []

    def add_points(P):
        p0 = sum(len(c) for c in cpts)
        cpts.append(P)
        return p0, len(P)

    for c in spec["contacts"]:
        if c["type"] == "ground":
            if spec["ground"] is None:
                raise ValueError("ground_contact without <ground>")
            la, Tj, Bd = body_link_jointpose(c["body"])
            n = np.asarray(spec["ground"]["normal"], dtype=np.float64)
            n /= np.linalg.norm(n)
            t0 = np.cross(n, [1.0, 0, 0]) if abs(n[0]) < 0.9 else np.cross(n, [0, 1.0, 0])
            t0 /= np.linalg.norm(t0)
            Rg = np.stack([t0, np.cross(n, t0), n], axis=1)      # columns: tangent, tangent, normal
            if Bd["type"] == "sphere":
                # [CHOICE] moving contact point: lowest point of the sphere; the "point" stored is the
                # sphere centre, the radius goes in the shape params.
                _, Tb, _ = body_link_pose(c["body"])
                p0, npt = add_points(np.asarray([Tb.p]))
                shape, flags = [Bd["radius"], 0, 0, 0], 1 | 2
            else:
                p0, npt = add_points(Tj.apply(_body_contact_points(Bd)))
                shape, flags = [0, 0, 0, 0], 1
            pairs.append({"la": la, "lb": 0, "prim": B.TSIM_P_PLANE, "pt0": p0, "npt": npt, "flags": flags,
                          "T": Pose(Rg, spec["ground"]["pos"]), "shape": shape,
                          "k": [c["kn"], c["kt"], c["mu"], c["damping"]],
                          "key": ("ground", c["body"])})
        else:
            la, Tj, Bd = body_link_jointpose(c["general_body"])
            lb, Tb, Bp = body_link_pose(c["primitive_body"])
            prim, shape = _primitive_of(Bp)
            p0, npt = add_points(Tj.apply(_body_contact_points(Bd)))
            pairs.append({"la": la, "lb": lb, "prim": prim, "pt0": p0, "npt": npt, "flags": 1, "T": Tb,
                          "shape": shape, "k": [c["kn"], c["kt"], c["mu"], c["damping"]],
                          "key": (c["general_body"], c["primitive_body"])})

    # ---- tactile sensors
    sensors, taxels, sprims, image_pos = [], [], [], {}
    for s in spec["sensors"]:
        ls, Tj, Bd = body_link_jointpose(s["body"])
        if s["type"] == "rect_array":
            _, Tbody, _ = body_link_pose(s["body"])
            p0, p1 = np.asarray(s["rect_pos0"]), np.asarray(s["rect_pos1"])
            a0, a1 = np.asarray(s["axis0"], dtype=np.float64), np.asarray(s["axis1"], dtype=np.float64)
            a0, a1 = a0 / np.linalg.norm(a0), a1 / np.linalg.norm(a1)
            nrm = np.cross(a1, a0)     # [CHOICE] matches the explicit normals of the abstract spec file
            R_, C_ = s["resolution"]
            e0, e1 = np.dot(p1 - p0, a0), np.dot(p1 - p0, a1)
            tl, img = [], []
            for i in range(R_):
                for jx in range(C_):
                    pos = p0 + a0 * (e0 * i / max(R_ - 1, 1)) + a1 * (e1 * jx / max(C_ - 1, 1))
                    tl.append(np.concatenate([Tbody.apply(pos), Tbody.rotate(a0), Tbody.rotate(a1), Tbody.rotate(nrm)]))
                    img.append((i, jx))
            rows, cols = R_, C_
        else:
            _, Tbody, _ = body_link_pose(s["body"])          # sensor pos/quat are relative to the body frame
            Ts = Tbody * Pose.from_pos_quat(s["pos"], s["quat"])
            tl, img = [], []
            for t in s["taxels"]:
                tl.append(np.concatenate([Ts.apply(t["pos"]), Ts.rotate(t["axis0"]), Ts.rotate(t["axis1"]),
                                          Ts.rotate(t["normal"])]))
                img.append(tuple(t["img"]))
            rows = max(r for r, _ in img) + 1
            cols = max(c_ for _, c_ in img) + 1
        # [CHOICE] a sensor's taxels are tested against the primitive of every general_primitive pair whose
        # general body is the sensor's body.
        plist = [pi for pi, p in enumerate(pairs) if p["key"][0] == s["body"] and p["key"][0] != "ground"]
        sensors.append({"link": ls, "tax0": sum(len(t) for t in taxels), "ntax": len(tl), "sprim0": len(sprims),
                        "nsprim": len(plist), "rows": rows, "cols": cols,
                        "k": [s["kn"], s["kt"], s["mu"], s["damping"]], "name": s["name"]})
        sprims.extend(plist)
        taxels.append(np.asarray(tl, dtype=np.float64).reshape(-1, 12))
        image_pos[s["name"]] = img

    # ---- motors -> one record per entry of u
    motors = []
    for m in spec["motors"]:
        j = jidx_by_name[m["joint"]]
        if B.JOINT_NDOF[joints[j]["type"]] == 0:
            raise ValueError("motor on fixed joint %r" % m["joint"])
        for L in links[1:]:
            if L["joint"] != j:
                continue
            for k in range(L["ndof"]):
                motors.append({"dof": L["dof0"] + k, "ctrl": 0 if m["ctrl"] == "force" else 1,
                               "f": [m["ctrl_range"][0], m["ctrl_range"][1], m["P"], m["D"]]})
    nu = len(motors)

    # ---- variables
    variables = []
    for e in spec["endeffectors"]:
        j = jidx_by_name[e["joint"]]
        variables.append({"link": link_of_joint[j], "pos": T_link_joint[j].apply(e["pos"]), "name": e["name"]})

    # ---- emit
    cpt = np.concatenate(cpts, axis=0) if cpts else np.zeros((0, 3))
    tax = np.concatenate(taxels, axis=0) if taxels else np.zeros((0, 12))
    ncpt, ntax = len(cpt), len(tax)

    Ih = np.zeros(B.TSIM_IH_SIZE, dtype=np.int64)
    Il, Fl = [], []

    def section(irecs, frecs, isz, fsz):
        Io, Fo = B.TSIM_IH_SIZE + sum(len(a) for a in Il), B.TSIM_FH_SIZE + sum(len(a) for a in Fl)
        ia = np.zeros(len(irecs) * isz, dtype=np.int64)
        fa = np.zeros(len(frecs) * fsz, dtype=np.float64)
        for n, r in enumerate(irecs):
            ia[n * isz:n * isz + len(r)] = r
        for n, r in enumerate(frecs):
            fa[n * fsz:n * fsz + len(r)] = r
        Il.append(ia)
        Fl.append(fa)
        return Io, Fo

    lrec_i, lrec_f = [], []
    for i in range(1, nl + 1):
        L = links[i]
        lrec_i.append([L["parent"], B.JOINT_TYPES[L["jtype"]], L["dof0"], L["ndof"], L["ancmask"]])
        axes = np.zeros((3, 3))
        for k, a in enumerate(L["axes"]):
            a = np.asarray(a, dtype=np.float64)
            axes[k] = a / np.linalg.norm(a)
        mp = L["mp"]
        Ic = mp.Ic
        lrec_f.append(np.concatenate([L["E_pj0"].R.reshape(-1), L["E_pj0"].p, axes.reshape(-1), [mp.m], mp.c,
                                      [Ic[0, 0], Ic[1, 1], Ic[2, 2], Ic[0, 1], Ic[0, 2], Ic[1, 2]]]))
    Ih[B.TSIM_IH_OFF_LINK], Ih[B.TSIM_IH_FOFF_LINK] = section(lrec_i, lrec_f, B.TSIM_LI_SIZE, B.TSIM_LF_SIZE)

    drec_i, drec_f = [], []
    for i in range(1, nl + 1):
        L = links[i]
        for k in range(L["ndof"]):
            drec_i.append([i])
            if L["lim"] is not None and L["lim_stiffness"] > 0:
                drec_f.append([L["damping"], L["lim"][0], L["lim"][1], L["lim_stiffness"]])
            else:
                drec_f.append([L["damping"], 0.0, 0.0, 0.0])
    Ih[B.TSIM_IH_OFF_DOF], Ih[B.TSIM_IH_FOFF_DOF] = section(drec_i, drec_f, B.TSIM_DI_SIZE, B.TSIM_DF_SIZE)
    Ih[B.TSIM_IH_OFF_MOTOR], Ih[B.TSIM_IH_FOFF_MOTOR] = section(
        [[m["dof"], m["ctrl"]] for m in motors], [m["f"] for m in motors], B.TSIM_MI_SIZE, B.TSIM_MF_SIZE)
    Ih[B.TSIM_IH_OFF_VAR], Ih[B.TSIM_IH_FOFF_VAR] = section(
        [[v["link"]] for v in variables], [v["pos"] for v in variables], B.TSIM_VI_SIZE, B.TSIM_VF_SIZE)
    Ih[B.TSIM_IH_OFF_PAIR], Ih[B.TSIM_IH_FOFF_PAIR] = section(
        [[p["la"], p["lb"], p["prim"], p["pt0"], p["npt"], p["flags"]] for p in pairs],
        [np.concatenate([p["T"].R.reshape(-1), p["T"].p, p["shape"], p["k"]]) for p in pairs],
        B.TSIM_PI_SIZE, B.TSIM_PF_SIZE)
    Ih[B.TSIM_IH_OFF_SENSOR], Ih[B.TSIM_IH_FOFF_SENSOR] = section(
        [[s["link"], s["tax0"], s["ntax"], s["sprim0"], s["nsprim"], s["rows"], s["cols"]] for s in sensors],
        [s["k"] for s in sensors], B.TSIM_SI_SIZE, B.TSIM_SF_SIZE)
    Ih[B.TSIM_IH_OFF_SPRIM], _ = section([[p] for p in sprims], [], 1, 1)
    Ih[B.TSIM_IH_FOFF_CPT] = B.TSIM_FH_SIZE + sum(len(a) for a in Fl)
    Fl.append(cpt.T.reshape(-1).copy())
    Ih[B.TSIM_IH_FOFF_TAXEL] = B.TSIM_FH_SIZE + sum(len(a) for a in Fl)
    Fl.append(tax.T.reshape(-1).copy())

    integ = {"BDF1": 1, "BDF2": 2}.get(spec["options"]["integrator"])
    if integ is None:
        raise NotImplementedError("integrator %r" % spec["options"]["integrator"])
    Ih[B.TSIM_IH_MAGIC], Ih[B.TSIM_IH_VERSION] = B.TSIM_MAGIC, B.TSIM_VERSION
    Ih[B.TSIM_IH_NL], Ih[B.TSIM_IH_NR], Ih[B.TSIM_IH_NU], Ih[B.TSIM_IH_NVAR] = nl, nr, nu, len(variables)
    Ih[B.TSIM_IH_NPAIR], Ih[B.TSIM_IH_NCPT], Ih[B.TSIM_IH_NSENSOR] = len(pairs), ncpt, len(sensors)
    Ih[B.TSIM_IH_NTAXEL], Ih[B.TSIM_IH_NSPRIM] = ntax, len(sprims)
    Ih[B.TSIM_IH_INTEGRATOR] = integ
    Ih[B.TSIM_IH_MAX_ITER], Ih[B.TSIM_IH_MAX_LS] = spec["options"]["max_iter"], spec["options"]["max_ls"]
    Ih[B.TSIM_IH_NDOF_TACTILE] = 3 * ntax
    Fh = np.zeros(B.TSIM_FH_SIZE)
    Fh[B.TSIM_FH_H] = spec["options"]["h"]
    Fh[B.TSIM_FH_GX:B.TSIM_FH_GX + 3] = spec["options"]["gravity"]
    Fh[B.TSIM_FH_TOL] = spec["options"]["tol"]
    I = np.concatenate([Ih] + Il)
    F = np.concatenate([Fh] + Fl)
    I[B.TSIM_IH_NI], I[B.TSIM_IH_NF] = len(I), len(F)

    meta = {
        "joint_names": [J["name"] for J in joints],
        "link_of_joint": link_of_joint,
        "dof_of_joint": {J["name"]: (min(L["dof0"] for L in links[1:] if L["joint"] == j), B.JOINT_NDOF[J["type"]])
                         for j, J in enumerate(joints) if B.JOINT_NDOF[J["type"]] > 0},
        "pair_keys": [p["key"] for p in pairs],
        "sensor_names": [s["name"] for s in sensors],
        "sensor_taxels": [(s["tax0"], s["ntax"], s["rows"], s["cols"]) for s in sensors],
        "image_pos": image_pos,
        "variable_names": [v["name"] for v in variables],
        "link_mass": [links[i]["mp"].m for i in range(1, nl + 1)],
    }
    return CompiledModel(spec, I.astype(np.int32), F.astype(np.float64), meta)


# =============================================================================== spec edits (update_*)
def _find_body(spec, name):
    for J in spec["joints"]:
        if J["body"]["name"] == name:
            return J["body"]
    raise KeyError("unknown body %r" % name)


def _find_joint(spec, name):
    for J in spec["joints"]:
        if J["name"] == name:
            return J
    raise KeyError("unknown joint %r" % name)


def edit_spec(spec, what, name, *args, **kw):
    """In-place edits behind the binding's update_* methods."""
    if what == "joint_damping":
        _find_joint(spec, name)["damping"] = float(args[0])
    elif what == "joint_location":
        _find_joint(spec, name)["pos"] = [float(x) for x in args[0]]
    elif what == "body_density":
        _find_body(spec, name)["density"] = float(args[0])
    elif what == "body_size":
        Bd, v = _find_body(spec, name), [float(x) for x in np.asarray(args[0]).reshape(-1)]
        if Bd["type"] == "cuboid":
            Bd["size"] = v[:3]
        elif Bd["type"] == "sphere":
            Bd["radius"] = v[0]
        elif Bd["type"] == "cylinder":          # (length, radius) order as passed by dclaw_rotate_env.py:175
            Bd["length"], Bd["radius"] = v[0], v[1]
        else:
            raise NotImplementedError("update_body_size on %s body" % Bd["type"])
    elif what == "endeffector_position":
        for e in spec["endeffectors"]:
            if e["name"] == name:
                e["pos"] = [float(x) for x in args[0]]
                break
        else:
            raise KeyError("unknown endeffector %r" % name)
    elif what == "contact_parameters":
        gb, pb = name
        hit = False
        for c in spec["contacts"]:
            if c["type"] == "general_primitive" and c["general_body"] == gb and c["primitive_body"] == pb:
                for k in ("kn", "kt", "mu", "damping"):
                    if kw.get(k) is not None:
                        c[k] = float(kw[k])
                hit = True
        if not hit:
            raise KeyError("no general_primitive_contact %s -> %s" % (gb, pb))
    elif what == "tactile_parameters":
        hit = False
        for s in spec["sensors"]:
            if s["body"] == name or s["name"] == name:
                for k in ("kn", "kt", "mu", "damping"):
                    if kw.get(k) is not None:
                        s[k] = float(kw[k])
                hit = True
        if not hit:
            raise KeyError("no tactile sensor on %r" % name)
    elif what == "virtual_object":
        v = np.asarray(args[0], dtype=np.float64).reshape(-1)
        spec["virtual"].setdefault(name, {})
        spec["virtual"][name]["pos"], spec["virtual"][name]["quat"] = v[:3].tolist(), v[3:7].tolist()
    else:
        raise KeyError(what)
