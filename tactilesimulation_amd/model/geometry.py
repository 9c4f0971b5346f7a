"""Host-side geometry helpers for the model compiler: SE(3), mass properties, point sampling.

Everything here is float64 numpy and runs once per model load; none of it is on the hot path.
Reference counterparts live in the (absent) DiffRedMax C++ — the choices made here are this
build's own and are documented in DESIGN.md §Model.
"""
import numpy as np


# ----------------------------------------------------------------------------- SE(3)
def quat_to_R(q):
    """Rotation matrix of a (w, x, y, z) quaternion (the XML convention, cf. the reference's
    envs/tactile_push_env.py:149-150). The quaternion is normalised first (the XMLs carry
    3-digit values such as 0.707)."""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q)
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class Pose:
    """x_parent = R x_child + p."""
    __slots__ = ("R", "p")

    def __init__(self, R=None, p=None):
        self.R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64).reshape(3, 3)
        self.p = np.zeros(3) if p is None else np.asarray(p, dtype=np.float64).reshape(3)

    @staticmethod
    def from_pos_quat(pos, quat):
        return Pose(quat_to_R(quat), pos)

    def __mul__(self, o):
        return Pose(self.R @ o.R, self.R @ o.p + self.p)

    def inv(self):
        return Pose(self.R.T, -self.R.T @ self.p)

    def apply(self, x):
        x = np.asarray(x, dtype=np.float64)
        return x @ self.R.T + self.p

    def rotate(self, v):
        return np.asarray(v, dtype=np.float64) @ self.R.T


# ----------------------------------------------------------------------------- mass properties
class MassProps:
    """mass, centre of mass c and rotational inertia Ic (3x3, about c), all in one frame."""
    __slots__ = ("m", "c", "Ic")

    def __init__(self, m=0.0, c=None, Ic=None):
        self.m = float(m)
        self.c = np.zeros(3) if c is None else np.asarray(c, dtype=np.float64)
        self.Ic = np.zeros((3, 3)) if Ic is None else np.asarray(Ic, dtype=np.float64)

    def transformed(self, T):
        return MassProps(self.m, T.apply(self.c), T.R @ self.Ic @ T.R.T)

    def scaled(self, s):
        return MassProps(self.m * s, self.c, self.Ic * s)

    def __add__(self, o):
        m = self.m + o.m
        if m == 0.0:
            return MassProps()
        c = (self.m * self.c + o.m * o.c) / m

        def shift(mp):
            d = mp.c - c
            return mp.Ic + mp.m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        return MassProps(m, c, shift(self) + shift(o))


def cuboid_props(size, density):
    sx, sy, sz = size
    m = density * sx * sy * sz
    return MassProps(m, np.zeros(3), np.diag([m / 12 * (sy * sy + sz * sz), m / 12 * (sx * sx + sz * sz),
                                              m / 12 * (sx * sx + sy * sy)]))


def sphere_props(r, density):
    m = density * 4.0 / 3.0 * np.pi * r ** 3
    return MassProps(m, np.zeros(3), np.eye(3) * (0.4 * m * r * r))


def cylinder_props(r, length, density):
    """Solid cylinder, axis = local z."""
    m = density * np.pi * r * r * length
    ixy = m * (3 * r * r + length * length) / 12.0
    return MassProps(m, np.zeros(3), np.diag([ixy, ixy, 0.5 * m * r * r]))


def load_obj(path):
    """Minimal OBJ reader: vertices + triangulated faces."""
    V, Fc = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                V.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(t.split("/")[0]) for t in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(V) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    Fc.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(V, dtype=np.float64), np.asarray(Fc, dtype=np.int64)


def mesh_props(V, Fc):
    """Unit-density mass properties of a closed triangle mesh by signed tetrahedra against the
    origin (divergence theorem). Known-answer values for the reference's meshes are in
    SURVEY.md Appendix D (wsg50_base 4.910e-4 m^3, guide_left 9.129e-6 m^3, gelslim_left 2.374e-5 m^3)."""
    a, b, c = V[Fc[:, 0]], V[Fc[:, 1]], V[Fc[:, 2]]
    det = np.einsum("ij,ij->i", a, np.cross(b, c))          # 6 * signed tet volume
    vol = det.sum() / 6.0
    com = ((a + b + c) * det[:, None]).sum(0) / (24.0 * vol)
    # second moments  integral x_i x_j dV  over each tet (origin, a, b, c)
    S = np.zeros((3, 3))
    for P, Q in ((a, a), (b, b), (c, c)):
        S += 2.0 * np.einsum("n,ni,nj->ij", det, P, Q)
    for P, Q in ((a, b), (a, c), (b, c)):
        S += np.einsum("n,ni,nj->ij", det, P, Q) + np.einsum("n,ni,nj->ij", det, Q, P)
    S /= 120.0
    I0 = np.trace(S) * np.eye(3) - S                         # inertia about the origin
    Ic = I0 - vol * (np.dot(com, com) * np.eye(3) - np.outer(com, com))
    if vol < 0:                                              # inward-facing winding
        vol, Ic = -vol, -Ic
    return MassProps(vol, com, Ic)


# ----------------------------------------------------------------------------- point sampling
def cuboid_surface_lattice(size, res):
    """Lattice of res = (nx, ny, nz) points per axis spanning the cuboid; only the points on the
    surface are kept. res = (2,2,2) gives the 8 corners."""
    res = [max(int(r), 2) for r in res]
    axes = [np.linspace(-0.5 * s, 0.5 * s, n) for s, n in zip(size, res)]
    pts = []
    for i in range(res[0]):
        for j in range(res[1]):
            for k in range(res[2]):
                if i in (0, res[0] - 1) or j in (0, res[1] - 1) or k in (0, res[2] - 1):
                    pts.append([axes[0][i], axes[1][j], axes[2][k]])
    return np.asarray(pts, dtype=np.float64)


def cylinder_cap_points(radius, length, n_angle, n_radius):
    """Both end caps of a z-axis cylinder: centre point + n_radius rings of n_angle points."""
    pts = []
    for z in (0.5 * length, -0.5 * length):
        pts.append([0.0, 0.0, z])
        for ir in range(1, n_radius + 1):
            r = radius * ir / n_radius
            for ia in range(n_angle):
                th = 2.0 * np.pi * ia / n_angle
                pts.append([r * np.cos(th), r * np.sin(th), z])
    return np.asarray(pts, dtype=np.float64)
