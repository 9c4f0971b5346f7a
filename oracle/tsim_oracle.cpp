// tsim_oracle.cpp — fp64 CPU ORACLE for the batched tactile-simulation hot path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing in the product (tactilesimulation_amd/) may import,
// link or execute this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// *** PARITY UNPINNED. ***  The reference's arithmetic for this path lives in an un-vendored git
// submodule (externals/DiffHand -> python module redmax_py, .gitmodules:1-3; pinned SHA unknown, no
// network).  This file is therefore a restatement of the PUBLISHED formulation (RedMax reduced
// coordinates + DiffHand penalty contact, recalled from the papers — SURVEY.md §8c [BACKGROUND]) anchored
// on the reference's own call sites:
//   forward(num_steps)            envs/redmax_torch_functions.py:49,132
//   get_q / get_variables / get_tactile_force_vector     :51-57,134-136
//   backward_steps(n) / backward()                        :92,167  (gradient seeding :151-165)
//   model semantics               envs/assets/pusher/pusher.xml:2-67
// It is pinned only by analytic known-answer tests and finite-difference self-consistency
// (tests/test_oracle_physics.py), the same *method* the reference uses (algorithms/gd.py:407-468).
//
// Formulation:
//   r(q, qd, qdd, u) = ID(q, qd, qdd) - Q_contact(q, qd) - tau_joint(q, qd, u)          (RNEA, world frame)
//   BDF1:  q1 solves  g(q1) = h^2 r(q1, (q1-q0)/h, (q1-q0-h qd0)/h^2, u) = 0
//          == M(q1)(q1 - q0 - h qd0) - h^2 f_r(q1, qd1)        (RedMax, pusher.xml:2 integrator="BDF1")
//   Newton with backtracking line search on ||g||_2, tol / max_iter / max_ls from <solver_option>.
//   All Jacobians by forward-mode AD (dual numbers) — exact, no hand-derived derivative anywhere here.
//   Adjoint: (dg/dq1)^T z = lam_q + lam_v/h ; dL/du = -(dg/du)^T z ; lam_q0 = -(dg/dq0)^T z - lam_v/h ;
//            lam_v0 = -(dg/dqd0)^T z.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../include/tsim_blob.h"

#define MAXL 24
#define MAXR 32
#define NDMAX 32

// ------------------------------------------------------------------------------------------ dual numbers
static thread_local int g_nd = 0;   // active tangent directions
struct Dual {
  double v; double d[NDMAX];
  Dual() : v(0) { for (int i = 0; i < g_nd; ++i) d[i] = 0; }
  Dual(double x) : v(x) { for (int i = 0; i < g_nd; ++i) d[i] = 0; }
};
static inline Dual operator+(const Dual& a, const Dual& b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < g_nd; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static inline Dual operator-(const Dual& a, const Dual& b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < g_nd; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
static inline Dual operator-(const Dual& a) { Dual r; r.v = -a.v; for (int i = 0; i < g_nd; ++i) r.d[i] = -a.d[i]; return r; }
static inline Dual operator*(const Dual& a, const Dual& b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < g_nd; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
static inline Dual operator/(const Dual& a, const Dual& b) { Dual r; double iv = 1.0 / b.v; r.v = a.v * iv; for (int i = 0; i < g_nd; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * iv; return r; }
static inline Dual& operator+=(Dual& a, const Dual& b) { a = a + b; return a; }
static inline Dual& operator-=(Dual& a, const Dual& b) { a = a - b; return a; }
static inline Dual sqrt(const Dual& a) { Dual r; r.v = std::sqrt(a.v); double k = 0.5 / r.v; for (int i = 0; i < g_nd; ++i) r.d[i] = a.d[i] * k; return r; }
static inline Dual sin(const Dual& a) { Dual r; r.v = std::sin(a.v); double c = std::cos(a.v); for (int i = 0; i < g_nd; ++i) r.d[i] = a.d[i] * c; return r; }
static inline Dual cos(const Dual& a) { Dual r; r.v = std::cos(a.v); double s = -std::sin(a.v); for (int i = 0; i < g_nd; ++i) r.d[i] = a.d[i] * s; return r; }
static inline double val(double x) { return x; }
static inline double val(const Dual& x) { return x.v; }

// one-direction dual over an arbitrary scalar (used only for d/ds J_l(theta + s thetadot) inside the exponential joint)
template <class T> struct D1 { T v, d; D1() {} D1(double x) : v(x), d(0.0) {} D1(const T& a, const T& b) : v(a), d(b) {} };
template <class T> static inline D1<T> operator+(const D1<T>& a, const D1<T>& b) { return D1<T>(a.v + b.v, a.d + b.d); }
template <class T> static inline D1<T> operator-(const D1<T>& a, const D1<T>& b) { return D1<T>(a.v - b.v, a.d - b.d); }
template <class T> static inline D1<T> operator*(const D1<T>& a, const D1<T>& b) { return D1<T>(a.v * b.v, a.d * b.v + a.v * b.d); }
template <class T> static inline D1<T> operator/(const D1<T>& a, const D1<T>& b) { T q = a.v / b.v; return D1<T>(q, (a.d - q * b.d) / b.v); }
template <class T> static inline D1<T> sqrt(const D1<T>& a) { T s = sqrt(a.v); return D1<T>(s, a.d / (s + s)); }
template <class T> static inline D1<T> sin(const D1<T>& a) { return D1<T>(sin(a.v), a.d * cos(a.v)); }
template <class T> static inline D1<T> cos(const D1<T>& a) { return D1<T>(cos(a.v), T(0.0) - a.d * sin(a.v)); }
template <class T> static inline double val(const D1<T>& x) { return val(x.v); }
using std::sqrt; using std::sin; using std::cos;

// ------------------------------------------------------------------------------------------ small linear algebra
template <class T> struct V3 { T x, y, z; };
template <class T> static inline V3<T> mk(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <class T> static inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return mk<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> static inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return mk<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> static inline V3<T> operator*(const V3<T>& a, const T& s) { return mk<T>(a.x * s, a.y * s, a.z * s); }
template <class T> static inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> static inline V3<T> cross(const V3<T>& a, const V3<T>& b) { return mk<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class T> struct M3 { T m[9]; };   // row-major
template <class T> static inline V3<T> mul(const M3<T>& A, const V3<T>& v) { return mk<T>(A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z, A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z); }
template <class T> static inline V3<T> mulT(const M3<T>& A, const V3<T>& v) { return mk<T>(A.m[0] * v.x + A.m[3] * v.y + A.m[6] * v.z, A.m[1] * v.x + A.m[4] * v.y + A.m[7] * v.z, A.m[2] * v.x + A.m[5] * v.y + A.m[8] * v.z); }
template <class T> static inline M3<T> mul(const M3<T>& A, const M3<T>& B) { M3<T> C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j]; return C; }
template <class T> static inline M3<T> cmat(const double* r) { M3<T> A; for (int i = 0; i < 9; ++i) A.m[i] = T(r[i]); return A; }
template <class T> static inline V3<T> cvec(const double* r) { return mk<T>(T(r[0]), T(r[1]), T(r[2])); }
// Rodrigues rotation about a unit axis
template <class T> static inline M3<T> rot_axis(const double* a, const T& th) {
  T c = cos(th), s = sin(th), t = T(1.0) - c; M3<T> R;
  R.m[0] = t * T(a[0] * a[0]) + c;          R.m[1] = t * T(a[0] * a[1]) - s * T(a[2]); R.m[2] = t * T(a[0] * a[2]) + s * T(a[1]);
  R.m[3] = t * T(a[0] * a[1]) + s * T(a[2]); R.m[4] = t * T(a[1] * a[1]) + c;          R.m[5] = t * T(a[1] * a[2]) - s * T(a[0]);
  R.m[6] = t * T(a[0] * a[2]) - s * T(a[1]); R.m[7] = t * T(a[1] * a[2]) + s * T(a[0]); R.m[8] = t * T(a[2] * a[2]) + c;
  return R;
}

// exp([th]) and the left Jacobian J_l(th) of SO(3):  omega_spatial = J_l(th) thdot.   U is any scalar with + - * / sqrt sin cos.
template <class U> static inline void so3_coeffs(const U& x, const U& y, const U& z, U& s1, U& s2, U& c2) {
  U p2 = x * x + y * y + z * z;
  if (val(p2) < 1e-8) {   // series in phi^2 (differentiable through the duals)
    s1 = U(1.0) - p2 * U(1.0 / 6) + p2 * p2 * U(1.0 / 120);
    s2 = U(0.5) - p2 * U(1.0 / 24) + p2 * p2 * U(1.0 / 720);
    c2 = U(1.0 / 6) - p2 * U(1.0 / 120) + p2 * p2 * U(1.0 / 5040);
  } else {
    U p = sqrt(p2);
    s1 = sin(p) / p; s2 = (U(1.0) - cos(p)) / p2; c2 = (p - sin(p)) / (p2 * p);
  }
}
template <class U> static inline void so3_mat(const U& x, const U& y, const U& z, const U& a, const U& b, U* M) {
  // M = I + a [th]x + b [th]x^2
  M[0] = U(1.0) - b * (y * y + z * z); M[1] = b * x * y - a * z;          M[2] = b * x * z + a * y;
  M[3] = b * x * y + a * z;          M[4] = U(1.0) - b * (x * x + z * z); M[5] = b * y * z - a * x;
  M[6] = b * x * z - a * y;          M[7] = b * y * z + a * x;          M[8] = U(1.0) - b * (x * x + y * y);
}

// ------------------------------------------------------------------------------------------ model view
struct Model {
  std::vector<int> I; std::vector<double> F;
  int nl, nr, nu, nvar, npair, ncpt, nsensor, ntax, integrator, max_iter, max_ls;
  double h, grav[3], tol;
  const int* li(int i) const { return &I[I[TSIM_IH_OFF_LINK] + (i - 1) * TSIM_LI_SIZE]; }
  const double* lf(int i) const { return &F[I[TSIM_IH_FOFF_LINK] + (i - 1) * TSIM_LF_SIZE]; }
  const double* df(int k) const { return &F[I[TSIM_IH_FOFF_DOF] + k * TSIM_DF_SIZE]; }
  const int* mi(int k) const { return &I[I[TSIM_IH_OFF_MOTOR] + k * TSIM_MI_SIZE]; }
  const double* mf(int k) const { return &F[I[TSIM_IH_FOFF_MOTOR] + k * TSIM_MF_SIZE]; }
  const int* vi(int k) const { return &I[I[TSIM_IH_OFF_VAR] + k * TSIM_VI_SIZE]; }
  const double* vf(int k) const { return &F[I[TSIM_IH_FOFF_VAR] + k * TSIM_VF_SIZE]; }
  const int* pi(int k) const { return &I[I[TSIM_IH_OFF_PAIR] + k * TSIM_PI_SIZE]; }
  const double* pf(int k) const { return &F[I[TSIM_IH_FOFF_PAIR] + k * TSIM_PF_SIZE]; }
  const int* si(int k) const { return &I[I[TSIM_IH_OFF_SENSOR] + k * TSIM_SI_SIZE]; }
  const double* sf(int k) const { return &F[I[TSIM_IH_FOFF_SENSOR] + k * TSIM_SF_SIZE]; }
  int sprim(int j) const { return I[I[TSIM_IH_OFF_SPRIM] + j]; }
  double cpt(int c, int i) const { return F[I[TSIM_IH_FOFF_CPT] + c * ncpt + i]; }
  double tax(int c, int i) const { return F[I[TSIM_IH_FOFF_TAXEL] + c * ntax + i]; }
};

template <class T> struct Link {
  M3<T> R; V3<T> p;       // pose
  V3<T> w, v;             // spatial velocity (world frame, about world origin)
  V3<T> aw, av;           // spatial acceleration
  V3<T> fn, ff;           // net wrench the joint must transmit: inertial - external  (moment about origin; force)
};

// ------------------------------------------------------------------------------------------ contact law
// DiffHand penalty model: d<0: fn = (-kn + kd ddot) d ; ft = -min(kt|vt|, mu|fn|) vt/|vt|.
// Returns the world-frame force on the point fixed to link A at x_w (and -F on link B).
// branch (optional): the smooth piece of the law the point is on: bit 0 = sticking, bits 1.. = face of the primitive
// (cuboid: 2 axis + (negative side); cylinder: 0 side, 1 / 2 caps; plane, sphere: 0) — see orc_signature.
template <class T>
static bool contact_force(int prim, const double* shape, const double* k, const M3<T>& RP, const V3<T>& pP,
                          const V3<T>& xw, const V3<T>& vrel_w, V3<T>& Fw, int* branch = nullptr) {
  V3<T> x = mulT(RP, xw - pP);
  T d; V3<T> n;
  int face = 0;
  if (prim == TSIM_P_PLANE) { d = x.z; n = mk<T>(T(0.0), T(0.0), T(1.0)); }
  else if (prim == TSIM_P_CUBOID) {
    double ex = std::fabs(val(x.x)) - shape[0], ey = std::fabs(val(x.y)) - shape[1], ez = std::fabs(val(x.z)) - shape[2];
    if (ex >= ey && ex >= ez) { double s = val(x.x) >= 0 ? 1.0 : -1.0; d = x.x * T(s) - T(shape[0]); n = mk<T>(T(s), T(0.0), T(0.0)); face = s > 0 ? 0 : 1; }
    else if (ey >= ez)        { double s = val(x.y) >= 0 ? 1.0 : -1.0; d = x.y * T(s) - T(shape[1]); n = mk<T>(T(0.0), T(s), T(0.0)); face = s > 0 ? 2 : 3; }
    else                      { double s = val(x.z) >= 0 ? 1.0 : -1.0; d = x.z * T(s) - T(shape[2]); n = mk<T>(T(0.0), T(0.0), T(s)); face = s > 0 ? 4 : 5; }
  } else if (prim == TSIM_P_SPHERE) {
    T r2 = dot(x, x);
    if (val(r2) < 1e-24) return false;
    T r = sqrt(r2); d = r - T(shape[0]); n = x * (T(1.0) / r);
  } else {  // cylinder, axis z
    T rho2 = x.x * x.x + x.y * x.y;
    double rho = std::sqrt(val(rho2));
    double dr = rho - shape[0], dz = std::fabs(val(x.z)) - shape[1];
    if (dr > dz && rho > 1e-12) { T rr = sqrt(rho2); d = rr - T(shape[0]); T ir = T(1.0) / rr; n = mk<T>(x.x * ir, x.y * ir, T(0.0)); }
    else { double s = val(x.z) >= 0 ? 1.0 : -1.0; d = x.z * T(s) - T(shape[1]); n = mk<T>(T(0.0), T(0.0), T(s)); face = s > 0 ? 1 : 2; }
  }
  if (!(val(d) < 0.0)) return false;
  V3<T> xd = mulT(RP, vrel_w);
  T ddot = dot(n, xd);
  T fn = (T(-k[0]) + T(k[3]) * ddot) * d;
  V3<T> vt = xd - n * ddot;
  T vt2 = dot(vt, vt);
  V3<T> F = n * fn;
  double vtn = std::sqrt(val(vt2));
  double fnabs = std::fabs(val(fn));
  const bool stick = k[1] * vtn <= k[2] * fnabs || vtn < 1e-14;
  if (branch) *branch = (stick ? 1 : 0) | (face << 1);
  if (stick) {
    F = F - vt * T(k[1]);                                  // "sticking": viscous
  } else {
    T s = (val(fn) >= 0 ? fn : -fn) * T(k[2]) / sqrt(vt2);  // sliding: Coulomb
    F = F - vt * s;
  }
  Fw = mul(RP, F);
  return true;
}

// world pose of a pair's primitive frame and the spatial velocity of link B
template <class T> static inline void prim_pose(const Model& m, int pk, const Link<T>* L, M3<T>& RP, V3<T>& pP) {
  const int* pi = m.pi(pk); const double* pf = m.pf(pk);
  const Link<T>& B = L[pi[TSIM_PI_LINKB]];
  RP = mul(B.R, cmat<T>(pf + TSIM_PF_R));
  pP = mul(B.R, cvec<T>(pf + TSIM_PF_P)) + B.p;
}

// ------------------------------------------------------------------------------------------ kinematics + RNEA residual
template <class T>
static void kinematics(const Model& m, const T* q, const T* qd, const T* qdd, Link<T>* L, V3<T>* Ww, V3<T>* Wv, bool dyn) {
  Link<T>& W0 = L[0];
  for (int i = 0; i < 9; ++i) W0.R.m[i] = T(i % 4 == 0 ? 1.0 : 0.0);
  W0.p = mk<T>(T(0.0), T(0.0), T(0.0)); W0.w = W0.p; W0.v = W0.p; W0.aw = W0.p;
  W0.av = mk<T>(T(-m.grav[0]), T(-m.grav[1]), T(-m.grav[2]));   // gravity as base acceleration
  W0.fn = W0.p; W0.ff = W0.p;
  for (int i = 1; i <= m.nl; ++i) {
    const int* li = m.li(i); const double* lf = m.lf(i);
    const Link<T>& P = L[li[TSIM_LI_PARENT]];
    Link<T>& X = L[i];
    int k0 = li[TSIM_LI_DOF0], nd = li[TSIM_LI_NDOF], jt = li[TSIM_LI_JTYPE];
    M3<T> R0 = mul(P.R, cmat<T>(lf + TSIM_LF_R));           // joint-0 frame in the world
    V3<T> p0 = mul(P.R, cvec<T>(lf + TSIM_LF_P)) + P.p;
    const double* ax = lf + TSIM_LF_AXES;
    V3<T> extra_b = mk<T>(T(0.0), T(0.0), T(0.0)); bool has_b = false;
    // joint motion + world-frame twist columns  W_k = Ad(E_0i) S_k
    if (jt == TSIM_J_REVOLUTE) {
      X.R = mul(R0, rot_axis<T>(ax, q[k0])); X.p = p0;
      V3<T> a = mul(X.R, cvec<T>(ax));
      Ww[k0] = a; Wv[k0] = cross(X.p, a);
    } else if (jt == TSIM_J_PRISMATIC || jt == TSIM_J_PLANAR || jt == TSIM_J_TRANSLATIONAL) {
      X.R = R0; X.p = p0;
      for (int k = 0; k < nd; ++k) {
        double e[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
        V3<T> a = mul(R0, cvec<T>(jt == TSIM_J_TRANSLATIONAL ? e : ax + 3 * k));
        X.p = X.p + a * q[k0 + k];
        Ww[k0 + k] = mk<T>(T(0.0), T(0.0), T(0.0)); Wv[k0 + k] = a;
      }
    } else if (jt == TSIM_J_SPHERICAL_EXP) {
      // R = R0 exp([th]);  world angular columns a_m = R0 J_l(th) e_m;  extra acceleration term b = R0 (d/dt J_l) thdot
      T s1, s2, c2, E[9], Jl[9];
      so3_coeffs<T>(q[k0], q[k0 + 1], q[k0 + 2], s1, s2, c2);
      so3_mat<T>(q[k0], q[k0 + 1], q[k0 + 2], s1, s2, E);
      so3_mat<T>(q[k0], q[k0 + 1], q[k0 + 2], s2, c2, Jl);
      M3<T> Em; for (int e = 0; e < 9; ++e) Em.m[e] = E[e];
      X.R = mul(R0, Em); X.p = p0;
      for (int mcol = 0; mcol < 3; ++mcol) {
        V3<T> a = mul(R0, mk<T>(Jl[mcol], Jl[3 + mcol], Jl[6 + mcol]));
        Ww[k0 + mcol] = a; Wv[k0 + mcol] = cross(X.p, a);
      }
      typedef D1<T> U;
      U t0(q[k0], qd[k0]), t1(q[k0 + 1], qd[k0 + 1]), t2(q[k0 + 2], qd[k0 + 2]), u1, u2, uc, JU[9];
      so3_coeffs<U>(t0, t1, t2, u1, u2, uc);
      so3_mat<U>(t0, t1, t2, u2, uc, JU);
      V3<T> jd = mk<T>(JU[0].d * qd[k0] + JU[1].d * qd[k0 + 1] + JU[2].d * qd[k0 + 2],
                       JU[3].d * qd[k0] + JU[4].d * qd[k0 + 1] + JU[5].d * qd[k0 + 2],
                       JU[6].d * qd[k0] + JU[7].d * qd[k0 + 1] + JU[8].d * qd[k0 + 2]);
      extra_b = mul(R0, jd); has_b = true;
    } else {
      fprintf(stderr, "oracle: joint type %d not implemented\n", jt); abort();
    }
    // V_i = V_p + sum W_k qd_k ;  A_i = A_p + sum W_k qdd_k + V_i x^ V_J   (constant-S joints)
    V3<T> jw = mk<T>(T(0.0), T(0.0), T(0.0)), jv = jw, bw = jw, bv = jw;
    for (int k = k0; k < k0 + nd; ++k) { jw = jw + Ww[k] * qd[k]; jv = jv + Wv[k] * qd[k]; bw = bw + Ww[k] * qdd[k]; bv = bv + Wv[k] * qdd[k]; }
    X.w = P.w + jw; X.v = P.v + jv;
    X.aw = P.aw + bw + cross(X.w, jw);
    X.av = P.av + bv + cross(X.w, jv) + cross(X.v, jw);
    if (has_b) { X.aw = X.aw + extra_b; X.av = X.av + cross(X.p, extra_b); }
    if (!dyn) continue;
    // inertial wrench about the world origin
    T mass = T(lf[TSIM_LF_MASS]);
    V3<T> c = mul(X.R, cvec<T>(lf + TSIM_LF_COM)) + X.p;
    const double* ii = lf + TSIM_LF_INERTIA;
    M3<T> Il; Il.m[0] = T(ii[0]); Il.m[4] = T(ii[1]); Il.m[8] = T(ii[2]); Il.m[1] = Il.m[3] = T(ii[3]); Il.m[2] = Il.m[6] = T(ii[4]); Il.m[5] = Il.m[7] = T(ii[5]);
    V3<T> vc = X.v + cross(X.w, c);
    V3<T> ac = X.av + cross(X.aw, c) + cross(X.w, vc);
    V3<T> f = ac * mass;
    V3<T> wl = mulT(X.R, X.w), al = mulT(X.R, X.aw);
    V3<T> nc = mul(X.R, mul(Il, al) + cross(wl, mul(Il, wl)));
    X.ff = f; X.fn = nc + cross(c, f);
  }
}

template <class T>
static void residual(const Model& m, const T* q, const T* qd, const T* qdd, const T* u, T* r, Link<T>* L) {
  V3<T> Ww[MAXR], Wv[MAXR];
  kinematics(m, q, qd, qdd, L, Ww, Wv, true);
  // contacts (dynamics-active pairs)
  for (int pk = 0; pk < m.npair; ++pk) {
    const int* pi = m.pi(pk); const double* pf = m.pf(pk);
    if (!(pi[TSIM_PI_FLAGS] & 1)) continue;
    int la = pi[TSIM_PI_LINKA], lb = pi[TSIM_PI_LINKB];
    M3<T> RP; V3<T> pP; prim_pose(m, pk, L, RP, pP);
    Link<T>& A = L[la]; Link<T>& Bk = L[lb];
    for (int c = pi[TSIM_PI_PT0]; c < pi[TSIM_PI_PT0] + pi[TSIM_PI_NPT]; ++c) {
      V3<T> xw = mul(A.R, mk<T>(T(m.cpt(0, c)), T(m.cpt(1, c)), T(m.cpt(2, c)))) + A.p;
      if (pi[TSIM_PI_FLAGS] & 2) {   // sphere on plane: lowest point
        V3<T> nw = mk<T>(RP.m[2], RP.m[5], RP.m[8]);
        xw = xw - nw * T(pf[TSIM_PF_SHAPE]);
      }
      V3<T> vrel = (A.v + cross(A.w, xw)) - (Bk.v + cross(Bk.w, xw));
      V3<T> Fw;
      if (!contact_force<T>(pi[TSIM_PI_PRIM], pf + TSIM_PF_SHAPE, pf + TSIM_PF_KN, RP, pP, xw, vrel, Fw)) continue;
      V3<T> mo = cross(xw, Fw);
      A.ff = A.ff - Fw; A.fn = A.fn - mo;
      if (lb > 0) { Bk.ff = Bk.ff + Fw; Bk.fn = Bk.fn + mo; }
    }
  }
  // project on the joints, leaf to root
  for (int i = m.nl; i >= 1; --i) {
    const int* li = m.li(i);
    Link<T>& X = L[i];
    for (int k = li[TSIM_LI_DOF0]; k < li[TSIM_LI_DOF0] + li[TSIM_LI_NDOF]; ++k) r[k] = dot(Ww[k], X.fn) + dot(Wv[k], X.ff);
    int p = li[TSIM_LI_PARENT];
    if (p > 0) { L[p].fn = L[p].fn + X.fn; L[p].ff = L[p].ff + X.ff; }
  }
  // joint-space forces: damping, limits, motors
  for (int k = 0; k < m.nr; ++k) {
    const double* df = m.df(k);
    r[k] = r[k] + qd[k] * T(df[TSIM_DF_DAMPING]);
    if (df[TSIM_DF_LIM_K] > 0) {
      if (val(q[k]) < df[TSIM_DF_LIM_LO]) r[k] = r[k] - (T(df[TSIM_DF_LIM_LO]) - q[k]) * T(df[TSIM_DF_LIM_K]);
      else if (val(q[k]) > df[TSIM_DF_LIM_HI]) r[k] = r[k] + (q[k] - T(df[TSIM_DF_LIM_HI])) * T(df[TSIM_DF_LIM_K]);
    }
  }
  for (int j = 0; j < m.nu; ++j) {
    const int* mi = m.mi(j); const double* mf = m.mf(j);
    int k = mi[TSIM_MI_DOF];
    if (mi[TSIM_MI_CTRL] == 0) {
      T uc = u[j];
      if (val(uc) > 1.0) uc = T(1.0); else if (val(uc) < -1.0) uc = T(-1.0);
      r[k] = r[k] - (T(mf[TSIM_MF_LO]) + (uc + T(1.0)) * T(0.5 * (mf[TSIM_MF_HI] - mf[TSIM_MF_LO])));
    } else {
      r[k] = r[k] - ((u[j] - q[k]) * T(mf[TSIM_MF_P]) - qd[k] * T(mf[TSIM_MF_D]));
    }
  }
}

// tactile force vector + end-effector variables at state (q, qd)
template <class T>
static void outputs(const Model& m, const T* q, const T* qd, T* var, T* tac, bool want_tac) {
  Link<T> L[MAXL]; V3<T> Ww[MAXR], Wv[MAXR];
  T zero[MAXR]; for (int k = 0; k < m.nr; ++k) zero[k] = T(0.0);
  kinematics(m, q, qd, zero, L, Ww, Wv, false);
  for (int e = 0; e < m.nvar; ++e) {
    const Link<T>& X = L[m.vi(e)[TSIM_VI_LINK]];
    V3<T> x = mul(X.R, cvec<T>(m.vf(e))) + X.p;
    var[3 * e] = x.x; var[3 * e + 1] = x.y; var[3 * e + 2] = x.z;
  }
  if (!want_tac) return;
  for (int s = 0; s < m.nsensor; ++s) {
    const int* si = m.si(s); const double* sf = m.sf(s);
    const Link<T>& A = L[si[TSIM_SI_LINK]];
    M3<T> RP[16]; V3<T> pP[16];
    int np = si[TSIM_SI_NSPRIM];
    for (int j = 0; j < np; ++j) prim_pose(m, m.sprim(si[TSIM_SI_SPRIM0] + j), L, RP[j], pP[j]);
    for (int t = si[TSIM_SI_TAX0]; t < si[TSIM_SI_TAX0] + si[TSIM_SI_NTAX]; ++t) {
      V3<T> xw = mul(A.R, mk<T>(T(m.tax(0, t)), T(m.tax(1, t)), T(m.tax(2, t)))) + A.p;
      V3<T> F = mk<T>(T(0.0), T(0.0), T(0.0));
      for (int j = 0; j < np; ++j) {
        int pk = m.sprim(si[TSIM_SI_SPRIM0] + j);
        const int* pi = m.pi(pk); const double* pf = m.pf(pk);
        const Link<T>& Bk = L[pi[TSIM_PI_LINKB]];
        V3<T> vrel = (A.v + cross(A.w, xw)) - (Bk.v + cross(Bk.w, xw));
        V3<T> Fw;
        if (contact_force<T>(pi[TSIM_PI_PRIM], pf + TSIM_PF_SHAPE, sf, RP[j], pP[j], xw, vrel, Fw)) F = F + Fw;
      }
      V3<T> Fl = mulT(A.R, F);     // sensor-link frame
      tac[3 * t + 0] = Fl.x * T(m.tax(3, t)) + Fl.y * T(m.tax(4, t)) + Fl.z * T(m.tax(5, t));
      tac[3 * t + 1] = Fl.x * T(m.tax(6, t)) + Fl.y * T(m.tax(7, t)) + Fl.z * T(m.tax(8, t));
      tac[3 * t + 2] = Fl.x * T(m.tax(9, t)) + Fl.y * T(m.tax(10, t)) + Fl.z * T(m.tax(11, t));
    }
  }
}

// Branch signature of the state (q, qd) (test diagnostics; the HIP library states the same two numbers in
// tsim_debug_signature): over the dynamics-active contact points and the (taxel, paired primitive) items, count those that
// penetrate and add up mix(group, index, 1 + branch) mod 2^32.
static inline unsigned sig_mix(unsigned group, unsigned index, unsigned code) {
  unsigned x = (group * 0x9E3779B1u) ^ (index * 0x85EBCA77u) ^ (code * 0xC2B2AE3Du);
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
static void signature(const Model& m, const double* q, const double* qd, unsigned* out) {
  Link<double> L[MAXL]; V3<double> Ww[MAXR], Wv[MAXR];
  double zero[MAXR]; for (int k = 0; k < m.nr; ++k) zero[k] = 0.0;
  kinematics<double>(m, q, qd, zero, L, Ww, Wv, false);
  unsigned cnt = 0, sum = 0;
  for (int pk = 0; pk < m.npair; ++pk) {
    const int* pi = m.pi(pk); const double* pf = m.pf(pk);
    if (!(pi[TSIM_PI_FLAGS] & 1)) continue;
    M3<double> RP; V3<double> pP; prim_pose(m, pk, L, RP, pP);
    const Link<double>& A = L[pi[TSIM_PI_LINKA]]; const Link<double>& Bk = L[pi[TSIM_PI_LINKB]];
    for (int i = 0; i < pi[TSIM_PI_NPT]; ++i) {
      int c = pi[TSIM_PI_PT0] + i;
      V3<double> xw = mul(A.R, mk<double>(m.cpt(0, c), m.cpt(1, c), m.cpt(2, c))) + A.p;
      if (pi[TSIM_PI_FLAGS] & 2) xw = xw - mk<double>(RP.m[2], RP.m[5], RP.m[8]) * pf[TSIM_PF_SHAPE];
      V3<double> vrel = (A.v + cross(A.w, xw)) - (Bk.v + cross(Bk.w, xw)), Fw; int br = 0;
      if (contact_force<double>(pi[TSIM_PI_PRIM], pf + TSIM_PF_SHAPE, pf + TSIM_PF_KN, RP, pP, xw, vrel, Fw, &br)) { ++cnt; sum += sig_mix((unsigned)pk, (unsigned)i, (unsigned)(1 + br)); }
    }
  }
  for (int s = 0; s < m.nsensor; ++s) {
    const int* si = m.si(s); const double* sf = m.sf(s);
    const Link<double>& A = L[si[TSIM_SI_LINK]];
    for (int j = 0; j < si[TSIM_SI_NSPRIM]; ++j) {
      int pk = m.sprim(si[TSIM_SI_SPRIM0] + j);
      const int* pi = m.pi(pk); const double* pf = m.pf(pk);
      M3<double> RP; V3<double> pP; prim_pose(m, pk, L, RP, pP);
      const Link<double>& Bk = L[pi[TSIM_PI_LINKB]];
      for (int i = 0; i < si[TSIM_SI_NTAX]; ++i) {
        int t = si[TSIM_SI_TAX0] + i;
        V3<double> xw = mul(A.R, mk<double>(m.tax(0, t), m.tax(1, t), m.tax(2, t))) + A.p;
        V3<double> vrel = (A.v + cross(A.w, xw)) - (Bk.v + cross(Bk.w, xw)), Fw; int br = 0;
        if (contact_force<double>(pi[TSIM_PI_PRIM], pf + TSIM_PF_SHAPE, sf, RP, pP, xw, vrel, Fw, &br)) { ++cnt; sum += sig_mix(0x10000u + (unsigned)(si[TSIM_SI_SPRIM0] + j), (unsigned)i, (unsigned)(1 + br)); }
      }
    }
  }
  out[0] = cnt; out[1] = sum;
}

// ------------------------------------------------------------------------------------------ dense LU
static bool lu_factor(int n, double* A, int* piv) {
  for (int c = 0; c < n; ++c) {
    int p = c; double best = std::fabs(A[c * n + c]);
    for (int r = c + 1; r < n; ++r) if (std::fabs(A[r * n + c]) > best) { best = std::fabs(A[r * n + c]); p = r; }
    piv[c] = p;
    if (best == 0.0) return false;
    if (p != c) for (int j = 0; j < n; ++j) std::swap(A[c * n + j], A[p * n + j]);
    for (int r = c + 1; r < n; ++r) {
      double f = A[r * n + c] / A[c * n + c]; A[r * n + c] = f;
      for (int j = c + 1; j < n; ++j) A[r * n + j] -= f * A[c * n + j];
    }
  }
  return true;
}
static void lu_solve(int n, const double* A, const int* piv, double* b) {
  // full rows (L part included) were swapped during factorisation, so the whole permutation is applied first
  for (int c = 0; c < n; ++c) if (piv[c] != c) std::swap(b[c], b[piv[c]]);
  for (int c = 0; c < n; ++c) for (int r = c + 1; r < n; ++r) b[r] -= A[r * n + c] * b[c];
  for (int c = n - 1; c >= 0; --c) { b[c] /= A[c * n + c]; for (int r = 0; r < c; ++r) b[r] -= A[r * n + c] * b[c]; }
}
static bool solve_dense(int n, const double* A, const double* b, double* x, bool transpose) {
  double M[MAXR * MAXR]; int piv[MAXR];
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) M[i * n + j] = transpose ? A[j * n + i] : A[i * n + j];
  if (!lu_factor(n, M, piv)) return false;
  for (int i = 0; i < n; ++i) x[i] = b[i];
  lu_solve(n, M, piv, x);
  return true;
}

// ------------------------------------------------------------------------------------------ simulation object
struct Rec { std::vector<double> q0, qd0, q1, qd1, u, qm1, qdm1; bool bdf2 = false; };
struct Sim {
  Model m;
  std::vector<double> q, qd, u, lam_q, lam_v, qm1, qdm1;   // qm1 / qdm1: state before the last sub-step (BDF2)
  std::vector<double> lam_q1, lam_v1;                       // BDF2 adjoint: what later sub-steps already contributed to the state one step further back
  bool has_prev = false;
  std::vector<Rec> tape;
  std::vector<std::vector<Rec>> cache;
  bool record = false;
  int solver = 1;          // 1 (default): LITERAL — exactly what the XML states, and what the HIP kernels run;  0: the round-2 globalisation (legacy, kept to document what it did)
  long newton_iters = 0, substeps = 0, nonconv = 0, evals = 0;
  long kicks = 0, restarts = 0, trust = 0, ls_exhausted = 0;   // how often each globalisation device acted (kernel mode) / a line search ran out (literal mode)
};

// One implicit step in predictor form (covers BDF1 and BDF2):
//   qd1 = qdpred + cv (q1 - qpred),  qdd1 = ca (q1 - qpred),  g = r(q1, qd1, qdd1, u) / ca
//   BDF1: qpred = q0 + h qd0, qdpred = qd0, cv = 1/h, ca = 1/h^2            (g = h^2 r, RedMax BDF1)
//   BDF2: qpred = 4/3 q0 - 1/3 q_1 + 8/9 h qd0 - 2/9 h qd_1, qdpred = (3 qpred - 4 q0 + q_1)/(2h),
//         cv = 3/(2h), ca = 9/(4h^2)                                        (RedMax BDF2; first step after reset: BDF1)
struct StepCoef {
  double qpred[MAXR], qdpred[MAXR], cv, ca;
  // d qpred / d (q0, qd0, q_1, qd_1) and d qdpred / d (the same): multiples of the identity (adjoint: eval_g_jac_c which = 1, 2, 4, 5)
  double dqp[4], dqdp[4];
};
static void make_coef(const Model& m, const double* q0, const double* qd0, const double* qm1, const double* qdm1, StepCoef& c) {
  double h = m.h;
  if (qm1) {
    c.cv = 1.5 / h; c.ca = 2.25 / (h * h);
    c.dqp[0] = 4.0 / 3; c.dqp[1] = 8.0 / 9 * h; c.dqp[2] = -1.0 / 3; c.dqp[3] = -2.0 / 9 * h;
    c.dqdp[0] = (3 * c.dqp[0] - 4) / (2 * h); c.dqdp[1] = 3 * c.dqp[1] / (2 * h); c.dqdp[2] = (3 * c.dqp[2] + 1) / (2 * h); c.dqdp[3] = 3 * c.dqp[3] / (2 * h);
    for (int k = 0; k < m.nr; ++k) {
      c.qpred[k] = 4.0 / 3 * q0[k] - 1.0 / 3 * qm1[k] + 8.0 / 9 * h * qd0[k] - 2.0 / 9 * h * qdm1[k];
      c.qdpred[k] = (3 * c.qpred[k] - 4 * q0[k] + qm1[k]) / (2 * h);
    }
  } else {
    c.cv = 1.0 / h; c.ca = 1.0 / (h * h);
    c.dqp[0] = 1.0; c.dqp[1] = h; c.dqp[2] = c.dqp[3] = 0.0;
    c.dqdp[0] = 0.0; c.dqdp[1] = 1.0; c.dqdp[2] = c.dqdp[3] = 0.0;
    for (int k = 0; k < m.nr; ++k) { c.qpred[k] = q0[k] + h * qd0[k]; c.qdpred[k] = qd0[k]; }
  }
}
static void eval_g_c(const Model& m, const double* q1, const StepCoef& c, const double* u, double* g) {
  double qd[MAXR], qa[MAXR]; Link<double> L[MAXL];
  for (int k = 0; k < m.nr; ++k) { double d = q1[k] - c.qpred[k]; qd[k] = c.qdpred[k] + c.cv * d; qa[k] = c.ca * d; }
  residual<double>(m, q1, qd, qa, u, g, L);
  for (int k = 0; k < m.nr; ++k) g[k] /= c.ca;
}
// which: 0 = d/dq1, 1 = d/dq0, 2 = d/dqd0, 3 = d/du, 4 = d/dq_1, 5 = d/dqd_1 (BDF2 history) — q1 held fixed for 1, 2, 4, 5 ;  J[row*ncol+col]
static void eval_g_jac_c(const Model& m, const double* q1, const StepCoef& c, const double* u, int which, double* g, double* J) {
  int nr = m.nr, ncol = which == 3 ? m.nu : nr;
  for (int c0 = 0; c0 < ncol; c0 += NDMAX) {
    int nd = std::min(NDMAX, ncol - c0);
    g_nd = nd;
    Dual q[MAXR], qd[MAXR], qa[MAXR], uu[MAXR], r[MAXR]; Link<Dual> L[MAXL];
    for (int k = 0; k < nr; ++k) { double d = q1[k] - c.qpred[k]; q[k] = Dual(q1[k]); qd[k] = Dual(c.qdpred[k] + c.cv * d); qa[k] = Dual(c.ca * d); }
    for (int j = 0; j < m.nu; ++j) uu[j] = Dual(u[j]);
    for (int d = 0; d < nd; ++d) {
      int k = c0 + d;
      if (which == 0) { q[k].d[d] = 1.0; qd[k].d[d] = c.cv; qa[k].d[d] = c.ca; }
      else if (which == 3) uu[k].d[d] = 1.0;
      else {                                                  // through the predictor: qd = qdpred + cv (q1 - qpred), qdd = ca (q1 - qpred)
        const int w = which == 1 ? 0 : which == 2 ? 1 : which == 4 ? 2 : 3;
        qd[k].d[d] = c.dqdp[w] - c.cv * c.dqp[w]; qa[k].d[d] = -c.ca * c.dqp[w];
      }
    }
    residual<Dual>(m, q, qd, qa, uu, r, L);
    for (int i = 0; i < nr; ++i) { g[i] = r[i].v / c.ca; for (int d = 0; d < nd; ++d) J[i * ncol + c0 + d] = r[i].d[d] / c.ca; }
    g_nd = 0;
  }
}
// BDF1 wrappers in terms of (q0, qd0) (adjoint, diagnostics)
static void eval_g(const Model& m, const double* q1, const double* q0, const double* qd0, const double* u, double* g) {
  StepCoef c; make_coef(m, q0, qd0, nullptr, nullptr, c); eval_g_c(m, q1, c, u, g);
}
static void eval_g_jac(const Model& m, const double* q1, const double* q0, const double* qd0, const double* u, int which, double* g, double* J) {
  StepCoef c; make_coef(m, q0, qd0, nullptr, nullptr, c); eval_g_jac_c(m, q1, c, u, which, g, J);
}

static double norm2(int n, const double* x) { double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return std::sqrt(s); }


// Constants of the LEGACY (round-2) globalisation only — solver mode 0 below; the literal solver reads none of them.
#ifndef TSIM_LS_SHORT
#define TSIM_LS_SHORT 4
#endif
#define TSIM_KICK_MAX 6
#define TSIM_STEP_MAX 0.5

// LITERAL solver (the default, orc_set_solver(h, 1)): Newton with monotone backtracking exactly as the model file states it and nothing
// else — `<solver_option tol="1e-8" max_iter="100" max_ls="20"/>` (envs/assets/pusher/pusher.xml:4; the same line in
// dclaw_position_control.xml, tactile_insertion.xml, stable_grasp.xml): up to max_iter Newton iterations; each halves the
// step until ||g|| decreases, at most max_ls times; converged when ||g||_2 < tol.  No non-monotone steps, no restart, no trust
// region, no "100 tol" acceptance: none of the legacy constants above is read here.  When no trial of a line search reduces ||g|| the smallest one (alpha = 2^-max_ls) is taken and the iteration
// goes on — the loop has no exit the XML does not name [CHOICE: DiffRedMax's own behaviour there is unknown, source absent].
// Since round 3 the HIP kernels run this very loop (k_forward's Newton state machine), so the fp64 kernels take the same iterates.
// History: rounds 1-2 ran a performance-driven globalisation in kernel AND oracle (mode 0 below: backtracking cut short after 4
// halvings, then the full Newton step taken anyway, restart, trust region).  Checked against this literal loop in round 3 it turned
// out to land on ANOTHER ROOT — 0.15 rad / 2 cm away, |g| < tol — on 11 of 4096 TactileInsertion grasps (tests/test_oracle_literal.py
// keeps one of them as a regression), and was removed from the kernels.
static int substep_literal(Sim& S, const double* u, const StepCoef& c, double* q1) {
  const Model& m = S.m; int nr = m.nr;
  double g[MAXR], H[MAXR * MAXR], dq[MAXR], qn[MAXR], gn[MAXR];
  for (int k = 0; k < nr; ++k) q1[k] = c.qpred[k];
  int it = 0; bool ok = false;
  for (; it <= m.max_iter; ++it) {
    eval_g_jac_c(m, q1, c, u, 0, g, H); ++S.evals;
    const double gnorm = norm2(nr, g);
    if (gnorm < m.tol) { ok = true; break; }
    if (!(gnorm == gnorm) || it == m.max_iter) break;
    for (int k = 0; k < nr; ++k) g[k] = -g[k];
    if (!solve_dense(nr, H, g, dq, false)) break;
    double alpha = 1.0;
    for (int ls = 0;; ++ls) {
      for (int k = 0; k < nr; ++k) qn[k] = q1[k] + alpha * dq[k];
      eval_g_c(m, qn, c, u, gn); ++S.evals;
      if (norm2(nr, gn) < gnorm) break;
      if (ls >= m.max_ls) {
        ++S.ls_exhausted;
        static const bool trace_l = getenv("TSIM_ORACLE_TRACE") != nullptr;
        if (trace_l) { double mx = 0; for (int k = 0; k < nr; ++k) mx = std::max(mx, std::fabs(dq[k])); fprintf(stderr, "LSX it %d gn %.3e step_inf %.3e\n", it, gnorm, mx); }
        break;
      }
      alpha *= 0.5;
    }
    for (int k = 0; k < nr; ++k) q1[k] = qn[k];
  }
  return ok ? it : -it - 1;
}

// one implicit sub-step (BDF1, or BDF2 once a previous state exists); returns Newton iterations used, negative if not converged
static int substep(Sim& S, const double* u) {
  const Model& m = S.m; int nr = m.nr; double h = m.h;
  double q0[MAXR] = {0}, qd0[MAXR] = {0}, q1[MAXR], g[MAXR], H[MAXR * MAXR], dq[MAXR], qn[MAXR], gn[MAXR];
  for (int k = 0; k < nr; ++k) { q0[k] = S.q[k]; qd0[k] = S.qd[k]; }
  const bool bdf2 = m.integrator == 2 && S.has_prev;
  StepCoef c; make_coef(m, q0, qd0, bdf2 ? S.qm1.data() : nullptr, bdf2 ? S.qdm1.data() : nullptr, c);
  for (int k = 0; k < nr; ++k) q1[k] = c.qpred[k];
  int it = 0, kicks = 0; bool ok = false, deep = false;
  static const bool trace = getenv("TSIM_ORACLE_TRACE") != nullptr;
  if (S.solver == 1) {
    const int r = substep_literal(S, u, c, q1);
    ok = r >= 0; it = ok ? r : -r - 1;
  } else
  for (; it <= m.max_iter; ++it) {
    eval_g_jac_c(m, q1, c, u, 0, g, H); ++S.evals;
    double gnorm = norm2(nr, g);
    if (trace) fprintf(stderr, "  it %d gn %.3e kicks %d deep %d\n", it, gnorm, kicks, (int)deep);
    if (gnorm < m.tol) { ok = true; break; }
    if (it == m.max_iter) break;
    for (int k = 0; k < nr; ++k) g[k] = -g[k];
    if (!solve_dense(nr, H, g, dq, false)) break;
    {                                            // trust region (legacy constant TSIM_STEP_MAX)
      double mx = 0.0;
      for (int k = 0; k < nr; ++k) mx = std::max(mx, std::fabs(dq[k]));
      if (mx > TSIM_STEP_MAX) { ++S.trust; for (int k = 0; k < nr; ++k) dq[k] *= TSIM_STEP_MAX / mx; }
    }
    // Globalisation: backtracking on ||g||. ||g|| has non-smooth local minima next to contact / friction
    // kinks where no short step along the Newton direction reduces it; there the full Newton step is taken anyway (it
    // lands across the kink, from where the iteration normally converges in two or three steps).  A sub-step that needs
    // more than TSIM_KICK_MAX such steps (a cycle) is restarted from the predictor with plain monotone backtracking
    // down to 2^-max_ls, which stops at the last accepted iterate when even that finds no decrease.
    double alpha = 1.0; bool accepted = false, kick = false, restart = false;
    for (int ls = 0;; ++ls) {
      for (int k = 0; k < nr; ++k) qn[k] = q1[k] + alpha * dq[k];
      eval_g_c(m, qn, c, u, gn); ++S.evals;
      if (norm2(nr, gn) < gnorm) { accepted = true; if (trace) fprintf(stderr, "    accept alpha %.3g\n", alpha); break; }
      if (!deep && ls >= std::min(m.max_ls, TSIM_LS_SHORT)) {
        if (kicks < TSIM_KICK_MAX) { ++kicks; ++S.kicks; kick = true; if (trace) { double mx = 0; for (int k = 0; k < nr; ++k) mx = std::max(mx, std::fabs(dq[k])); fprintf(stderr, "KICK it %d gn %.3e step_inf %.3e\n", it, gnorm, mx); } } else { deep = true; restart = true; ++S.restarts; }
        break;
      }
      if (ls >= m.max_ls) break;
      alpha *= 0.5;
    }
    if (restart) { for (int k = 0; k < nr; ++k) q1[k] = c.qpred[k]; it = -1; continue; }
    if (kick) { for (int k = 0; k < nr; ++k) qn[k] = q1[k] + dq[k]; }
    else if (!accepted) { ok = gnorm < 100.0 * m.tol; break; }   // stays at the last accepted iterate
    for (int k = 0; k < nr; ++k) q1[k] = qn[k];
  }
  double qd1[MAXR];
  for (int k = 0; k < nr; ++k) qd1[k] = c.qdpred[k] + c.cv * (q1[k] - c.qpred[k]);
  if (S.record) {
    Rec r; r.q0.assign(q0, q0 + nr); r.qd0.assign(qd0, qd0 + nr); r.q1.assign(q1, q1 + nr); r.u.assign(u, u + m.nu);
    r.qd1.assign(qd1, qd1 + nr); r.bdf2 = bdf2;
    if (bdf2) { r.qm1 = S.qm1; r.qdm1 = S.qdm1; }
    S.tape.push_back(r);
  }
  S.qm1.assign(q0, q0 + nr); S.qdm1.assign(qd0, qd0 + nr); S.has_prev = true;
  for (int k = 0; k < nr; ++k) { S.q[k] = q1[k]; S.qd[k] = qd1[k]; }
  (void)h;
  S.newton_iters += it; S.substeps += 1; if (!ok) S.nonconv += 1;
  return ok ? it : -it - 1;
}

// lam_q += (dvar/dq)^T wvar + (dtac/dq)^T wtac ; lam_v += (dtac/dqd)^T wtac   at state (q, qd)
static void output_vjp(const Model& m, const double* q0, const double* qd0, const double* wvar, const double* wtac, double* lam_q, double* lam_v) {
  int nr = m.nr, nvar3 = 3 * m.nvar, ntac3 = 3 * m.ntax;
  bool any_tac = false; for (int i = 0; i < ntac3 && wtac; ++i) if (wtac[i] != 0.0) { any_tac = true; break; }
  bool any_var = false; for (int i = 0; i < nvar3 && wvar; ++i) if (wvar[i] != 0.0) { any_var = true; break; }
  if (!any_tac && !any_var) return;
  int ncol = 2 * nr;   // columns: d/dq (nr) then d/dqd (nr)
  std::vector<Dual> var(std::max(nvar3, 1)), tac(std::max(ntac3, 1));
  for (int c0 = 0; c0 < ncol; c0 += NDMAX) {
    int nd = std::min(NDMAX, ncol - c0);
    g_nd = nd;
    Dual q[MAXR], qd[MAXR];
    for (int k = 0; k < nr; ++k) { q[k] = Dual(q0[k]); qd[k] = Dual(qd0[k]); }
    for (int d = 0; d < nd; ++d) { int c = c0 + d; if (c < nr) q[c].d[d] = 1.0; else qd[c - nr].d[d] = 1.0; }
    outputs<Dual>(m, q, qd, var.data(), tac.data(), any_tac);
    for (int d = 0; d < nd; ++d) {
      double s = 0;
      if (any_var) for (int i = 0; i < nvar3; ++i) s += wvar[i] * var[i].d[d];
      if (any_tac) for (int i = 0; i < ntac3; ++i) s += wtac[i] * tac[i].d[d];
      int c = c0 + d; if (c < nr) lam_q[c] += s; else lam_v[c - nr] += s;
    }
    g_nd = 0;
  }
}

extern "C" {

void* orc_create(const int* I, const double* F) {
  if (I[TSIM_IH_MAGIC] != TSIM_MAGIC || I[TSIM_IH_VERSION] != TSIM_VERSION) return nullptr;
  Sim* S = new Sim();
  Model& m = S->m;
  m.I.assign(I, I + I[TSIM_IH_NI]); m.F.assign(F, F + I[TSIM_IH_NF]);
  m.nl = I[TSIM_IH_NL]; m.nr = I[TSIM_IH_NR]; m.nu = I[TSIM_IH_NU]; m.nvar = I[TSIM_IH_NVAR];
  m.npair = I[TSIM_IH_NPAIR]; m.ncpt = I[TSIM_IH_NCPT]; m.nsensor = I[TSIM_IH_NSENSOR]; m.ntax = I[TSIM_IH_NTAXEL];
  m.integrator = I[TSIM_IH_INTEGRATOR]; m.max_iter = I[TSIM_IH_MAX_ITER]; m.max_ls = I[TSIM_IH_MAX_LS];
  m.h = F[TSIM_FH_H]; m.grav[0] = F[TSIM_FH_GX]; m.grav[1] = F[TSIM_FH_GY]; m.grav[2] = F[TSIM_FH_GZ]; m.tol = F[TSIM_FH_TOL];
  if (m.nl + 1 > MAXL || m.nr > MAXR || m.nu > MAXR) { delete S; return nullptr; }
  S->q.assign(m.nr, 0.0); S->qd.assign(m.nr, 0.0); S->u.assign(m.nu, 0.0);
  S->lam_q.assign(m.nr, 0.0); S->lam_v.assign(m.nr, 0.0);
  return S;
}
void orc_destroy(void* h) { delete (Sim*)h; }

void orc_reset(void* h, const double* q, const double* qd, int record) {
  Sim& S = *(Sim*)h;
  for (int k = 0; k < S.m.nr; ++k) { S.q[k] = q[k]; S.qd[k] = qd[k]; }
  S.tape.clear(); S.record = record != 0; S.has_prev = false;
  std::fill(S.lam_q.begin(), S.lam_q.end(), 0.0); std::fill(S.lam_v.begin(), S.lam_v.end(), 0.0);
  std::fill(S.lam_q1.begin(), S.lam_q1.end(), 0.0); std::fill(S.lam_v1.begin(), S.lam_v1.end(), 0.0);
}
// BDF2 models: the state before the previous sub-step (teacher-forced replays start in the middle of a roll-out); call after orc_reset
void orc_set_prev(void* h, const double* qm1, const double* qdm1) {
  Sim& S = *(Sim*)h;
  S.qm1.assign(qm1, qm1 + S.m.nr); S.qdm1.assign(qdm1, qdm1 + S.m.nr); S.has_prev = true;
}
// nsub implicit sub-steps with u held; returns number of non-converged sub-steps
int orc_forward(void* h, const double* u, int nsub) {
  Sim& S = *(Sim*)h; int bad = 0;
  for (int s = 0; s < nsub; ++s) if (substep(S, u) < 0) ++bad;
  return bad;
}
// the same, recording the branch signature (count, hash) after every sub-step: sig [nsub][2]
int orc_forward_sig(void* h, const double* u, int nsub, unsigned* sig) {
  Sim& S = *(Sim*)h; int bad = 0;
  for (int s = 0; s < nsub; ++s) { if (substep(S, u) < 0) ++bad; signature(S.m, S.q.data(), S.qd.data(), sig + 2 * s); }
  return bad;
}
void orc_get_state(void* h, double* q, double* qd) { Sim& S = *(Sim*)h; for (int k = 0; k < S.m.nr; ++k) { q[k] = S.q[k]; qd[k] = S.qd[k]; } }
void orc_outputs(void* h, double* var, double* tac) {
  Sim& S = *(Sim*)h;
  std::vector<double> v(std::max(3 * S.m.nvar, 1)), t(std::max(3 * S.m.ntax, 1));
  outputs<double>(S.m, S.q.data(), S.qd.data(), v.data(), t.data(), tac != nullptr);
  if (var) std::copy(v.begin(), v.begin() + 3 * S.m.nvar, var);
  if (tac) std::copy(t.begin(), t.begin() + 3 * S.m.ntax, tac);
}
int orc_tape_len(void* h) { return (int)((Sim*)h)->tape.size(); }
void orc_stats(void* h, long* out) { Sim& S = *(Sim*)h; out[0] = S.newton_iters; out[1] = S.substeps; out[2] = S.nonconv; out[3] = S.evals;
  out[4] = S.kicks; out[5] = S.restarts; out[6] = S.trust; out[7] = S.ls_exhausted; }
// 1 (default): literal XML solver (substep_literal above) = what the HIP kernels run;  0: the round-2 globalisation (legacy)
int orc_set_solver(void* h, int mode) { if (mode != 0 && mode != 1) return -1; ((Sim*)h)->solver = mode; return 0; }
int orc_get_solver(void* h) { return ((Sim*)h)->solver; }

// Adjoint over the newest n taped sub-steps, newest first, continuing the carried adjoint (lam_q, lam_v).
// df_dq [n*nr], df_dvar [n*3nvar], df_dtac [n*3ntax]: direct partials of the loss w.r.t. the outputs after each of
// those sub-steps, step-major, oldest first (layout of envs/redmax_torch_functions.py:153-165). df_du out [n*nu].
int orc_backward_steps(void* h, int n, const double* df_dq, const double* df_dvar, const double* df_dtac, double* df_du) {
  Sim& S = *(Sim*)h; const Model& m = S.m; int nr = m.nr, nu = m.nu;
  if ((int)S.tape.size() < n) return -1;
  if ((int)S.lam_q1.size() != nr) { S.lam_q1.assign(nr, 0.0); S.lam_v1.assign(nr, 0.0); }
  std::vector<double> g(nr), H(nr * nr), J[4], Ju(nr * std::max(nu, 1)), rhs(nr), z(nr);
  for (auto& j : J) j.assign(nr * nr, 0.0);
  static const int which_of[4] = {1, 2, 4, 5};      // d/dq0, d/dqd0, d/dq_1, d/dqd_1
  for (int j = n - 1; j >= 0; --j) {
    Rec& r = S.tape.back();
    // One sub-step in predictor form (make_coef): the new state (q1, qd1 = qdpred + cv (q1 - qpred)) depends on the history
    // p = (q0, qd0, q_1, qd_1) through g(q1; p, u) = 0 and through qd1's explicit dependence on p.  (lam_q, lam_v): total derivative of
    // the loss w.r.t. (q1, qd1) with everything later accounted for.  All Jacobians by dual numbers; nothing here uses the identities
    // the kernels use (H = K + (cv R_v + ca M) / ca, ...), so the GPU adjoint test also verifies those.
    StepCoef c; make_coef(m, r.q0.data(), r.qd0.data(), r.bdf2 ? r.qm1.data() : nullptr, r.bdf2 ? r.qdm1.data() : nullptr, c);
    for (int k = 0; k < nr; ++k) S.lam_q[k] += df_dq ? df_dq[j * nr + k] : 0.0;
    output_vjp(m, r.q1.data(), r.qd1.data(), df_dvar ? df_dvar + (size_t)j * 3 * m.nvar : nullptr,
               df_dtac ? df_dtac + (size_t)j * 3 * m.ntax : nullptr, S.lam_q.data(), S.lam_v.data());
    eval_g_jac_c(m, r.q1.data(), c, r.u.data(), 0, g.data(), H.data());
    const int nhist = r.bdf2 ? 4 : 2;
    for (int w = 0; w < nhist; ++w) eval_g_jac_c(m, r.q1.data(), c, r.u.data(), which_of[w], g.data(), J[w].data());
    if (nu > 0) eval_g_jac_c(m, r.q1.data(), c, r.u.data(), 3, g.data(), Ju.data());
    for (int k = 0; k < nr; ++k) rhs[k] = S.lam_q[k] + c.cv * S.lam_v[k];               // d qd1 / d q1 = cv
    if (!solve_dense(nr, H.data(), rhs.data(), z.data(), true)) return -2;
    for (int cc = 0; cc < nu; ++cc) { double s = 0; for (int i = 0; i < nr; ++i) s += Ju[i * nu + cc] * z[i]; df_du[j * nu + cc] = -s; }
    std::vector<double> out[4];
    for (int w = 0; w < 4; ++w) {
      out[w].assign(nr, 0.0);
      if (w >= nhist) continue;
      const double dv = c.dqdp[w] - c.cv * c.dqp[w];                                     // explicit d qd1 / d p_w (q1 held)
      for (int cc = 0; cc < nr; ++cc) {
        double s = 0;
        for (int i = 0; i < nr; ++i) s += J[w][i * nr + cc] * z[i];
        out[w][cc] = -s + dv * S.lam_v[cc];
      }
    }
    // the state one step back (q0, qd0) also carries what LATER sub-steps contributed to it as THEIR q_1 / qd_1
    for (int k = 0; k < nr; ++k) {
      S.lam_q[k] = out[0][k] + S.lam_q1[k]; S.lam_v[k] = out[1][k] + S.lam_v1[k];
      S.lam_q1[k] = out[2][k]; S.lam_v1[k] = out[3][k];
    }
    // restore the simulator state to the start of this sub-step (so outputs()/forward() stay consistent)
    S.q = r.q0; S.qd = r.qd0;
    if (r.bdf2) { S.qm1 = r.qm1; S.qdm1 = r.qdm1; S.has_prev = true; } else S.has_prev = false;
    S.tape.pop_back();
  }
  return 0;
}
void orc_get_adjoint(void* h, double* lam_q, double* lam_v) { Sim& S = *(Sim*)h; for (int k = 0; k < S.m.nr; ++k) { lam_q[k] = S.lam_q[k]; lam_v[k] = S.lam_v[k]; } }
void orc_clear_adjoint(void* h) { Sim& S = *(Sim*)h; std::fill(S.lam_q.begin(), S.lam_q.end(), 0.0); std::fill(S.lam_v.begin(), S.lam_v.end(), 0.0); std::fill(S.lam_q1.begin(), S.lam_q1.end(), 0.0); std::fill(S.lam_v1.begin(), S.lam_v1.end(), 0.0); }

// diagnostics: the penetrating dynamics contact points of the state (q, qd), up to `max` rows of (pair, point index, branch, cylinder / cuboid
// medial distance): out_i [max][3], out_d [max][2] = (penetration depth d < 0, for a cylinder dr - dz: 0 = equidistant from side and cap,
// where the normal of the penalty force JUMPS from radial to axial — see tools/dclaw_nonconv_probe.py).  Returns the number of penetrating points.
int orc_contact_list(void* h, const double* q, const double* qd, int max, int* out_i, double* out_d) {
  Sim& S = *(Sim*)h; const Model& m = S.m;
  Link<double> L[MAXL]; V3<double> Ww[MAXR], Wv[MAXR];
  double zero[MAXR]; for (int k = 0; k < m.nr; ++k) zero[k] = 0.0;
  kinematics<double>(m, q, qd, zero, L, Ww, Wv, false);
  int n = 0;
  for (int pk = 0; pk < m.npair; ++pk) {
    const int* pi = m.pi(pk); const double* pf = m.pf(pk);
    if (!(pi[TSIM_PI_FLAGS] & 1)) continue;
    M3<double> RP; V3<double> pP; prim_pose(m, pk, L, RP, pP);
    const Link<double>& A = L[pi[TSIM_PI_LINKA]]; const Link<double>& Bk = L[pi[TSIM_PI_LINKB]];
    for (int i = 0; i < pi[TSIM_PI_NPT]; ++i) {
      int c = pi[TSIM_PI_PT0] + i;
      V3<double> xw = mul(A.R, mk<double>(m.cpt(0, c), m.cpt(1, c), m.cpt(2, c))) + A.p;
      if (pi[TSIM_PI_FLAGS] & 2) xw = xw - mk<double>(RP.m[2], RP.m[5], RP.m[8]) * pf[TSIM_PF_SHAPE];
      V3<double> vrel = (A.v + cross(A.w, xw)) - (Bk.v + cross(Bk.w, xw)), Fw; int br = 0;
      if (!contact_force<double>(pi[TSIM_PI_PRIM], pf + TSIM_PF_SHAPE, pf + TSIM_PF_KN, RP, pP, xw, vrel, Fw, &br)) continue;
      if (n < max) {
        const double* sh = pf + TSIM_PF_SHAPE;
        V3<double> x = mulT(RP, xw - pP);
        double d = 0, med = 0;
        if (pi[TSIM_PI_PRIM] == TSIM_P_CYLINDER) { double dr = std::sqrt(x.x * x.x + x.y * x.y) - sh[0], dz = std::fabs(x.z) - sh[1]; d = std::max(dr, dz); med = dr - dz; }
        else if (pi[TSIM_PI_PRIM] == TSIM_P_CUBOID) { double e[3] = {std::fabs(x.x) - sh[0], std::fabs(x.y) - sh[1], std::fabs(x.z) - sh[2]}; std::sort(e, e + 3); d = e[2]; med = e[2] - e[1]; }
        else if (pi[TSIM_PI_PRIM] == TSIM_P_PLANE) d = x.z;
        else d = std::sqrt(x.x * x.x + x.y * x.y + x.z * x.z) - sh[0];
        out_i[3 * n] = pk; out_i[3 * n + 1] = i; out_i[3 * n + 2] = br; out_d[2 * n] = d; out_d[2 * n + 1] = med;
      }
      ++n;
    }
  }
  return n;
}

// diagnostics for the tests: residual g and Jacobian `which` at an arbitrary point
void orc_residual(void* h, const double* q1, const double* q0, const double* qd0, const double* u, int which, double* g, double* J) {
  Sim& S = *(Sim*)h;
  if (which < 0) eval_g(S.m, q1, q0, qd0, u, g); else eval_g_jac(S.m, q1, q0, qd0, u, which, g, J);
}
// inverse-dynamics residual r(q, qd, qdd, u) (used by known-answer tests: mass matrix, gravity, ...)
void orc_inverse_dynamics(void* h, const double* q, const double* qd, const double* qdd, const double* u, double* r) {
  Sim& S = *(Sim*)h; Link<double> L[MAXL];
  residual<double>(S.m, q, qd, qdd, u, r, L);
}

// timed rollout for bench.py's cpu_baseline: nenv independent envs run serially on the calling thread.
// u_tab [nenv][nstep][nu]; q0 [nenv][nr]. Returns total env-steps done. If with_backward, runs the adjoint of
// L = sum of all returned q, var, tactile (a dense seed) after each rollout.
long orc_bench_rollout(void* h, int nenv, int nstep, int nsub, const double* q0, const double* u_tab, int with_backward, double* checksum) {
  Sim& S = *(Sim*)h; const Model& m = S.m; int nr = m.nr, nu = m.nu;
  std::vector<double> zero(nr, 0.0), var(std::max(3 * m.nvar, 1)), tac(std::max(3 * m.ntax, 1));
  std::vector<double> wq(nsub * nr, 0.0), wv(nsub * 3 * m.nvar, 0.0), wt((size_t)nsub * 3 * m.ntax, 0.0), du(nsub * std::max(nu, 1));
  for (int k = 0; k < nr; ++k) wq[(nsub - 1) * nr + k] = 1.0;
  for (int k = 0; k < 3 * m.nvar; ++k) wv[(nsub - 1) * 3 * m.nvar + k] = 1.0;
  for (int k = 0; k < 3 * m.ntax; ++k) wt[(size_t)(nsub - 1) * 3 * m.ntax + k] = 1.0;
  double cs = 0; long steps = 0;
  for (int e = 0; e < nenv; ++e) {
    orc_reset(h, q0 + (size_t)e * nr, zero.data(), with_backward);
    for (int t = 0; t < nstep; ++t) {
      orc_forward(h, u_tab + ((size_t)e * nstep + t) * nu, nsub);
      outputs<double>(m, S.q.data(), S.qd.data(), var.data(), tac.data(), true);
      cs += S.q[0] + (m.ntax ? tac[2] : 0.0);
      ++steps;
    }
    if (with_backward) for (int t = nstep - 1; t >= 0; --t) { orc_backward_steps(h, nsub, wq.data(), wv.data(), wt.data(), du.data()); cs += du[0]; }
  }
  if (checksum) *checksum = cs;
  return steps;
}

}  // extern "C"
