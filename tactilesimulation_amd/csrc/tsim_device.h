// tsim_device.h — device-side building blocks of the batched tactile-simulation step (gfx950 / CDNA4).
//
// Execution model: ONE ENVIRONMENT PER 64-LANE WAVEFRONT (block = 64 threads).  The environment's reduced
// state (q, qd), the per-link world transforms / spatial velocities / wrenches and the Newton matrix live
// in LDS for the whole kernel (all sub-steps of an env-step); HBM is touched only for the action, the
// outputs and the tape.  Two lane mappings alternate inside one residual evaluation:
//
//   lanes = tangent directions  (phase 1: kinematics + inertial wrenches, phase 3: projection on joints)
//       lane k carries dual numbers (value, d/d(direction k)); the few links are walked serially, all
//       lanes in lock-step on the same link, so there is no divergence.  Exact Newton matrix, no
//       hand-derived second-order kinematics.
//   lanes = contact points / taxels  (phase 2, tactile read-out)
//       each lane owns one sampled surface point, loops over the relevant directions reading the link
//       tangents as LDS broadcasts, and the per-link wrench (value + tangents) is combined with
//       wavefront butterfly reductions.
//
// Replaces the per-sub-step C++ of the reference's absent DiffRedMax behind `sim.forward()` /
// `sim.backward_steps()` (envs/redmax_torch_functions.py:132,167).  Formulation: DESIGN.md §Physics.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tsim_blob.h"

#define TS_WAVE 64
// per-link record in LDS (reals)
enum { LK_R = 0, LK_P = 9, LK_W = 12, LK_V = 15, LK_AW = 18, LK_AV = 21, LK_FN = 24, LK_FF = 27, LK_SIZE = 30 };

// ------------------------------------------------------------------------------------------------ dual numbers
template <class R> struct Du {
  R v, d;
  __device__ __forceinline__ Du() {}
  __device__ __forceinline__ Du(R a) : v(a), d(R(0)) {}
  __device__ __forceinline__ Du(R a, R b) : v(a), d(b) {}
};
template <class T> struct RealOf { typedef T type; };
template <class R> struct RealOf<Du<R>> { typedef R type; };

template <class R> __device__ __forceinline__ Du<R> operator+(Du<R> a, Du<R> b) { return Du<R>(a.v + b.v, a.d + b.d); }
template <class R> __device__ __forceinline__ Du<R> operator-(Du<R> a, Du<R> b) { return Du<R>(a.v - b.v, a.d - b.d); }
template <class R> __device__ __forceinline__ Du<R> operator-(Du<R> a) { return Du<R>(-a.v, -a.d); }
template <class R> __device__ __forceinline__ Du<R> operator*(Du<R> a, Du<R> b) { return Du<R>(a.v * b.v, a.d * b.v + a.v * b.d); }
template <class R> __device__ __forceinline__ Du<R> operator*(Du<R> a, R b) { return Du<R>(a.v * b, a.d * b); }
template <class R> __device__ __forceinline__ Du<R> operator/(Du<R> a, Du<R> b) { R iv = R(1) / b.v; R q = a.v * iv; return Du<R>(q, (a.d - q * b.d) * iv); }

__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double t_sqrt(double x) { return sqrt(x); }
template <class R> __device__ __forceinline__ Du<R> t_sqrt(Du<R> a) { R s = t_sqrt(a.v); return Du<R>(s, a.d * (R(0.5) / s)); }
__device__ __forceinline__ void t_sincos(float x, float& s, float& c) { sincosf(x, &s, &c); }
__device__ __forceinline__ void t_sincos(double x, double& s, double& c) { sincos(x, &s, &c); }
template <class R> __device__ __forceinline__ void t_sincos(Du<R> a, Du<R>& s, Du<R>& c) { R sv, cv; t_sincos(a.v, sv, cv); s = Du<R>(sv, a.d * cv); c = Du<R>(cv, -a.d * sv); }
__device__ __forceinline__ float pv(float x) { return x; }
__device__ __forceinline__ double pv(double x) { return x; }
template <class R> __device__ __forceinline__ R pv(Du<R> x) { return x.v; }
__device__ __forceinline__ float t_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double t_abs(double x) { return fabs(x); }

// ------------------------------------------------------------------------------------------------ 3-vectors / 3x3
template <class T> struct V3 { T x, y, z; };
template <class T> __device__ __forceinline__ V3<T> mk3(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <class T> __device__ __forceinline__ V3<T> operator+(V3<T> a, V3<T> b) { return mk3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> __device__ __forceinline__ V3<T> operator-(V3<T> a, V3<T> b) { return mk3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> __device__ __forceinline__ V3<T> operator*(V3<T> a, T s) { return mk3<T>(a.x * s, a.y * s, a.z * s); }
template <class T> __device__ __forceinline__ T dot3(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> __device__ __forceinline__ V3<T> cross3(V3<T> a, V3<T> b) { return mk3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class T> struct M3 { T m[9]; };  // row-major
template <class T> __device__ __forceinline__ V3<T> mulMv(const M3<T>& A, V3<T> v) {
  return mk3<T>(A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z, A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z);
}
template <class T> __device__ __forceinline__ V3<T> mulMtv(const M3<T>& A, V3<T> v) {
  return mk3<T>(A.m[0] * v.x + A.m[3] * v.y + A.m[6] * v.z, A.m[1] * v.x + A.m[4] * v.y + A.m[7] * v.z, A.m[2] * v.x + A.m[5] * v.y + A.m[8] * v.z);
}
template <class T> __device__ __forceinline__ M3<T> mulMM(const M3<T>& A, const M3<T>& B) {
  M3<T> C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}
// products with model constants (plain reals read from the blob): no tangent on the constant side
template <class T, class R> __device__ __forceinline__ V3<T> mulMc(const M3<T>& A, const R* c) {
  return mk3<T>(A.m[0] * c[0] + A.m[1] * c[1] + A.m[2] * c[2], A.m[3] * c[0] + A.m[4] * c[1] + A.m[5] * c[2], A.m[6] * c[0] + A.m[7] * c[1] + A.m[8] * c[2]);
}
template <class T, class R> __device__ __forceinline__ M3<T> mulMcM(const M3<T>& A, const R* c) {
  M3<T> C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * c[j] + A.m[3 * i + 1] * c[3 + j] + A.m[3 * i + 2] * c[6 + j];
  return C;
}

// ------------------------------------------------------------------------------------------------ LDS access
// value part at P[idx]; tangent of direction k at T[idx * nd + k] (direction fastest -> conflict-free for
// lanes = directions, broadcast for lanes = points).
template <class T> struct Lds;
template <> struct Lds<float> {
  static __device__ __forceinline__ float ld(const float* P, const float*, int idx, int, int) { return P[idx]; }
};
template <> struct Lds<double> {
  static __device__ __forceinline__ double ld(const double* P, const double*, int idx, int, int) { return P[idx]; }
};
template <class R> struct Lds<Du<R>> {
  static __device__ __forceinline__ Du<R> ld(const R* P, const R* T, int idx, int nd, int k) { return Du<R>(P[idx], T[idx * nd + k]); }
};
template <class T, class R> __device__ __forceinline__ V3<T> ld3(const R* P, const R* Tg, int idx, int nd, int k) {
  return mk3<T>(Lds<T>::ld(P, Tg, idx, nd, k), Lds<T>::ld(P, Tg, idx + 1, nd, k), Lds<T>::ld(P, Tg, idx + 2, nd, k));
}
template <class T, class R> __device__ __forceinline__ M3<T> ld9(const R* P, const R* Tg, int idx, int nd, int k) {
  M3<T> A;
#pragma unroll
  for (int i = 0; i < 9; ++i) A.m[i] = Lds<T>::ld(P, Tg, idx + i, nd, k);
  return A;
}
template <class R> __device__ __forceinline__ void st(R* P, R* T, int idx, int nd, int k, bool wp, Du<R> x) {
  if (wp) P[idx] = x.v;
  T[idx * nd + k] = x.d;
}
template <class R> __device__ __forceinline__ void st3(R* P, R* T, int idx, int nd, int k, bool wp, V3<Du<R>> x) {
  st(P, T, idx, nd, k, wp, x.x); st(P, T, idx + 1, nd, k, wp, x.y); st(P, T, idx + 2, nd, k, wp, x.z);
}

// ------------------------------------------------------------------------------------------------ wave reductions
// Cross-lane traffic stays in the VALU: DPP row operations (quad_perm / row_mirror / row_bcast) instead of
// ds_bpermute round trips through the LDS crossbar, and v_readlane for broadcasts of a wave-uniform lane.
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dpp_i(int x) {
  return __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xf, false);
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ float dpp_r(float x) {
  return __builtin_bit_cast(float, dpp_i<CTRL, ROWMASK>(__builtin_bit_cast(int, x)));
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_r(double x) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = dpp_i<CTRL, ROWMASK>((int)(b & 0xffffffffll)), hi = dpp_i<CTRL, ROWMASK>((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane_bcast(float x, int lane_uniform) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane_uniform));
}
__device__ __forceinline__ double lane_bcast(double x, int lane_uniform) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane_uniform), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane_uniform);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// sum over the 64 lanes, result in every lane
template <class R> __device__ __forceinline__ R wave_sum(R x) {
  x += dpp_r<0xB1, 0xf>(x);    // quad_perm [1,0,3,2]
  x += dpp_r<0x4E, 0xf>(x);    // quad_perm [2,3,0,1]
  x += dpp_r<0x141, 0xf>(x);   // row_half_mirror
  x += dpp_r<0x140, 0xf>(x);   // row_mirror      -> every lane of a 16-lane row holds the row total
  x += dpp_r<0x142, 0xa>(x);   // row_bcast:15    -> rows 1 and 3 add the total of the row below
  x += dpp_r<0x143, 0xc>(x);   // row_bcast:31    -> rows 2 and 3 add the total of rows 0-1; lane 63 has the sum
  return lane_bcast(x, 63);
}

// ------------------------------------------------------------------------------------------------ per-block context
template <class R> struct Ctx {
  const int* I; const R* F;               // model blob (global memory, shared by all environments)
  int nl, nr, nu, nvar, npair, ncpt, nsensor, ntax, nd;
  int off_link, off_dof, off_motor, off_var, off_pair, off_sensor, off_sprim;
  int foff_link, foff_dof, foff_motor, foff_var, foff_pair, foff_sensor, foff_cpt, foff_tax;
  R h, gx, gy, gz, tol;
  int max_iter, max_ls;
  // LDS
  R *q, *q0, *qd0, *u, *qd, *qa, *g, *dq, *dl, *H, *lamq, *lamv, *z, *rhs;
  R *LP, *LT, *WP, *WT, *scr;
};

// number of LDS reals a block needs (host and device must agree)
__host__ __device__ inline int ts_lds_reals(int nl, int nr, int nu) {
  int nd = nr;
  int n = 0;
  n += 10 * nr + nu;           // q q0 qd0 qd qa g dq dl-base(2) dl ; u
  n += nr * nr;                // H
  n += 4 * nr;                 // lamq lamv z rhs
  n += (nl + 1) * LK_SIZE;     // LP
  n += nr * 6;                 // WP
  n += (nl + 1) * LK_SIZE * nd;  // LT
  n += nr * 6 * nd;            // WT
  n += (nl + 1) * 12;          // scratch (M z pass)
  return n + 8;
}

template <class R> __device__ inline void ctx_init(Ctx<R>& c, const int* I, const R* F, R* lds) {
  c.I = I; c.F = F;
  c.nl = I[TSIM_IH_NL]; c.nr = I[TSIM_IH_NR]; c.nu = I[TSIM_IH_NU]; c.nvar = I[TSIM_IH_NVAR];
  c.npair = I[TSIM_IH_NPAIR]; c.ncpt = I[TSIM_IH_NCPT]; c.nsensor = I[TSIM_IH_NSENSOR]; c.ntax = I[TSIM_IH_NTAXEL];
  c.nd = c.nr;
  c.off_link = I[TSIM_IH_OFF_LINK]; c.off_dof = I[TSIM_IH_OFF_DOF]; c.off_motor = I[TSIM_IH_OFF_MOTOR];
  c.off_var = I[TSIM_IH_OFF_VAR]; c.off_pair = I[TSIM_IH_OFF_PAIR]; c.off_sensor = I[TSIM_IH_OFF_SENSOR];
  c.off_sprim = I[TSIM_IH_OFF_SPRIM];
  c.foff_link = I[TSIM_IH_FOFF_LINK]; c.foff_dof = I[TSIM_IH_FOFF_DOF]; c.foff_motor = I[TSIM_IH_FOFF_MOTOR];
  c.foff_var = I[TSIM_IH_FOFF_VAR]; c.foff_pair = I[TSIM_IH_FOFF_PAIR]; c.foff_sensor = I[TSIM_IH_FOFF_SENSOR];
  c.foff_cpt = I[TSIM_IH_FOFF_CPT]; c.foff_tax = I[TSIM_IH_FOFF_TAXEL];
  c.h = F[TSIM_FH_H]; c.gx = F[TSIM_FH_GX]; c.gy = F[TSIM_FH_GY]; c.gz = F[TSIM_FH_GZ]; c.tol = F[TSIM_FH_TOL];
  c.max_iter = I[TSIM_IH_MAX_ITER]; c.max_ls = I[TSIM_IH_MAX_LS];
  int nr = c.nr, nl = c.nl, nd = c.nd;
  R* p = lds;
  c.q = p; p += nr; c.q0 = p; p += nr; c.qd0 = p; p += nr; c.qd = p; p += nr; c.qa = p; p += nr;
  c.g = p; p += nr; c.dq = p; p += 2 * nr; c.dl = p; p += nr; c.u = p; p += c.nu;
  c.H = p; p += nr * nr;
  c.lamq = p; p += nr; c.lamv = p; p += nr; c.z = p; p += nr; c.rhs = p; p += nr;
  c.LP = p; p += (nl + 1) * LK_SIZE;
  c.WP = p; p += nr * 6;
  c.LT = p; p += (nl + 1) * LK_SIZE * nd;
  c.WT = p; p += nr * 6 * nd;
  c.scr = p;
}

// world link: identity pose, zero velocity, gravity as base acceleration, zero wrench; all tangents zero.
template <class R> __device__ inline void init_world(const Ctx<R>& c, int lane) {
  for (int i = lane; i < LK_SIZE; i += TS_WAVE) {
    R v = R(0);
    if (i == 0 || i == 4 || i == 8) v = R(1);
    if (i == LK_AV) v = -c.gx;
    if (i == LK_AV + 1) v = -c.gy;
    if (i == LK_AV + 2) v = -c.gz;
    c.LP[i] = v;
  }
  for (int i = lane; i < LK_SIZE * c.nd; i += TS_WAVE) c.LT[i] = R(0);
}

// ------------------------------------------------------------------------------------------------ contact law
// DiffHand penalty model: d < 0:  fn = (-kn + kd ddot) d,  ft = -min(kt |vt|, mu |fn|) vt/|vt|.
// Force on the point of link A (world frame); link B receives the opposite force at the same point.
template <class T, class R>
__device__ __forceinline__ bool contact_law(int prim, const R* shape, R kn, R kt, R mu, R kd, const M3<T>& RP, V3<T> pP,
                                            V3<T> xw, V3<T> vrel, V3<T>& Fw) {
  V3<T> x = mulMtv(RP, xw - pP);
  T d; V3<T> n;
  if (prim == TSIM_P_PLANE) { d = x.z; n = mk3<T>(T(R(0)), T(R(0)), T(R(1))); }
  else if (prim == TSIM_P_CUBOID) {
    R ex = t_abs(pv(x.x)) - shape[0], ey = t_abs(pv(x.y)) - shape[1], ez = t_abs(pv(x.z)) - shape[2];
    if (ex >= ey && ex >= ez) { R s = pv(x.x) >= R(0) ? R(1) : R(-1); d = x.x * s - T(shape[0]); n = mk3<T>(T(s), T(R(0)), T(R(0))); }
    else if (ey >= ez)        { R s = pv(x.y) >= R(0) ? R(1) : R(-1); d = x.y * s - T(shape[1]); n = mk3<T>(T(R(0)), T(s), T(R(0))); }
    else                      { R s = pv(x.z) >= R(0) ? R(1) : R(-1); d = x.z * s - T(shape[2]); n = mk3<T>(T(R(0)), T(R(0)), T(s)); }
  } else if (prim == TSIM_P_SPHERE) {
    T r2 = dot3(x, x);
    if (pv(r2) < R(1e-24)) return false;
    T r = t_sqrt(r2); d = r - T(shape[0]); n = x * (T(R(1)) / r);
  } else {
    T rho2 = x.x * x.x + x.y * x.y;
    R rho = t_sqrt(pv(rho2));
    R dr = rho - shape[0], dz = t_abs(pv(x.z)) - shape[1];
    if (dr > dz && rho > R(1e-12)) { T rr = t_sqrt(rho2); d = rr - T(shape[0]); T ir = T(R(1)) / rr; n = mk3<T>(x.x * ir, x.y * ir, T(R(0))); }
    else { R s = pv(x.z) >= R(0) ? R(1) : R(-1); d = x.z * s - T(shape[1]); n = mk3<T>(T(R(0)), T(R(0)), T(s)); }
  }
  if (!(pv(d) < R(0))) return false;
  V3<T> xd = mulMtv(RP, vrel);
  T ddot = dot3(n, xd);
  T fn = (T(-kn) + ddot * kd) * d;
  V3<T> vt = xd - n * ddot;
  T vt2 = dot3(vt, vt);
  V3<T> Fl = n * fn;
  R vtn = t_sqrt(pv(vt2));
  R fna = t_abs(pv(fn));
  if (kt * vtn <= mu * fna || vtn < R(1e-14)) {
    Fl = Fl - vt * T(kt);
  } else {
    T fabs_ = pv(fn) >= R(0) ? fn : -fn;
    T s = fabs_ * mu / t_sqrt(vt2);
    Fl = Fl - vt * s;
  }
  Fw = mulMv(RP, Fl);
  return true;
}
