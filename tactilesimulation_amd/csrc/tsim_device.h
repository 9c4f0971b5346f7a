// tsim_device.h — device-side building blocks of the batched tactile-simulation step (gfx950 / CDNA4).
//
// Every device function here and in tsim_eval.h is __forceinline__: a kernel is one function.  (Out-of-line calls would
// pass the LDS pointers of Ctx as flat pointers — slower, and a backend assertion in this toolchain; the AMDGPU inliner
// stops inlining into callers above a basic-block budget unless told so explicitly.)
//
// Execution model: block = ONE 64-LANE WAVEFRONT that carries 64 / LPE ENVIRONMENTS ("slots") of LPE = 64, 32 or 16
// lanes each (template parameter; chosen per launch from the batch size).  An environment's reduced state (q, qd),
// the per-link world transforms / spatial velocities / wrenches and the Newton matrix live in the slot's LDS region
// for the whole kernel (all sub-steps of an env-step / episode); HBM is touched only for the action, the outputs and
// the tape.  Inside the device functions `lane` is the lane index INSIDE the slot; every slot runs the same
// instruction stream on its own environment (the model is shared, so model-dependent control flow is wave-uniform;
// state-dependent decisions are per-slot predicates).  Most phases use only nr <= 16 lanes of a slot (lanes =
// directions), which is why several slots per wavefront pay: the instruction count per environment is what bounds
// this path.  Lane mappings inside one residual evaluation:
//
//   phase 1  (lanes = directions,     link values + exact tangents of link twist / acceleration / inertial wrench w.r.t.
//             root branches together)  dof k, root -> leaf, world-frame spatial algebra (6-vectors); lane k walks only the
//                                     links of the root branch of its dof (host-built schedule in LDS)
//   phase 2  (lanes = pairs,          pair staging in the primitive's frame;
//             lanes = contact points) penalty contact: force + its 3x3 local Jacobians -> per pair the wrench and its
//                                     6 x 12 derivative w.r.t. the pair's relative displacement / twist (fp32; fp64
//                                     evaluates per direction), segmented DPP reductions, applied by lanes = directions
//   phase 3  (lanes = (direction,     projection on the joints, leaf -> root:  g  and  H[:,k]
//             root branch))
//
// Replaces the per-sub-step C++ of the reference's absent DiffRedMax behind `sim.forward()` /
// `sim.backward_steps()` (envs/redmax_torch_functions.py:132,167).  Formulation: DESIGN.md §1.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include "../../include/tsim_blob.h"

#define TS_WAVE 64
// A block of the simulation kernels is ONE wavefront, so a workgroup barrier degenerates to ordering the wavefront's own LDS
// traffic — which the hardware does anyway (the DS instructions of a wavefront execute in issue order).  TS_SYNC() therefore
// only stops the COMPILER from moving memory accesses across it.  __syncthreads() would add a workgroup-scope fence, i.e.
// s_waitcnt vmcnt(0): the lone wavefront then sits out the full latency of every outstanding global store (tape records, outputs)
// and load at each of the ~20 sync points of an evaluation.  -DTS_REAL_BARRIERS restores __syncthreads() (A/B, debugging).
#ifdef TS_REAL_BARRIERS
#define TS_SYNC() __syncthreads()
#else
#define TS_SYNC() asm volatile("" ::: "memory")
#endif
#define TS_PAIR_GROUP 4      // contact pairs staged in LDS at a time
// per-link value record in LDS (reals)
enum { LK_R = 0, LK_P = 9, LK_W = 12, LK_V = 15, LK_AW = 18, LK_AV = 21, LK_FN = 24, LK_FF = 27, LK_C = 30, LK_IC = 33,
       LK_JW = 39, LK_JV = 42, LK_SIZE = 45 };
// per (link, direction) tangent record: d(twist) d(acceleration) d(wrench)
enum { DT_VW = 0, DT_VV = 3, DT_AW = 6, DT_AV = 9, DT_FN = 12, DT_FF = 15, DT_SIZE = 18 };
// staged pair, value part: pose of A in the primitive frame P, relative twist in P, pose of P in the world, wrench sum (P frame)
enum { PP_RPA = 0, PP_PPA = 9, PP_WREL = 12, PP_VREL = 15, PP_RP = 18, PP_PP = 27, PP_WN = 30, PP_WF = 33, PP_SIZE = 36 };
// staged pair, per direction: relative displacement (P frame), d(relative twist), d(wrench sum)
enum { PT_DTH = 0, PT_DRHO = 3, PT_DW = 6, PT_DV = 9, PT_WN = 12, PT_WF = 15, PT_SIZE = 18 };

// Finite tests on the BIT PATTERN, laundered through an empty asm: the translation unit of the statically specialised kernels is built with
// -ffinite-math-only -fno-signed-zeros (tsim_static.h), under which the compiler may assume that no floating-point value is ever a NaN or an
// infinity — x == x, x - x == 0, !(a < b), even a mask test on the bits of a COMPUTED value may then be folded to "finite".  Every place
// that must notice a NaN / inf (a non-finite control, a residual norm that blew up, a bad multiplier in the pivot-free solve) asks these: the
// asm makes the bits opaque to the optimiser.
__device__ __forceinline__ bool ts_finite(float x) {
  unsigned b = __builtin_bit_cast(unsigned, x);
  asm volatile("" : "+v"(b));
  return (b & 0x7f800000u) != 0x7f800000u;
}
__device__ __forceinline__ bool ts_finite(double x) {
  unsigned hi = (unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32);
  asm volatile("" : "+v"(hi));
  return (hi & 0x7ff00000u) != 0x7ff00000u;
}
__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double t_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ void t_sincos(float x, float& s, float& c) { sincosf(x, &s, &c); }
__device__ __forceinline__ void t_sincos(double x, double& s, double& c) { sincos(x, &s, &c); }
__device__ __forceinline__ float t_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double t_abs(double x) { return fabs(x); }
__device__ __forceinline__ float t_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double t_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float t_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double t_min(double a, double b) { return fmin(a, b); }

// ------------------------------------------------------------------------------------------------ 3-vectors / 3x3
template <class T> struct V3 { T x, y, z; };
template <class T> __device__ __forceinline__ V3<T> mk3(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <class T> __device__ __forceinline__ V3<T> zero3() { return mk3<T>(T(0), T(0), T(0)); }
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
template <class T> __device__ __forceinline__ M3<T> mulMtM(const M3<T>& A, const M3<T>& B) {   // A^T B
  M3<T> C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[i] * B.m[j] + A.m[3 + i] * B.m[3 + j] + A.m[6 + i] * B.m[6 + j];
  return C;
}
template <class T> __device__ __forceinline__ V3<T> ldv(const T* p) { return mk3<T>(p[0], p[1], p[2]); }
template <class T> __device__ __forceinline__ M3<T> ldm(const T* p) { M3<T> A;
#pragma unroll
  for (int i = 0; i < 9; ++i) A.m[i] = p[i];
  return A; }
template <class T> __device__ __forceinline__ void stv(T* p, V3<T> v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
template <class T> __device__ __forceinline__ void stm(T* p, const M3<T>& A) {
#pragma unroll
  for (int i = 0; i < 9; ++i) p[i] = A.m[i];
}
// symmetric 3x3 stored as (xx yy zz xy xz yz) times a vector
template <class T> __device__ __forceinline__ V3<T> symv(const T* s, V3<T> v) {
  return mk3<T>(s[0] * v.x + s[3] * v.y + s[4] * v.z, s[3] * v.x + s[1] * v.y + s[5] * v.z, s[4] * v.x + s[5] * v.y + s[2] * v.z);
}

// ------------------------------------------------------------------------------------------------ spatial 6-vectors
// motion vectors (angular; linear) and force vectors (moment; force), world frame, about the world origin
template <class T> struct S6 { V3<T> a, l; };
template <class T> __device__ __forceinline__ S6<T> mk6(V3<T> a, V3<T> l) { S6<T> r; r.a = a; r.l = l; return r; }
template <class T> __device__ __forceinline__ S6<T> zero6() { return mk6<T>(zero3<T>(), zero3<T>()); }
template <class T> __device__ __forceinline__ S6<T> operator+(S6<T> x, S6<T> y) { return mk6<T>(x.a + y.a, x.l + y.l); }
template <class T> __device__ __forceinline__ S6<T> operator-(S6<T> x, S6<T> y) { return mk6<T>(x.a - y.a, x.l - y.l); }
template <class T> __device__ __forceinline__ S6<T> operator*(S6<T> x, T s) { return mk6<T>(x.a * s, x.l * s); }
template <class T> __device__ __forceinline__ T dot6(S6<T> m, S6<T> f) { return dot3(m.a, f.a) + dot3(m.l, f.l); }
template <class T> __device__ __forceinline__ S6<T> crm(S6<T> v, S6<T> m) { return mk6<T>(cross3(v.a, m.a), cross3(v.a, m.l) + cross3(v.l, m.a)); }   // v x m
template <class T> __device__ __forceinline__ S6<T> crf(S6<T> v, S6<T> f) { return mk6<T>(cross3(v.a, f.a) + cross3(v.l, f.l), cross3(v.a, f.l)); }   // v x* f
// precision changes (the pose chain and the penetration depths are kept in double inside the fp32 kernels, see Ctx)
template <class T, class U> __device__ __forceinline__ V3<T> cvt3(V3<U> v) { return mk3<T>((T)v.x, (T)v.y, (T)v.z); }
template <class T, class U> __device__ __forceinline__ M3<T> cvtm(const M3<U>& A) { M3<T> B;
#pragma unroll
  for (int i = 0; i < 9; ++i) B.m[i] = (T)A.m[i];
  return B; }
template <class T, class U> __device__ __forceinline__ V3<T> ldv_as(const U* p) { return mk3<T>((T)p[0], (T)p[1], (T)p[2]); }
template <class T, class U> __device__ __forceinline__ M3<T> ldm_as(const U* p) { M3<T> A;
#pragma unroll
  for (int i = 0; i < 9; ++i) A.m[i] = (T)p[i];
  return A; }
// sin and cos of a joint angle in double, ~45 instructions and branch-free: round-to-nearest multiple of pi/2 removed with three FMAs
// (pi/2 as a 3 x 53-bit sum: exact enough for |x| < ~1e6 rad, i.e. any joint angle a simulation can reach), fdlibm's kernel
// polynomials on [-pi/4, pi/4], quadrant fix-up by selects.  Within 1 - 2 ulp of libm.  The library sincos() is several hundred
// instructions with data-dependent branches (Payne-Hanek path for huge arguments) and sits on the critical path of every
// evaluation (the revolute step of the pose chain, which a lone wavefront executes serially).  -DTS_LIBM_SINCOS restores it (A/B).
__device__ __forceinline__ void t_sincos_d(double x, double& s, double& c) {
#ifdef TS_LIBM_SINCOS
  sincos(x, &s, &c);
#else
  const double kf = rint(x * 6.36619772367581382433e-01);              // 2 / pi
  double r = fma(-kf, 1.57079632679489655800e+00, x);                  // pi/2 = hi + mid + lo
  r = fma(-kf, 6.12323399573676603587e-17, r);
  r = fma(-kf, -1.49738490485916983834e-33, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)kf & 3;
  const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
#endif
}
template <class T> __device__ __forceinline__ S6<T> ld6(const T* p) { return mk6<T>(ldv(p), ldv(p + 3)); }
template <class T> __device__ __forceinline__ void st6(T* p, S6<T> v) { stv(p, v.a); stv(p + 3, v.l); }
// p[0..6) += s * v   (read-modify-write of a 6-vector in LDS)
template <class T> __device__ __forceinline__ void acc6(T* p, S6<T> v, T s) {
  p[0] += s * v.a.x; p[1] += s * v.a.y; p[2] += s * v.a.z; p[3] += s * v.l.x; p[4] += s * v.l.y; p[5] += s * v.l.z;
}
// spatial inertia (mass, world COM c, world rotational inertia about the COM) times a motion vector
template <class T> __device__ __forceinline__ S6<T> imul(T mass, V3<T> c, const T* Ic, S6<T> m) {
  V3<T> f = (m.l + cross3(m.a, c)) * mass;
  return mk6<T>(symv(Ic, m.a) + cross3(c, f), f);
}
// world twist about the world origin -> frame P (pose R, p) about P's origin, and the dual map for wrenches
template <class T> __device__ __forceinline__ S6<T> to_frame(const M3<T>& R, V3<T> p, S6<T> m) {
  return mk6<T>(mulMtv(R, m.a), mulMtv(R, m.l + cross3(m.a, p)));
}
template <class T> __device__ __forceinline__ S6<T> wrench_to_world(const M3<T>& R, V3<T> p, S6<T> w) {
  V3<T> f = mulMv(R, w.l);
  return mk6<T>(mulMv(R, w.a) + cross3(p, f), f);
}

// ------------------------------------------------------------------------------------------------ SO(3) exponential joint
// Small forward-mode jets, used ONLY inside the rotation-vector joint to differentiate the 3x3 left Jacobian J_l(theta)
// (first derivatives w.r.t. theta, and their directional derivative along thetadot). Everything else in the kernels
// propagates tangents analytically.
template <class T, int N> struct Jet { T v; T d[N]; };
__device__ __forceinline__ float t_sin(float x) { return sinf(x); }
__device__ __forceinline__ double t_sin(double x) { return sin(x); }
__device__ __forceinline__ float t_cos(float x) { return cosf(x); }
__device__ __forceinline__ double t_cos(double x) { return cos(x); }
__device__ __forceinline__ float jval(float x) { return x; }
__device__ __forceinline__ double jval(double x) { return x; }
template <class T, int N> __device__ __forceinline__ auto jval(const Jet<T, N>& a) { return jval(a.v); }
template <class T> struct JetC { static __device__ __forceinline__ T c(double x) { return T(x); } };
template <class T, int N> struct JetC<Jet<T, N>> {
  static __device__ __forceinline__ Jet<T, N> c(double x) { Jet<T, N> r; r.v = JetC<T>::c(x);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = JetC<T>::c(0.0);
    return r; }
};
template <class T, int N> __device__ __forceinline__ Jet<T, N> operator+(const Jet<T, N>& a, const Jet<T, N>& b) { Jet<T, N> r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
  return r; }
template <class T, int N> __device__ __forceinline__ Jet<T, N> operator-(const Jet<T, N>& a, const Jet<T, N>& b) { Jet<T, N> r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
  return r; }
template <class T, int N> __device__ __forceinline__ Jet<T, N> operator*(const Jet<T, N>& a, const Jet<T, N>& b) { Jet<T, N> r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r; }
template <class T, int N> __device__ __forceinline__ Jet<T, N> operator/(const Jet<T, N>& a, const Jet<T, N>& b) { Jet<T, N> r; r.v = a.v / b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
  return r; }
template <class T, int N> __device__ __forceinline__ Jet<T, N> t_sqrt(const Jet<T, N>& a) { Jet<T, N> r; r.v = t_sqrt(a.v); const T k = JetC<T>::c(0.5) / r.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k;
  return r; }
template <class T, int N> __device__ __forceinline__ Jet<T, N> t_sin(const Jet<T, N>& a) { Jet<T, N> r; r.v = t_sin(a.v); const T c = t_cos(a.v);
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c;
  return r; }
template <class T, int N> __device__ __forceinline__ Jet<T, N> t_cos(const Jet<T, N>& a) { Jet<T, N> r; r.v = t_cos(a.v); const T s = JetC<T>::c(0.0) - t_sin(a.v);
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s;
  return r; }
// exp([th]) = I + s1 [th]x + s2 [th]x^2 ;  J_l(th) = I + s2 [th]x + c2 [th]x^2   (omega_spatial = J_l thdot)
template <class U> __device__ __forceinline__ void so3_coeffs(const U& x, const U& y, const U& z, U& s1, U& s2, U& c2) {
  const U p2 = x * x + y * y + z * z;
  if (jval(p2) < 1e-8) {          // series in phi^2, differentiable through the jets
    s1 = JetC<U>::c(1.0) - p2 * JetC<U>::c(1.0 / 6) + p2 * p2 * JetC<U>::c(1.0 / 120);
    s2 = JetC<U>::c(0.5) - p2 * JetC<U>::c(1.0 / 24) + p2 * p2 * JetC<U>::c(1.0 / 720);
    c2 = JetC<U>::c(1.0 / 6) - p2 * JetC<U>::c(1.0 / 120) + p2 * p2 * JetC<U>::c(1.0 / 5040);
  } else {
    const U p = t_sqrt(p2);
    s1 = t_sin(p) / p; s2 = (JetC<U>::c(1.0) - t_cos(p)) / p2; c2 = (p - t_sin(p)) / (p2 * p);
  }
}
template <class U> __device__ __forceinline__ void so3_mat(const U& x, const U& y, const U& z, const U& a, const U& b, U* M) {
  const U one = JetC<U>::c(1.0);
  M[0] = one - b * (y * y + z * z); M[1] = b * x * y - a * z;        M[2] = b * x * z + a * y;
  M[3] = b * x * y + a * z;        M[4] = one - b * (x * x + z * z); M[5] = b * y * z - a * x;
  M[6] = b * x * z - a * y;        M[7] = b * y * z + a * x;        M[8] = one - b * (x * x + y * y);
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
__device__ __forceinline__ int lane_bcast(int x, int lane_uniform) { return __builtin_amdgcn_readlane(x, lane_uniform); }
__device__ __forceinline__ float lane_bcast(float x, int lane_uniform) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane_uniform));
}
__device__ __forceinline__ double lane_bcast(double x, int lane_uniform) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane_uniform), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane_uniform);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// value of x in wavefront lane src (per-lane source, any pattern): LDS crossbar, no memory access
__device__ __forceinline__ int lane_gather(int x, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, x); }
__device__ __forceinline__ float lane_gather(float x, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, x)));
}
__device__ __forceinline__ double lane_gather(double x, int src) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b & 0xffffffffll)), hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// sum over the LPE lanes of the caller's slot, result in every lane of the slot
template <int LPE, class R> __device__ __forceinline__ R seg_sum(R x) {
  x += dpp_r<0xB1, 0xf>(x);    // quad_perm [1,0,3,2]
  x += dpp_r<0x4E, 0xf>(x);    // quad_perm [2,3,0,1]
  x += dpp_r<0x141, 0xf>(x);   // row_half_mirror
  x += dpp_r<0x140, 0xf>(x);   // row_mirror      -> every lane of a 16-lane row holds the row total
  if (LPE == 16) return x;
  if (LPE == 32) return x + lane_gather(x, (int)threadIdx.x ^ 16);     // the other row of the slot
  x += dpp_r<0x142, 0xa>(x);   // row_bcast:15    -> rows 1 and 3 add the total of the row below
  x += dpp_r<0x143, 0xc>(x);   // row_bcast:31    -> rows 2 and 3 add the total of rows 0-1; lane 63 has the sum
  return lane_bcast(x, 63);
}
// N independent sums at once, STEP by step over all of them: the N chains interleave, so no DPP instruction waits for the one before it.
// (Written per value, the scheduler of a register-hungry kernel serialises each chain through one temporary: four dependent DPP adds with
// their hazard nops per value — k_forward<float, 8, false, 32> after round 4's loop change: 121 extra s_nop, 8 % of the kernel's time.)
// The scheduling barriers keep the steps apart.
template <int LPE, int N, class R> __device__ __forceinline__ void seg_sum_many(R (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_r<0xB1, 0xf>(v[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_r<0x4E, 0xf>(v[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_r<0x141, 0xf>(v[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_r<0x140, 0xf>(v[i]);
  __builtin_amdgcn_sched_barrier(0);
  if (LPE == 16) return;
  if (LPE == 32) {
    R o[N];
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = lane_gather(v[i], (int)threadIdx.x ^ 16);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += o[i];
    return;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_r<0x142, 0xa>(v[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_r<0x143, 0xc>(v[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = lane_bcast(v[i], 63);
}
// sum over lanes 0..7 of each 8-lane half of a 16-lane row (three DPP steps): for per-pair sums whose points all sit in the first 8 lanes of
// the slot (the other lanes hold zeros), result in lanes 0..7
template <class R> __device__ __forceinline__ R half_row_sum(R x) {
  x += dpp_r<0xB1, 0xf>(x);    // quad_perm [1,0,3,2]
  x += dpp_r<0x4E, 0xf>(x);    // quad_perm [2,3,0,1]
  x += dpp_r<0x141, 0xf>(x);   // row_half_mirror
  return x;
}
// maximum over the LPE lanes of the caller's slot, result in every lane of the slot (x >= 0)
template <int LPE, class R> __device__ __forceinline__ R seg_max(R x) {
  x = t_max(x, dpp_r<0xB1, 0xf>(x));
  x = t_max(x, dpp_r<0x4E, 0xf>(x));
  x = t_max(x, dpp_r<0x141, 0xf>(x));
  x = t_max(x, dpp_r<0x140, 0xf>(x));
  if (LPE == 16) return x;
  if (LPE == 32) return t_max(x, lane_gather(x, (int)threadIdx.x ^ 16));
  x = t_max(x, lane_gather(x, (int)threadIdx.x ^ 16));
  return t_max(x, lane_gather(x, (int)threadIdx.x ^ 32));
}
// value of x in lane src (inside the slot; the same for all lanes of a slot) of the caller's slot
template <int LPE, class R> __device__ __forceinline__ R seg_bcast(R x, int src) {
  if (LPE == TS_WAVE) return lane_bcast(x, __builtin_amdgcn_readfirstlane(src));
  return lane_gather(x, ((int)threadIdx.x & ~(LPE - 1)) + src);
}

// A value that is the same in every lane of the wavefront (model constants read from the staged tables), moved to a scalar
// register explicitly.  The tables sit in LDS, and what comes back from a ds_read is a vector register the compiler does not
// always prove uniform: branches on it are then built from exec masks (v_cmp + s_and_saveexec + s_xor + s_or per arm, every arm
// visited) instead of one s_cbranch — the primitive-type switch of contact_law alone was ~40 such instructions per chunk of contact
// points.  A lone wavefront pays 4 cycles for every instruction it issues, scalar or vector.
__device__ __forceinline__ int ts_u(int x) { return __builtin_amdgcn_readfirstlane(x); }

// ------------------------------------------------------------------------------------------------ per-block context
template <class R> struct Ctx {
  const int* I; const R* F;               // model records (int blob; link / dof / motor / pair / sensor float tables): staged in LDS
  const R* Fg;                            // whole float blob in global memory (taxel SoA arrays; large contact-point arrays)
  const R* CPT;                           // contact-point SoA arrays x[] y[] z[] in global memory ...
  __attribute__((address_space(3))) const R* CPTl; bool cpt_lds;   // ... and their LDS copy when staged (typed as an LDS pointer).  Two pointers, one flag: a single pointer that may be either
                                          // makes every point load a flat_load (vector-memory latency, waits on vmcnt) instead of a ds_read
  int nl, nr, nu, nvar, npair, ncpt, nsensor, ntax, nd;
  int off_link, off_dof, off_motor, off_var, off_pair, off_sensor, off_sprim;
  int foff_link, foff_dof, foff_motor, foff_var, foff_pair, foff_sensor, foff_cpt, foff_tax;
  R h, gx, gy, gz, tol;
  R cv, ca;                               // qd = qdp + cv*dl, qdd = ca*dl (BDF1: 1/h, 1/h^2; BDF2: 3/(2h), 9/(4h^2)); g = r/ca
  int max_iter, max_ls;
  // LDS
  R *q, *q0, *qd0, *u, *qd, *qa, *g, *dq, *dl, *H, *H2, *lamq, *lamv, *z, *rhs;
  R *qp, *qdp, *qm1, *qdm1;               // predictor of the implicit step; state before the previous sub-step (BDF2)
  R *expw;                                // rotation-vector joint: d W_m / d theta_k, 9 x 6 reals
  R *LP, *WP, *DT, *PP, *PT, *scr;
  // High-precision side of the geometry (double also in the fp32 kernels): a penetration depth of 1e-4 m is the
  // difference of positions of 0.1 ... 0.3 m — 2e-4 relative error if the pose chain is fp32, which is what flips
  // contact / friction branches over a long roll-out (profiles/r01_fp32_vs_fp64_scale.json).  Positions q, link poses,
  // staged pair poses and the point -> primitive-frame transform are double; velocities, forces, tangents are R.
  double *qD, *q0D, *qpD, *qm1D;          // q1 = qpD + dl, state before the sub-step, predictor, state before that (BDF2)
  double *LPd;                            // per link: R (9) + p (3)
  double *PPd;                            // per staged pair: R_PA (9) + p_PA (3)
  const int* LI;                          // sweep schedule + per-link int records in LDS (ts_sched layout below)
  int cull;               // phase 2 skips contact pairs whose bounding sphere is clear of the primitive (tsim_set_option TSIM_OPT_PAIR_CULL)
  long long* stamps;      // optional per-env array of shader-clock stamps (debug kernel only), else null
  mutable int nstamp;
};
#ifdef TS_ISA_MARKS        // static analysis only (tools/isa_phases.py): every stamp site becomes a unique s_sleep marker in the ISA
#define TS_MARK_(n) asm volatile("s_sleep %0" ::"n"(n) : "memory")
#define TS_STAMP(c) TS_MARK_(__COUNTER__ + 20)
#define TS_STAMP2(c) TS_MARK_(__COUNTER__ + 20)
#else
#ifdef TS_FINE_STAMPS      // A/B builds only (tools/fine_stamps.py): extra stamps inside the phases
#define TS_STAMP2(c) TS_STAMP(c)
#else
#define TS_STAMP2(c) do { } while (0)
#endif
#define TS_STAMP(c) do { if ((c).stamps) { if (threadIdx.x == 0 && (c).nstamp < 32) (c).stamps[(c).nstamp] = clock64(); (c).nstamp++; } } while (0)
#endif

// LDS reals of one environment's state (host and device must agree)
__host__ __device__ inline int ts_lds_env_doubles(int nl, int nr) { return 4 * nr + (nl + 1) * 12 + TS_PAIR_GROUP * 12; }
__host__ __device__ inline int ts_lds_env_reals(int nl, int nr, int nu, int esz) {
  const int nd = nr;
  int n = ts_lds_env_doubles(nl, nr) * (8 / esz);   // the double block comes first (8-byte aligned)
  n += 15 * nr + nu;                       // q q0 qd0 qd qa g dq(2) dl(2) qp qdp qm1 qdm1 spare ; u
  n += 2 * nr * nr;                        // H, H2 (taped Newton matrix in the adjoint kernel)
  n += 4 * nr;                             // lamq lamv z rhs
  n += (nl + 1) * LK_SIZE;                 // LP
  n += nr * 6;                             // WP
  n += (nl + 1) * nd * DT_SIZE;            // DT
  n += TS_PAIR_GROUP * PP_SIZE;            // PP
  n += TS_PAIR_GROUP * nd * PT_SIZE;       // PT
  n += 16 + 54;                            // scratch, expw
  return (n + 8 + 3) & ~3;                 // 16-byte multiples keep every slot's records equally aligned
}
// LDS reals of a block of nslot environments. nfrec: leading reals of the model blob that are staged in LDS (everything
// except the per-point SoA arrays); one copy per block, or one per slot with per-environment tables.
// The contact-point SoA arrays (3 ncpt reals right behind the tables in the blob) are staged with the shared tables when
// the host says so (kernel argument stage_cpt, decided per launch: they are small and do not cost the launch its lanes-per-environment shape):
// every residual evaluation reads all of them, and a lone wavefront cannot hide ~600-cycle L2 latencies.
#define TS_CPT_LDS_BYTES 8192
// (Round 4: also next to per-environment tables — the tables hold the float RECORDS of an environment; the contact-point arrays are geometry
// of the general bodies, shared by every environment, and one copy per block serves all its slots.  Measured on D'Claw collection with 16
// randomised variants: 0.63 M env-steps/s either way (TSIM_NO_ENVTAB_CPT=1 restores the global loads) — what randomisation costs there is
// Newton effort on the randomised models, 871 of 204 800 env-steps at the evaluation budget, not these loads.)
__host__ __device__ inline int ts_cpt_staged(int ncpt, bool env_tables, bool stage) {
  (void)env_tables;
  return stage ? 3 * ncpt : 0;
}
// reals of the staged-table region of a block, a multiple of 4 (what follows holds doubles: keep it 8-byte aligned)
__host__ __device__ inline int ts_tab_reals(int nfrec, int ncpt, int nslot, bool env_tables, bool stage_cpt) {
  return ((env_tables ? nslot : 1) * (nfrec + 2) + ts_cpt_staged(ncpt, env_tables, stage_cpt) + 3) & ~3;
}
__host__ __device__ inline int ts_lds_reals(int nl, int nr, int nu, int nfrec, int ncpt, bool stage_cpt, int nslot, bool env_tables, int nsched, int esz) {
  return ts_tab_reals(nfrec, ncpt, nslot, env_tables, stage_cpt)
       + ((nsched * 4 + esz - 1) / esz + 3) / 4 * 4 + nslot * ts_lds_env_reals(nl, nr, nu, esz) + 8;
}

// Sweep schedule (built on the host from the link parents, appended to the device copy of the int blob at I[TSIM_IH_NI]):
// the links of different root branches (sub-trees hanging off the world) are independent in the root->leaf sweep, so
// lane k < nr walks only the links of the branch of its own dof k — all branches advance together, and the sweep takes
// max(branch size) steps instead of nl.
//   S[0] = number of ints, S[1] = steps, S[2 + l] = branch of lane l (l < 16; -1: lane has no dof),
//   S[18 + b] = leader lane of branch b, S[34] = number of branches, S[35] = offset of the taxel staging table (ts_tax_table),
//   S[36] = offset of the contact pairs' bounding spheres (ts_pair_bound),
//   S[TS_SCHED_ENT + step * 16 + l] = link visited by lane l at that step | leader << 8 (0: none; the leader lane of a
//   branch stores the link's value record), then per link 8 ints: parent, joint type, dof0, ndof, ancestor mask, branch.
// The leaf->root projection (phase 3) uses the same lists backwards, one lane per (direction, branch).
enum { TS_SCHED_BRANCH = 2, TS_SCHED_LEADER = 18, TS_SCHED_NB = 34, TS_SCHED_TAXTAB = 35, TS_SCHED_PBOUND = 36, TS_SCHED_ENT = 37, TS_LR_PARENT = 0, TS_LR_JTYPE, TS_LR_DOF0, TS_LR_NDOF, TS_LR_ANCMASK, TS_LR_BRANCH, TS_LR_SIZE = 8 };
__device__ __forceinline__ int ts_sched_rec(const int* S) { return TS_SCHED_ENT + S[1] * 16; }
// ... followed by a copy of the contact-pair int records (TSIM_PI_*), for the lanes = pairs staging of phase 2
template <class C> __device__ __forceinline__ const int* ts_pair_rec(const C& c, int pk) {
  return c.LI + ts_sched_rec(c.LI) + c.nl * TS_LR_SIZE + pk * TSIM_PI_SIZE;
}
// ... then 16 ints: the motor acting on dof j (-1: none, -2: several — walk the records), and a copy of the motor int
// records (TSIM_MI_*): the joint-space part of phase 3 and the adjoint's dL/du read them per lane, and a lone wavefront
// cannot hide the latency of dependent global loads
template <class C> __device__ __forceinline__ const int* ts_dof_motor(const C& c) {
  return c.LI + ts_sched_rec(c.LI) + c.nl * TS_LR_SIZE + c.npair * TSIM_PI_SIZE;
}
template <class C> __device__ __forceinline__ const int* ts_motor_rec(const C& c, int m) { return ts_dof_motor(c) + 16 + m * TSIM_MI_SIZE; }
// ... and, last, the taxel staging table of k_taxels (tsim_readout): per sensor 3 ints (end of its taxel range, first (sensor,
// primitive) record, number of records), then per record 2 ints (primitive type, contact pair)
__device__ __forceinline__ const int* ts_tax_table(const int* S) { return S + S[TS_SCHED_TAXTAB]; }
// ... and per contact pair 4 floats: the bounding sphere of the pair's contact points in the frame of link A (centre, radius; radius < 0: no
// bound — the moving contact point of a sphere on a plane).  Built by the host from the contact-point arrays (build_sched); phase 2 skips a pair
// whose sphere is farther from the primitive than its radius in every environment of the wavefront (pair_stage_value, phase2).
template <class C> __device__ __forceinline__ const float* ts_pair_bound(const C& c, int pk) {
  return reinterpret_cast<const float*>(c.LI + c.LI[TS_SCHED_PBOUND]) + 4 * pk;
}

// A wavefront is about to read global memory it (or a wavefront of its CU) stored to earlier in the launch: wait until the stores
// have reached L2 (s_waitcnt vmcnt(0): one CU, one XCD, one L2), then drop the vector L1's lines (buffer_inv sc1) — the L1 keeps a
// line it fetched before the store.  __threadfence() does that too, but releases at device scope first: `buffer_wbl2 sc1`, a write-back
// of the XCD's L2 for the benefit of the other XCDs, which nobody here needs (1.5 % of the closed-loop epoch;
// profiles/r03_closed_loop_policy.md).
__device__ __forceinline__ void ts_own_stores_visible() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// LDS layout of a block: [model float tables: one copy, or one per slot when the environments have their own tables]
// [slot 0 state][slot 1 state]...   (ts_lds_env_reals each)
// An int of the model blob's header for the context: read from the blob, or — for a kernel instantiated for a compiled-in model (MS: tsim_static.h), whose
// batch carries that model's ints (tsim_hip.hip blob_has_structure; the taxel layout excepted) — the compiled-in constant: the sizes, offsets and LDS array
// offsets of the context then fold, and the scalar registers that held them are free (round 6).
template <class MS> __device__ __forceinline__ int ts_ctx_int(const int* I, int idx) {
  if constexpr (std::is_void<MS>::value) return I[idx]; else return MS::Iv(idx);
}
template <class R, class MS = void> __device__ __forceinline__ void ctx_init(Ctx<R>& c, const int* I, const R* F, R* lds, int nslot, int slot, int lane, int lpe, bool stage_cpt, const R* Fenv = nullptr) {
#define TS_IV(idx) ts_ctx_int<MS>(I, idx)
  // Stage the model's tables in LDS — the float records (link / dof / motor / pair / sensor), the sweep schedule and the whole
  // int blob: later reads are ds_read broadcasts instead of ~500-cycle global loads (a lone wavefront cannot hide those).
  const int nfrec = TS_IV(TSIM_IH_FOFF_CPT);
  {
    R* mf = lds;
    R* cpt_l = lds + nfrec;                      // shared tables: the contact points follow the records, as in the blob
    if (Fenv) {                                  // per-environment float tables (domain randomisation): one copy per slot ...
      mf += slot * (nfrec + 2);
      for (int i = lane; i < nfrec; i += lpe) mf[i] = Fenv[i];
      cpt_l = lds + nslot * (nfrec + 2);         // ... and ONE copy of the (shared) contact-point arrays behind them
      const int nc = ts_cpt_staged(TS_IV(TSIM_IH_NCPT), true, stage_cpt);
      for (int i = threadIdx.x; i < nc; i += TS_WAVE) cpt_l[i] = F[nfrec + i];
    } else {
      const int nst = nfrec + ts_cpt_staged(TS_IV(TSIM_IH_NCPT), false, stage_cpt);   // tables (+ contact points)
      for (int i = threadIdx.x; i < nst; i += TS_WAVE) mf[i] = F[i];
    }
    lds += ts_tab_reals(nfrec, TS_IV(TSIM_IH_NCPT), nslot, Fenv != nullptr, stage_cpt);
    {                                            // sweep schedule + the whole int blob (one copy per block)
      // The int tables are wave-uniform, but loads through a plain global pointer are not scalar loads here (the kernel
      // also writes global memory, so the compiler issues global_load_dword + s_waitcnt vmcnt(0) + v_readfirstlane): ~400
      // cycles of exposed latency each for a lone wavefront, a dozen times per evaluation.  From LDS they cost a ds_read.
      const int* S = I + TS_IV(TSIM_IH_NI);
      const int ns = S[0], ni = TS_IV(TSIM_IH_NI);
      int* li = reinterpret_cast<int*>(lds);
      for (int i = threadIdx.x; i < ns; i += TS_WAVE) li[i] = S[i];
      for (int i = threadIdx.x; i < ni; i += TS_WAVE) li[ns + i] = I[i];
      c.LI = li;
      c.I = li + ns;
      lds += (((ns + ni) * 4 + (int)sizeof(R) - 1) / (int)sizeof(R) + 3) / 4 * 4;
    }
    TS_SYNC();
    c.Fg = F; c.F = mf;
    c.cpt_lds = ts_u(ts_cpt_staged(TS_IV(TSIM_IH_NCPT), Fenv != nullptr, stage_cpt)) != 0;
    c.CPT = F + TS_IV(TSIM_IH_FOFF_CPT);
    c.CPTl = (__attribute__((address_space(3))) const R*)(c.cpt_lds ? cpt_l : mf);
    F = mf;
  }
  c.stamps = nullptr; c.nstamp = 0; c.cull = 0;
  c.nl = TS_IV(TSIM_IH_NL); c.nr = TS_IV(TSIM_IH_NR); c.nu = TS_IV(TSIM_IH_NU); c.nvar = TS_IV(TSIM_IH_NVAR);
  c.npair = TS_IV(TSIM_IH_NPAIR); c.ncpt = TS_IV(TSIM_IH_NCPT); c.nsensor = TS_IV(TSIM_IH_NSENSOR); c.ntax = I[TSIM_IH_NTAXEL];
  c.nd = c.nr;
  c.off_link = TS_IV(TSIM_IH_OFF_LINK); c.off_dof = TS_IV(TSIM_IH_OFF_DOF); c.off_motor = TS_IV(TSIM_IH_OFF_MOTOR);
  c.off_var = TS_IV(TSIM_IH_OFF_VAR); c.off_pair = TS_IV(TSIM_IH_OFF_PAIR); c.off_sensor = TS_IV(TSIM_IH_OFF_SENSOR);
  c.off_sprim = TS_IV(TSIM_IH_OFF_SPRIM);
  c.foff_link = TS_IV(TSIM_IH_FOFF_LINK); c.foff_dof = TS_IV(TSIM_IH_FOFF_DOF); c.foff_motor = TS_IV(TSIM_IH_FOFF_MOTOR);
  c.foff_var = TS_IV(TSIM_IH_FOFF_VAR); c.foff_pair = TS_IV(TSIM_IH_FOFF_PAIR); c.foff_sensor = TS_IV(TSIM_IH_FOFF_SENSOR);
  c.foff_cpt = TS_IV(TSIM_IH_FOFF_CPT); c.foff_tax = TS_IV(TSIM_IH_FOFF_TAXEL);
  c.h = F[TSIM_FH_H]; c.gx = F[TSIM_FH_GX]; c.gy = F[TSIM_FH_GY]; c.gz = F[TSIM_FH_GZ]; c.tol = F[TSIM_FH_TOL];
  // (the header FLOATS stay run-time reads also for a fully static model: folding them made the forward kernel 3 % slower — scheduling noise
  // of a 50 KB straight-line loop, measured in round 6)
  c.max_iter = TS_IV(TSIM_IH_MAX_ITER); c.max_ls = TS_IV(TSIM_IH_MAX_LS);
  c.cv = R(1) / c.h; c.ca = R(1) / (c.h * c.h);
  int nr = c.nr, nl = c.nl, nd = c.nd;
  R* p = lds + slot * ts_lds_env_reals(nl, nr, c.nu, (int)sizeof(R));
  {
    double* d = reinterpret_cast<double*>(p);
    c.qD = d; d += nr; c.q0D = d; d += nr; c.qpD = d; d += nr; c.qm1D = d; d += nr;
    c.LPd = d; d += (nl + 1) * 12;
    c.PPd = d; d += TS_PAIR_GROUP * 12;
    p = reinterpret_cast<R*>(d);
  }
  c.q = p; p += nr; c.q0 = p; p += nr; c.qd0 = p; p += nr; c.qd = p; p += nr; c.qa = p; p += nr;
  c.g = p; p += nr; c.dq = p; p += 2 * nr; c.dl = p; p += 2 * nr;
  c.qp = p; p += nr; c.qdp = p; p += nr; c.qm1 = p; p += nr; c.qdm1 = p; p += nr; c.u = p; p += c.nu;
  c.H = p; p += nr * nr; c.H2 = p; p += nr * nr;
  c.lamq = p; p += nr; c.lamv = p; p += nr; c.z = p; p += nr; c.rhs = p; p += nr;
  c.LP = p; p += (nl + 1) * LK_SIZE;
  c.WP = p; p += nr * 6;
  c.DT = p; p += (nl + 1) * nd * DT_SIZE;
  c.PP = p; p += TS_PAIR_GROUP * PP_SIZE;
  c.PT = p; p += TS_PAIR_GROUP * nd * PT_SIZE;
  c.expw = p; p += 54;
  c.scr = p;
}
#undef TS_IV

// world link: identity pose, zero velocity, gravity as base acceleration, zero wrench; all tangents zero.
template <class R> __device__ __forceinline__ void init_world(const Ctx<R>& c, int lane, int lpe) {
  for (int i = lane; i < LK_SIZE; i += lpe) {
    R v = R(0);
    if (i == 0 || i == 4 || i == 8) v = R(1);
    if (i == LK_AV) v = -c.gx;
    if (i == LK_AV + 1) v = -c.gy;
    if (i == LK_AV + 2) v = -c.gz;
    c.LP[i] = v;
  }
  for (int i = lane; i < (c.nl + 1) * c.nd * DT_SIZE; i += lpe) c.DT[i] = R(0);    // incl. the (link, dof) records no sweep writes
  for (int i = lane; i < 12; i += lpe) c.LPd[i] = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
}

// contact point i (SoA planes x, y, z) from its LDS copy or from global memory (wave-uniform choice)
template <class R> __device__ __forceinline__ V3<R> ld_cpt(const Ctx<R>& c, int i) {
  if (c.cpt_lds) return mk3<R>(c.CPTl[i], c.CPTl[i + c.ncpt], c.CPTl[i + 2 * c.ncpt]);
  return mk3<R>(c.CPT[i], c.CPT[i + c.ncpt], c.CPT[i + 2 * c.ncpt]);
}

__device__ __forceinline__ int anc_of(const int* I, int off_link, int link) {
  return link > 0 ? I[off_link + (link - 1) * TSIM_LI_SIZE + TSIM_LI_ANCMASK] : 0;
}

// ------------------------------------------------------------------------------------------------ branch signature
// Per (environment, sub-step): which contact points / taxels penetrate and on which smooth piece of the penalty law each
// of them is (tsim_debug_signature in include/tsim.h; the oracle computes the same two numbers).  Item id = (pair or
// sensor-primitive list index, point index); term = mix(id, code), code = 1 + branch of contact_law; the signature is
// (number of penetrating items, sum of terms mod 2^32) — commutative, so lanes can add in any order.
__host__ __device__ inline unsigned ts_sig_mix(unsigned group, unsigned index, unsigned code) {
  unsigned x = (group * 0x9E3779B1u) ^ (index * 0x85EBCA77u) ^ (code * 0xC2B2AE3Du);
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

// ------------------------------------------------------------------------------------------------ contact law
// DiffHand penalty model in the primitive's frame: d < 0:  fn = (-kn + kd ddot) d,  ft = -min(kt |vt|, mu |fn|) vt/|vt|.
// x = point in the primitive frame, v = its velocity relative to the primitive (same frame).
// Returns the force on the point; with JAC also Jx = dF/dx and Jv = dF/dv (exact: the law is piecewise smooth).
// xh: the same point in double — the signed distance (a small difference of large numbers) is taken from it.
// branch (diagnostics, tsim_debug_signature): which smooth piece of the law the point is on — bit 0 = sticking, bits 1.. =
// face of the primitive (cuboid: 2 axis + (negative side); cylinder: 0 side, 1 / 2 caps; plane, sphere: 0).
// Signed distance of a point to a primitive in the kernel's own precision: a cheap "certainly outside" test in front of contact_law
// (which measures the distance from a double-precision position): callers skip points whose R-precision distance exceeds
// TS_FAR_MARGIN, far more than the rounding of an R-precision pose product, so the set of points contact_law accepts is unchanged.
#define TS_FAR_MARGIN 1e-4
template <class R>
__device__ __forceinline__ R prim_distance(int prim, const R* shape, V3<R> x) {
  if (prim == TSIM_P_PLANE) return x.z;
  if (prim == TSIM_P_CUBOID) return t_max(t_abs(x.x) - shape[0], t_max(t_abs(x.y) - shape[1], t_abs(x.z) - shape[2]));
  if (prim == TSIM_P_SPHERE) return t_sqrt(dot3(x, x)) - shape[0];
  return t_max(t_sqrt(x.x * x.x + x.y * x.y) - shape[0], t_abs(x.z) - shape[1]);        // cylinder: inside iff both are negative
}

template <class R, bool JAC>
__device__ __forceinline__ bool contact_law(int prim, const R* shape, const R* kp, V3<R> x, V3<R> v, V3<R>& F, M3<R>& Jx, M3<R>& Jv, V3<double> xh, int* branch = nullptr, bool jac = true) {
  const R kn = kp[0], kt = kp[1], mu = kp[2], kd = kp[3];
  R d; V3<R> n;
  R ncurv = R(0);          // N = dn/dx = ncurv * (Pm - n n^T), Pm = diag(1, 1, pz)
  R pz = R(1);
  int face = 0;
  if (prim == TSIM_P_PLANE) { d = (R)xh.z; n = mk3<R>(R(0), R(0), R(1)); }
  else if (prim == TSIM_P_CUBOID) {
    const double ex = fabs(xh.x) - (double)shape[0], ey = fabs(xh.y) - (double)shape[1], ez = fabs(xh.z) - (double)shape[2];
    if (ex >= ey && ex >= ez) { const R s = xh.x >= 0.0 ? R(1) : R(-1); d = (R)ex; n = mk3<R>(s, R(0), R(0)); face = xh.x >= 0.0 ? 0 : 1; }
    else if (ey >= ez)        { const R s = xh.y >= 0.0 ? R(1) : R(-1); d = (R)ey; n = mk3<R>(R(0), s, R(0)); face = xh.y >= 0.0 ? 2 : 3; }
    else                      { const R s = xh.z >= 0.0 ? R(1) : R(-1); d = (R)ez; n = mk3<R>(R(0), R(0), s); face = xh.z >= 0.0 ? 4 : 5; }
  } else if (prim == TSIM_P_SPHERE) {
    const double r2 = xh.x * xh.x + xh.y * xh.y + xh.z * xh.z;
    if (r2 < 1e-24) return false;
    const double rr = sqrt(r2);
    const R r = (R)rr; d = (R)(rr - (double)shape[0]); n = x * (R(1) / r); ncurv = R(1) / r;
  } else {
    const double rhod = sqrt(xh.x * xh.x + xh.y * xh.y);
    const double dr = rhod - (double)shape[0], dz = fabs(xh.z) - (double)shape[1];
    const R rho = (R)rhod;
    if (dr > dz && rhod > 1e-12) { d = (R)dr; const R ir = R(1) / rho; n = mk3<R>(x.x * ir, x.y * ir, R(0)); ncurv = ir; pz = R(0); }
    else { const R s = xh.z >= 0.0 ? R(1) : R(-1); d = (R)dz; n = mk3<R>(R(0), R(0), s); face = xh.z >= 0.0 ? 1 : 2; }
  }
  if (!(d < R(0))) return false;
  const R dd = dot3(n, v);
  const R a = -kn + kd * dd;
  const R fn = a * d;
  const V3<R> vt = v - n * dd;
  const R vtn = t_sqrt(dot3(vt, vt));
  const bool stick = kt * vtn <= mu * t_abs(fn) || vtn < R(1e-14);
  if (branch) *branch = (stick ? 1 : 0) | (face << 1);
  R s = kt;
  const R sg = fn >= R(0) ? R(1) : R(-1);
  if (!stick) s = mu * sg * fn / vtn;
  F = n * fn - vt * s;
  if (JAC && jac) {      // jac: a wave-uniform run-time switch on top (value-only evaluations of line-search trials, evaluate())
    // N w = ncurv (Pm w - n (n.w))
    const V3<R> Nv = (mk3<R>(v.x, v.y, pz * v.z) - n * dd) * ncurv;
    const V3<R> dfx = n * a + Nv * (kd * d);       // d fn / dx
    const V3<R> dfv = n * (kd * d);                // d fn / dv
    V3<R> dsx = zero3<R>(), dsv = zero3<R>();
    if (!stick) {
      const R c1 = mu * sg / vtn, c2 = s / (vtn * vtn);
      const V3<R> Nvt = (mk3<R>(vt.x, vt.y, pz * vt.z)) * ncurv;     // n.vt = 0
      dsx = dfx * c1 + Nvt * (c2 * dd);
      dsv = dfv * c1 - vt * c2;
    }
    // Jx = n dfx^T + fn N - vt dsx^T - s dvt/dx ,  dvt/dx = -n Nv^T - dd N
    // Jv = n dfv^T        - vt dsv^T - s (I - n n^T)
    const R nn[3] = {n.x, n.y, n.z}, vv[3] = {vt.x, vt.y, vt.z};
    const R ax[3] = {dfx.x + s * Nv.x, dfx.y + s * Nv.y, dfx.z + s * Nv.z};     // n (dfx + s Nv)^T
    const R bx[3] = {dsx.x, dsx.y, dsx.z};
    const R av[3] = {dfv.x, dfv.y, dfv.z}, bv[3] = {dsv.x, dsv.y, dsv.z};
    const R kN = (fn + s * dd) * ncurv;                                          // (fn + s dd) N
    const R pm[3] = {R(1), R(1), pz};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const R nnij = nn[i] * nn[j];
        Jx.m[3 * i + j] = nn[i] * ax[j] - vv[i] * bx[j] + kN * ((i == j ? pm[i] : R(0)) - nnij);
        Jv.m[3 * i + j] = nn[i] * av[j] - vv[i] * bv[j] - s * ((i == j ? R(1) : R(0)) - nnij);
      }
  }
  return true;
}
