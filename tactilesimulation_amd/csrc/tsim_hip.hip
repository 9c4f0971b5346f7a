// tsim_hip.hip — kernels + C ABI (include/tsim.h) of the MI355X-native batched tactile-simulation step.
//
// Kernels (one environment per 64-lane wavefront; see tsim_device.h):
//   k_forward   : num_steps implicit BDF1 sub-steps (Newton + line search) with the action held, tape append,
//                 q / qd / variables / tactile read-out          <- sim.set_u + sim.forward + getters
//                                                                   (envs/redmax_torch_functions.py:131-136)
//   k_backward  : adjoint of the newest n taped sub-steps, carrying (lam_q, lam_v) across calls
//                                                                <- sim.backward_steps(n)  (:151-170)
//   k_readout   : variables + tactile at the current state       <- get_variables / get_tactile_force_vector
//   k_debug_eval: one residual + Newton-matrix evaluation (parity tests)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/tsim.h"
#include "tsim_eval.h"

// tape record per (sub-step, env): q[nr] qd[nr] H[nr*nr] u[nu]
__host__ __device__ inline int ts_rec(int nr, int nu) { return 2 * nr + nr * nr + nu; }

// ================================================================================================ read-out
// variables: lanes = end-effector points; tactile: lanes = taxels (coalesced SoA loads of position / frame,
// 12 B per lane contiguous stores).
template <class R>
__device__ void readout(const Ctx<R>& c, int lane, int env, R* var_out, R* tac_out) {
  if (var_out) {
    for (int e = lane; e < c.nvar; e += TS_WAVE) {
      const int l = c.I[c.off_var + e * TSIM_VI_SIZE + TSIM_VI_LINK];
      const R* vp = c.F + c.foff_var + e * TSIM_VF_SIZE;
      M3<R> RA = ld9<R>(c.LP, c.LT, l * LK_SIZE + LK_R, c.nd, 0);
      V3<R> x = mulMc(RA, vp) + ld3<R>(c.LP, c.LT, l * LK_SIZE + LK_P, c.nd, 0);
      R* o = var_out + (size_t)env * 3 * c.nvar + 3 * e;
      o[0] = x.x; o[1] = x.y; o[2] = x.z;
    }
  }
  if (!tac_out) return;
  for (int s = 0; s < c.nsensor; ++s) {
    const int* si = c.I + c.off_sensor + s * TSIM_SI_SIZE;
    const R* sf = c.F + c.foff_sensor + s * TSIM_SF_SIZE;
    const int la = si[TSIM_SI_LINK], t0 = si[TSIM_SI_TAX0], nt = si[TSIM_SI_NTAX], sp0 = si[TSIM_SI_SPRIM0], nsp = si[TSIM_SI_NSPRIM];
    M3<R> RA = ld9<R>(c.LP, c.LT, la * LK_SIZE + LK_R, c.nd, 0);
    for (int base = 0; base < nt; base += TS_WAVE) {
      const int t = t0 + base + lane;
      if (base + lane >= nt) continue;
      const R* tp = c.F + c.foff_tax + t;
      V3<R> xa = mk3<R>(tp[0], tp[c.ntax], tp[2 * c.ntax]);
      V3<R> F = mk3<R>(R(0), R(0), R(0));
      for (int j = 0; j < nsp; ++j) {
        const int pk = c.I[c.off_sprim + sp0 + j];
        const int* pi = c.I + c.off_pair + pk * TSIM_PI_SIZE;
        const R* pf = c.F + c.foff_pair + pk * TSIM_PF_SIZE;
        V3<R> Fw, mo;
        if (pair_point_force<R, R>(c, pi, pf, sf, la, pi[TSIM_PI_LINKB], 0, xa, false, Fw, mo)) F = F + Fw;
      }
      V3<R> Fl = mulMtv(RA, F);
      R* o = tac_out + (size_t)env * 3 * c.ntax + 3 * t;
      o[0] = Fl.x * tp[3 * c.ntax] + Fl.y * tp[4 * c.ntax] + Fl.z * tp[5 * c.ntax];
      o[1] = Fl.x * tp[6 * c.ntax] + Fl.y * tp[7 * c.ntax] + Fl.z * tp[8 * c.ntax];
      o[2] = Fl.x * tp[9 * c.ntax] + Fl.y * tp[10 * c.ntax] + Fl.z * tp[11 * c.ntax];
    }
  }
}

// ================================================================================================ forward kernel
template <class R> struct FwdArgs {
  const int* I; const R* F;
  int B, nsub, record, t0;
  R* tape; const R* u;
  R *q_out, *qd_out, *var_out, *tac_out; int* status; int* evals;
};

template <class R, int NRM>
__global__ void __launch_bounds__(TS_WAVE) k_forward(FwdArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  const int env = blockIdx.x, lane = threadIdx.x;
  Ctx<R> c; ctx_init(c, a.I, a.F, lds);
  const int nr = c.nr, nu = c.nu, REC = ts_rec(nr, nu);
  init_world(c, lane);
  {
    const R* st = a.tape + ((size_t)a.t0 * a.B + env) * REC;
    if (lane < nr) { c.q0[lane] = st[lane]; c.qd0[lane] = st[nr + lane]; }
    if (lane < nu) c.u[lane] = a.u[(size_t)env * nu + lane];
  }
  __syncthreads();
  const R sq = R(1), sv = R(1) / c.h, sa = R(1) / (c.h * c.h);
  R* dlbase = c.dq + nr;
  int bad = 0; bool nonfinite = false;
  long long evals = 0;
  for (int s = 0; s < a.nsub; ++s) {
    if (lane < nr) c.dl[lane] = R(0);          // initial guess q1 = q0 + h qd0
    __syncthreads();
    evaluate(c, lane, sq, sv, sa); ++evals;
    R gn = block_norm2(c.g, nr, lane);
    int iter = 0; bool conv = false;
    while (true) {
      if (!(gn == gn)) { nonfinite = true; break; }
      if (gn < c.tol) { conv = true; break; }
      if (iter >= c.max_iter) break;
      if (lane < nr) { c.rhs[lane] = -c.g[lane]; dlbase[lane] = c.dl[lane]; }
      __syncthreads();
      solve_lanes<R, NRM>(c.H, c.rhs, c.dq, nr, false, lane);
      R alpha = R(1), gn2 = gn;
      bool stalled = false;
      for (int ls = 0; ls <= c.max_ls; ++ls) {
        if (lane < nr) c.dl[lane] = dlbase[lane] + alpha * c.dq[lane];
        __syncthreads();
        evaluate(c, lane, sq, sv, sa); ++evals;
        gn2 = block_norm2(c.g, nr, lane);
        if (gn2 < gn) break;
        if (ls == c.max_ls) { stalled = true; break; }
        alpha *= R(0.5);
      }
      gn = gn2; ++iter;
      // No step length down to 2^-max_ls reduces ||g||: the iterate sits on the round-off floor of the residual (or on
      // a kink).  Repeating the same failed search max_iter times cannot change it by more than 2^-max_ls |dq| per
      // pass, so stop here; it counts as converged when it is within two decades of tol.
      if (stalled) { conv = gn < R(100) * c.tol; break; }
    }
    if (!conv) ++bad;
    // commit the sub-step: c.q = q1, c.qd = (q1 - q0)/h, c.H = dg/dq1 at q1
    if (a.record) {
      R* rec = a.tape + ((size_t)(a.t0 + s + 1) * a.B + env) * REC;
      if (lane < nr) { rec[lane] = c.q[lane]; rec[nr + lane] = c.qd[lane]; }
      for (int e = lane; e < nr * nr; e += TS_WAVE) rec[2 * nr + e] = c.H[e];
      if (lane < nu) rec[2 * nr + nr * nr + lane] = c.u[lane];
    }
    __syncthreads();
    if (lane < nr) { c.q0[lane] = c.q[lane]; c.qd0[lane] = c.qd[lane]; }
    __syncthreads();
  }
  if (!a.record) {
    R* st = a.tape + ((size_t)a.t0 * a.B + env) * REC;
    if (lane < nr) { st[lane] = c.q0[lane]; st[nr + lane] = c.qd0[lane]; }
  }
  if (lane < nr) {
    if (a.q_out) a.q_out[(size_t)env * nr + lane] = c.q0[lane];
    if (a.qd_out) a.qd_out[(size_t)env * nr + lane] = c.qd0[lane];
  }
  if (a.status && lane == 0) a.status[env] = bad | (nonfinite ? (1 << 30) : 0);
  if (a.evals && lane == 0) a.evals[env] = (int)evals;
  // link poses / velocities in LDS are those of the accepted state (last evaluation)
  readout(c, lane, env, a.var_out, a.tac_out);
}

// ================================================================================================ read-out kernel
template <class R> struct ReadArgs { const int* I; const R* F; int B, t0; const R* tape; R *var_out, *tac_out; };

template <class R>
__global__ void __launch_bounds__(TS_WAVE) k_readout(ReadArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  const int env = blockIdx.x, lane = threadIdx.x;
  Ctx<R> c; ctx_init(c, a.I, a.F, lds);
  const int nr = c.nr, REC = ts_rec(nr, c.nu);
  init_world(c, lane);
  const R* st = a.tape + ((size_t)a.t0 * a.B + env) * REC;
  if (lane < nr) { c.q[lane] = st[lane]; c.qd[lane] = st[nr + lane]; c.qa[lane] = R(0); }
  __syncthreads();
  phase1(c, lane, R(0), R(0), R(0));
  readout(c, lane, env, a.var_out, a.tac_out);
}

// ================================================================================================ debug evaluation
template <class R> struct DbgArgs { const int* I; const R* F; int B; const R *q1, *q0, *qd0, *u; R *g, *H; long long* cyc; };

template <class R>
__global__ void __launch_bounds__(TS_WAVE) k_debug_eval(DbgArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  const int env = blockIdx.x, lane = threadIdx.x;
  Ctx<R> c; ctx_init(c, a.I, a.F, lds);
  const int nr = c.nr, nu = c.nu;
  init_world(c, lane);
  if (lane < nr) {
    c.q0[lane] = a.q0[(size_t)env * nr + lane]; c.qd0[lane] = a.qd0[(size_t)env * nr + lane];
    c.dl[lane] = a.q1[(size_t)env * nr + lane] - c.q0[lane] - c.h * c.qd0[lane];
  }
  if (lane < nu) c.u[lane] = a.u[(size_t)env * nu + lane];
  __syncthreads();
  if (a.cyc) {   // phase-by-phase shader-clock stamps (s_memtime), under whatever load the launch creates
    const R sq = R(1), sv = R(1) / c.h, sa = R(1) / (c.h * c.h);
    long long t0 = clock64();
    if (lane < nr) {
      const R d = c.dl[lane];
      c.qd[lane] = c.qd0[lane] + d / c.h; c.qa[lane] = d / (c.h * c.h); c.q[lane] = c.q0[lane] + (c.h * c.qd0[lane] + d);
    }
    __syncthreads();
    phase1(c, lane, sq, sv, sa);
    long long t1 = clock64();
    phase2(c, lane);
    long long t2 = clock64();
    phase3(c, lane, sq, sv);
    long long t3 = clock64();
    if (lane < nr) c.rhs[lane] = -c.g[lane];
    __syncthreads();
    solve_lanes<R, 16>(c.H, c.rhs, c.dq, nr, false, lane);
    long long t4 = clock64();
    if (lane == 0) { long long* o = a.cyc + (size_t)env * 4; o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; }
  } else {
    evaluate(c, lane, R(1), R(1) / c.h, R(1) / (c.h * c.h));
  }
  if (lane < nr) a.g[(size_t)env * nr + lane] = c.g[lane];
  for (int e = lane; e < nr * nr; e += TS_WAVE) a.H[(size_t)env * nr * nr + e] = c.H[e];
}

// ================================================================================================ backward kernel
template <class R> struct BwdArgs {
  const int* I; const R* F;
  int B, n, t_end, seed_mode;
  const R* tape;
  const R *df_dq, *df_dvar, *df_dtac;
  R *lamq, *lamv, *df_du;
};

// (M z)_j for lane j: direct sums over the links below dof j (no recursion, no scratch)
template <class R>
__device__ R mass_times_z(const Ctx<R>& c, int j) {
  R tau = R(0);
  for (int i = 1; i <= c.nl; ++i) {
    const int* li = c.I + c.off_link + (i - 1) * TSIM_LI_SIZE;
    const int anc = li[TSIM_LI_ANCMASK];
    if (!((anc >> j) & 1)) continue;
    const R* lf = c.F + c.foff_link + (i - 1) * TSIM_LF_SIZE;
    V3<R> aw = mk3<R>(R(0), R(0), R(0)), av = aw;
    for (int k = 0; k < c.nr; ++k) {
      if (!((anc >> k) & 1)) continue;
      const R zk = c.z[k];
      aw = aw + ld3<R>(c.WP, c.WT, k * 6, c.nd, 0) * zk;
      av = av + ld3<R>(c.WP, c.WT, k * 6 + 3, c.nd, 0) * zk;
    }
    M3<R> XR = ld9<R>(c.LP, c.LT, i * LK_SIZE + LK_R, c.nd, 0);
    V3<R> cw = mulMc(XR, lf + TSIM_LF_COM) + ld3<R>(c.LP, c.LT, i * LK_SIZE + LK_P, c.nd, 0);
    V3<R> f = (av + cross3(aw, cw)) * lf[TSIM_LF_MASS];
    const R* ii = lf + TSIM_LF_INERTIA;
    V3<R> al = mulMtv(XR, aw);
    V3<R> Ia = mk3<R>(al.x * ii[0] + al.y * ii[3] + al.z * ii[4], al.x * ii[3] + al.y * ii[1] + al.z * ii[5], al.x * ii[4] + al.y * ii[5] + al.z * ii[2]);
    V3<R> n = mulMv(XR, Ia) + cross3(cw, f);
    tau += dot3(ld3<R>(c.WP, c.WT, j * 6, c.nd, 0), n) + dot3(ld3<R>(c.WP, c.WT, j * 6 + 3, c.nd, 0), f);
  }
  return tau;
}

// lam_q += (dvar/dq)^T w_var + (dtac/dq)^T w_tac ; lam_v += (dtac/dqd)^T w_tac, at the state whose link values and
// q-tangents are in LDS (phase 1 with seeds (1,0,0)).
template <class R>
__device__ void output_vjp(const Ctx<R>& c, int lane, const R* wvar, const R* wtac) {
  typedef Du<R> D;
  const int nd = c.nd, nr = c.nr;
  if (wvar && lane < nr) {
    R acc = R(0);
    for (int e = 0; e < c.nvar; ++e) {
      const int l = c.I[c.off_var + e * TSIM_VI_SIZE + TSIM_VI_LINK];
      if (l == 0) continue;
      const int anc = c.I[c.off_link + (l - 1) * TSIM_LI_SIZE + TSIM_LI_ANCMASK];
      if (!((anc >> lane) & 1)) continue;
      const R* vp = c.F + c.foff_var + e * TSIM_VF_SIZE;
      M3<R> RA = ld9<R>(c.LP, c.LT, l * LK_SIZE + LK_R, nd, 0);
      V3<R> x = mulMc(RA, vp) + ld3<R>(c.LP, c.LT, l * LK_SIZE + LK_P, nd, 0);
      V3<R> J = cross3(ld3<R>(c.WP, c.WT, lane * 6, nd, 0), x) + ld3<R>(c.WP, c.WT, lane * 6 + 3, nd, 0);
      acc += wvar[3 * e] * J.x + wvar[3 * e + 1] * J.y + wvar[3 * e + 2] * J.z;
    }
    c.lamq[lane] += acc;
  }
  __syncthreads();
  if (!wtac) return;
  for (int s = 0; s < c.nsensor; ++s) {
    const int* si = c.I + c.off_sensor + s * TSIM_SI_SIZE;
    const R* sf = c.F + c.foff_sensor + s * TSIM_SF_SIZE;
    const int la = si[TSIM_SI_LINK], t0 = si[TSIM_SI_TAX0], nt = si[TSIM_SI_NTAX], sp0 = si[TSIM_SI_SPRIM0], nsp = si[TSIM_SI_NSPRIM];
    const int ancA = la > 0 ? c.I[c.off_link + (la - 1) * TSIM_LI_SIZE + TSIM_LI_ANCMASK] : 0;
    int ancAll = ancA;
    for (int j = 0; j < nsp; ++j) {
      const int lb = c.I[c.off_pair + c.I[c.off_sprim + sp0 + j] * TSIM_PI_SIZE + TSIM_PI_LINKB];
      if (lb > 0) ancAll |= c.I[c.off_link + (lb - 1) * TSIM_LI_SIZE + TSIM_LI_ANCMASK];
    }
    for (int base = 0; base < nt; base += TS_WAVE) {
      const bool valid = base + lane < nt;
      const int t = t0 + (valid ? base + lane : 0);
      const R* tp = c.F + c.foff_tax + t;
      V3<R> xa = mk3<R>(tp[0], tp[c.ntax], tp[2 * c.ntax]);
      R w0 = R(0), w1 = R(0), w2 = R(0);
      if (valid) { w0 = wtac[3 * t]; w1 = wtac[3 * t + 1]; w2 = wtac[3 * t + 2]; }
      // weight vector in the sensor-link frame: sum_c w_c axis_c
      V3<R> wl = mk3<R>(w0 * tp[3 * c.ntax] + w1 * tp[6 * c.ntax] + w2 * tp[9 * c.ntax],
                        w0 * tp[4 * c.ntax] + w1 * tp[7 * c.ntax] + w2 * tp[10 * c.ntax],
                        w0 * tp[5 * c.ntax] + w1 * tp[8 * c.ntax] + w2 * tp[11 * c.ntax]);
      const bool live = valid && (w0 != R(0) || w1 != R(0) || w2 != R(0));
      if (!__any(live)) continue;
      for (int dir = 0; dir < nr; ++dir) {
        if (!((ancAll >> dir) & 1)) continue;
        R sq_ = R(0), sv_ = R(0);
        if (live) {
          // (a) q-tangent: link tangents from LDS
          {
            M3<D> RA = ld9<D>(c.LP, c.LT, la * LK_SIZE + LK_R, nd, dir);
            V3<D> F = mk3<D>(D(R(0)), D(R(0)), D(R(0)));
            for (int j = 0; j < nsp; ++j) {
              const int pk = c.I[c.off_sprim + sp0 + j];
              const int* pi = c.I + c.off_pair + pk * TSIM_PI_SIZE;
              const R* pf = c.F + c.foff_pair + pk * TSIM_PF_SIZE;
              V3<D> Fw, mo;
              if (pair_point_force<D, R>(c, pi, pf, sf, la, pi[TSIM_PI_LINKB], dir, xa, false, Fw, mo)) F = F + Fw;
            }
            V3<D> Fl = mulMtv(RA, F);
            sq_ = Fl.x.d * wl.x + Fl.y.d * wl.y + Fl.z.d * wl.z;
          }
          // (b) qd-tangent: poses fixed, d(V_link)/d(qd_dir) = W_dir for links below dof dir
          {
            const int ab = la * LK_SIZE;
            M3<R> RAv = ld9<R>(c.LP, c.LT, ab + LK_R, nd, 0);
            M3<D> RA; for (int e = 0; e < 9; ++e) RA.m[e] = D(RAv.m[e]);
            V3<R> Ww = ld3<R>(c.WP, c.WT, dir * 6, nd, 0), Wv = ld3<R>(c.WP, c.WT, dir * 6 + 3, nd, 0);
            const R ina = ((ancA >> dir) & 1) ? R(1) : R(0);
            V3<R> pAv = ld3<R>(c.LP, c.LT, ab + LK_P, nd, 0), wAv = ld3<R>(c.LP, c.LT, ab + LK_W, nd, 0), vAv = ld3<R>(c.LP, c.LT, ab + LK_V, nd, 0);
            V3<D> pA = mk3<D>(D(pAv.x), D(pAv.y), D(pAv.z));
            V3<D> wA = mk3<D>(D(wAv.x, Ww.x * ina), D(wAv.y, Ww.y * ina), D(wAv.z, Ww.z * ina));
            V3<D> vA = mk3<D>(D(vAv.x, Wv.x * ina), D(vAv.y, Wv.y * ina), D(vAv.z, Wv.z * ina));
            V3<D> F = mk3<D>(D(R(0)), D(R(0)), D(R(0)));
            for (int j = 0; j < nsp; ++j) {
              const int pk = c.I[c.off_sprim + sp0 + j];
              const int* pi = c.I + c.off_pair + pk * TSIM_PI_SIZE;
              const R* pf = c.F + c.foff_pair + pk * TSIM_PF_SIZE;
              const int lb = pi[TSIM_PI_LINKB], bb = lb * LK_SIZE;
              const int ancB = lb > 0 ? c.I[c.off_link + (lb - 1) * TSIM_LI_SIZE + TSIM_LI_ANCMASK] : 0;
              const R inb = ((ancB >> dir) & 1) ? R(1) : R(0);
              M3<R> RBv = ld9<R>(c.LP, c.LT, bb + LK_R, nd, 0);
              M3<D> RB; for (int e = 0; e < 9; ++e) RB.m[e] = D(RBv.m[e]);
              V3<R> pBv = ld3<R>(c.LP, c.LT, bb + LK_P, nd, 0), wBv = ld3<R>(c.LP, c.LT, bb + LK_W, nd, 0), vBv = ld3<R>(c.LP, c.LT, bb + LK_V, nd, 0);
              V3<D> pB = mk3<D>(D(pBv.x), D(pBv.y), D(pBv.z));
              V3<D> wB = mk3<D>(D(wBv.x, Ww.x * inb), D(wBv.y, Ww.y * inb), D(wBv.z, Ww.z * inb));
              V3<D> vB = mk3<D>(D(vBv.x, Wv.x * inb), D(vBv.y, Wv.y * inb), D(vBv.z, Wv.z * inb));
              V3<D> Fw, xw;
              if (point_force<D, R>(pi[TSIM_PI_PRIM], pf, sf, false, RA, pA, wA, vA, RB, pB, wB, vB, xa, Fw, xw)) F = F + Fw;
            }
            V3<D> Fl = mulMtv(RA, F);
            sv_ = Fl.x.d * wl.x + Fl.y.d * wl.y + Fl.z.d * wl.z;
          }
        }
        sq_ = wave_sum(sq_); sv_ = wave_sum(sv_);
        if (lane == 0) { c.lamq[dir] += sq_; c.lamv[dir] += sv_; }
      }
    }
  }
  __syncthreads();
}

template <class R, int NRM>
__global__ void __launch_bounds__(TS_WAVE) k_backward(BwdArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  const int env = blockIdx.x, lane = threadIdx.x;
  Ctx<R> c; ctx_init(c, a.I, a.F, lds);
  const int nr = c.nr, nu = c.nu, REC = ts_rec(nr, nu);
  const int nvar3 = 3 * c.nvar, ntac3 = 3 * c.ntax;
  R* H2 = c.scr;   // nr*nr reals: taped Newton matrix of the sub-step (scratch region is (nl+1)*12 + 8 >= ... checked on host)
  init_world(c, lane);
  if (lane < nr) { c.lamq[lane] = a.lamq[(size_t)env * nr + lane]; c.lamv[lane] = a.lamv[(size_t)env * nr + lane]; }
  __syncthreads();
  for (int j = a.n - 1; j >= 0; --j) {
    const int t = a.t_end - (a.n - 1 - j);
    const R* r1 = a.tape + ((size_t)t * a.B + env) * REC;
    const R* r0 = a.tape + ((size_t)(t - 1) * a.B + env) * REC;
    if (lane < nr) {
      c.q[lane] = r1[lane]; c.q0[lane] = r0[lane]; c.qd0[lane] = r0[nr + lane];
      c.qd[lane] = r1[nr + lane];                               // taped (q1 - q0)/h
      c.qa[lane] = (r1[nr + lane] - r0[nr + lane]) / c.h;       // discrete acceleration, no position cancellation
    }
    if (lane < nu) c.u[lane] = r1[2 * nr + nr * nr + lane];
    for (int e = lane; e < nr * nr; e += TS_WAVE) H2[e] = r1[2 * nr + e];
    __syncthreads();
    phase1(c, lane, R(1), R(0), R(0));
    // direct partials of the loss w.r.t. this sub-step's outputs
    const bool seeded = a.seed_mode == 1 || j == a.n - 1;
    if (seeded) {
      const size_t so = a.seed_mode == 1 ? (size_t)env * a.n + j : (size_t)env;
      if (a.df_dq && lane < nr) c.lamq[lane] += a.df_dq[so * nr + lane];
      __syncthreads();
      output_vjp(c, lane, (a.df_dvar && nvar3) ? a.df_dvar + so * nvar3 : nullptr, (a.df_dtac && ntac3) ? a.df_dtac + so * ntac3 : nullptr);
    }
    if (lane < nr) c.rhs[lane] = c.lamq[lane] + c.lamv[lane] / c.h;
    __syncthreads();
    solve_lanes<R, NRM>(H2, c.rhs, c.z, nr, true, lane);
    phase2(c, lane);
    phase3(c, lane, R(1), R(0));       // c.H = h^2 dr/dq
    if (lane < nr) {
      R yq = R(0);
      for (int i = 0; i < nr; ++i) yq += c.z[i] * c.H[i * nr + lane];
      const R ym = mass_times_z(c, lane);
      c.lamq[lane] -= yq;
      c.lamv[lane] = c.h * ym;
    }
    if (lane < nu) {
      const int* mi = c.I + c.off_motor + lane * TSIM_MI_SIZE;
      const R* mf = c.F + c.foff_motor + lane * TSIM_MF_SIZE;
      R dtu;
      if (mi[TSIM_MI_CTRL] == 0) dtu = (c.u[lane] > R(-1) && c.u[lane] < R(1)) ? R(0.5) * (mf[TSIM_MF_HI] - mf[TSIM_MF_LO]) : R(0);
      else dtu = mf[TSIM_MF_P];
      a.df_du[((size_t)env * a.n + j) * nu + lane] = c.h * c.h * c.z[mi[TSIM_MI_DOF]] * dtu;
    }
    __syncthreads();
  }
  if (lane < nr) { a.lamq[(size_t)env * nr + lane] = c.lamq[lane]; a.lamv[(size_t)env * nr + lane] = c.lamv[lane]; }
}

// ================================================================================================ host side
static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct CacheEntry { void* buf; int len; int record; };
struct tsim_batch {
  int B, dtype, device, cap;
  std::vector<int32_t> I; std::vector<double> F;
  int nl, nr, nu, nvar, ntax, rec;
  int* dI; void* dF;             // model on device (dF in the batch's real type)
  void* tape;                    // [(cap+1)][B][rec]
  void *lamq, *lamv;             // carried adjoint [B][nr]
  int* evals;                    // residual evaluations of the last forward launch, per env (diagnostics)
  int t_cur, record;
  size_t lds_bytes, esz;
  std::vector<CacheEntry> cache;
};

static int upload_model(tsim_batch* b, hipStream_t st) {
  HIPCHK(hipMemcpyAsync(b->dI, b->I.data(), b->I.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
  if (b->dtype == TSIM_F32) {
    std::vector<float> f(b->F.begin(), b->F.end());
    HIPCHK(hipMemcpyAsync(b->dF, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // f is a temporary
  } else {
    HIPCHK(hipMemcpyAsync(b->dF, b->F.data(), b->F.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return 0;
}

// scatter [B][nr] q / qd into tape record 0
template <class R> __global__ void k_set_state(R* tape, const R* q, const R* qd, int B, int nr, int rec) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nr) return;
  int env = i / nr, k = i % nr;
  tape[(size_t)env * rec + k] = q[i];
  tape[(size_t)env * rec + nr + k] = qd ? qd[i] : R(0);
}
template <class R> __global__ void k_get_state(const R* tape_rec, R* q, R* qd, int B, int nr, int rec) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nr) return;
  int env = i / nr, k = i % nr;
  if (q) q[i] = tape_rec[(size_t)env * rec + k];
  if (qd) qd[i] = tape_rec[(size_t)env * rec + nr + k];
}

template <class R>
static int launch_forward(tsim_batch* b, const void* u, int nsub, void* q_out, void* qd_out, void* var_out, void* tac_out, int32_t* status, hipStream_t st) {
  FwdArgs<R> a;
  a.I = b->dI; a.F = (const R*)b->dF; a.B = b->B; a.nsub = nsub; a.record = b->record; a.t0 = b->t_cur;
  a.tape = (R*)b->tape; a.u = (const R*)u;
  a.q_out = (R*)q_out; a.qd_out = (R*)qd_out; a.var_out = (R*)var_out; a.tac_out = (R*)tac_out; a.status = status; a.evals = b->evals;
  if (b->nr <= 8) hipLaunchKernelGGL((k_forward<R, 8>), dim3(b->B), dim3(TS_WAVE), b->lds_bytes, st, a);
  else hipLaunchKernelGGL((k_forward<R, 16>), dim3(b->B), dim3(TS_WAVE), b->lds_bytes, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

template <class R>
static int launch_backward(tsim_batch* b, int n, int seed_mode, const void* df_dq, const void* df_dvar, const void* df_dtac, void* df_du, hipStream_t st) {
  BwdArgs<R> a;
  a.I = b->dI; a.F = (const R*)b->dF; a.B = b->B; a.n = n; a.t_end = b->t_cur; a.seed_mode = seed_mode;
  a.tape = (const R*)b->tape; a.df_dq = (const R*)df_dq; a.df_dvar = (const R*)df_dvar; a.df_dtac = (const R*)df_dtac;
  a.lamq = (R*)b->lamq; a.lamv = (R*)b->lamv; a.df_du = (R*)df_du;
  if (b->nr <= 8) hipLaunchKernelGGL((k_backward<R, 8>), dim3(b->B), dim3(TS_WAVE), b->lds_bytes, st, a);
  else hipLaunchKernelGGL((k_backward<R, 16>), dim3(b->B), dim3(TS_WAVE), b->lds_bytes, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" {

const char* tsim_last_error(void) { return g_err.c_str(); }

int tsim_batch_create(const int32_t* I, const double* F, int B, int tape_capacity, int dtype, int device, tsim_batch** out) {
  if (!I || !F || !out) return fail("null argument");
  if (I[TSIM_IH_MAGIC] != TSIM_MAGIC || I[TSIM_IH_VERSION] != TSIM_VERSION) return fail("model blob: bad magic/version");
  if (B <= 0 || tape_capacity < 0) return fail("bad batch size / tape capacity");
  if (dtype != TSIM_F32 && dtype != TSIM_F64) return fail("bad dtype");
  if (I[TSIM_IH_INTEGRATOR] != 1) return fail("integrator not supported by the HIP path yet (BDF1 only)");
  const int nl = I[TSIM_IH_NL], nr = I[TSIM_IH_NR], nu = I[TSIM_IH_NU];
  if (nr > 16 || nr < 1 || nu > TS_WAVE) return fail("ndof_r must be in 1..16");
  for (int i = 1; i <= nl; ++i) {
    int jt = I[I[TSIM_IH_OFF_LINK] + (i - 1) * TSIM_LI_SIZE + TSIM_LI_JTYPE];
    if (jt != TSIM_J_REVOLUTE && jt != TSIM_J_PRISMATIC && jt != TSIM_J_PLANAR && jt != TSIM_J_TRANSLATIONAL)
      return fail("joint type not supported by the HIP path yet");
  }
  for (int s = 0; s < I[TSIM_IH_NSENSOR]; ++s)
    if (I[I[TSIM_IH_OFF_SENSOR] + s * TSIM_SI_SIZE + TSIM_SI_NSPRIM] > 16) return fail("too many primitives per sensor");
  HIPCHK(hipSetDevice(device));
  tsim_batch* b = new tsim_batch();
  b->B = B; b->dtype = dtype; b->device = device; b->cap = tape_capacity;
  b->I.assign(I, I + I[TSIM_IH_NI]); b->F.assign(F, F + I[TSIM_IH_NF]);
  b->nl = nl; b->nr = nr; b->nu = nu; b->nvar = I[TSIM_IH_NVAR]; b->ntax = I[TSIM_IH_NTAXEL];
  b->rec = ts_rec(nr, nu);
  b->esz = dtype == TSIM_F32 ? 4 : 8;
  int reals = ts_lds_reals(nl, nr, nu);
  int scr_have = (nl + 1) * 12 + 8;
  if (scr_have < nr * nr) reals += nr * nr - scr_have;     // the adjoint kernel keeps the taped H in the scratch region
  b->lds_bytes = ((size_t)reals * b->esz + 15) / 16 * 16;
  if (b->lds_bytes > 64 * 1024) { delete b; return fail("model needs more than 64 KiB of LDS per environment"); }
  b->t_cur = 0; b->record = 0;
  b->dI = nullptr; b->dF = nullptr; b->tape = nullptr; b->lamq = nullptr; b->lamv = nullptr; b->evals = nullptr;
  size_t tape_bytes = (size_t)(tape_capacity + 1) * B * b->rec * b->esz;
  if (hipMalloc(&b->dI, b->I.size() * sizeof(int32_t)) != hipSuccess || hipMalloc(&b->dF, b->F.size() * b->esz) != hipSuccess ||
      hipMalloc(&b->tape, tape_bytes) != hipSuccess || hipMalloc(&b->lamq, (size_t)B * nr * b->esz) != hipSuccess ||
      hipMalloc(&b->lamv, (size_t)B * nr * b->esz) != hipSuccess || hipMalloc(&b->evals, (size_t)B * sizeof(int)) != hipSuccess) {
    tsim_batch_destroy(b);
    return fail("hipMalloc failed (tape bytes = " + std::to_string(tape_bytes) + ")");
  }
  if (hipMemset(b->tape, 0, tape_bytes) != hipSuccess || hipMemset(b->lamq, 0, (size_t)B * nr * b->esz) != hipSuccess ||
      hipMemset(b->lamv, 0, (size_t)B * nr * b->esz) != hipSuccess) { tsim_batch_destroy(b); return fail("hipMemset failed"); }
  if (upload_model(b, nullptr)) { tsim_batch_destroy(b); return 1; }
  *out = b;
  return 0;
}

void tsim_batch_destroy(tsim_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  for (auto& e : b->cache) (void)hipFree(e.buf);
  (void)hipFree(b->dI); (void)hipFree(b->dF); (void)hipFree(b->tape); (void)hipFree(b->lamq); (void)hipFree(b->lamv); (void)hipFree(b->evals);
  delete b;
}

int tsim_ndof_r(const tsim_batch* b) { return b->nr; }
int tsim_ndof_u(const tsim_batch* b) { return b->nu; }
int tsim_ndof_var(const tsim_batch* b) { return 3 * b->nvar; }
int tsim_ndof_tactile(const tsim_batch* b) { return 3 * b->ntax; }
int tsim_batch_size(const tsim_batch* b) { return b->B; }
int tsim_dtype(const tsim_batch* b) { return b->dtype; }
double tsim_timestep(const tsim_batch* b) { return b->F[TSIM_FH_H]; }
int tsim_tape_len(const tsim_batch* b) { return b->record ? b->t_cur : 0; }
int tsim_launch_info(const tsim_batch* b, int32_t* out) { out[0] = (int32_t)b->lds_bytes; out[1] = TS_WAVE; out[2] = b->B; return 0; }
int tsim_last_evals(tsim_batch* b, int32_t* host_out) {
  if (hipSetDevice(b->device) != hipSuccess || hipMemcpy(host_out, b->evals, (size_t)b->B * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail("last_evals: copy failed");
  return 0;
}

int tsim_update_model(tsim_batch* b, const int32_t* I, const double* F, void* stream) {
  if (I[TSIM_IH_NI] != (int)b->I.size() || I[TSIM_IH_NF] != (int)b->F.size()) return fail("update_model: blob size changed");
  for (int i = 0; i < TSIM_IH_SIZE; ++i) if (I[i] != b->I[i]) return fail("update_model: topology changed");
  HIPCHK(hipSetDevice(b->device));
  b->I.assign(I, I + I[TSIM_IH_NI]); b->F.assign(F, F + I[TSIM_IH_NF]);
  return upload_model(b, (hipStream_t)stream);
}

int tsim_reset(tsim_batch* b, const void* q0, const void* qd0, int backward_flag, void* stream) {
  if (!q0) return fail("reset: q0 is null");
  HIPCHK(hipSetDevice(b->device));
  hipStream_t st = (hipStream_t)stream;
  int n = b->B * b->nr, blk = 256, grd = (n + blk - 1) / blk;
  if (b->dtype == TSIM_F32) hipLaunchKernelGGL(k_set_state<float>, dim3(grd), dim3(blk), 0, st, (float*)b->tape, (const float*)q0, (const float*)qd0, b->B, b->nr, b->rec);
  else hipLaunchKernelGGL(k_set_state<double>, dim3(grd), dim3(blk), 0, st, (double*)b->tape, (const double*)q0, (const double*)qd0, b->B, b->nr, b->rec);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(b->lamq, 0, (size_t)b->B * b->nr * b->esz, st));
  HIPCHK(hipMemsetAsync(b->lamv, 0, (size_t)b->B * b->nr * b->esz, st));
  b->t_cur = 0; b->record = backward_flag ? 1 : 0;
  return 0;
}

int tsim_step(tsim_batch* b, const void* u, int num_steps, void* q_out, void* qd_out, void* var_out, void* tac_out, int32_t* status, void* stream) {
  if (num_steps <= 0) return fail("step: num_steps must be positive");
  if (!u && b->nu > 0) return fail("step: u is null");
  if (b->record && b->t_cur + num_steps > b->cap) return fail("step: tape capacity exceeded (" + std::to_string(b->cap) + " sub-steps)");
  HIPCHK(hipSetDevice(b->device));
  int rc = b->dtype == TSIM_F32 ? launch_forward<float>(b, u, num_steps, q_out, qd_out, var_out, tac_out, status, (hipStream_t)stream)
                                : launch_forward<double>(b, u, num_steps, q_out, qd_out, var_out, tac_out, status, (hipStream_t)stream);
  if (rc) return rc;
  if (b->record) b->t_cur += num_steps;
  return 0;
}

int tsim_get_state(tsim_batch* b, void* q_out, void* qd_out, void* stream) {
  HIPCHK(hipSetDevice(b->device));
  int n = b->B * b->nr, blk = 256, grd = (n + blk - 1) / blk;
  size_t off = (size_t)b->t_cur * b->B * b->rec;
  if (b->dtype == TSIM_F32) hipLaunchKernelGGL(k_get_state<float>, dim3(grd), dim3(blk), 0, (hipStream_t)stream, (const float*)b->tape + off, (float*)q_out, (float*)qd_out, b->B, b->nr, b->rec);
  else hipLaunchKernelGGL(k_get_state<double>, dim3(grd), dim3(blk), 0, (hipStream_t)stream, (const double*)b->tape + off, (double*)q_out, (double*)qd_out, b->B, b->nr, b->rec);
  HIPCHK(hipGetLastError());
  return 0;
}

int tsim_readout(tsim_batch* b, void* var_out, void* tac_out, void* stream) {
  HIPCHK(hipSetDevice(b->device));
  if (b->dtype == TSIM_F32) {
    ReadArgs<float> a{b->dI, (const float*)b->dF, b->B, b->t_cur, (const float*)b->tape, (float*)var_out, (float*)tac_out};
    hipLaunchKernelGGL(k_readout<float>, dim3(b->B), dim3(TS_WAVE), b->lds_bytes, (hipStream_t)stream, a);
  } else {
    ReadArgs<double> a{b->dI, (const double*)b->dF, b->B, b->t_cur, (const double*)b->tape, (double*)var_out, (double*)tac_out};
    hipLaunchKernelGGL(k_readout<double>, dim3(b->B), dim3(TS_WAVE), b->lds_bytes, (hipStream_t)stream, a);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int tsim_backward_steps(tsim_batch* b, int n, int seed_mode, const void* df_dq, const void* df_dvar, const void* df_dtac, void* df_du, void* stream) {
  if (!b->record) return fail("backward_steps: reset(backward_flag=True) was not called");
  if (n <= 0 || n > b->t_cur) return fail("backward_steps: only " + std::to_string(b->t_cur) + " sub-steps on the tape");
  if (!df_du) return fail("backward_steps: df_du is null");
  if (seed_mode != 0 && seed_mode != 1) return fail("backward_steps: bad seed_mode");
  HIPCHK(hipSetDevice(b->device));
  int rc = b->dtype == TSIM_F32 ? launch_backward<float>(b, n, seed_mode, df_dq, df_dvar, df_dtac, df_du, (hipStream_t)stream)
                                : launch_backward<double>(b, n, seed_mode, df_dq, df_dvar, df_dtac, df_du, (hipStream_t)stream);
  if (rc) return rc;
  b->t_cur -= n;
  return 0;
}

int tsim_get_adjoint(tsim_batch* b, void* df_dq0, void* df_dqd0, void* stream) {
  HIPCHK(hipSetDevice(b->device));
  size_t bytes = (size_t)b->B * b->nr * b->esz;
  if (df_dq0) HIPCHK(hipMemcpyAsync(df_dq0, b->lamq, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (df_dqd0) HIPCHK(hipMemcpyAsync(df_dqd0, b->lamv, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int tsim_cache_save(tsim_batch* b, void* stream) {
  HIPCHK(hipSetDevice(b->device));
  CacheEntry e; e.len = b->t_cur; e.record = b->record; e.buf = nullptr;
  size_t bytes = (size_t)(b->t_cur + 1) * b->B * b->rec * b->esz;
  HIPCHK(hipMalloc(&e.buf, bytes));
  HIPCHK(hipMemcpyAsync(e.buf, b->tape, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  b->cache.push_back(e);
  return 0;
}
int tsim_cache_pop(tsim_batch* b, void* stream) {
  if (b->cache.empty()) return fail("popBackwardCache: cache is empty");
  HIPCHK(hipSetDevice(b->device));
  CacheEntry e = b->cache.back(); b->cache.pop_back();
  size_t bytes = (size_t)(e.len + 1) * b->B * b->rec * b->esz;
  HIPCHK(hipMemcpyAsync(b->tape, e.buf, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipFree(e.buf));
  b->t_cur = e.len; b->record = e.record;
  HIPCHK(hipMemsetAsync(b->lamq, 0, (size_t)b->B * b->nr * b->esz, (hipStream_t)stream));
  HIPCHK(hipMemsetAsync(b->lamv, 0, (size_t)b->B * b->nr * b->esz, (hipStream_t)stream));
  return 0;
}
int tsim_cache_clear(tsim_batch* b) {
  (void)hipSetDevice(b->device);
  for (auto& e : b->cache) (void)hipFree(e.buf);
  b->cache.clear();
  return 0;
}

int tsim_debug_eval(tsim_batch* b, const void* q1, const void* q0, const void* qd0, const void* u, void* g_out, void* H_out, long long* cycles, void* stream) {
  HIPCHK(hipSetDevice(b->device));
  if (b->dtype == TSIM_F32) {
    DbgArgs<float> a{b->dI, (const float*)b->dF, b->B, (const float*)q1, (const float*)q0, (const float*)qd0, (const float*)u, (float*)g_out, (float*)H_out, cycles};
    hipLaunchKernelGGL(k_debug_eval<float>, dim3(b->B), dim3(TS_WAVE), b->lds_bytes, (hipStream_t)stream, a);
  } else {
    DbgArgs<double> a{b->dI, (const double*)b->dF, b->B, (const double*)q1, (const double*)q0, (const double*)qd0, (const double*)u, (double*)g_out, (double*)H_out, cycles};
    hipLaunchKernelGGL(k_debug_eval<double>, dim3(b->B), dim3(TS_WAVE), b->lds_bytes, (hipStream_t)stream, a);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"
