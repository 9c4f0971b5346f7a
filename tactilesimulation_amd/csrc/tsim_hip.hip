// tsim_hip.hip — kernels + C ABI (include/tsim.h) of the MI355X-native batched tactile-simulation step.
//
// Kernels (block = one 64-lane wavefront carrying 64 / LPE environments of LPE lanes each; see tsim_device.h):
//   k_forward   : nframes env-steps of num_steps implicit BDF1 / BDF2 sub-steps (Newton + line search) with the action of
//                 the frame held, tape append, q / qd / variables / tactile read-out per frame
//                                                                <- sim.set_u + sim.forward + getters
//                                                                   (envs/redmax_torch_functions.py:131-136; nframes > 1:
//                                                                   the episode loop of :46-57)
//   k_backward  : adjoint of the newest n taped sub-steps, carrying (lam_q, lam_v) across calls
//                                                                <- sim.backward_steps(n)  (:151-170), sim.backward() (:92)
//   k_readout   : variables + tactile at the current state       <- get_variables / get_tactile_force_vector
//   k_debug_eval: one residual + Newton-matrix evaluation (parity tests)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/tsim.h"
#include "tsim_eval.h"
#include "tsim_policy_push.h"


// tape record per (sub-step, env), in reals: q[nr] as DOUBLE (the pose chain is double also in the fp32 kernels),
// qd[nr], H[nr*nr], u[nu]; padded to an even count so that every record starts 8-byte aligned
__host__ __device__ inline int ts_qw(int esz) { return 8 / esz; }                      // reals per double
__host__ __device__ inline int ts_rec(int nr, int nu, int esz) { return (ts_qw(esz) * nr + nr + nr * nr + nu + 1) & ~1; }
template <class R> __device__ __forceinline__ double* rec_q(R* rec) { return reinterpret_cast<double*>(rec); }
template <class R> __device__ __forceinline__ const double* rec_q(const R* rec) { return reinterpret_cast<const double*>(rec); }
template <class R> __device__ __forceinline__ int rec_qd(int nr) { return ts_qw((int)sizeof(R)) * nr; }           // offset of qd
template <class R> __device__ __forceinline__ int rec_H(int nr) { return ts_qw((int)sizeof(R)) * nr + nr; }
template <class R> __device__ __forceinline__ int rec_u(int nr) { return ts_qw((int)sizeof(R)) * nr + nr + nr * nr; }

// ================================================================================================ read-out
// variables: lanes = end-effector points; tactile: lanes = taxels (coalesced SoA loads of position / frame,
// 12 B per lane contiguous stores).  Each taxel is evaluated in the frame of the primitive it is tested against.
template <int LPE, class R>
__device__ __forceinline__ void readout(const Ctx<R>& c, int lane, int env, bool wr_var, bool wr, bool has_var, bool has_tac, R* var_out, R* tac_out, int tb = 0, int te = 0x7fffffff) {
  // has_var / has_tac are wave-uniform (the loops below contain fences); wr_var / wr are per slot: in k_forward the slots of a
  // wavefront reach the end of a frame in different rounds, and only those that did write
  if (has_var && wr_var) {
    for (int e = lane; e < c.nvar; e += LPE) {
      const int l = c.I[c.off_var + e * TSIM_VI_SIZE + TSIM_VI_LINK];
      const V3<R> x = mulMv(ldm(c.LP + l * LK_SIZE + LK_R), ldv(c.F + c.foff_var + e * TSIM_VF_SIZE)) + ldv(c.LP + l * LK_SIZE + LK_P);
      R* o = var_out + (size_t)env * 3 * c.nvar + 3 * e;
      o[0] = x.x; o[1] = x.y; o[2] = x.z;
    }
  }
  if (!has_tac) return;
  for (int s = 0; s < c.nsensor; ++s) {
    const int* si = c.I + c.off_sensor + s * TSIM_SI_SIZE;
    const R* sf = c.F + c.foff_sensor + s * TSIM_SF_SIZE;
    const int t0 = ts_u(si[TSIM_SI_TAX0]), nt = ts_u(si[TSIM_SI_NTAX]), sp0 = ts_u(si[TSIM_SI_SPRIM0]), nsp = ts_u(si[TSIM_SI_NSPRIM]);
    for (int j0 = 0; j0 < nsp || j0 == 0; j0 += TS_PAIR_GROUP) {
      const int je = min(j0 + TS_PAIR_GROUP, nsp);
      TS_SYNC();
      for (int j = j0; j < je; ++j) pair_stage_value(c, ts_u(c.I[c.off_sprim + sp0 + j]), j - j0, lane == 0);
      TS_SYNC();
      // this block's slice [tb, te) of the global taxel range, intersected with the sensor
      const int lo = max(tb, t0) - t0, hi = min(te, t0 + nt) - t0;
      for (int base = lo; base < hi; base += LPE) {
        const int t = t0 + base + lane;
        if (base + lane >= hi || !wr) continue;
        const R* tp = c.Fg + c.foff_tax + t;
        const V3<R> xa = mk3<R>(tp[0], tp[c.ntax], tp[2 * c.ntax]);
        V3<R> Fl = zero3<R>();                          // force on the taxel, sensor-link frame
        for (int j = j0; j < je; ++j) {
          const int pk = ts_u(c.I[c.off_sprim + sp0 + j]);
          const int prim = ts_u(c.I[c.off_pair + pk * TSIM_PI_SIZE + TSIM_PI_PRIM]);
          const R* pf = c.F + c.foff_pair + pk * TSIM_PF_SIZE;
          const R* S = c.PP + (j - j0) * PP_SIZE;
          const M3<R> RPA = ldm(S + PP_RPA);
          const V3<double> xPd = mulMv(ldm(c.PPd + (j - j0) * 12), cvt3<double>(xa)) + ldv(c.PPd + (j - j0) * 12 + 9);
          const V3<R> xP = cvt3<R>(xPd);
          V3<R> F; M3<R> Jx, Jv;
          if (contact_law<R, false>(prim, pf + TSIM_PF_SHAPE, sf, xP, ldv(S + PP_VREL) + cross3(ldv(S + PP_WREL), xP), F, Jx, Jv, xPd))
            Fl = Fl + mulMtv(RPA, F);
        }
        R* o = tac_out + (size_t)env * 3 * c.ntax + 3 * t;
        R o0 = R(0), o1 = R(0), o2 = R(0);
        if (Fl.x != R(0) || Fl.y != R(0) || Fl.z != R(0)) {       // the nine axis constants only for taxels that carry a force
          o0 = Fl.x * tp[3 * c.ntax] + Fl.y * tp[4 * c.ntax] + Fl.z * tp[5 * c.ntax];
          o1 = Fl.x * tp[6 * c.ntax] + Fl.y * tp[7 * c.ntax] + Fl.z * tp[8 * c.ntax];
          o2 = Fl.x * tp[9 * c.ntax] + Fl.y * tp[10 * c.ntax] + Fl.z * tp[11 * c.ntax];
        }
        if (j0 == 0) { o[0] = o0; o[1] = o1; o[2] = o2; } else { o[0] += o0; o[1] += o1; o[2] += o2; }
      }
    }
  }
}

// ================================================================================================ forward kernel
enum { TP_R_SIZE = 18, TP_D_SIZE = 12 };      // pose record of a (sensor, primitive) combination: R part, double part (k_readout)
template <class R> struct FwdArgs {
  const int* I; const R* F; const R* Fenv; int fstride;
  int B, nsub, record, t0;
  int nframes;            // env-steps in this launch; frame f reads u[f][B][nu] and writes *_out[f][B][...] (tsim_rollout)
  const int* tac_slot;    // [nframes] slot of frame f in tac_out, < 0: no tactile read-out for that frame; null: slot f
  R* tape; const R* u;
  R *q_out, *qd_out, *var_out, *tac_out; int* status; int* evals;
  const int* order;       // block -> environment map (longest-processing-time-first scheduling), or null
  double* prev; int has_prev;  // state before the previous sub-step [B][2 nr] doubles (BDF2 history across launches)
  int stage_cpt;               // contact-point arrays staged in LDS with the shared tables (sized into the launch's LDS)
  int cross_kinks;             // full Newton step at an exhausted line search close to convergence (tsim_set_solver_options)
  int eval_budget;             // residual evaluations a sub-step may take before it is flagged and left (0: the XML's max_iter / max_ls only)
  float* gnorm;                // [B] largest ||g|| a sub-step of this launch ended with (diagnostics, tsim_last_gnorm)
  PushPolicy<R> pol;           // POLICY instantiations only (tsim_push_closed_rollout): the TactilePush policy between the frames
  R* poseR = nullptr; double* poseD = nullptr; int nspt = 0;   // large pads: pose records of the final state for tsim_readout's k_taxels (see k_readout)
};

// -DTS_WAVES_PER_EU=n (A/B builds): ask the compiler for n wavefronts per SIMD in the two simulation kernels (2 -> at most 256 registers)
#ifdef TS_WAVES_PER_EU
#define TS_KLB __launch_bounds__(TS_WAVE, TS_WAVES_PER_EU)
#else
#define TS_KLB __launch_bounds__(TS_WAVE)
#endif
template <class R, int NRM, bool EXPJ, int LPE, bool POLICY = false>
__global__ void TS_KLB k_forward(FwdArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  constexpr int NS = TS_WAVE / LPE;
  const int slot = threadIdx.x / LPE, lane = threadIdx.x % LPE;       // lane: inside the slot
  // Stragglers set the kernel time (all environments wait for the one with the most Newton work), so environments that
  // were expensive in the previous env-step are dispatched first: slot s of block b runs environment order[b NS + s]
  // (neighbours in that order have similar work, which also keeps the slots of one wavefront together).
  const int eidx = blockIdx.x * NS + slot;
  const bool valid = eidx < a.B;                                        // a batch that is no multiple of NS: idle slot
  const int env = a.order ? a.order[min(eidx, a.B - 1)] : min(eidx, a.B - 1);
  Ctx<R> c; ctx_init(c, a.I, a.F, lds, NS, slot, lane, LPE, a.stage_cpt != 0, a.Fenv ? a.Fenv + (size_t)env * a.fstride : nullptr);
  const int nr = c.nr, nu = c.nu, REC = ts_rec(nr, nu, (int)sizeof(R));
  init_world(c, lane, LPE);
  {
    const R* st = a.tape + ((size_t)a.t0 * a.B + env) * REC;
    if (lane < nr) { c.q0D[lane] = rec_q(st)[lane]; c.q0[lane] = (R)c.q0D[lane]; c.qd0[lane] = st[rec_qd<R>(nr) + lane]; }
  }
  TS_SYNC();
  R* dlbase = c.dq + nr;
  int bad = 0; bool nonfinite = false;
  int evals = 0;
  R gmax = R(0);
  const bool bdf2_model = ts_u(c.I[TSIM_IH_INTEGRATOR]) == 2;
  // BDF2 history (the state before the previous sub-step).  While recording it is tape record t0 - 1 — so taped sub-step t is a BDF2
  // step exactly when t >= 2, which is what the adjoint kernel assumes (also after the tape was swapped by the backward cache);
  // without a tape it is the batch's `prev` buffer.
  bool has_prev = a.record ? a.t0 >= 1 : a.has_prev != 0;
  if (bdf2_model && has_prev && lane < nr) {
    if (a.record) {
      const R* pr = a.tape + ((size_t)(a.t0 - 1) * a.B + env) * REC;
      c.qm1D[lane] = rec_q(pr)[lane]; c.qm1[lane] = (R)c.qm1D[lane]; c.qdm1[lane] = pr[rec_qd<R>(nr) + lane];
    } else {
      c.qm1D[lane] = a.prev[(size_t)env * 2 * nr + lane]; c.qm1[lane] = (R)c.qm1D[lane];
      c.qdm1[lane] = (R)a.prev[(size_t)env * 2 * nr + nr + lane];
    }
  }
  TS_SYNC();
  // A launch covers nframes env-steps (1 for tsim_step).  With nframes > 1 an environment never waits for the slowest
  // environment of the batch between env-steps: Newton stragglers average out over the episode.
#ifdef TS_PP_TIME      // A/B builds only: share of the launch spent in the policy call, left in gnorm (tools/closed_loop_breakdown.py)
  long long pp_cycles_ = 0; const long long pp_t0_ = clock64();
#endif
  for (int f = 0; f < a.nframes; ++f) {
  {
    R uv = R(0);
    if (POLICY) {
      // closed loop: the action comes from the policy, evaluated by this slot on the observation the previous frame left
      // (tsim_policy_push.h).  The tactile frame was written by this slot: make the stores visible to its own loads first.
      ts_own_stores_visible();
      const R* tprev = a.pol.mode != TSIM_PUSH_OBS_TACTILE ? nullptr : (f == 0 ? a.pol.tac0 + (size_t)env * PP_NTAC : a.tac_out + ((size_t)(f - 1) * a.B + env) * PP_NTAC);
#ifdef TS_PP_TIME
      const long long tp0_ = clock64();
#endif
      push_policy_forward<LPE>(c, lane, valid, a.pol, (size_t)f * a.B + env, env, tprev);
      TS_SYNC();
#ifdef TS_PP_TIME
      pp_cycles_ += clock64() - tp0_;
#endif
      if (lane < nu) uv = c.u[lane];
    } else
    if (lane < nu) { uv = a.u[((size_t)f * a.B + env) * nu + lane]; c.u[lane] = uv; }
    // a NaN / inf control would be clamped away silently by the motor law's fmin / fmax: flag it (status bit 30) instead
    if (seg_sum<LPE>((uv - uv == R(0)) ? R(0) : R(1)) > R(0)) nonfinite = true;
  }
  TS_SYNC();
  for (int s = 0; s < a.nsub; ++s) {
    // force-free predictor of the implicit step and the coefficients of qd1, qdd1 in the increment
    if (bdf2_model && has_prev) {
      c.cv = R(1.5) / c.h; c.ca = R(2.25) / (c.h * c.h);
      if (lane < nr) {
        const double hD = (double)c.h;
        const double qp = 4.0 / 3 * c.q0D[lane] - 1.0 / 3 * c.qm1D[lane] + hD * (8.0 / 9 * (double)c.qd0[lane] - 2.0 / 9 * (double)c.qdm1[lane]);
        c.qpD[lane] = qp; c.qp[lane] = (R)qp;
        c.qdp[lane] = (R)((3.0 * qp - 4.0 * c.q0D[lane] + c.qm1D[lane]) / (2.0 * hD));
      }
    } else {
      c.cv = R(1) / c.h; c.ca = R(1) / (c.h * c.h);
      if (lane < nr) { c.qpD[lane] = c.q0D[lane] + (double)c.h * (double)c.qd0[lane]; c.qp[lane] = (R)c.qpD[lane]; c.qdp[lane] = c.qd0[lane]; }
    }
    const R sq = R(1), sv = c.cv, sa = c.ca;
    if (lane < nr) c.dl[lane] = R(0);          // initial guess: the predictor
    TS_SYNC();
    // Newton with backtracking EXACTLY as the model file states it (<solver_option tol max_iter max_ls>, pusher.xml:4): up to max_iter
    // iterations; each halves the step until ||g|| decreases, at most max_ls times, and takes the last trial if none did; converged when
    // ||g||_2 < tol.  Nothing else: no non-monotone steps, no restart, no trust region (rounds 1-2 had all three, tuned for the slowest
    // wavefront; on the stiff TactileInsertion grasp their full Newton step across a kink "converged" to a root 0.15 rad away from the one
    // plain backtracking reaches — found by the oracle's literal solver in round 3).  The oracle (oracle/tsim_oracle.cpp
    // substep_literal) is the same loop in fp64; the fp64 kernels take its iterates.
    // Written as a state machine around ONE evaluate call site (code size matters: the evaluation is ~6k instructions and two inlined
    // copies overflow the instruction cache): an accepted trial's evaluation is the next iteration's Jacobian evaluation.  The state
    // is per slot (identical in all lanes of a slot); a slot that has finished its sub-step keeps evaluating at its final iterate
    // (same numbers again) until every slot of the wavefront has finished.
    // Around that loop, two options (tsim_set_solver_options), both visible to the caller and both OFF for fp64 batches by default —
    // the fp64 kernels ARE the loop:
    //  * cross_kinks (fp32 default: on).  ||g|| has non-smooth local minima at contact / friction kinks: the iterate sits on the kink,
    //    every step along the Newton direction lands on the other piece with a larger ||g||.  The literal loop halves its way down to
    //    step lengths of 1e-6, takes the last trial anyway (it IS non-monotone there), which puts the iterate just across the kink, and
    //    converges from the other side — after 130 - 190 evaluations in fp64 (18 of 819 200 TactilePush sub-steps).  In fp32 the
    //    comparisons at those step lengths drown in rounding: noise-sized "decreases" are accepted for up to max_iter iterations (8 of
    //    those 18 sub-steps ended non-converged after ~2000 evaluations each, k_forward 5x slower; profiles/r03_solver_probe.md).
    //    With the option, and ONLY close to convergence (||g|| < TSIM_KINK_FACTOR x tol, where the Newton step is small: <= 1.3e-3 on
    //    those 18), a trial still rejected after TSIM_KINK_LS halvings is followed by the FULL Newton step across the kink, at most
    //    TSIM_KINK_MAX times per sub-step: ~20 evaluations, the same root as the literal loop wherever that converges
    //    (tests/test_gpu_literal.py).  Far from convergence nothing changes: rounds 1-2 took such steps anywhere, and on the stiff
    //    TactileInsertion grasp (||g|| ~ 1e-3, steps of 0.02 - 0.5) that reached roots 0.15 rad away from the literal one.
    //  * eval_budget (default 0 = none): an upper bound on the evaluations of one sub-step for throughput-minded roll-out collection;
    //    a sub-step cut short is flagged non-converged in status.
    R gn = R(0), alpha = R(1);
    int iter = 0, ls = -1, sub_evals = 0, crossings = 0;       // ls < 0: the evaluation just done is not a line-search trial
    bool conv = false, fin = false, forced = false;
    while (true) {
      evaluate<R, NRM, EXPJ, LPE>(c, lane, sq, sv, sa);
      const R gnew = block_norm2<LPE>(c.g, nr, lane);
      bool solve = false;
      if (!fin) {
        ++evals; ++sub_evals;
        bool take = false;                     // the point just evaluated becomes the iterate
        if (ls >= 0 && !forced && !(gnew < gn)) {                                  // a rejected trial
          if (a.cross_kinks && ls >= min(c.max_ls, TSIM_KINK_LS) && crossings < TSIM_KINK_MAX && gn < R(TSIM_KINK_FACTOR) * c.tol) {
            ++crossings; forced = true;        // close to convergence and no decrease down to 2^-TSIM_KINK_LS: the full step across the kink
            if (lane < nr) c.dl[lane] = dlbase[lane] + c.dq[lane];
          } else if (ls < c.max_ls) {          // halve the step
            alpha *= R(0.5); ++ls;
            if (lane < nr) c.dl[lane] = dlbase[lane] + alpha * c.dq[lane];
          } else take = true;                  // the literal loop: the last trial is taken anyway
        } else take = true;                    // the first evaluation of the sub-step, an accepted trial, or the step across a kink
        if (take) {
          forced = false;
          if (ls >= 0) ++iter;
          gn = gnew;
          if (!(gn == gn)) { nonfinite = true; fin = true; }
          else if (gn < c.tol) { conv = true; fin = true; }
          else if (iter >= c.max_iter || (a.eval_budget > 0 && sub_evals >= a.eval_budget)) fin = true;
          else {
            solve = true;
            if (lane < nr) { c.rhs[lane] = -c.g[lane]; dlbase[lane] = c.dl[lane]; }
          }
        }
      }
      TS_SYNC();
      if (__any(solve)) {
        solve_lanes<R, NRM, LPE, double>(c.H, c.rhs, c.dq, nr, false, lane, solve);
        if (solve) {
          alpha = R(1); ls = 0;
          if (lane < nr) c.dl[lane] = dlbase[lane] + c.dq[lane];
        }
        TS_SYNC();
      }
      if (__all(fin)) break;
    }
    if (!conv) ++bad;
    gmax = t_max(gmax, gn);
    // commit the sub-step: c.q = q1, c.qd = (q1 - q0)/h, c.H = dg/dq1 at q1
    if (a.record && valid) {
      R* rec = a.tape + ((size_t)(a.t0 + f * a.nsub + s + 1) * a.B + env) * REC;
      if (lane < nr) { rec_q(rec)[lane] = c.qD[lane]; rec[rec_qd<R>(nr) + lane] = c.qd[lane]; }
      for (int e = lane; e < nr * nr; e += LPE) rec[rec_H<R>(nr) + e] = c.H[e];
      if (lane < nu) rec[rec_u<R>(nr) + lane] = c.u[lane];
    }
    TS_SYNC();
    if (lane < nr) {
      c.qm1[lane] = c.q0[lane]; c.qm1D[lane] = c.q0D[lane]; c.qdm1[lane] = c.qd0[lane];
      c.q0[lane] = c.q[lane]; c.q0D[lane] = c.qD[lane]; c.qd0[lane] = c.qd[lane];
    }
    has_prev = true;
    TS_SYNC();
  }
  if (lane < nr && valid) {
    const size_t o = ((size_t)f * a.B + env) * nr + lane;
    if (a.q_out) a.q_out[o] = (R)c.q0D[lane];        // the double position rounded once (== tsim_get_state)
    if (a.qd_out) a.qd_out[o] = c.qd0[lane];
  }
  // link poses / velocities in LDS are those of the accepted state (last evaluation)
  const int tslot = a.tac_slot ? a.tac_slot[f] : f;
  readout<LPE>(c, lane, env, valid, valid && tslot >= 0, a.var_out != nullptr, a.tac_out != nullptr && tslot >= 0,
               a.var_out ? a.var_out + (size_t)f * a.B * 3 * c.nvar : nullptr,
               (a.tac_out && tslot >= 0) ? a.tac_out + (size_t)tslot * a.B * 3 * c.ntax : nullptr);
  TS_SYNC();
  }
  if (a.poseR) {
    // Large pads are read out on demand (tsim_readout), by a kernel whose lanes are taxels and which needs, per (sensor, primitive)
    // combination, the pose of the sensor link in the primitive's frame and the relative twist there.  The link records in LDS are those
    // of the state this launch ends in: leave the pose records here and the read-out needs no kinematics kernel of its own.
    int k = 0;
    for (int s = 0; s < c.nsensor; ++s) {
      const int* si = c.I + c.off_sensor + s * TSIM_SI_SIZE;
      const int nsp = ts_u(si[TSIM_SI_NSPRIM]), sp0 = ts_u(si[TSIM_SI_SPRIM0]);
      for (int j = 0; j < nsp; ++j, ++k) {
        TS_SYNC();
        pair_stage_value(c, ts_u(c.I[c.off_sprim + sp0 + j]), 0, lane == 0);
        TS_SYNC();
        const size_t rec = (size_t)env * a.nspt + k;
        if (valid) {
          for (int e = lane; e < TP_R_SIZE; e += LPE) a.poseR[rec * TP_R_SIZE + e] = c.PP[e];
          if (lane < TP_D_SIZE) a.poseD[rec * TP_D_SIZE + lane] = c.PPd[lane];
        }
      }
    }
  }
  if (valid) {
    if (bdf2_model && lane < nr) { a.prev[(size_t)env * 2 * nr + lane] = c.qm1D[lane]; a.prev[(size_t)env * 2 * nr + nr + lane] = (double)c.qdm1[lane]; }
    if (!a.record) {
      R* st = a.tape + ((size_t)a.t0 * a.B + env) * REC;
      if (lane < nr) { rec_q(st)[lane] = c.q0D[lane]; st[rec_qd<R>(nr) + lane] = c.qd0[lane]; }
    }
    if (a.status && lane == 0) a.status[env] = bad | (nonfinite ? (1 << 30) : 0);
    if (a.evals && lane == 0) a.evals[env] = evals;
    if (a.gnorm && lane == 0) a.gnorm[env] = (float)gmax;
#ifdef TS_PP_TIME
    if (a.gnorm && lane == 0) a.gnorm[env] = (float)((double)pp_cycles_ / (double)(clock64() - pp_t0_));
#endif
  }
}

// ================================================================================================ LPT ordering
// One block: counting sort of the environments by their residual-evaluation count of the last launch, descending
// (64 bins; order inside a bin is irrelevant), dealt out to the wavefronts like cards: rank r goes to slot r / nwaves of
// wavefront r % nwaves.  The expensive environments start first AND sit in different wavefronts — the slots of a
// wavefront are sub-step-synchronous, so two expensive environments in one wavefront cost the sum of their per-sub-step
// maxima (measured: a batch sorted by work runs 22 % slower than the unsorted one, profiles/r01_imbalance_exp.json).
// Runs on the same stream right after k_forward.
// Episode launches (nframes > 1) are ordered by the per-environment TOTALS of the previous episode launch of the same length — useful
// when consecutive episodes resemble each other (the bench replays one table; a GD epoch with fixed start states and a slowly
// changing policy), harmless when they do not (any order is a valid one).  bin = (evals - lo) * 64 / span maps the totals onto the
// 64 bins (lo = 2 evaluations per sub-step, the minimum of a converging Newton loop; span = 2.5 per sub-step); per-step launches keep
// bin = evals (lo 0, span 64).
__global__ void __launch_bounds__(1024) k_order_by_evals(const int* evals, int* order, int B, int ns, int lo, int span) {
  __shared__ int hist[64], base[64];
  const int t = threadIdx.x;
  if (t < 64) hist[t] = 0;
  __syncthreads();
  auto bin = [lo, span](int ev) { return min(max((ev - lo) * 64 / span, 0), 63); };
  for (int e = t; e < B; e += 1024) atomicAdd(&hist[bin(evals[e])], 1);
  __syncthreads();
  if (t == 0) { int acc = 0; for (int k = 63; k >= 0; --k) { base[k] = acc; acc += hist[k]; } }
  __syncthreads();
  const int nwaves = (B + ns - 1) / ns;
  for (int e = t; e < B; e += 1024) {
    const int r = atomicAdd(&base[bin(evals[e])], 1);           // rank of environment e, 0 = most expensive
    order[(r % nwaves) * ns + r / nwaves] = e;                  // slot r / nwaves of wavefront r % nwaves (B % ns == 0)
  }
}

// ================================================================================================ read-out kernels
// get_variables / get_tactile_force_vector on demand (tsim_readout), in two launches:
//   k_readout   one wavefront per environment: forward kinematics from the taped state, the end-effector variables, and — for the
//               taxel kernel — the pose record of every (sensor, primitive) combination: pose of the sensor link in the primitive's
//               frame (R precision and double) and the relative twist there (what pair_stage_value stages in LDS for the in-kernel
//               read-out of k_forward);
//   k_taxels    lanes = taxels, nothing else: 3 position constants, an fp32 "certainly outside" test, the double-precision position,
//               the penalty law, 12 B out (the 9 axis constants only where a force acts).  ~40 registers instead of the 178 the
//               kinematics need, so 8+ wavefronts per SIMD cover the L2 latency of the constants: this is the one kernel of the path
//               whose time is memory traffic (RollingBall: 40 000 taxels, 480 KB per environment and read-out).
template <class R> struct ReadArgs { const int* I; const R* F; const R* Fenv; int fstride; int B, t0; const R* tape; R* var_out; R* poseR; double* poseD; int nspt; int stage_cpt; };

template <class R>
__global__ void __launch_bounds__(TS_WAVE) k_readout(ReadArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  const int env = blockIdx.x, lane = threadIdx.x;
  Ctx<R> c; ctx_init(c, a.I, a.F, lds, 1, 0, lane, TS_WAVE, a.stage_cpt != 0, a.Fenv ? a.Fenv + (size_t)blockIdx.x * a.fstride : nullptr);
  const int nr = c.nr, REC = ts_rec(nr, c.nu, (int)sizeof(R));
  init_world(c, lane, TS_WAVE);
  const R* st = a.tape + ((size_t)a.t0 * a.B + env) * REC;
  if (lane < nr) { c.qD[lane] = rec_q(st)[lane]; c.q[lane] = (R)c.qD[lane]; c.qd[lane] = st[rec_qd<R>(nr) + lane]; c.qa[lane] = R(0); }
  TS_SYNC();
  phase1<R, false, true>(c, lane, R(0), R(0), R(0));
  readout<TS_WAVE>(c, lane, env, true, false, a.var_out != nullptr, false, a.var_out, (R*)nullptr);       // variables only
  if (!a.poseR) return;
  int k = 0;
  for (int s = 0; s < c.nsensor; ++s) {
    const int* si = c.I + c.off_sensor + s * TSIM_SI_SIZE;
    for (int j = 0; j < si[TSIM_SI_NSPRIM]; ++j, ++k) {
      TS_SYNC();
      pair_stage_value(c, c.I[c.off_sprim + si[TSIM_SI_SPRIM0] + j], 0, lane == 0);
      TS_SYNC();
      const size_t rec = (size_t)env * a.nspt + k;
      if (lane < TP_R_SIZE) a.poseR[rec * TP_R_SIZE + lane] = c.PP[lane];          // PP_RPA (9), PP_PPA (3), PP_WREL (3), PP_VREL (3)
      if (lane < TP_D_SIZE) a.poseD[rec * TP_D_SIZE + lane] = c.PPd[lane];
    }
  }
}

template <class R> struct TaxArgs { const int* I; const R* F; const R* Fenv; int fstride; const R* poseR; const double* poseD; int nspt; R* tac_out; int slice;
  // model constants the host knows (header entries and the offset of the staging table): as kernel arguments they cost the prologue no
  // dependent global loads (header -> table offset -> table -> record was four L2 round trips per block, ~2 us of a 28 us kernel)
  int ntax, nsensor, foff_sensor, foff_pair, foff_taxel, tt_off; };

// Per-block staging of everything k_taxels needs that does not depend on the taxel: the block works on ONE environment, so the sensor
// ranges, the primitive of every (sensor, primitive) record with its shape, the sensors' penalty parameters and the pose records are
// put in LDS once; inside the taxel loop the only global accesses left are the taxel's own constants and its 12 output bytes.  (Read
// from global memory in the loop they are chains of dependent ~600-cycle loads — index -> record -> value — and the loop was exactly
// that latency: 55 us for 256 x 40 000 taxels.)  The integer part of the staging (per sensor: end of its taxel range, first record,
// number of records; per record: primitive type, contact pair) is a MODEL constant: the host builds it once (build_sched, ts_tax_table)
// so that the prologue is one cooperative copy instead of a serial walk of dependent global loads by thread 0.
enum { TX_MAXS = 16, TX_MAXK = 64 };
// the three outputs of a taxel, as NON-TEMPORAL stores: the read-out is a pure write stream (0.49 GB at 1024 environments, 1.97 GB at 4096: 2 - 7 x
// the Infinity Cache) that nothing on the device reads back.  Measured (profiles/r04_readout_hbm.md): 1024 environments 4.45 -> 5.40 TB/s,
// 4096 environments 4.4 - 4.8 -> 4.7 - 5.0 TB/s; -DTS_TAX_PLAIN restores ordinary stores (A/B).
template <class R> __device__ __forceinline__ void ts_store3(R* o, R a, R b, R c) {
#ifdef TS_TAX_PLAIN
  o[0] = a; o[1] = b; o[2] = c;
#else
  __builtin_nontemporal_store(a, o); __builtin_nontemporal_store(b, o + 1); __builtin_nontemporal_store(c, o + 2);
#endif
}
#ifndef TS_TAX_UNROLL
#define TS_TAX_UNROLL 2      // taxels per thread and loop iteration: their loads are in flight together (A/B: profiles/r03_readout_ab.md)
#endif
template <class R>
__global__ void __launch_bounds__(256) k_taxels(TaxArgs<R> a) {
  const int env = blockIdx.x;
  const int* I = a.I;
  const R* F = a.Fenv ? a.Fenv + (size_t)env * a.fstride : a.F;        // this environment's float records (domain randomisation)
  const int ntax = a.ntax, nsensor = a.nsensor;
  __shared__ int sEnd[TX_MAXS], sKb[TX_MAXS], sNsp[TX_MAXS], sPrim[TX_MAXK];
  __shared__ R sSf[TX_MAXS * TSIM_SF_SIZE], sShape[TX_MAXK * 4], sP[TX_MAXK * TP_R_SIZE];
  __shared__ double sD[TX_MAXK * TP_D_SIZE];
  __shared__ __attribute__((aligned(16))) R sC[TX_MAXK * 4];          // per record: the primitive's centre in the sensor-link frame, (bounding radius + margin)^2
  {
    const int* TT = I + a.tt_off;
    const int foff_sensor = a.foff_sensor, foff_pair = a.foff_pair;
    for (int i = threadIdx.x; i < nsensor; i += 256) { sEnd[i] = TT[3 * i]; sKb[i] = TT[3 * i + 1]; sNsp[i] = TT[3 * i + 2]; }
    for (int i = threadIdx.x; i < a.nspt; i += 256) sPrim[i] = TT[3 * nsensor + 2 * i];
    for (int i = threadIdx.x; i < a.nspt * 4; i += 256) sShape[i] = F[foff_pair + TT[3 * nsensor + 2 * (i >> 2) + 1] * TSIM_PF_SIZE + TSIM_PF_SHAPE + (i & 3)];
    for (int i = threadIdx.x; i < nsensor * TSIM_SF_SIZE; i += 256) sSf[i] = F[foff_sensor + i];
    for (int i = threadIdx.x; i < a.nspt * TP_R_SIZE; i += 256) sP[i] = a.poseR[(size_t)env * a.nspt * TP_R_SIZE + i];
    for (int i = threadIdx.x; i < a.nspt * TP_D_SIZE; i += 256) sD[i] = a.poseD[(size_t)env * a.nspt * TP_D_SIZE + i];
    __syncthreads();
    // "Certainly outside" in 7 instructions: a taxel at x_A (sensor-link frame) cannot touch a primitive whose bounding sphere (centre
    // c_A = -R_PA^T p_PA, radius r_b) it is farther from than r_b + TS_FAR_MARGIN.  Most taxels of a large pad are nowhere near the
    // primitive, and for them this replaces the rotation into the primitive's frame (12 LDS reads, ~25 instructions); the margin is
    // far above the rounding of the test, so the set of taxels contact_law accepts is unchanged.  Planes have no bound (radius < 0).
    for (int k = threadIdx.x; k < a.nspt; k += 256) {
      const R* P = sP + k * TP_R_SIZE;
      const V3<R> cA = mulMtv(ldm(P), ldv(P + 9)) * R(-1);
      const R* sh = sShape + k * 4;
      const int prim = sPrim[k];
      R rb = R(-1);
      if (prim == TSIM_P_SPHERE) rb = sh[0];
      else if (prim == TSIM_P_CUBOID) rb = t_sqrt(sh[0] * sh[0] + sh[1] * sh[1] + sh[2] * sh[2]);
      else if (prim == TSIM_P_CYLINDER) rb = t_sqrt(sh[0] * sh[0] + sh[1] * sh[1]);
      sC[4 * k] = cA.x; sC[4 * k + 1] = cA.y; sC[4 * k + 2] = cA.z;
      sC[4 * k + 3] = rb < R(0) ? R(-1) : (rb + R(TS_FAR_MARGIN)) * (rb + R(TS_FAR_MARGIN));
    }
    __syncthreads();
  }
  const R* tax = a.F + a.foff_taxel;                          // SoA planes: position (3), axis0, axis1, normal (9); shared
  R* out = a.tac_out + (size_t)env * 3 * ntax;
  const int te = min(ntax, ((int)blockIdx.y + 1) * a.slice);
  // One taxel: its three outputs go out as ONE 12-byte (fp64: 24-byte) store per lane — consecutive lanes, consecutive addresses: a
  // wavefront's store instruction covers 768 contiguous bytes (global_store_dwordx3 in the ISA).  Routing them through LDS for 16-byte
  // vectors instead was measured twice and is slower both ways (block-wide with barriers: round 2; wave-private without: round 3,
  // 2.45 -> 2.12 TB/s on the bench leg, profiles/r03_readout_ab.md).
  auto taxel = [&](int t, V3<R> xa) {
    int s = 0;                                                         // sensor of taxel t
    while (s < nsensor - 1 && t >= sEnd[s]) ++s;
    const int kb = sKb[s], nsp = sNsp[s];
    const R* sf = sSf + s * TSIM_SF_SIZE;
    const R* tp = tax + t;
    V3<R> Fl = zero3<R>();                                             // force on the taxel, sensor-link frame
    for (int j = 0; j < nsp; ++j) {
      const R* Cc = sC + (kb + j) * 4;
      const V3<R> dc = xa - ldv(Cc);
      if (Cc[3] >= R(0) && dot3(dc, dc) > Cc[3]) continue;              // outside the primitive's bounding sphere (+ margin)
      const int prim = sPrim[kb + j];
      const R* shape = sShape + (kb + j) * 4;
      const R* P = sP + (kb + j) * TP_R_SIZE;
      const double* D = sD + (kb + j) * TP_D_SIZE;
      const M3<R> RPA = ldm(P);
      // fp32 kernels: the exact shape's distance from an fp32 position, before the double-precision one
      if (sizeof(R) == 4 && !(prim_distance<R>(prim, shape, mulMv(RPA, xa) + ldv(P + 9)) < R(TS_FAR_MARGIN))) continue;
      const V3<double> xPd = mulMv(ldm(D), cvt3<double>(xa)) + ldv(D + 9);
      const V3<R> xP = cvt3<R>(xPd);
      V3<R> Fc; M3<R> Jx, Jv;
      if (contact_law<R, false>(prim, shape, sf, xP, ldv(P + 15) + cross3(ldv(P + 12), xP), Fc, Jx, Jv, xPd)) Fl = Fl + mulMtv(RPA, Fc);
    }
    R o0 = R(0), o1 = R(0), o2 = R(0);
    if (Fl.x != R(0) || Fl.y != R(0) || Fl.z != R(0)) {                // the nine axis constants only for taxels that carry a force
      o0 = Fl.x * tp[3 * ntax] + Fl.y * tp[4 * ntax] + Fl.z * tp[5 * ntax];
      o1 = Fl.x * tp[6 * ntax] + Fl.y * tp[7 * ntax] + Fl.z * tp[8 * ntax];
      o2 = Fl.x * tp[9 * ntax] + Fl.y * tp[10 * ntax] + Fl.z * tp[11 * ntax];
    }
#ifdef TS_TAX_ZEROS        // A/B only: the store pattern alone (no taxel arithmetic) — the ceiling of this write stream
    o0 = o1 = o2 = R(0);
#endif
    ts_store3(out + 3 * t, o0, o1, o2);
  };
  if (nsensor == 1 && a.nspt == 1 && sC[3] >= R(0)) {
    // One sensor against one bounded primitive (RollingBall's pad and ball, TactilePush's pad and box), in two passes per 1024 taxels:
    //   pass 1  lanes = taxels: 3 loads, the bounding-sphere test against centre / radius^2 held in registers (7 instructions), zeros
    //           stored for the taxels outside; the few inside are appended to a list in LDS;
    //   pass 2  lanes = LISTED taxels: the penalty law, with full wavefronts.
    // Without the list a wavefront runs the ~1000-instruction law whenever ONE of its 64 taxels is near the primitive — 15 % of the
    // wavefronts of the RollingBall pad for 5 % of its taxels, and that was the kernel's time (12.4 M vector instructions for
    // 10.2 M taxels; profiles/r03_readout_ab.md).  A taxel's result does not depend on the lane that computes it: same bits as before.
#ifndef TS_TAX_CH
#define TS_TAX_CH 8
#endif
    enum { CH = TS_TAX_CH };
    __shared__ int sList[256 * CH];
    __shared__ int sCount;
    const V3<R> cA = ldv(sC);
    const R r2 = sC[3];
    const R* px = tax; const R* py = tax + ntax; const R* pz = tax + 2 * ntax;
    // chunks are dealt to the environment's blocks round-robin, not as contiguous slices: the taxels near the primitive lie in a band of
    // the pad (RollingBall: ~50 of 200 rows), and with contiguous slices two of an environment's eight blocks ran the law for all of them
    // while the others stored zeros — the kernel lasted as long as those two
    const int te = ntax;
    for (int base = (int)blockIdx.y * 256 * CH; base < te; base += (int)gridDim.y * 256 * CH) {
      if (threadIdx.x == 0) sCount = 0;
      __syncthreads();
      V3<R> xa[CH];
#pragma unroll
      for (int r = 0; r < CH; ++r) {                                   // all loads of the chunk in flight together
        const int tr = min(base + (int)threadIdx.x + 256 * r, te - 1);
        xa[r] = mk3<R>(px[tr], py[tr], pz[tr]);
      }
#pragma unroll
      for (int r = 0; r < CH; ++r) {
        const int tr = base + (int)threadIdx.x + 256 * r;
        if (tr < te) {
          const V3<R> dc = xa[r] - cA;
          if (dot3(dc, dc) > r2) ts_store3(out + 3 * tr, R(0), R(0), R(0));
          else sList[atomicAdd(&sCount, 1)] = tr;
        }
      }
      __syncthreads();
      const int n = sCount;
      for (int i = threadIdx.x; i < n; i += 256) {
        const int tr = sList[i];
        taxel(tr, mk3<R>(px[tr], py[tr], pz[tr]));
      }
      __syncthreads();
    }
    return;
  }
  for (int t = (int)blockIdx.y * a.slice + (int)threadIdx.x; t < te; t += 256 * TS_TAX_UNROLL) {
    V3<R> xa[TS_TAX_UNROLL];
#pragma unroll
    for (int r = 0; r < TS_TAX_UNROLL; ++r) {                          // positions of all of this iteration's taxels first: loads in flight together
      const int tr = min(t + 256 * r, te - 1);
      xa[r] = mk3<R>(tax[tr], tax[tr + ntax], tax[tr + 2 * ntax]);
    }
#pragma unroll
    for (int r = 0; r < TS_TAX_UNROLL; ++r)
      if (t + 256 * r < te) taxel(t + 256 * r, xa[r]);
  }
}

// ================================================================================================ branch signature
// Diagnostics (tsim_debug_signature): for the taped sub-steps t_first+1 .. t_first+n of every environment, which contact
// points / taxels penetrate and on which smooth piece of the penalty law (stick / slip, face of the primitive) each one is,
// recomputed from the taped state (q as double, qd) exactly as the backward kernel re-evaluates it.  Two trajectories with
// equal signatures went through the same smooth pieces, so their gradients are comparable; where they differ, one of them
// crossed a contact / friction kink.  One environment per 64-lane block; lanes = points.
template <class R> struct SigArgs { const int* I; const R* F; const R* Fenv; int fstride; int B, t_first, n; const R* tape; unsigned* out; int stage_cpt; };

__device__ __forceinline__ unsigned wave_sum_u32(unsigned x) {
  for (int off = 32; off > 0; off >>= 1) x += (unsigned)lane_gather((int)x, (int)threadIdx.x ^ off);
  return x;
}

template <class R>
__global__ void __launch_bounds__(TS_WAVE) k_signature(SigArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  const int env = blockIdx.x, lane = threadIdx.x;
  Ctx<R> c; ctx_init(c, a.I, a.F, lds, 1, 0, lane, TS_WAVE, a.stage_cpt != 0, a.Fenv ? a.Fenv + (size_t)blockIdx.x * a.fstride : nullptr);
  const int nr = c.nr, REC = ts_rec(nr, c.nu, (int)sizeof(R));
  init_world(c, lane, TS_WAVE);
  for (int j = 0; j < a.n; ++j) {
    const R* st = a.tape + ((size_t)(a.t_first + 1 + j) * a.B + env) * REC;
    TS_SYNC();
    if (lane < nr) { c.qD[lane] = rec_q(st)[lane]; c.q[lane] = (R)c.qD[lane]; c.qd[lane] = st[rec_qd<R>(nr) + lane]; c.qa[lane] = R(0); }
    TS_SYNC();
    phase1<R, false, true>(c, lane, R(0), R(0), R(0));
    unsigned cnt = 0, sum = 0;
    for (int pk = 0; pk < c.npair; ++pk) {                     // dynamics-active contact pairs
      const int* pi = c.I + c.off_pair + pk * TSIM_PI_SIZE;
      if (!(pi[TSIM_PI_FLAGS] & 1)) continue;
      const R* pf = c.F + c.foff_pair + pk * TSIM_PF_SIZE;
      TS_SYNC();
      pair_stage_value(c, pk, 0, lane == 0);
      TS_SYNC();
      const R* S = c.PP;
      const M3<double> RPAd = ldm(c.PPd); const V3<double> pPAd = ldv(c.PPd + 9);
      const V3<R> wrel = ldv(S + PP_WREL), vrel = ldv(S + PP_VREL);
      for (int i = lane; i < pi[TSIM_PI_NPT]; i += TS_WAVE) {
        V3<double> xPd = mulMv(RPAd, cvt3<double>(ld_cpt(c, pi[TSIM_PI_PT0] + i))) + pPAd;
        if (pi[TSIM_PI_FLAGS] & 2) xPd.z -= (double)pf[TSIM_PF_SHAPE];
        const V3<R> xP = cvt3<R>(xPd);
        V3<R> F; M3<R> Jx, Jv; int br = 0;
        if (contact_law<R, false>(pi[TSIM_PI_PRIM], pf + TSIM_PF_SHAPE, pf + TSIM_PF_KN, xP, vrel + cross3(wrel, xP), F, Jx, Jv, xPd, &br)) {
          ++cnt; sum += ts_sig_mix((unsigned)pk, (unsigned)i, (unsigned)(1 + br));
        }
      }
    }
    for (int s = 0; s < c.nsensor; ++s) {                      // taxels against the primitives paired with their sensor
      const int* si = c.I + c.off_sensor + s * TSIM_SI_SIZE;
      const R* sf = c.F + c.foff_sensor + s * TSIM_SF_SIZE;
      for (int jp = 0; jp < si[TSIM_SI_NSPRIM]; ++jp) {
        const int pk = c.I[c.off_sprim + si[TSIM_SI_SPRIM0] + jp];
        const int* pi = c.I + c.off_pair + pk * TSIM_PI_SIZE;
        const R* pf = c.F + c.foff_pair + pk * TSIM_PF_SIZE;
        TS_SYNC();
        pair_stage_value(c, pk, 0, lane == 0);
        TS_SYNC();
        const R* S = c.PP;
        const V3<R> wrel = ldv(S + PP_WREL), vrel = ldv(S + PP_VREL);
        for (int i = lane; i < si[TSIM_SI_NTAX]; i += TS_WAVE) {
          const R* tp = c.Fg + c.foff_tax + si[TSIM_SI_TAX0] + i;
          const V3<double> xPd = mulMv(ldm(c.PPd), mk3<double>((double)tp[0], (double)tp[c.ntax], (double)tp[2 * c.ntax])) + ldv(c.PPd + 9);
          const V3<R> xP = cvt3<R>(xPd);
          V3<R> F; M3<R> Jx, Jv; int br = 0;
          if (contact_law<R, false>(pi[TSIM_PI_PRIM], pf + TSIM_PF_SHAPE, sf, xP, vrel + cross3(wrel, xP), F, Jx, Jv, xPd, &br)) {
            ++cnt; sum += ts_sig_mix(0x10000u + (unsigned)(si[TSIM_SI_SPRIM0] + jp), (unsigned)i, (unsigned)(1 + br));
          }
        }
      }
    }
    cnt = wave_sum_u32(cnt); sum = wave_sum_u32(sum);
    if (lane == 0) { unsigned* o = a.out + ((size_t)j * a.B + env) * 2; o[0] = cnt; o[1] = sum; }
  }
}

// ================================================================================================ debug evaluation
template <class R> struct DbgArgs { const int* I; const R* F; const R* Fenv; int fstride; int B; const R *q1, *q0, *qd0, *u; R *g, *H; long long* cyc; int stage_cpt; };

template <class R, int LPE>
__global__ void __launch_bounds__(TS_WAVE) k_debug_eval(DbgArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  constexpr int NS = TS_WAVE / LPE;
  const int slot = threadIdx.x / LPE, lane = threadIdx.x % LPE;
  const bool valid = (int)blockIdx.x * NS + slot < a.B;
  const int env = min((int)blockIdx.x * NS + slot, a.B - 1);
  Ctx<R> c; ctx_init(c, a.I, a.F, lds, NS, slot, lane, LPE, a.stage_cpt != 0, a.Fenv ? a.Fenv + (size_t)env * a.fstride : nullptr);
  const int nr = c.nr, nu = c.nu;
  init_world(c, lane, LPE);
  if (lane < nr) {
    c.q0[lane] = a.q0[(size_t)env * nr + lane]; c.qd0[lane] = a.qd0[(size_t)env * nr + lane];
    c.qp[lane] = c.q0[lane] + c.h * c.qd0[lane]; c.qdp[lane] = c.qd0[lane];
    c.dl[lane] = a.q1[(size_t)env * nr + lane] - c.qp[lane];
    c.qpD[lane] = (double)a.q1[(size_t)env * nr + lane] - (double)c.dl[lane];      // so that qD = qpD + dl is the given q1
  }
  if (lane < nu) c.u[lane] = a.u[(size_t)env * nu + lane];
  TS_SYNC();
  if (a.cyc) {   // shader-clock stamps (s_memtime) at the TS_STAMP points of one evaluation + the dense solve; one row
                 // per wavefront (the row of its first environment), the other rows stay zero
    c.stamps = a.cyc + (size_t)env * 32;
    evaluate<R, 8, false, LPE>(c, lane, R(1), R(1) / c.h, R(1) / (c.h * c.h));
    if (lane < nr) c.rhs[lane] = -c.g[lane];
    TS_SYNC();
    solve_lanes<R, 8, LPE>(c.H, c.rhs, c.dq, nr, false, lane);
    TS_STAMP(c);
    if (lane == 0 && valid) for (int i = (slot == 0 ? c.nstamp : 0); i < 32; ++i) c.stamps[i] = 0;
  } else {
    evaluate<R, 16, true, LPE>(c, lane, R(1), R(1) / c.h, R(1) / (c.h * c.h));
  }
  if (lane < nr && valid) a.g[(size_t)env * nr + lane] = c.g[lane];
  if (valid) for (int e = lane; e < nr * nr; e += LPE) a.H[(size_t)env * nr * nr + e] = c.H[e];
}

// ================================================================================================ backward kernel
template <class R> struct BwdArgs {
  const int* I; const R* F; const R* Fenv; int fstride;
  int B, n, t_end;
  int seed_stride;        // sub-step j (0 = oldest of the n) carries direct loss partials iff (j + 1) % seed_stride == 0
  int frames;             // 0: seeds [B][n / seed_stride][.], df_du [B][n][nu] per sub-step (tsim_backward_steps)
                          // 1: seeds [n / seed_stride][B][.], df_du [n / seed_stride][B][nu] summed per env-step (tsim_backward_episode)
  const int* tac_slot;    // frames mode: slot of frame f in df_dtac (< 0: no tactile seed), null: slot f
  const R* tape;
  const R *df_dq, *df_dvar, *df_dtac;
  R *lamq, *lamv, *df_du;
  int stage_cpt;
  long long* cyc;         // diagnostics: shader-clock stamps of the first sub-steps of wavefront 0 (tsim_debug_stamps), or null
  PushPolicy<R> pol;      // POLICY instantiations only (tsim_push_closed_backward)
};

// (M z)_j for lane j, M = sum_i J_i^T I_i J_i:  lanes = links form f_i = I_i (sum_{k above i} W_k z_k) in the (idle) pair-staging
// scratch, then lanes = dofs add up W_j . f_i over the links below dof j.  (The direct double loop per lane was ~600
// instructions, a tenth of an adjoint sub-step.)
template <int LPE, class R>
__device__ __forceinline__ R mass_times_z(const Ctx<R>& c, int lane) {
  const int* LR = c.LI + ts_sched_rec(c.LI);
  R* fi = c.PT;                                    // [nl + 1][6], free between phase 2 and the next staging
  for (int i = 1 + lane; i <= c.nl; i += LPE) {
    const int anc = LR[(i - 1) * TS_LR_SIZE + TS_LR_ANCMASK];
    S6<R> A = zero6<R>();
    for (int k = 0; k < c.nr; ++k)
      if ((anc >> k) & 1) A = A + ld6(c.WP + k * 6) * c.z[k];
    const R* X = c.LP + i * LK_SIZE;
    st6(fi + i * 6, imul(c.F[c.foff_link + (i - 1) * TSIM_LF_SIZE + TSIM_LF_MASS], ldv(X + LK_C), X + LK_IC, A));
  }
  TS_SYNC();
  R tau = R(0);
  if (lane < c.nr) {
    const S6<R> Wj = ld6(c.WP + lane * 6);
    for (int i = 1; i <= c.nl; ++i)
      if ((LR[(i - 1) * TS_LR_SIZE + TS_LR_ANCMASK] >> lane) & 1) tau += dot6(Wj, ld6(fi + i * 6));
  }
  TS_SYNC();
  return tau;
}

// lam_q += (dvar/dq)^T w_var + (dtac/dq)^T w_tac ; lam_v += (dtac/dqd)^T w_tac, at the state whose link values and
// q-tangents (seeds (1,0,0)) are in LDS.  Tactile: reverse mode at the taxel level — each lane forms the gradient of
// w . out w.r.t. the pair's relative displacement and relative twist (12 numbers, primitive frame); one reduction
// per (sensor, primitive); lanes = directions then dot it with the pair's per-direction records.
template <int LPE, class R>
__device__ __forceinline__ void output_vjp(const Ctx<R>& c, int lane, const R* wvar, const R* wtac) {
  const int nr = c.nr;
  if (wvar && lane < nr) {
    R acc = R(0);
    const S6<R> Wk = ld6(c.WP + lane * 6);
    for (int e = 0; e < c.nvar; ++e) {
      const int l = c.I[c.off_var + e * TSIM_VI_SIZE + TSIM_VI_LINK];
      if (!((anc_of(c.I, c.off_link, l) >> lane) & 1)) continue;
      const V3<R> x = mulMv(ldm(c.LP + l * LK_SIZE + LK_R), ldv(c.F + c.foff_var + e * TSIM_VF_SIZE)) + ldv(c.LP + l * LK_SIZE + LK_P);
      const V3<R> J = cross3(Wk.a, x) + Wk.l;
      acc += wvar[3 * e] * J.x + wvar[3 * e + 1] * J.y + wvar[3 * e + 2] * J.z;
    }
    c.lamq[lane] += acc;
  }
  TS_SYNC();
  if (!wtac) return;
  for (int s = 0; s < c.nsensor; ++s) {
    const int* si = c.I + c.off_sensor + s * TSIM_SI_SIZE;
    const R* sf = c.F + c.foff_sensor + s * TSIM_SF_SIZE;
    const int t0 = ts_u(si[TSIM_SI_TAX0]), nt = ts_u(si[TSIM_SI_NTAX]), sp0 = ts_u(si[TSIM_SI_SPRIM0]), nsp = ts_u(si[TSIM_SI_NSPRIM]);
    for (int j = 0; j < nsp; ++j) {
      const int pk = ts_u(c.I[c.off_sprim + sp0 + j]);
      const int prim = ts_u(c.I[c.off_pair + pk * TSIM_PI_SIZE + TSIM_PI_PRIM]);
      const R* pf = c.F + c.foff_pair + pk * TSIM_PF_SIZE;
      TS_SYNC();
      pair_stage_value(c, pk, 0, lane == 0);
      TS_SYNC();
      const R* S = c.PP;
      const M3<R> RPA = ldm(S + PP_RPA);
      const V3<R> pPA = ldv(S + PP_PPA), wrel = ldv(S + PP_WREL), vrel = ldv(S + PP_VREL);
      R g[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) g[e] = R(0);
      bool any_live = false;
      // the seed and the position of a chunk's taxels are fetched one chunk ahead: a lone wavefront cannot hide the two dependent
      // global-memory latencies per chunk (seed -> live? -> position) otherwise
      R nw0 = R(0), nw1 = R(0), nw2 = R(0), nx0 = R(0), nx1 = R(0), nx2 = R(0);
      if (lane < nt) {
        const int t = t0 + lane; const R* tp = c.Fg + c.foff_tax + t;
        nw0 = wtac[3 * t]; nw1 = wtac[3 * t + 1]; nw2 = wtac[3 * t + 2]; nx0 = tp[0]; nx1 = tp[c.ntax]; nx2 = tp[2 * c.ntax];
      }
      for (int base = 0; base < nt; base += LPE) {
        const bool valid = base + lane < nt;
        const int t = t0 + (valid ? base + lane : 0);
        const R* tp = c.Fg + c.foff_tax + t;
        const R w0 = valid ? nw0 : R(0), w1 = valid ? nw1 : R(0), w2 = valid ? nw2 : R(0);
        const R x0 = nx0, x1 = nx1, x2 = nx2;
        if (base + LPE + lane < nt) {
          const int tn = t0 + base + LPE + lane; const R* tq = c.Fg + c.foff_tax + tn;
          nw0 = wtac[3 * tn]; nw1 = wtac[3 * tn + 1]; nw2 = wtac[3 * tn + 2]; nx0 = tq[0]; nx1 = tq[c.ntax]; nx2 = tq[2 * c.ntax];
        }
        bool live = valid && (w0 != R(0) || w1 != R(0) || w2 != R(0));
        V3<R> xP, F; M3<R> Jx, Jv;
        if (live) {
          const V3<double> xPd = mulMv(ldm(c.PPd), mk3<double>((double)x0, (double)x1, (double)x2)) + ldv(c.PPd + 9);
          xP = cvt3<R>(xPd);
          live = contact_law<R, true>(prim, pf + TSIM_PF_SHAPE, sf, xP, vrel + cross3(wrel, xP), F, Jx, Jv, xPd);
        }
        if (!__any(live)) continue;
        any_live = true;
        if (live) {
          // weight in the sensor-link frame, then in the primitive frame:  s = wP . F
          const V3<R> wl = mk3<R>(w0 * tp[3 * c.ntax] + w1 * tp[6 * c.ntax] + w2 * tp[9 * c.ntax],
                                  w0 * tp[4 * c.ntax] + w1 * tp[7 * c.ntax] + w2 * tp[10 * c.ntax],
                                  w0 * tp[5 * c.ntax] + w1 * tp[8 * c.ntax] + w2 * tp[11 * c.ntax]);
          const V3<R> wP = mulMv(RPA, wl);
          const V3<R> gv = mulMtv(Jv, wP);
          const V3<R> gx = mulMtv(Jx, wP) + cross3(gv, wrel);       // d s / d(point displacement)
          const V3<R> ath = cross3(xP, gx) + cross3(wP, F);         // d s / d(relative rotation)
          const V3<R> bw = cross3(xP, gv);                           // d s / d(relative angular velocity)
          g[0] += ath.x; g[1] += ath.y; g[2] += ath.z; g[3] += gx.x; g[4] += gx.y; g[5] += gx.z;
          g[6] += bw.x; g[7] += bw.y; g[8] += bw.z; g[9] += gv.x; g[10] += gv.y; g[11] += gv.z;
        }
      }
      if (!any_live) continue;
#pragma unroll
      for (int e = 0; e < 12; ++e) g[e] = seg_sum<LPE>(g[e]);
      // lanes = directions
      pair_stage_tangent(c, pk, 0, lane, R(1), 0);
      if (lane < nr) {
        const R* T = c.PT + lane * PT_SIZE;
        R sq_ = R(0);
#pragma unroll
        for (int e = 0; e < 12; ++e) sq_ += g[e] * T[e];
        c.lamq[lane] += sq_;
      }
      pair_stage_tangent(c, pk, 0, lane, R(1), 1);
      if (lane < nr) {
        const R* T = c.PT + lane * PT_SIZE;
        R sv_ = R(0);
#pragma unroll
        for (int e = 6; e < 12; ++e) sv_ += g[e] * T[e];
        c.lamv[lane] += sv_;
      }
    }
  }
  TS_SYNC();
}

template <class R, int NRM, bool EXPJ, int LPE, bool POLICY = false>
__global__ void TS_KLB k_backward(BwdArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  constexpr int NS = TS_WAVE / LPE;
  const int slot = threadIdx.x / LPE, lane = threadIdx.x % LPE;
  const bool valid = (int)blockIdx.x * NS + slot < a.B;
  const int env = min((int)blockIdx.x * NS + slot, a.B - 1);
  Ctx<R> c; ctx_init(c, a.I, a.F, lds, NS, slot, lane, LPE, a.stage_cpt != 0, a.Fenv ? a.Fenv + (size_t)env * a.fstride : nullptr);
  const int nr = c.nr, nu = c.nu, REC = ts_rec(nr, nu, (int)sizeof(R));
  const int nvar3 = 3 * c.nvar, ntac3 = 3 * c.ntax;
  R* H2 = c.H2;    // taped Newton matrix of the sub-step
  init_world(c, lane, LPE);
  if (a.cyc && blockIdx.x == 0) c.stamps = a.cyc;
  if (lane < nr) { c.lamq[lane] = a.lamq[(size_t)env * nr + lane]; c.lamv[lane] = a.lamv[(size_t)env * nr + lane]; }
  // BDF2 models: taped sub-step t >= 2 is a BDF2 step (the first one after a reset is the BDF1 start-up, k_forward).  Its new state
  // depends on the TWO states before it, so next to the adjoint of the state one step back (lamq, lamv) the kernel carries what later
  // sub-steps already contributed to the state two steps back (lq1, lv1: one value per lane, in registers; second half of the buffers).
  const bool bdf2_model = ts_u(c.I[TSIM_IH_INTEGRATOR]) == 2;
  const size_t half = (size_t)a.B * nr;
  R lq1 = R(0), lv1 = R(0);
  if (bdf2_model && lane < nr) { lq1 = a.lamq[half + (size_t)env * nr + lane]; lv1 = a.lamv[half + (size_t)env * nr + lane]; }
  TS_SYNC();
  R du_frame = R(0);
  R pol_dq = R(0); bool pol_have = false;      // POLICY: what the NEXT frame's observation put on this frame's final state (q[0..2]; tactile: pol.dobs_tac)
  // The tape record of sub-step t (q1, qd1, u, H) and the state before it (q, qd of record t - 1) are fetched ONE ITERATION AHEAD
  // into registers: a lone wavefront cannot hide the ~2 x 1.5 k cycles of HBM latency of dependent loads at the top of every
  // sub-step, but the loads for the next sub-step fly during the whole of this one.  (Record t - 1 supplies q0, qd0 now and
  // q1, qd1 of the next iteration, so each iteration fetches u, H of record t - 1 and q, qd of record t - 2.)
  constexpr int NHL = (NRM * NRM + LPE - 1) / LPE;
  const int oqd = rec_qd<R>(nr), oH = rec_H<R>(nr), ou = rec_u<R>(nr);
  double pq1 = 0.0, pq0 = 0.0; R pqd1 = R(0), pqd0 = R(0), pqdm = R(0), pu = R(0), pH[NHL];     // pqdm: qd two records back (BDF2)
  {
    const R* r1 = a.tape + ((size_t)a.t_end * a.B + env) * REC;
    const R* r0 = a.tape + ((size_t)(a.t_end - 1) * a.B + env) * REC;
    if (lane < nr) { pq1 = rec_q(r1)[lane]; pqd1 = r1[oqd + lane]; pq0 = rec_q(r0)[lane]; pqd0 = r0[oqd + lane]; }
    if (bdf2_model && a.t_end >= 2 && lane < nr) pqdm = a.tape[((size_t)(a.t_end - 2) * a.B + env) * REC + oqd + lane];
    if (lane < nu) pu = r1[ou + lane];
#pragma unroll
    for (int i = 0; i < NHL; ++i) { const int e = lane + i * LPE; pH[i] = e < nr * nr ? r1[oH + e] : R(0); }
  }
  for (int j = a.n - 1; j >= 0; --j) {
    const int t = a.t_end - (a.n - 1 - j);
    const bool bdf2 = bdf2_model && t >= 2;
    c.cv = bdf2 ? R(1.5) / c.h : R(1) / c.h;
    c.ca = bdf2 ? R(2.25) / (c.h * c.h) : R(1) / (c.h * c.h);
    if (lane < nr) {
      c.qD[lane] = pq1; c.q[lane] = (R)pq1; c.q0[lane] = (R)pq0; c.qd0[lane] = pqd0;
      c.qd[lane] = pqd1;                              // taped velocity of the new state
      // discrete acceleration from the taped velocities, no position cancellation: BDF1 (qd1 - qd0) / h, BDF2 (3 qd1 - 4 qd0 + qd_1) / 2h
      c.qa[lane] = bdf2 ? (R(3) * pqd1 - R(4) * pqd0 + pqdm) / (R(2) * c.h) : (pqd1 - pqd0) / c.h;
    }
    if (lane < nu) c.u[lane] = pu;
#pragma unroll
    for (int i = 0; i < NHL; ++i) { const int e = lane + i * LPE; if (e < nr * nr) H2[e] = pH[i]; }
    TS_STAMP(c);
    if (j > 0) {                                      // next iteration: sub-step t - 1
      const R* r1 = a.tape + ((size_t)(t - 1) * a.B + env) * REC;
      const R* r0 = a.tape + ((size_t)(t - 2) * a.B + env) * REC;
      pq1 = pq0; pqd1 = pqd0;
      if (lane < nr) { pq0 = rec_q(r0)[lane]; pqd0 = r0[oqd + lane]; }
      if (bdf2_model && t >= 3 && lane < nr) pqdm = a.tape[((size_t)(t - 3) * a.B + env) * REC + oqd + lane];
      if (lane < nu) pu = r1[ou + lane];
#pragma unroll
      for (int i = 0; i < NHL; ++i) { const int e = lane + i * LPE; pH[i] = e < nr * nr ? r1[oH + e] : R(0); }
    }
    TS_SYNC();
    TS_STAMP(c);
    phase1<R, true, EXPJ>(c, lane, R(1), R(0), R(0));
    TS_STAMP(c);
    // direct partials of the loss w.r.t. this sub-step's outputs
    const bool seeded = (j + 1) % a.seed_stride == 0;
    if (seeded) {
      const int fr = j / a.seed_stride;
      const size_t so = a.frames ? (size_t)fr * a.B + env : (size_t)env * (a.n / a.seed_stride) + fr;
      const int tslot = (a.frames && a.tac_slot) ? a.tac_slot[fr] : 0;
      const size_t sot = (a.frames && a.tac_slot) ? (size_t)max(tslot, 0) * a.B + env : so;
      if (a.df_dq && lane < nr) c.lamq[lane] += a.df_dq[so * nr + lane];
      if (POLICY && pol_have && lane < nr) c.lamq[lane] += pol_dq;       // state part of the next frame's observation (goal; privilege: box pose)
      TS_SYNC();
      const R* wtac_ = (a.df_dtac && ntac3 && tslot >= 0) ? a.df_dtac + sot * ntac3 : nullptr;
      if (POLICY) wtac_ = (pol_have && a.pol.mode == TSIM_PUSH_OBS_TACTILE) ? a.pol.dobs_tac + ((size_t)(fr + 1) * a.B + env) * PP_NTAC : nullptr;   // tactile part (frame fr + 1's observation)
      output_vjp<LPE>(c, lane, (a.df_dvar && nvar3) ? a.df_dvar + so * nvar3 : nullptr, wtac_);
    }
    TS_STAMP(c);
    if (lane < nr) c.rhs[lane] = c.lamq[lane] + c.cv * c.lamv[lane];      // d qd1 / d q1 = cv
    TS_SYNC();
    solve_lanes<R, NRM, LPE>(H2, c.rhs, c.z, nr, true, lane);
    TS_STAMP(c);
    phase2<R, NRM, LPE>(c, lane, R(1));
    TS_STAMP(c);
    phase3<R, EXPJ, LPE>(c, lane, R(1), R(0));       // c.H = h^2 dr/dq
    TS_STAMP(c);
    const R ym = mass_times_z<LPE>(c, lane);
    TS_STAMP(c);
    if (lane < nr) {
      R yq = R(0);
      for (int i = 0; i < nr; ++i) yq += c.z[i] * c.H[i * nr + lane];
      if (!bdf2) {                                    // BDF1: new state from (q0, qd0) only
        c.lamq[lane] = c.lamq[lane] - yq + lq1;       // lq1, lv1: what a later BDF2 step put on this sub-step's (q0, qd0) as ITS (q_1, qd_1)
        c.lamv[lane] = c.h * ym + lv1;
        lq1 = R(0); lv1 = R(0);
      } else {
        // BDF2 in predictor form: with a_w = d qd1 / d p_w and dqp_w = d qpred / d p_w for p = (q0, qd0, q_1, qd_1),
        //   -(dg/dp_w)^T z + a_w lam_v = a_w (lam_v - R_v^T z / ca) + dqp_w M z ,   R_v^T z / ca = (rhs - K^T z - M z) / cv
        // (H = K + (cv R_v + ca M) / ca; rhs = H^T z).  a = (-2/h, 0, 1/2h, 0), dqp = (4/3, 8h/9, -1/3, -2h/9).
        const R d = c.lamv[lane] - (c.rhs[lane] - yq - ym) / c.cv;
        const R o0 = R(-2) / c.h * d + R(4.0 / 3) * ym, o1 = R(8.0 / 9) * c.h * ym;
        const R o2 = R(0.5) / c.h * d - R(1.0 / 3) * ym, o3 = R(-2.0 / 9) * c.h * ym;
        c.lamq[lane] = o0 + lq1; c.lamv[lane] = o1 + lv1;
        lq1 = o2; lv1 = o3;
      }
    }
    if (lane < nu) {
      const int* mi = ts_motor_rec(c, lane);
      const R* mf = c.F + c.foff_motor + lane * TSIM_MF_SIZE;
      R dtu;
      if (mi[TSIM_MI_CTRL] == 0) dtu = (c.u[lane] >= R(-1) && c.u[lane] <= R(1)) ? R(0.5) * (mf[TSIM_MF_HI] - mf[TSIM_MF_LO]) : R(0);
      else dtu = mf[TSIM_MF_P];
      const R du = c.z[mi[TSIM_MI_DOF]] * dtu / c.ca;         // -(dg/du)^T z, g = r / ca
      if (!a.frames) { if (valid) a.df_du[((size_t)env * a.n + j) * nu + lane] = du; }
      else {
        du_frame += du;
        if (j % a.seed_stride == 0 && valid && a.df_du) a.df_du[((size_t)(j / a.seed_stride) * a.B + env) * nu + lane] = du_frame;
      }
    }
    if (a.frames && j % a.seed_stride == 0) {              // a frame is undone
      if (POLICY) {
        // ... and so is the policy call in front of it: dL/d(action) -> MLP -> observation -> the state / tactile frame before it
        TS_SYNC();
        const int fr0 = j / a.seed_stride;
        pol_dq = push_policy_backward<LPE>(c, lane, valid, a.pol, (size_t)fr0 * a.B + env, env, du_frame, c.q0);
        pol_have = true;
        ts_own_stores_visible();                           // dobs_tac is read back by this slot as the previous frame's tactile seed
      }
      du_frame = R(0);
    }
    TS_SYNC();
  }
  if (lane < nr && valid) {
    a.lamq[(size_t)env * nr + lane] = c.lamq[lane]; a.lamv[(size_t)env * nr + lane] = c.lamv[lane];
    if (bdf2_model) { a.lamq[half + (size_t)env * nr + lane] = lq1; a.lamv[half + (size_t)env * nr + lane] = lv1; }
  }
}

// ================================================================================================ host side
static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// Every entry point runs on the batch's device and leaves the calling thread's current device as it found it (a process
// may drive several GPUs, and torch's current device / current_stream() follow the thread's HIP device).
struct DeviceGuard {
  int prev = -1; bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define TS_DEVICE(b) DeviceGuard guard_((b)->device); if (!guard_.ok) return fail("hipSetDevice(" + std::to_string((b)->device) + ") failed")

struct CacheEntry { void* buf; int len; int record; };
struct tsim_batch {
  int B, dtype, device, cap;
  std::vector<int32_t> I; std::vector<double> F;
  int nl, nr, nu, nvar, ntax, rec;
  int* dI; void* dF;             // model on device (dF in the batch's real type)
  void* dFenv; int nfrec;        // optional per-environment float tables [B][nfrec] (domain randomisation)
  void* tape;                    // [(cap+1)][B][rec]
  void *lamq, *lamv;             // carried adjoint [2][B][nr]: of the state the next adjoint sub-step starts from, and (BDF2) what later
                                 // sub-steps already contributed to the state one step further back
  int* evals;                    // residual evaluations of the last forward launch, per env
  float* gnorm = nullptr;        // largest ||g|| a sub-step of the last forward launch ended with, per env
  int cross_kinks = 0, eval_budget = 0;     // tsim_set_solver_options
  int* order; int order_valid;   // block -> env map for the next forward launch (LPT scheduling)
  int* order_ep = nullptr; int order_ep_n = 0;   // ... for the next EPISODE launch of order_ep_n sub-steps (from the previous one's totals; survives reset)
  void* prev; int has_prev;      // BDF2: state before the previous sub-step [B][2 nr]
  void* poseR; double* poseD; int nspt;   // tsim_readout: pose records [B][nspt] of the (sensor, primitive) combinations (k_readout -> k_taxels)
  int has_exp;                   // model contains a rotation-vector joint
  int t_cur, record;
  int lpe_forced;                // lanes per environment forced by TSIM_LPE (0 = choose from the batch size)
  int nsched;                    // ints of the sweep schedule appended to dI
  int stage_cpt;                 // the contact-point arrays are staged in LDS with the shared tables
  int n_simd;                    // SIMDs of the device (CUs x 4)
  int pose_valid = 0;            // the pose records are those of the current state (left by the last forward launch)
  int pose_off = 0;              // a launch of this batch was captured in a HIP graph: replays change the state behind the host's back, no reuse
  int tt_off = 0;                // offset of the taxel staging table in the device int blob (I[NI] + S[TS_SCHED_TAXTAB])
  int tax_slots = 0;             // blocks of k_taxels the device holds at once (occupancy x CUs), queried on first use
  size_t esz;
  std::vector<CacheEntry> cache;   // saved tapes, newest last
  std::vector<void*> pool;          // spare tape buffers
  long long* bwd_stamps = nullptr;  // diagnostics (tsim_debug_stamps)
};
// The state changed other than by a forward launch (or is about to, in a captured graph): tsim_readout recomputes the kinematics.
static void pose_invalidate(tsim_batch* b, hipStream_t st) {
  b->pose_valid = 0;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) b->pose_off = 1;
}


// sweep schedule of the link tree (layout: ts_sched in tsim_device.h)
static std::vector<int32_t> build_sched(const std::vector<int32_t>& I) {
  const int nl = I[TSIM_IH_NL], nr = I[TSIM_IH_NR], ol = I[TSIM_IH_OFF_LINK];
  std::vector<int> parent(nl + 1, 0), branch(nl + 1, -1), dof_link(nr, 0);
  std::vector<std::vector<int>> links;                       // links of each root branch, parents before children
  for (int i = 1; i <= nl; ++i) {
    const int32_t* li = &I[ol + (i - 1) * TSIM_LI_SIZE];
    parent[i] = li[TSIM_LI_PARENT];
    if (parent[i] == 0) { branch[i] = (int)links.size(); links.emplace_back(); }
    else branch[i] = branch[parent[i]];
    links[branch[i]].push_back(i);
    for (int k = li[TSIM_LI_DOF0]; k < li[TSIM_LI_DOF0] + li[TSIM_LI_NDOF]; ++k) dof_link[k] = i;
  }
  int nsteps = 0;
  for (auto& l : links) nsteps = std::max(nsteps, (int)l.size());
  std::vector<int> leader(links.size(), -1);                 // lowest dof lane of each branch
  for (int k = nr - 1; k >= 0; --k) leader[branch[dof_link[k]]] = k;
  const int npair = I[TSIM_IH_NPAIR], op = I[TSIM_IH_OFF_PAIR];
  const int nu = I[TSIM_IH_NU], om = I[TSIM_IH_OFF_MOTOR];
  std::vector<int32_t> S(TS_SCHED_ENT + nsteps * 16 + nl * TS_LR_SIZE + npair * TSIM_PI_SIZE + 16 + nu * TSIM_MI_SIZE, 0);
  S[0] = (int32_t)S.size(); S[1] = nsteps;
  for (int l = 0; l < 16; ++l) S[TS_SCHED_BRANCH + l] = l < nr ? branch[dof_link[l]] : -1;
  for (size_t b = 0; b < links.size() && b < 16; ++b) S[TS_SCHED_LEADER + b] = leader[b];
  S[TS_SCHED_NB] = (int32_t)links.size();
  for (int st = 0; st < nsteps; ++st)
    for (int l = 0; l < nr && l < 16; ++l) {
      const int br = branch[dof_link[l]];
      if (st < (int)links[br].size()) S[TS_SCHED_ENT + st * 16 + l] = links[br][st] | ((leader[br] == l) ? 0x100 : 0);
    }
  const int rec0 = TS_SCHED_ENT + nsteps * 16;
  for (int i = 1; i <= nl; ++i) {
    const int32_t* li = &I[ol + (i - 1) * TSIM_LI_SIZE];
    int32_t* r = &S[rec0 + (i - 1) * TS_LR_SIZE];
    r[TS_LR_PARENT] = li[TSIM_LI_PARENT]; r[TS_LR_JTYPE] = li[TSIM_LI_JTYPE]; r[TS_LR_DOF0] = li[TSIM_LI_DOF0];
    r[TS_LR_NDOF] = li[TSIM_LI_NDOF]; r[TS_LR_ANCMASK] = li[TSIM_LI_ANCMASK]; r[TS_LR_BRANCH] = branch[i];
  }
  for (int e = 0; e < npair * TSIM_PI_SIZE; ++e) S[rec0 + nl * TS_LR_SIZE + e] = I[op + e];      // pair int records
  {                                                                                              // taxel staging table (k_taxels), offset in S[TS_SCHED_TAXTAB]
    const int ns_ = I[TSIM_IH_NSENSOR], os = I[TSIM_IH_OFF_SENSOR], osp = I[TSIM_IH_OFF_SPRIM];
    S[TS_SCHED_TAXTAB] = (int32_t)S.size();
    int kb = 0;
    for (int s = 0; s < ns_; ++s) {
      const int32_t* si = &I[os + s * TSIM_SI_SIZE];
      S.push_back(si[TSIM_SI_TAX0] + si[TSIM_SI_NTAX]); S.push_back(kb); S.push_back(si[TSIM_SI_NSPRIM]);
      kb += si[TSIM_SI_NSPRIM];
    }
    for (int s = 0; s < ns_; ++s) {
      const int32_t* si = &I[os + s * TSIM_SI_SIZE];
      for (int j = 0; j < si[TSIM_SI_NSPRIM]; ++j) {
        const int pk = I[osp + si[TSIM_SI_SPRIM0] + j];
        S.push_back(I[op + pk * TSIM_PI_SIZE + TSIM_PI_PRIM]); S.push_back(pk);
      }
    }
    S[0] = (int32_t)S.size();
  }
  const int dm0 = rec0 + nl * TS_LR_SIZE + npair * TSIM_PI_SIZE;                                 // dof -> motor, motor int records
  for (int j = 0; j < 16; ++j) S[dm0 + j] = -1;
  for (int m = 0; m < nu; ++m) {
    const int j = I[om + m * TSIM_MI_SIZE + TSIM_MI_DOF];
    if (j >= 0 && j < 16) S[dm0 + j] = S[dm0 + j] == -1 ? m : -2;
    for (int e = 0; e < TSIM_MI_SIZE; ++e) S[dm0 + 16 + m * TSIM_MI_SIZE + e] = I[om + m * TSIM_MI_SIZE + e];
  }
  return S;
}

static int upload_model(tsim_batch* b, hipStream_t st) {
  HIPCHK(hipMemcpyAsync(b->dI, b->I.data(), b->I.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
  {
    std::vector<int32_t> S = build_sched(b->I);              // appended to the device copy at I[TSIM_IH_NI]
    if ((int)S.size() != b->nsched) return fail("sweep schedule size changed");
    b->tt_off = b->I[TSIM_IH_NI] + S[TS_SCHED_TAXTAB];
    HIPCHK(hipMemcpyAsync(b->dI + b->I.size(), S.data(), S.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
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

// Zero-fill by a plain kernel.  hipMemsetAsync nodes did NOT reliably take effect when a captured HIP graph was replayed
// (the carried adjoint kept the previous episode's final value at B >= 2048: policy gradients of 1e16 ... NaN in the graphed GD
// loop as soon as an episode's data changed; profiles/r02_graphed_rollout_fix.md) — a kernel node does.
__global__ void k_zero_words(uint32_t* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
static int zero_async(void* p, size_t bytes, hipStream_t st) {
  const size_t n = bytes / 4;                       // all buffers zeroed here hold 4- or 8-byte reals
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_zero_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint32_t*)p, n);
  HIPCHK(hipGetLastError());
  return 0;
}

// scatter [B][nr] q / qd into tape record 0
template <class R> __global__ void k_set_state(R* tape, const R* q, const R* qd, int B, int nr, int rec) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nr) return;
  int env = i / nr, k = i % nr;
  rec_q(tape + (size_t)env * rec)[k] = (double)q[i];
  tape[(size_t)env * rec + rec_qd<R>(nr) + k] = qd ? qd[i] : R(0);
}
// masked variant: only environments with mask[env] != 0 get the new state (record t of the tape)
template <class R> __global__ void k_set_state_masked(R* tape_rec, const R* q, const R* qd, const int32_t* mask, double* prev, R h, int B, int nr, int rec) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nr) return;
  int env = i / nr, k = i % nr;
  if (!mask[env]) return;
  rec_q(tape_rec + (size_t)env * rec)[k] = (double)q[i];
  tape_rec[(size_t)env * rec + rec_qd<R>(nr) + k] = qd ? qd[i] : R(0);
  if (prev) { const double v = qd ? (double)qd[i] : 0.0; prev[(size_t)env * 2 * nr + k] = (double)q[i] - (double)h * v; prev[(size_t)env * 2 * nr + nr + k] = v; }
}
template <class R> __global__ void k_get_state(const R* tape_rec, R* q, R* qd, int B, int nr, int rec) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nr) return;
  int env = i / nr, k = i % nr;
  if (q) q[i] = (R)rec_q(tape_rec + (size_t)env * rec)[k];
  if (qd) qd[i] = tape_rec[(size_t)env * rec + rec_qd<R>(nr) + k];
}

// dynamic LDS of a block of nslot environments
static size_t lds_bytes_for(const tsim_batch* b, int nslot) {
  const int reals = ts_lds_reals(b->nl, b->nr, b->nu, b->nfrec, b->I[TSIM_IH_NCPT], b->stage_cpt != 0, nslot, b->dFenv != nullptr, b->nsched + (int)b->I.size(), (int)b->esz);
  return ((size_t)reals * b->esz + 15) / 16 * 16;
}
// Launch shape of the forward / backward kernels.
// Lanes per environment (LPE): one environment per wavefront uses <= nr of the 64 lanes in most phases; packing 2 or 4
// environments into a wavefront divides the instruction count per environment.  These kernels hold one wavefront per
// SIMD (~270 registers; holding them to 256 for two per SIMD spills and is slower at every batch size measured,
// profiles/r01_launch_shape_ab.txt) and a lone wavefront is latency-bound, so the time of a launch is
//     rounds x latency(LPE),   rounds = ceil(wavefronts / #SIMDs),   latency(64 : 32 : 16) ~ 1 : 0.93 : 1.02
// (phase stamps, tools/phase_cycles.py).  The launch takes the LPE that minimises it, subject to the block's LDS
// leaving room for four blocks per CU.
struct LaunchShape { int lpe; unsigned grid; size_t lds; };
static LaunchShape launch_shape(const tsim_batch* b) {
  LaunchShape L;
  int lpe = b->lpe_forced;
  const size_t lds_cap = b->lpe_forced ? 64 * 1024 : 40 * 1024;
  if (!lpe) {
    lpe = TS_WAVE;
    double best = 1e30;
    const int cand[3] = {64, 32, 16};
    const double lat[3] = {1.0, 0.93, 1.02};
    for (int i = 0; i < 3; ++i) {
      const int ns = TS_WAVE / cand[i];
      if (i > 0 && lds_bytes_for(b, ns) > lds_cap) break;
      const long long waves = ((long long)b->B + ns - 1) / ns;
      const double t = (double)((waves + b->n_simd - 1) / b->n_simd) * lat[i];
      if (t < best) { best = t; lpe = cand[i]; }
    }
  }
  if (b->has_exp) lpe = TS_WAVE;
  while (lpe < TS_WAVE && lds_bytes_for(b, TS_WAVE / lpe) > lds_cap) lpe *= 2;
  L.lpe = lpe;
  const int ns = TS_WAVE / lpe;
  L.grid = (unsigned)((b->B + ns - 1) / ns);
  L.lds = lds_bytes_for(b, ns);
  return L;
}
// Stage the contact-point arrays in LDS (one copy per block) if they are small and the launch keeps its lanes-per-environment shape with
// them; re-decided whenever what the shape depends on changes (forced lanes, per-environment tables on / off).
static void decide_stage_cpt(tsim_batch* b) {
  b->stage_cpt = 0;
  if ((size_t)3 * b->I[TSIM_IH_NCPT] * b->esz > TS_CPT_LDS_BYTES) return;
  if (b->dFenv && getenv("TSIM_NO_ENVTAB_CPT")) return;      // A/B: the round-3 behaviour (contact points from global memory next to per-environment tables)
  const int lpe0 = launch_shape(b).lpe;
  b->stage_cpt = 1;
  if (launch_shape(b).lpe != lpe0 || lds_bytes_for(b, 1) > 64 * 1024) b->stage_cpt = 0;
}
// kernel variants: NRM = 8 / 16 rows in the register solve; EXPJ = model has a rotation-vector joint (its code is
// compiled out otherwise: it costs registers in every evaluation); LPE as above
#define TS_LAUNCH_L(KERNEL, R, NRM, L, st, a) do {                                                                       \
    if (L.lpe == 64) hipLaunchKernelGGL((KERNEL<R, NRM, false, 64>), dim3(L.grid), dim3(TS_WAVE), L.lds, st, a);         \
    else if (L.lpe == 32) hipLaunchKernelGGL((KERNEL<R, NRM, false, 32>), dim3(L.grid), dim3(TS_WAVE), L.lds, st, a);    \
    else hipLaunchKernelGGL((KERNEL<R, NRM, false, 16>), dim3(L.grid), dim3(TS_WAVE), L.lds, st, a);                     \
  } while (0)
#define TS_LAUNCH(KERNEL, R, b, st, a) do {                                                                              \
    const LaunchShape L = launch_shape(b);                                                                               \
    if (b->has_exp) hipLaunchKernelGGL((KERNEL<R, 16, true, 64>), dim3(L.grid), dim3(TS_WAVE), L.lds, st, a);            \
    else if (b->nr <= 8) TS_LAUNCH_L(KERNEL, R, 8, L, st, a);                                                            \
    else TS_LAUNCH_L(KERNEL, R, 16, L, st, a);                                                                           \
  } while (0)

// Pads too large for the in-kernel read-out (lanes of one environment over its taxels) are read on demand by tsim_readout; for those
// the forward launch leaves the pose records of its final state (k_forward, end of the launch).  Not under stream capture: a graph
// replay changes the state without the host seeing it, so a batch that was ever captured always recomputes the kinematics.
enum { TS_POSE_EMIT_MIN_TAXELS = 4096 };
static bool pose_emit(tsim_batch* b, hipStream_t st) {
  pose_invalidate(b, st);
  return !b->pose_off && b->nspt > 0 && b->poseR && b->ntax >= TS_POSE_EMIT_MIN_TAXELS;
}

template <class R>
static int launch_forward(tsim_batch* b, const void* u, int nframes, const int32_t* tac_slot, int nsub, void* q_out, void* qd_out, void* var_out, void* tac_out, int32_t* status, hipStream_t st) {
  FwdArgs<R> a;
  a.I = b->dI; a.F = (const R*)b->dF; a.Fenv = (const R*)b->dFenv; a.fstride = b->nfrec; a.B = b->B; a.nsub = nsub; a.record = b->record; a.t0 = b->t_cur; a.nframes = nframes; a.tac_slot = tac_slot;
  a.tape = (R*)b->tape; a.u = (const R*)u;
  a.q_out = (R*)q_out; a.qd_out = (R*)qd_out; a.var_out = (R*)var_out; a.tac_out = (R*)tac_out; a.status = status; a.evals = b->evals; a.order = (b->B >= 256 && nframes == 1 && b->order_valid) ? b->order : (b->B >= 256 && nframes > 1 && b->order_ep_n == nframes * nsub && !getenv("TSIM_NO_EPISODE_LPT")) ? b->order_ep : nullptr;
  a.prev = (double*)b->prev; a.has_prev = b->has_prev; a.stage_cpt = b->stage_cpt;
  a.cross_kinks = b->cross_kinks; a.eval_budget = b->eval_budget; a.gnorm = b->gnorm;
  const bool emit = pose_emit(b, st);
  if (emit) { a.poseR = (R*)b->poseR; a.poseD = b->poseD; a.nspt = b->nspt; }
  TS_LAUNCH(k_forward, R, b, st, a);
  HIPCHK(hipGetLastError());
  b->pose_valid = emit ? 1 : 0;
  if (b->B >= 256) {
    const int ns = TS_WAVE / launch_shape(b).lpe, nsv = (b->B % ns == 0) ? ns : 1;
    if (nframes > 1) {          // episode totals: the order of the next episode launch of this length (kept across resets)
      const int n = nframes * nsub;
      hipLaunchKernelGGL(k_order_by_evals, dim3(1), dim3(1024), 0, st, (const int*)b->evals, b->order_ep, b->B, nsv, 2 * n, std::max(1, 5 * n / 2));
      b->order_ep_n = n; b->order_valid = 0;
    } else {
      hipLaunchKernelGGL(k_order_by_evals, dim3(1), dim3(1024), 0, st, (const int*)b->evals, b->order, b->B, nsv, 0, 64);
      b->order_valid = 1;
    }
    HIPCHK(hipGetLastError());
  } else if (nframes > 1) b->order_valid = 0;
  return 0;
}

template <class R>
static int launch_backward(tsim_batch* b, int n, int seed_stride, int frames, const int32_t* tac_slot, const void* df_dq, const void* df_dvar, const void* df_dtac, void* df_du, hipStream_t st) {
  BwdArgs<R> a;
  a.I = b->dI; a.F = (const R*)b->dF; a.Fenv = (const R*)b->dFenv; a.fstride = b->nfrec; a.B = b->B; a.n = n; a.t_end = b->t_cur; a.seed_stride = seed_stride; a.frames = frames; a.tac_slot = tac_slot;
  a.tape = (const R*)b->tape; a.df_dq = (const R*)df_dq; a.df_dvar = (const R*)df_dvar; a.df_dtac = (const R*)df_dtac;
  a.lamq = (R*)b->lamq; a.lamv = (R*)b->lamv; a.df_du = (R*)df_du; a.stage_cpt = b->stage_cpt; a.cyc = b->bwd_stamps;
  TS_LAUNCH(k_backward, R, b, st, a);
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
  if (I[TSIM_IH_INTEGRATOR] != 1 && I[TSIM_IH_INTEGRATOR] != 2) return fail("unknown integrator");
  const int nl = I[TSIM_IH_NL], nr = I[TSIM_IH_NR], nu = I[TSIM_IH_NU];
  int n_exp = 0;
  if (nr > 16 || nr < 1 || nu > 16) return fail("ndof_r must be in 1..16 and ndof_u <= 16");
  for (int i = 1; i <= nl; ++i) {
    int jt = I[I[TSIM_IH_OFF_LINK] + (i - 1) * TSIM_LI_SIZE + TSIM_LI_JTYPE];
    if (jt != TSIM_J_REVOLUTE && jt != TSIM_J_PRISMATIC && jt != TSIM_J_PLANAR && jt != TSIM_J_TRANSLATIONAL && jt != TSIM_J_SPHERICAL_EXP)
      return fail("joint type not supported by the HIP path");
    if (jt == TSIM_J_SPHERICAL_EXP && ++n_exp > 1) return fail("at most one rotation-vector joint per model on the HIP path");
  }
  for (int s = 0; s < I[TSIM_IH_NSENSOR]; ++s)
    if (I[I[TSIM_IH_OFF_SENSOR] + s * TSIM_SI_SIZE + TSIM_SI_NSPRIM] > 16) return fail("too many primitives per sensor");
  DeviceGuard guard_(device);
  if (!guard_.ok) return fail("hipSetDevice(" + std::to_string(device) + ") failed");
  tsim_batch* b = new tsim_batch();
  b->B = B; b->dtype = dtype; b->device = device; b->cap = tape_capacity;
  b->I.assign(I, I + I[TSIM_IH_NI]); b->F.assign(F, F + I[TSIM_IH_NF]);
  b->nl = nl; b->nr = nr; b->nu = nu; b->nvar = I[TSIM_IH_NVAR]; b->ntax = I[TSIM_IH_NTAXEL];
  b->esz = dtype == TSIM_F32 ? 4 : 8;
  b->rec = ts_rec(nr, nu, (int)b->esz);
  b->t_cur = 0; b->record = 0; b->has_exp = n_exp > 0;
  b->dFenv = nullptr; b->nfrec = I[TSIM_IH_FOFF_CPT];
  b->lpe_forced = 0;
  { const std::vector<int32_t> S_ = build_sched(b->I); b->nsched = (int)S_.size(); b->tt_off = b->I[TSIM_IH_NI] + S_[TS_SCHED_TAXTAB]; }
  if (const char* e = getenv("TSIM_LPE")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) b->lpe_forced = v; }
  b->cross_kinks = dtype == TSIM_F32 ? 1 : 0;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    b->n_simd = 4 * cus;
  }
  b->stage_cpt = 0;
  if (lds_bytes_for(b, 1) > 64 * 1024) { delete b; return fail("model needs more than 64 KiB of LDS per environment"); }
  decide_stage_cpt(b);
  b->dI = nullptr; b->dF = nullptr; b->tape = nullptr; b->lamq = nullptr; b->lamv = nullptr; b->evals = nullptr; b->order = nullptr; b->order_valid = 0; b->prev = nullptr; b->has_prev = 0; b->poseR = nullptr; b->poseD = nullptr; b->nspt = b->I[TSIM_IH_NSPRIM];
  size_t tape_bytes = (size_t)(tape_capacity + 1) * B * b->rec * b->esz;
  if (hipMalloc(&b->dI, (b->I.size() + b->nsched) * sizeof(int32_t)) != hipSuccess || hipMalloc(&b->dF, b->F.size() * b->esz) != hipSuccess ||
      hipMalloc(&b->tape, tape_bytes) != hipSuccess || hipMalloc(&b->lamq, (size_t)2 * B * nr * b->esz) != hipSuccess ||
      hipMalloc(&b->lamv, (size_t)2 * B * nr * b->esz) != hipSuccess || hipMalloc(&b->evals, (size_t)B * sizeof(int)) != hipSuccess || hipMalloc((void**)&b->gnorm, (size_t)B * sizeof(float)) != hipSuccess || hipMalloc(&b->order, (size_t)B * sizeof(int)) != hipSuccess || hipMalloc((void**)&b->order_ep, (size_t)B * sizeof(int)) != hipSuccess || hipMalloc(&b->prev, (size_t)B * 2 * nr * sizeof(double)) != hipSuccess ||
      (b->nspt > 0 && (hipMalloc(&b->poseR, (size_t)B * b->nspt * TP_R_SIZE * b->esz) != hipSuccess || hipMalloc((void**)&b->poseD, (size_t)B * b->nspt * TP_D_SIZE * sizeof(double)) != hipSuccess))) {
    tsim_batch_destroy(b);
    return fail("hipMalloc failed (tape bytes = " + std::to_string(tape_bytes) + ")");
  }
  if (hipMemset(b->tape, 0, tape_bytes) != hipSuccess || hipMemset(b->lamq, 0, (size_t)2 * B * nr * b->esz) != hipSuccess ||
      hipMemset(b->lamv, 0, (size_t)2 * B * nr * b->esz) != hipSuccess) { tsim_batch_destroy(b); return fail("hipMemset failed"); }
  if (upload_model(b, nullptr)) { tsim_batch_destroy(b); return 1; }
  *out = b;
  return 0;
}

void tsim_batch_destroy(tsim_batch* b) {
  if (!b) return;
  DeviceGuard guard_(b->device);
  for (auto& e : b->cache) (void)hipFree(e.buf);
  for (void* p : b->pool) (void)hipFree(p);
  (void)hipFree(b->dFenv); (void)hipFree(b->dI); (void)hipFree(b->dF); (void)hipFree(b->tape); (void)hipFree(b->lamq); (void)hipFree(b->lamv); (void)hipFree(b->evals); (void)hipFree(b->gnorm); (void)hipFree(b->order); (void)hipFree(b->order_ep); (void)hipFree(b->prev); (void)hipFree(b->poseR); (void)hipFree(b->poseD);
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
int tsim_launch_info(const tsim_batch* b, int32_t* out) {
  const LaunchShape L = launch_shape(b);
  out[0] = (int32_t)L.lds; out[1] = TS_WAVE; out[2] = (int32_t)L.grid; out[3] = L.lpe;
  return 0;
}
int tsim_set_lanes_per_env(tsim_batch* b, int lanes) {
  if (lanes != 0 && lanes != 16 && lanes != 32 && lanes != 64) return fail("set_lanes_per_env: 0 (automatic), 16, 32 or 64");
  b->lpe_forced = lanes;
  b->order_valid = 0; b->order_ep_n = 0;
  decide_stage_cpt(b);      // depends on the shape; the flag travels with every launch as a kernel argument: nothing on the device to update
  return 0;
}
int tsim_set_solver_options(tsim_batch* b, int cross_kinks, int eval_budget) {
  if (eval_budget < 0) return fail("set_solver_options: negative evaluation budget");
  b->cross_kinks = cross_kinks != 0;
  b->eval_budget = eval_budget;
  return 0;
}
int tsim_last_gnorm(tsim_batch* b, float* host_out) {
  TS_DEVICE(b);
  if (hipMemcpy(host_out, b->gnorm, (size_t)b->B * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return fail("last_gnorm: copy failed");
  return 0;
}
int tsim_last_evals(tsim_batch* b, int32_t* host_out) {
  TS_DEVICE(b);
  if (hipMemcpy(host_out, b->evals, (size_t)b->B * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail("last_evals: copy failed");
  return 0;
}

int tsim_update_model(tsim_batch* b, const int32_t* I, const double* F, void* stream) {
  if (I[TSIM_IH_NI] != (int)b->I.size() || I[TSIM_IH_NF] != (int)b->F.size()) return fail("update_model: blob size changed");
  for (int i = 0; i < TSIM_IH_SIZE; ++i) if (I[i] != b->I[i]) return fail("update_model: topology changed");
  TS_DEVICE(b);
  pose_invalidate(b, (hipStream_t)stream);
  b->I.assign(I, I + I[TSIM_IH_NI]); b->F.assign(F, F + I[TSIM_IH_NF]);
  if (b->dFenv) { HIPCHK(hipFree(b->dFenv)); b->dFenv = nullptr; decide_stage_cpt(b); }     // per-environment tables refer to the old model
  return upload_model(b, (hipStream_t)stream);
}

int tsim_set_env_tables(tsim_batch* b, const void* tables, void* stream) {
  TS_DEVICE(b);
  pose_invalidate(b, (hipStream_t)stream);
  if (!tables) { if (b->dFenv) { HIPCHK(hipFree(b->dFenv)); b->dFenv = nullptr; decide_stage_cpt(b); } return 0; }
  size_t bytes = (size_t)b->B * b->nfrec * b->esz;
  if (!b->dFenv) { HIPCHK(hipMalloc(&b->dFenv, bytes)); decide_stage_cpt(b); }      // the block's LDS layout changes with per-environment tables
  HIPCHK(hipMemcpyAsync(b->dFenv, tables, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
int tsim_table_size(const tsim_batch* b) { return b->nfrec; }

int tsim_reset(tsim_batch* b, const void* q0, const void* qd0, int backward_flag, void* stream) {
  if (!q0) return fail("reset: q0 is null");
  TS_DEVICE(b);
  hipStream_t st = (hipStream_t)stream;
  int n = b->B * b->nr, blk = 256, grd = (n + blk - 1) / blk;
  if (b->dtype == TSIM_F32) hipLaunchKernelGGL(k_set_state<float>, dim3(grd), dim3(blk), 0, st, (float*)b->tape, (const float*)q0, (const float*)qd0, b->B, b->nr, b->rec);
  else hipLaunchKernelGGL(k_set_state<double>, dim3(grd), dim3(blk), 0, st, (double*)b->tape, (const double*)q0, (const double*)qd0, b->B, b->nr, b->rec);
  HIPCHK(hipGetLastError());
  if (zero_async(b->lamq, (size_t)2 * b->B * b->nr * b->esz, st) || zero_async(b->lamv, (size_t)2 * b->B * b->nr * b->esz, st)) return 1;
  b->t_cur = 0; b->record = backward_flag ? 1 : 0; b->order_valid = 0; b->has_prev = 0;
  pose_invalidate(b, st);
  return 0;
}

int tsim_reset_masked(tsim_batch* b, const void* q0, const void* qd0, const int32_t* mask, void* stream) {
  if (!q0 || !mask) return fail("reset_masked: q0 / mask is null");
  if (b->record) return fail("reset_masked: not while recording (the tape is shared by the batch): use reset");
  TS_DEVICE(b);
  hipStream_t st = (hipStream_t)stream;
  pose_invalidate(b, st);
  int n = b->B * b->nr, blk = 256, grd = (n + blk - 1) / blk;
  size_t off = (size_t)b->t_cur * b->B * b->rec;
  // BDF2 models: the environment restarts with a constant-velocity history (q_-1 = q0 - h qd0, qd_-1 = qd0) [CHOICE]
  void* prev = (b->I[TSIM_IH_INTEGRATOR] == 2 && b->has_prev) ? b->prev : nullptr;
  if (b->dtype == TSIM_F32) hipLaunchKernelGGL(k_set_state_masked<float>, dim3(grd), dim3(blk), 0, st, (float*)b->tape + off, (const float*)q0, (const float*)qd0, mask, (double*)prev, (float)b->F[TSIM_FH_H], b->B, b->nr, b->rec);
  else hipLaunchKernelGGL(k_set_state_masked<double>, dim3(grd), dim3(blk), 0, st, (double*)b->tape + off, (const double*)q0, (const double*)qd0, mask, (double*)prev, b->F[TSIM_FH_H], b->B, b->nr, b->rec);
  HIPCHK(hipGetLastError());
  return 0;
}

int tsim_step(tsim_batch* b, const void* u, int num_steps, void* q_out, void* qd_out, void* var_out, void* tac_out, int32_t* status, void* stream) {
  if (num_steps <= 0) return fail("step: num_steps must be positive");
  if (!u && b->nu > 0) return fail("step: u is null");
  if (b->record && b->t_cur + num_steps > b->cap) return fail("step: tape capacity exceeded (" + std::to_string(b->cap) + " sub-steps)");
  TS_DEVICE(b);
  int rc = b->dtype == TSIM_F32 ? launch_forward<float>(b, u, 1, nullptr, num_steps, q_out, qd_out, var_out, tac_out, status, (hipStream_t)stream)
                                : launch_forward<double>(b, u, 1, nullptr, num_steps, q_out, qd_out, var_out, tac_out, status, (hipStream_t)stream);
  if (rc) return rc;
  if (b->record) b->t_cur += num_steps;
  b->has_prev = 1;
  return 0;
}

int tsim_get_state(tsim_batch* b, void* q_out, void* qd_out, void* stream) {
  TS_DEVICE(b);
  int n = b->B * b->nr, blk = 256, grd = (n + blk - 1) / blk;
  size_t off = (size_t)b->t_cur * b->B * b->rec;
  if (b->dtype == TSIM_F32) hipLaunchKernelGGL(k_get_state<float>, dim3(grd), dim3(blk), 0, (hipStream_t)stream, (const float*)b->tape + off, (float*)q_out, (float*)qd_out, b->B, b->nr, b->rec);
  else hipLaunchKernelGGL(k_get_state<double>, dim3(grd), dim3(blk), 0, (hipStream_t)stream, (const double*)b->tape + off, (double*)q_out, (double*)qd_out, b->B, b->nr, b->rec);
  HIPCHK(hipGetLastError());
  return 0;
}

int tsim_readout(tsim_batch* b, void* var_out, void* tac_out, void* stream) {
  TS_DEVICE(b);
  hipStream_t st = (hipStream_t)stream;
  const bool tac = tac_out && b->ntax > 0 && b->nspt > 0;
  // taxels that are paired with no primitive (a sensor body without a general_primitive_contact) read zero, as in k_forward's read-out
  if (tac_out && b->ntax > 0 && b->nspt == 0 && zero_async(tac_out, (size_t)b->B * 3 * b->ntax * b->esz, st)) return 1;
  if (tac && (b->nspt > TX_MAXK || b->I[TSIM_IH_NSENSOR] > TX_MAXS)) return fail("readout: more than " + std::to_string((int)TX_MAXK) + " (sensor, primitive) combinations or " + std::to_string((int)TX_MAXS) + " sensors (k_taxels staging)");
  // Taxels per block of k_taxels.  A block's prologue (staging the pose records and the model tables of its environment, two barriers)
  // is a chain of dependent global loads, ~3 us whatever the slice; with many short blocks per SIMD slot the kernel WAS that prologue
  // (8192 blocks of 5 taxels per thread: 4.6 rounds of ~7 us for 256 x 40 000 taxels).  So: ONE round — as many blocks as the device
  // holds at once (occupancy x CUs), each with a slice long enough to cover the batch; never less than 256 taxels per block, so the
  // single environment of test_sim_speed.py still spreads its 40 000 taxels over 157 blocks.
  int slice;
  {
    if (b->tax_slots == 0) {
      int per_cu = 0;
      const hipError_t e_ = b->dtype == TSIM_F32 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_taxels<float>, 256, 0)
                                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_taxels<double>, 256, 0);
      b->tax_slots = (e_ == hipSuccess && per_cu > 0 ? per_cu : 4) * (b->n_simd / 4);
    }
    const int per_env = std::max(1, std::min(b->tax_slots / b->B, (b->ntax + 255) / 256));
    slice = ((b->ntax + per_env - 1) / per_env + 255) / 256 * 256;
  }
  const dim3 tgrid(b->B, tac ? (b->ntax + slice - 1) / slice : 1);
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { b->pose_off = 1; b->pose_valid = 0; }
  }
  const bool fk = var_out || !tac || !b->pose_valid;      // the forward launch left the pose records of this state: k_taxels alone
  if (b->dtype == TSIM_F32) {
    ReadArgs<float> a{b->dI, (const float*)b->dF, (const float*)b->dFenv, b->nfrec, b->B, b->t_cur, (const float*)b->tape, (float*)var_out, tac ? (float*)b->poseR : nullptr, b->poseD, b->nspt, b->stage_cpt};
    if (fk) hipLaunchKernelGGL(k_readout<float>, dim3(b->B), dim3(TS_WAVE), lds_bytes_for(b, 1), st, a);
    if (tac) {
      TaxArgs<float> t{b->dI, (const float*)b->dF, (const float*)b->dFenv, b->nfrec, (const float*)b->poseR, b->poseD, b->nspt, (float*)tac_out, slice,
                       b->ntax, b->I[TSIM_IH_NSENSOR], b->I[TSIM_IH_FOFF_SENSOR], b->I[TSIM_IH_FOFF_PAIR], b->I[TSIM_IH_FOFF_TAXEL], b->tt_off};
      hipLaunchKernelGGL(k_taxels<float>, tgrid, dim3(256), 0, st, t);
    }
  } else {
    ReadArgs<double> a{b->dI, (const double*)b->dF, (const double*)b->dFenv, b->nfrec, b->B, b->t_cur, (const double*)b->tape, (double*)var_out, tac ? (double*)b->poseR : nullptr, b->poseD, b->nspt, b->stage_cpt};
    if (fk) hipLaunchKernelGGL(k_readout<double>, dim3(b->B), dim3(TS_WAVE), lds_bytes_for(b, 1), st, a);
    if (tac) {
      TaxArgs<double> t{b->dI, (const double*)b->dF, (const double*)b->dFenv, b->nfrec, (const double*)b->poseR, b->poseD, b->nspt, (double*)tac_out, slice,
                        b->ntax, b->I[TSIM_IH_NSENSOR], b->I[TSIM_IH_FOFF_SENSOR], b->I[TSIM_IH_FOFF_PAIR], b->I[TSIM_IH_FOFF_TAXEL], b->tt_off};
      hipLaunchKernelGGL(k_taxels<double>, tgrid, dim3(256), 0, st, t);
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int tsim_backward_steps(tsim_batch* b, int n, int seed_mode, const void* df_dq, const void* df_dvar, const void* df_dtac, void* df_du, void* stream) {
  if (!b->record) return fail("backward_steps: reset(backward_flag=True) was not called");
  if (n <= 0 || n > b->t_cur) return fail("backward_steps: only " + std::to_string(b->t_cur) + " sub-steps on the tape");
  if (!df_du && b->nu > 0) return fail("backward_steps: df_du is null");
  if (seed_mode != 0 && seed_mode != 1) return fail("backward_steps: bad seed_mode");
  TS_DEVICE(b);
  const int stride = seed_mode == 1 ? 1 : n;
  int rc = b->dtype == TSIM_F32 ? launch_backward<float>(b, n, stride, 0, nullptr, df_dq, df_dvar, df_dtac, df_du, (hipStream_t)stream)
                                : launch_backward<double>(b, n, stride, 0, nullptr, df_dq, df_dvar, df_dtac, df_du, (hipStream_t)stream);
  if (rc) return rc;
  b->t_cur -= n;
  pose_invalidate(b, (hipStream_t)stream);      // the pose records a forward launch left are those of the state before the roll-back
  return 0;
}

int tsim_rollout(tsim_batch* b, const void* u, int num_frames, int num_steps, const int32_t* tactile_slot, void* q_out, void* qd_out, void* var_out, void* tac_out, int32_t* status, void* stream) {
  if (num_frames <= 0 || num_steps <= 0) return fail("rollout: num_frames and num_steps must be positive");
  if (!u && b->nu > 0) return fail("rollout: u is null");
  if (b->record && b->t_cur + (long long)num_frames * num_steps > b->cap) return fail("rollout: tape capacity exceeded (" + std::to_string(b->cap) + " sub-steps)");
  TS_DEVICE(b);
  int rc = b->dtype == TSIM_F32 ? launch_forward<float>(b, u, num_frames, tactile_slot, num_steps, q_out, qd_out, var_out, tac_out, status, (hipStream_t)stream)
                                : launch_forward<double>(b, u, num_frames, tactile_slot, num_steps, q_out, qd_out, var_out, tac_out, status, (hipStream_t)stream);
  if (rc) return rc;
  if (b->record) b->t_cur += num_frames * num_steps;
  b->has_prev = 1;
  return 0;
}

int tsim_backward_episode(tsim_batch* b, int num_frames, int num_steps, const int32_t* tactile_slot, const void* df_dq, const void* df_dvar, const void* df_dtac, void* df_du, void* stream) {
  if (!b->record) return fail("backward_episode: reset(backward_flag=True) was not called");
  if (num_frames <= 0 || num_steps <= 0) return fail("backward_episode: num_frames and num_steps must be positive");
  const long long n = (long long)num_frames * num_steps;
  if (n > b->t_cur) return fail("backward_episode: only " + std::to_string(b->t_cur) + " sub-steps on the tape");
  if (!df_du && b->nu > 0) return fail("backward_episode: df_du is null");
  TS_DEVICE(b);
  int rc = b->dtype == TSIM_F32 ? launch_backward<float>(b, (int)n, num_steps, 1, tactile_slot, df_dq, df_dvar, df_dtac, df_du, (hipStream_t)stream)
                                : launch_backward<double>(b, (int)n, num_steps, 1, tactile_slot, df_dq, df_dvar, df_dtac, df_du, (hipStream_t)stream);
  if (rc) return rc;
  b->t_cur -= (int)n;
  pose_invalidate(b, (hipStream_t)stream);
  return 0;
}

int tsim_get_adjoint(tsim_batch* b, void* df_dq0, void* df_dqd0, void* stream) {
  TS_DEVICE(b);
  size_t bytes = (size_t)b->B * b->nr * b->esz;
  if (df_dq0) HIPCHK(hipMemcpyAsync(df_dq0, b->lamq, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (df_dqd0) HIPCHK(hipMemcpyAsync(df_dqd0, b->lamv, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

// Backward cache = LIFO of tape BUFFERS swapped by pointer (no tape copy): save parks the live tape on the stack and takes a
// spare buffer from the pool (allocated on first use, or ahead of time by tsim_cache_reserve), pop swaps back.  After
// warm-up neither allocates nor synchronises, so both can sit inside a captured HIP graph region's host code path.
static size_t tape_bytes_of(const tsim_batch* b) { return (size_t)(b->cap + 1) * b->B * b->rec * b->esz; }
int tsim_cache_reserve(tsim_batch* b, int depth) {
  TS_DEVICE(b);
  while ((int)(b->pool.size() + b->cache.size()) < depth) {
    void* p = nullptr;
    if (hipMalloc(&p, tape_bytes_of(b)) != hipSuccess) return fail("cache_reserve: hipMalloc of a " + std::to_string(tape_bytes_of(b)) + " byte tape failed");
    b->pool.push_back(p);
  }
  return 0;
}
int tsim_cache_save(tsim_batch* b, void* stream) {
  TS_DEVICE(b);
  if (b->pool.empty()) { int rc = tsim_cache_reserve(b, (int)b->cache.size() + 1); if (rc) return rc; }
  void* spare = b->pool.back(); b->pool.pop_back();
  // the simulation goes on from its current state: carry the newest record (q, qd of all environments) over
  const size_t rec_bytes = (size_t)b->B * b->rec * b->esz, off = (size_t)b->t_cur * rec_bytes;
  HIPCHK(hipMemcpyAsync((char*)spare + off, (char*)b->tape + off, rec_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  // ... and, for BDF2 models, the record before it: while recording, k_forward takes the state before the previous sub-step from tape
  // record t0 - 1, so a forward launch that continues on the spare buffer must find it there
  if (b->I[TSIM_IH_INTEGRATOR] == 2 && b->t_cur >= 1)
    HIPCHK(hipMemcpyAsync((char*)spare + off - rec_bytes, (char*)b->tape + off - rec_bytes, rec_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  CacheEntry e; e.buf = b->tape; e.len = b->t_cur; e.record = b->record;
  b->cache.push_back(e);
  b->tape = spare;
  return 0;
}
int tsim_cache_pop(tsim_batch* b, void* stream) {
  if (b->cache.empty()) return fail("popBackwardCache: cache is empty");
  TS_DEVICE(b);
  CacheEntry e = b->cache.back(); b->cache.pop_back();
  b->pool.push_back(b->tape);                 // stream order keeps earlier kernels on the old buffer safe: it is only
  b->tape = e.buf;                            // handed out again by a later save on the same stream
  b->t_cur = e.len; b->record = e.record; b->has_prev = 0; b->order_valid = 0;
  pose_invalidate(b, (hipStream_t)stream);
  if (zero_async(b->lamq, (size_t)2 * b->B * b->nr * b->esz, (hipStream_t)stream) || zero_async(b->lamv, (size_t)2 * b->B * b->nr * b->esz, (hipStream_t)stream)) return 1;
  return 0;
}
int tsim_cache_clear(tsim_batch* b) {
  DeviceGuard guard_(b->device);
  for (auto& e : b->cache) b->pool.push_back(e.buf);
  b->cache.clear();
  while (b->pool.size() > 2) { (void)hipFree(b->pool.back()); b->pool.pop_back(); }     // keep two spares warm
  return 0;
}
int tsim_cache_depth(const tsim_batch* b) { return (int)b->cache.size(); }

int tsim_debug_signature(tsim_batch* b, int t_first, int n, uint32_t* out, void* stream) {
  if (!b->record) return fail("debug_signature: reset(backward_flag=True) was not called (the signature is taken from the tape)");
  if (t_first < 0 || n <= 0 || t_first + n > b->t_cur) return fail("debug_signature: sub-steps " + std::to_string(t_first) + "+" + std::to_string(n) + " are not on the tape (" + std::to_string(b->t_cur) + ")");
  if (!out) return fail("debug_signature: out is null");
  TS_DEVICE(b);
  if (b->dtype == TSIM_F32) {
    SigArgs<float> a{b->dI, (const float*)b->dF, (const float*)b->dFenv, b->nfrec, b->B, t_first, n, (const float*)b->tape, out, b->stage_cpt};
    hipLaunchKernelGGL(k_signature<float>, dim3(b->B), dim3(TS_WAVE), lds_bytes_for(b, 1), (hipStream_t)stream, a);
  } else {
    SigArgs<double> a{b->dI, (const double*)b->dF, (const double*)b->dFenv, b->nfrec, b->B, t_first, n, (const double*)b->tape, out, b->stage_cpt};
    hipLaunchKernelGGL(k_signature<double>, dim3(b->B), dim3(TS_WAVE), lds_bytes_for(b, 1), (hipStream_t)stream, a);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int tsim_debug_stamps(tsim_batch* b, long long* cycles) { b->bwd_stamps = cycles; return 0; }

int tsim_debug_eval(tsim_batch* b, const void* q1, const void* q0, const void* qd0, const void* u, void* g_out, void* H_out, long long* cycles, void* stream) {
  TS_DEVICE(b);
  // one environment per wavefront, unless TSIM_LPE forces a packed shape (nr <= 8 models: the stamped variant is NRM 8)
  const int lpe = (b->lpe_forced && !b->has_exp && (!cycles || b->nr <= 8)) ? b->lpe_forced : TS_WAVE;
  const int ns = TS_WAVE / lpe;
  const dim3 grid((b->B + ns - 1) / ns), blk(TS_WAVE);
  const size_t lds = lds_bytes_for(b, ns);
  if (b->dtype == TSIM_F32) {
    DbgArgs<float> a{b->dI, (const float*)b->dF, (const float*)b->dFenv, b->nfrec, b->B, (const float*)q1, (const float*)q0, (const float*)qd0, (const float*)u, (float*)g_out, (float*)H_out, cycles, b->stage_cpt};
    if (lpe == 16) hipLaunchKernelGGL((k_debug_eval<float, 16>), grid, blk, lds, (hipStream_t)stream, a);
    else if (lpe == 32) hipLaunchKernelGGL((k_debug_eval<float, 32>), grid, blk, lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_debug_eval<float, 64>), grid, blk, lds, (hipStream_t)stream, a);
  } else {
    DbgArgs<double> a{b->dI, (const double*)b->dF, (const double*)b->dFenv, b->nfrec, b->B, (const double*)q1, (const double*)q0, (const double*)qd0, (const double*)u, (double*)g_out, (double*)H_out, cycles, b->stage_cpt};
    if (lpe == 16) hipLaunchKernelGGL((k_debug_eval<double, 16>), grid, blk, lds, (hipStream_t)stream, a);
    else if (lpe == 32) hipLaunchKernelGGL((k_debug_eval<double, 32>), grid, blk, lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_debug_eval<double, 64>), grid, blk, lds, (hipStream_t)stream, a);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"

// ================================================================================================ TactilePush per-step formulas
#include "../../include/tsim_env.h"
#include "tsim_env_push.h"
