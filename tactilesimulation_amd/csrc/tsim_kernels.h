// tsim_kernels.h — the two simulation kernels (k_forward, k_backward) with their argument structs and the helpers only they use.
// A header, not a translation unit: the generic instantiations are compiled in tsim_hip.hip, the instantiations for statically known models
// (tsim_static.h) in their own translation unit (tsim_static_pusher.hip), which is built with -ffinite-math-only -fno-signed-zeros so that
// the structural zeros and ones of the compiled-in model fold away — flags the generic kernels, whose fp64 instantiations walk the
// oracle's iterates to round-off, are NOT built with.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tsim.h"
#include "tsim_eval.h"
#include "tsim_policy_push.h"

// the model's integrator (1 BDF1, 2 BDF2): read from the blob, or the compiled-in model's constant (the batch's int blob equals it: blob_has_structure)
template <class MS, class C> __device__ __forceinline__ int ts_integrator(const C& c) {
  if constexpr (std::is_void<MS>::value) return ts_u(c.I[TSIM_IH_INTEGRATOR]); else return MS::Iv(TSIM_IH_INTEGRATOR);
}
// number of dofs of a statically known model (0: generic kernels, the size is a run-time value)
template <class MS> constexpr int ts_static_nr() { if constexpr (std::is_void<MS>::value) return 0; else return MS::Iv(TSIM_IH_NR); }

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
// trial point of a backtracking line search, base + alpha dq, as ONE fused multiply-add wherever it is formed: the slot that owns the line search and
// a helper slot (k_forward) must arrive at the same bits
__device__ __forceinline__ float ts_trial_point(float base, float alpha, float dq) { return __builtin_fmaf(alpha, dq, base); }
__device__ __forceinline__ double ts_trial_point(double base, double alpha, double dq) { return __builtin_fma(alpha, dq, base); }
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
  int free_run = 0;            // the slots of a wavefront run their frames / sub-steps independently (k_forward, main loop)
  int lockstep = 0;            // ... or go through every sub-step together (a slot that has converged re-evaluates its iterate until all have)
  R* fposeR = nullptr; double* fposeD = nullptr;               // [nframes][B][nspt] pose records per frame: the tactile frames are evaluated by k_taxels after the launch
  int cull = 0;                // phase 2 skips contact pairs out of reach of their primitive (Ctx::cull)
  int vo_ls = 0;               // > 0: line-search trials after vo_ls rejected ones evaluate the residual only (k_forward, main loop; tsim_set_option TSIM_OPT_VALUE_TRIALS)
  int* helped = nullptr;        // [B] line-search trials of the launch that a helper slot evaluated for this environment (tsim_last_helper_trials)
  int vo_first = 0;            // forward-only launches: the first trial after a Newton step is evaluated without tangents where the previous sub-step converged in one step (k_forward; TSIM_OPT_VALUE_FIRST)
  int default_opts = 0;        // every option above is at its default (cross_kinks 1, eval_budget 0, vo_ls 2, vo_first 1, helpers 1, lockstep 0): the launcher of a
                               // compiled-in model may then pick the TsDefaultOpts<> instantiation, which has them as constants (tsim_static.h)
  int helpers = 0;             // slots that have finished their environment evaluate the NEXT line-search trials of a slot that is still in one (k_forward, main loop; TSIM_OPT_TRIAL_HELPERS)
};

// -DTS_WAVES_PER_EU=n (A/B builds): ask the compiler for n wavefronts per SIMD in the two simulation kernels (2 -> at most 256 registers)
#define TS_UNLIKELY(x) __builtin_expect(!!(x), 0)      // cold code: laid out behind the loop's straight-line path
#ifdef TS_WAVES_PER_EU
#define TS_KLB __launch_bounds__(TS_WAVE, TS_WAVES_PER_EU)
#else
// One wavefront per SIMD is what these kernels run at (their LDS footprint allows four blocks per CU): say so, or a kernel that happens to fit 256 registers
// is scheduled FOR two wavefronts per SIMD — shorter live ranges, less overlap of its own loads and arithmetic — which costs a lone wavefront
// (round 6: the adjoint kernel with the model's sizes folded fits 220 registers and ran 18 % slower than the 280-register one until this attribute)
#define TS_KLB __launch_bounds__(TS_WAVE) __attribute__((amdgpu_waves_per_eu(1, 1)))
#endif
template <class R, int NRM, bool EXPJ, int LPE, bool POLICY = false, class MS = void>
__global__ void TS_KLB k_forward(FwdArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  constexpr int NS = TS_WAVE / LPE;
  // a TsDefaultOpts<> instantiation (tsim_static.h): the options below are their defaults, as constants — the host launches it only then
  constexpr bool kFixed = TsHasDefaultOpts<MS>::value;
#define TS_OPT(field, fixed) (kFixed ? (fixed) : a.field)
  const int slot = threadIdx.x / LPE, lane = threadIdx.x % LPE;       // lane: inside the slot
  // Stragglers set the kernel time (all environments wait for the one with the most Newton work), so environments that
  // were expensive in the previous env-step are dispatched first: slot s of block b runs environment order[b NS + s]
  // (neighbours in that order have similar work, which also keeps the slots of one wavefront together).
  const int eidx = blockIdx.x * NS + slot;
  const bool valid = eidx < a.B;                                        // a batch that is no multiple of NS: idle slot
  const int env = a.order ? a.order[min(eidx, a.B - 1)] : min(eidx, a.B - 1);
  Ctx<R> c; ctx_init<R, MS>(c, a.I, a.F, lds, NS, slot, lane, LPE, a.stage_cpt != 0, a.Fenv ? a.Fenv + (size_t)env * a.fstride : nullptr);
  c.cull = a.cull;
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
  const bool bdf2_model = ts_integrator<MS>(c) == 2;      // (a compile-time constant for a compiled-in model: its BDF2 branches fold away)
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
  // A launch covers nframes env-steps (1 for tsim_step) of nsub sub-steps each, and every sub-step is a Newton iteration of a few
  // residual evaluations.  The evaluation is the one expensive thing here and it has ONE call site, in ONE loop; everything else is a
  // state machine PER SLOT around it (the state is identical in all lanes of a slot): each of the wavefront's environments is in its own
  // frame, its own sub-step and its own Newton iteration.  Nothing couples the environments of a batch inside a launch, so a slot whose
  // sub-step has converged starts its next one in the very next round instead of re-evaluating its final iterate until the slowest slot of
  // the wavefront has converged too: the wavefront lasts  max over its slots of (sum of the slot's evaluations)  instead of
  // sum over the sub-steps of (max over its slots)  — on BASELINE's headline batch 349 instead of 403 rounds for the slowest wavefront,
  // which is what a launch lasts (profiles/r04_free_running_slots.md).  The price: the code between two evaluations that is not
  // evaluation (commit of a sub-step: tape record, state shift, predictor; end of a frame: outputs) now runs once per slot that needs it
  // instead of once per wavefront.  That is why the tactile read-out is no longer part of a free-running launch: a.free_run launches leave,
  // per frame, the pose records of the (sensor, primitive) combinations (a.fposeR / a.fposeD), and the taxels of all frames are evaluated
  // by k_taxels afterwards, lanes = taxels, at full occupancy.
  // Launches that need the frames' outputs INSIDE the launch (POLICY: the next action is computed from this frame's tactile image) or
  // whose read-out cannot be deferred (a.free_run == 0) keep their slots together at the frame ends only: a slot that has finished the
  // last sub-step of its frame re-evaluates its final iterate until the others have.
#ifdef TS_PP_TIME      // A/B builds only: share of the launch spent in the policy call, left in gnorm (tools/closed_loop_breakdown.py)
  long long pp_cycles_ = 0; const long long pp_t0_ = clock64();
#endif
  const bool free_run = !POLICY && a.free_run != 0 && (a.tac_out == nullptr || a.fposeR != nullptr);
  int f = 0, s = 0, tslot = 0;
  bool done = a.nframes <= 0;
  bool fs = !done, ss = !done;               // this slot is at the start of a frame (fetch the action) / of a sub-step (predictor)
  bool held = false;                         // !free_run: the last sub-step of the frame is finished, waiting for the other slots
  R unext = R(0);                            // the next frame's action, fetched one frame ahead (a lone wavefront cannot hide the load)
  if (!POLICY && !done && lane < nu) unext = a.u[(size_t)env * nu + lane];
  R gn = R(0), alpha = R(1);
  int iter = 0, ls = -1, sub_evals = 0, crossings = 0;       // ls < 0: the evaluation just done is not a line-search trial
  bool conv = false, fin = false, forced = false;
  // Line-search trials need ||g|| only.  A slot that is deep in a backtracking (>= a.vo_ls rejected trials in this iteration) marks its next
  // evaluation `vo`; a round in which no live slot of the wavefront needs a Newton matrix evaluates the VALUES only (evaluate(..., tang = false):
  // no tangents, no contact Jacobians, no 72 of the 78 reductions per pair — about half a round).  A values-only trial that would be TAKEN
  // (accepted, or the last of max_ls) is evaluated once more, with tangents, before anything is decided: g comes out bit-identical, so the
  // decision repeats itself and H is the full evaluation's.  Iterates, convergence and the taped matrices are exactly those of the loop
  // without the option; `evals` counts trial points (the repeats are not counted).  Where it pays: the environments a launch waits for
  // are the ones in long line searches (a D'Claw fingertip jammed against the cap: 100 iterations x 12 trials; profiles/r05_helpers.md, profiles/r05_option_ab.jsonl).
  bool vo = false;
  // Helper slots (round 5).  What a launch waits for is its slowest environment's CHAIN of evaluations, and those chains are long line
  // searches (D'Claw: a fingertip jammed against the cap, 100 iterations x 12 trials; TactileInsertion: one sub-step of 60 - 300 trials) —
  // while the other slots of that wavefront have long finished their own environments and re-evaluate a final iterate for nothing.  The trial
  // points of a backtracking are known in advance: dlbase + 2^-t dq, t = 0, 1, 2 ...  So a slot that is `done` evaluates trial t + j of a slot
  // that is about to evaluate its trial t (same predictor, control, increment base and direction, read from that slot's LDS; the owner's
  // parameter table with per-environment tables), and the owner then judges the results IN ORDER with the decision code of the sequential
  // loop: a rejected own trial is followed at once by the helper's result for the next one, and so on.  A helper's point that is to be taken
  // is adopted (its g and H copied from the helper's LDS) when the round computed tangents and a Newton step follows; otherwise the owner
  // evaluates that point again itself, in full, exactly as the value-only trials do.  The evaluation is a function of its inputs only
  // (reductions inside a slot are symmetric), so iterates, convergence flags, taped matrices and `evals` (trial points judged) are those of
  // the loop without helpers, bit for bit (tests/test_gpu_exact_options.py); a line search of n trials takes ceil(n / (1 + helpers)) rounds.
  // Not in launches that leave their final link records for tsim_readout (a helper's records are not its environment's).
#define helpers_on (NS > 1 && !POLICY && TS_OPT(helpers, 1) != 0 && !TS_OPT(lockstep, 0) && a.poseR == nullptr)      /* (re-read from the kernel arguments where it is asked: no register held for it) */
  static_assert(NS <= 4, "helper slots: three result registers (gh0..gh2) and step factors 1/2, 1/4, 1/8 — at most three helpers per owner");
  if (helpers_on && !valid) { done = true; fs = false; ss = false; }      // an idle slot of the last wavefront helps from the start
  int helped = 0;
  // Value-first trials (round 5; launches that record no tape).  A sub-step whose Newton iteration converges in ONE step — nearly all of them:
  // 2.07 evaluations per sub-step on TactileInsertion, 2.5 on D'Claw and TactilePush — is an evaluation at the predictor (g and H: the Newton step)
  // and one at the new point, of which only ||g|| < tol is used: H of the final iterate goes to the tape, and there is none.  So where the
  // Newton step just solved is expected to END the sub-step (`kq` below), its first trial is marked value-only like the deep line-search trials: if it
  // ends the sub-step (converged, or out of iterations / budget) it is taken as it is — q, qd, the link records are those of a full evaluation —
  // otherwise it is evaluated again in full (H for the next step) exactly as the value-only trials are.  A round is value-only only if no live slot
  // needs H from it, and free-running slots drift out of phase (one at its predictor while the other is at its trial: every round full); a slot
  // about to START a sub-step therefore waits ONE round — at most once per sub-step — when that makes the round value-only, which puts it in
  // phase with the others.  Same iterates, flags and evaluation counts as without the option (tests/test_gpu_exact_options.py).
  const bool vfirst = TS_OPT(vo_first, 1) != 0 && a.record == 0 && !POLICY && !TS_OPT(lockstep, 0);
  // `kq`: the last measured contraction of a full Newton step of this slot, ||g_new|| / ||g||^2 (quadratic convergence: roughly a constant of
  // the problem) — the first trial after a step from residual gn is expected to end the sub-step if kq gn^2 is well below tol.
  R kq = R(1e30);
  bool waited = false;
#ifdef TS_ROUND_STATS   // A/B builds only (tools/round_stats.py): rounds of this wavefront and its shader clocks, left in status / gnorm
  int rounds_ = 0; const long long rs_t0_ = clock64();
#endif
  while (!__all(done)) {
#ifdef TS_ROUND_STATS
    ++rounds_;
#endif
    if (POLICY ? __any(fs) : fs) {
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
      } else {
        uv = unext;
        if (lane < nu) {
          c.u[lane] = uv;
          if (f + 1 < a.nframes) unext = a.u[((size_t)(f + 1) * a.B + env) * nu + lane];
        }
      }
      // a NaN / inf control would be clamped away silently by the motor law's fmin / fmax: flag it (status bit 30) instead
      if (seg_sum<LPE>(ts_finite(uv) ? R(0) : R(1)) > R(0)) nonfinite = true;
      tslot = a.tac_slot ? a.tac_slot[f] : f;      // used at the end of the frame: fetched here, the load is long done by then
      fs = false;
    }
    bool wait_ = false;                        // this slot sits this round out (see `vfirst` above): it re-evaluates its last point, nothing is judged
    if (vfirst) {
      const bool live = !done && !held && !fin;
      if (__any(live && !ss && vo && ls == 0) && !__any(live && !ss && !vo)) wait_ = ss && !done && !waited;      // (ls == 0: a slot at such a first trial — one deep in a line search is value-only round after round, there is no phase to meet)
      if (wait_) waited = true;
    }
    if (ss && !wait_) {
      waited = false;
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
      if (lane < nr) c.dl[lane] = R(0);          // initial guess: the predictor
      gn = R(0); alpha = R(1); iter = 0; ls = -1; sub_evals = 0; crossings = 0; conv = false; fin = false; forced = false; vo = false;
      ss = false;
    }
    TS_SYNC();
    // Newton with backtracking EXACTLY as the model file states it (<solver_option tol max_iter max_ls>, pusher.xml:4): up to max_iter
    // iterations; each halves the step until ||g|| decreases, at most max_ls times, and takes the last trial if none did; converged when
    // ||g||_2 < tol.  Nothing else: no non-monotone steps, no restart, no trust region (rounds 1-2 had all three, tuned for the slowest
    // wavefront; on the stiff TactileInsertion grasp their full Newton step across a kink "converged" to a root 0.15 rad away from the one
    // plain backtracking reaches — found by the oracle's literal solver in round 3).  The oracle (oracle/tsim_oracle.cpp
    // substep_literal) is the same loop in fp64; the fp64 kernels take its iterates.
    // An accepted trial's evaluation is the next iteration's Jacobian evaluation.  A slot that has nothing left to do (done, or held at
    // the end of a frame) keeps evaluating at its final iterate: the same numbers again.
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
    const bool tang = (TS_OPT(vo_ls, 2) <= 0 && !vfirst) || __any(!fin && !done && !held && !vo && !wait_);      // does any live slot need H from this round?
    // ---- helper slots: who evaluates whose next trials this round (wave-uniform masks over the slots; see above).  Nothing of this is live
    //      across the evaluation (the kernels hold one wavefront per SIMD on their register count): the helpers are set up here, the owners
    //      find theirs again afterwards, from the same masks
    auto slot_masks = [&](bool need, unsigned& cm, unsigned& nm) {
      const unsigned long long cb = __ballot(done), nb = __ballot(need);
      cm = 0; nm = 0;
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) { cm |= (unsigned)((cb >> (s_ * LPE)) & 1ull) << s_; nm |= (unsigned)((nb >> (s_ * LPE)) & 1ull) << s_; }
    };
    auto nth = [](unsigned mask, int n) { int r = 0;
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) { if ((mask >> s_) & 1u) { if (n == 0) r = s_; --n; } } return r; };
    // (nothing to set up — and nothing spent on it — while every slot still has its own environment)
    if (TS_UNLIKELY(helpers_on && __any(done))) {               // (cold code: laid out behind the loop's straight-line path)
      const bool need = !done && !fin && !held && !wait_ && ls >= 0 && !forced;      // in a line search: this round evaluates trial ls, the next would be ls + 1
      unsigned cm, nm; slot_masks(need, cm, nm);
      if (cm != 0 && nm != 0) {                   // wave-uniform
        const int m = __popc(nm);
        int owner = slot, off = 0;
        if (done) { const int j = __popc(cm & ((1u << slot) - 1u)); owner = nth(nm, j % m); off = 1 + j / m; }      // helper j serves owner j mod m with its trial + 1 + j / m
        // the owner's loop state (identical in all lanes of its slot) travels through the LDS crossbar; every lane takes part
        const int src = owner * LPE + lane;
        const int ols = lane_gather(ls, src);
        const R oalpha = lane_gather(alpha, src), ocv = lane_gather(c.cv, src), oca = lane_gather(c.ca, src);
        if (done && ols + off <= c.max_ls) {
          const int d = (owner - slot) * ts_lds_env_reals(c.nl, nr, nu, (int)sizeof(R));   // this slot's LDS arrays -> the owner's
          const R ah = oalpha * (off == 1 ? R(0.5) : off == 2 ? R(0.25) : R(0.125));
          if (lane < nr) {
            c.qp[lane] = c.qp[lane + d]; c.qdp[lane] = c.qdp[lane + d];
            c.qpD[lane] = (c.qpD + d * (int)sizeof(R) / 8)[lane];
            c.dl[lane] = ts_trial_point(dlbase[lane + d], ah, c.dq[lane + d]);
          }
          if (lane < nu) c.u[lane] = c.u[lane + d];
          c.cv = ocv; c.ca = oca;                 // (a finished slot has no use for its own any more)
          if (a.Fenv) c.F = lds + owner * (a.fstride + 2);      // the owner's parameter table (ctx_init: one per slot, at the start of the block's LDS)
        }
      }
    }
    TS_SYNC();
    evaluate<R, NRM, EXPJ, LPE, MS>(c, lane, R(1), c.cv, c.ca, tang);      // (seeds of the Newton matrix: (1, cv, ca))
    const R gnew = block_norm2<LPE>(c.g, nr, lane);
    // ---- this slot's helpers of this round (cold code, as above): their number, their slots in the order of the trials they evaluated, and the
    //      ||g|| they found — scalars, not arrays: a dynamically indexed array would live in scratch memory
    int nh = 0, hs0 = slot, hs1 = slot, hs2 = slot;
    R gh0 = gnew, gh1 = gnew, gh2 = gnew;
    if (TS_UNLIKELY(helpers_on && __any(done))) {
      if (a.Fenv) c.F = lds + slot * (a.fstride + 2);
      const bool need = !done && !fin && !held && !wait_ && ls >= 0 && !forced;      // (the state the helpers were assigned from: nothing has changed it)
      unsigned cm, nm; slot_masks(need, cm, nm);
      if (cm != 0 && nm != 0) {
        const int m = __popc(nm), kc = __popc(cm), r_ = __popc(nm & ((1u << slot) - 1u));
        if (need) nh = r_ < kc ? (kc - r_ + m - 1) / m : 0;
        hs0 = nh > 0 ? nth(cm, r_) : slot;
        gh0 = lane_gather(gnew, hs0 * LPE + lane);
        if (NS > 2) { hs1 = nh > 1 ? nth(cm, r_ + m) : slot; gh1 = lane_gather(gnew, hs1 * LPE + lane); }
        if (NS > 3) { hs2 = nh > 2 ? nth(cm, r_ + 2 * m) : slot; gh2 = lane_gather(gnew, hs2 * LPE + lane); }
      }
    }
    bool solve = false, take = false;
    R gtake = gnew;
    // What an evaluation means for this slot — gj = ||g|| at the trial point, evaluated by this slot itself (j = 0) or by its helper in slot hj (j > 0):
    // the full step across a kink comes next / the step is halved / the point is taken.  Returns true iff the step was halved (c.dl is the next
    // trial point then, and a helper may have evaluated exactly that point already).
    auto judge = [&](const R gj, const int j, const int hj) -> bool {
      int action = 2;                        // the first evaluation of the sub-step, an accepted trial, or the step across a kink
      if (ls >= 0 && !forced && (!ts_finite(gj) || gj >= gn)) {              // a rejected trial (a non-finite one is rejected too)
        if (TS_OPT(cross_kinks, 1) && ls >= min(c.max_ls, TSIM_KINK_LS) && crossings < TSIM_KINK_MAX && gn < R(TSIM_KINK_FACTOR) * c.tol) action = 0;
        else if (ls < c.max_ls) action = 1;
      }                                      // (else: the literal loop takes the last trial anyway)
      if (action == 2) {
        // to be taken — which needs H of this point in this slot's LDS.  Its own evaluation without tangents: the same point again, in full
        // (nothing else changes).  A helper's: adopted if it came with tangents and a Newton step follows from it; if the sub-step ends
        // there (its records, q and qd are the frame's), or without tangents, this slot evaluates the point again itself.
        // A launch that records no tape has no use for H of a point that ENDS the sub-step: this slot's own value-only evaluation of such a
        // point is taken as it is, a helper's is evaluated again by this slot (its link records are the frame's) — value-only.
        const bool ends = !ts_finite(gj) || gj < c.tol || (ls >= 0 && iter + 1 >= c.max_iter) || (TS_OPT(eval_budget, 0) > 0 && sub_evals + 1 >= TS_OPT(eval_budget, 0));
        const bool no_h = ends && a.record == 0;
        bool usable = j == 0 ? (tang || no_h) : (tang && !ends);
        vo = !usable && no_h && (TS_OPT(vo_ls, 2) > 0 || vfirst);
        if (usable) {
          ++evals; ++sub_evals; take = true; gtake = gj;
          if (j > 0) {
            const int d = (hj - slot) * ts_lds_env_reals(c.nl, nr, nu, (int)sizeof(R));
            if (lane < nr) c.g[lane] = c.g[lane + d];
            for (int e = lane; e < nr * nr; e += LPE) c.H[e] = c.H[e + d];
          }
        }
        return false;
      }
      ++evals; ++sub_evals;
      if (action == 0) {
        ++crossings; forced = true; vo = false;      // close to convergence and no decrease down to 2^-TSIM_KINK_LS: the full step across the kink
        if (lane < nr) c.dl[lane] = dlbase[lane] + c.dq[lane];
        return false;
      }
      alpha *= R(0.5); ++ls;                 // halve the step
      vo = TS_OPT(vo_ls, 2) > 0 && ls >= TS_OPT(vo_ls, 2);
      if (lane < nr) c.dl[lane] = ts_trial_point(dlbase[lane], alpha, c.dq[lane]);
      return true;
    };
    if (!fin && !done && !wait_) {
      bool halved = judge(gnew, 0, slot);      // this slot's own evaluation (the one straight-line call of every round) ...
      if (TS_UNLIKELY(nh > 0)) {               // ... then its helpers', in the order the sequential loop would have made those evaluations
#pragma unroll 1
        for (int j = 1; halved && j <= nh; ++j) {      // (a loop, not unrolled: the kernels are as large as the instruction cache)
          const R gj = (NS > 3 && j == 3) ? gh2 : (NS > 2 && j == 2) ? gh1 : gh0;
          const int hj = (NS > 3 && j == 3) ? hs2 : (NS > 2 && j == 2) ? hs1 : hs0;
          ++helped;
          halved = judge(gj, j, hj);
        }
      }
      if (take) {
        if (vfirst && ls >= 0) kq = (ls == 0 && !forced && gn > R(0)) ? gtake / (gn * gn) : R(1e30);      // a full Newton step taken at once: its contraction; a damped one: no prediction
        forced = false;
        if (ls >= 0) ++iter;
        gn = gtake;
        if (!ts_finite(gn)) { nonfinite = true; fin = true; }
        else if (gn < c.tol) { conv = true; fin = true; }
        else if (iter >= c.max_iter || (TS_OPT(eval_budget, 0) > 0 && sub_evals >= TS_OPT(eval_budget, 0))) fin = true;
        else {
          solve = true;
          if (lane < nr) { c.rhs[lane] = -c.g[lane]; dlbase[lane] = c.dl[lane]; }
        }
      }
    }
    TS_SYNC();
    if (__any(solve)) {
      solve_newton<R, NRM, LPE, double, ts_static_nr<MS>()>(c.H, c.rhs, c.dq, nr, false, lane, solve);      // (elimination in fp32: -0.7 %, not taken; profiles/r04_static_model.md)
      if (solve) {
        alpha = R(1); ls = 0;
        vo = vfirst && kq * gn * gn < R(0.25) * c.tol;      // the trial that will most likely end the sub-step: ||g|| only (see `vfirst` above)
        if (lane < nr) c.dl[lane] = dlbase[lane] + c.dq[lane];
      }
      TS_SYNC();
    }
    // ---- commit the sub-steps that ended with this evaluation: c.q = q1, c.qd = (q1 - q0)/h, c.H = dg/dq1 at q1
    bool commit = fin && !done && !held;
    if (TS_OPT(lockstep, 0) && !__all(fin || done)) commit = false;      // the slots of a wavefront go through the sub-steps together (see FwdArgs)
    if (!free_run) {                           // the last sub-step of a frame is committed by all slots together
      if (commit && s == a.nsub - 1) { held = true; commit = false; }
      if (__all(held || done)) { commit = held; held = false; }
    }
    bool frame_end = false;
    if (commit) {
      if (!conv) ++bad;
      gmax = t_max(gmax, gn);
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
      fin = false; ss = true;
      if (++s == a.nsub) { s = 0; frame_end = true; }
    }
    TS_SYNC();
    // ---- end of a frame (per slot in a free-running launch, all slots together otherwise): the link poses / velocities in LDS are
    //      those of the accepted state (last evaluation)
    if (TS_UNLIKELY(__any(frame_end))) {       // once per frame and slot: kept out of the loop's straight-line code
      // the fused static evaluation leaves no link records in LDS: write those of the frame's final state now (what the read-out reads)
      if constexpr (ts_static_fused<MS, R>()) { if (frame_end) ts_static_value_records<R, MS>(c, lane); }
      if (frame_end && lane < nr && valid) {
        const size_t o = ((size_t)f * a.B + env) * nr + lane;
        if (a.q_out) a.q_out[o] = (R)c.q0D[lane];        // the double position rounded once (== tsim_get_state)
        if (a.qd_out) a.qd_out[o] = c.qd0[lane];
      }
      const bool tac_here = a.tac_out != nullptr && !a.fposeR;       // in-kernel read-out: wave-uniform, and so are f and tslot then (!free_run)
      readout<LPE>(c, lane, env, frame_end && valid, frame_end && valid && tslot >= 0, a.var_out != nullptr, tac_here && tslot >= 0,
                   a.var_out ? a.var_out + (size_t)f * a.B * 3 * c.nvar : nullptr,
                   (tac_here && tslot >= 0) ? a.tac_out + (size_t)tslot * a.B * 3 * c.ntax : nullptr);
      if (a.fposeR && frame_end && tslot >= 0) {                     // deferred read-out: this frame's pose records (as k_readout leaves them)
        int k = 0;
        for (int sn = 0; sn < c.nsensor; ++sn) {
          const int* si = c.I + c.off_sensor + sn * TSIM_SI_SIZE;
          const int nsp = ts_u(si[TSIM_SI_NSPRIM]), sp0 = ts_u(si[TSIM_SI_SPRIM0]);
          for (int j = 0; j < nsp; ++j, ++k) {
            TS_SYNC();
            pair_stage_value(c, ts_u(c.I[c.off_sprim + sp0 + j]), 0, lane == 0);
            TS_SYNC();
            const size_t rec = ((size_t)f * a.B + env) * a.nspt + k;
            if (valid) {
              for (int e = lane; e < TP_R_SIZE; e += LPE) a.fposeR[rec * TP_R_SIZE + e] = c.PP[e];
              if (lane < TP_D_SIZE) a.fposeD[rec * TP_D_SIZE + lane] = c.PPd[lane];
            }
          }
        }
      }
      if (frame_end) { ++f; fs = true; if (f == a.nframes) { done = true; fs = false; ss = false; } }
      TS_SYNC();
    }
  }
  if constexpr (ts_static_fused<MS, R>()) { if (a.poseR && a.nframes <= 0) ts_static_value_records<R, MS>(c, lane); }
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
    if (a.helped && lane == 0) a.helped[env] = helped;
    if (a.gnorm && lane == 0) a.gnorm[env] = (float)gmax;
#ifdef TS_PP_TIME
    if (a.gnorm && lane == 0) a.gnorm[env] = (float)((double)pp_cycles_ / (double)(clock64() - pp_t0_));
#endif
#ifdef TS_ROUND_STATS
    if (lane == 0) { if (a.gnorm) a.gnorm[env] = (float)(clock64() - rs_t0_); if (a.status) a.status[env] = rounds_; }
#endif
  }
}
#undef helpers_on
#undef TS_OPT


// ================================================================================================ backward kernel
template <class R, int LPE, class MS> __device__ __forceinline__ void ts_static_output_vjp(const Ctx<R>& c, int lane, const R* wvar, const R* wtac);      // below
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
  int cull = 0;
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

// The taxel loop of the tactile adjoint: lanes = taxels t0 .. t0 + nt of one sensor against one primitive whose staged pose is P; every lane
// accumulates in g the gradient of  w . out  w.r.t. the pair's relative displacement (dth, drho) and relative twist (dw, dv), primitive frame.
// Returns whether any lane of the wavefront had a loaded, seeded taxel.  PRIMC >= 0: the primitive type as a constant (static models), with
// an fp32 "certainly outside" test in front of the double-precision position.  shape / sf: the primitive's shape and the sensor's penalty record.
template <int LPE, class R, int PRIMC = -1>
__device__ __forceinline__ bool vjp_taxels(const Ctx<R>& c, int lane, int t0, int nt, int prim, const R* shape, const R* sf, const PairPose<R>& P, const R* wtac, R (&g)[12]) {
  const M3<R> RPA = P.RPA;
  const V3<R> pPA = P.pPA, wrel = P.wrel, vrel = P.vrel;
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
    if (PRIMC >= 0 && sizeof(R) == 4 && live) live = prim_distance<R>(prim, shape, mulMv(RPA, mk3<R>(x0, x1, x2)) + pPA) < R(TS_FAR_MARGIN);
    V3<R> xP, F; M3<R> Jx, Jv;
    if (live) {
      const V3<double> xPd = mulMv(P.RPAd, mk3<double>((double)x0, (double)x1, (double)x2)) + P.pPAd;
      xP = cvt3<R>(xPd);
      live = contact_law<R, true>(prim, shape, sf, xP, vrel + cross3(wrel, xP), F, Jx, Jv, xPd);
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
  return any_live;
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
      PairPose<R> P;
      P.RPAd = ldm(c.PPd); P.pPAd = ldv(c.PPd + 9);
      P.RPA = ldm(S + PP_RPA); P.pPA = ldv(S + PP_PPA); P.wrel = ldv(S + PP_WREL); P.vrel = ldv(S + PP_VREL);
      R g[12];
      const bool any_live = vjp_taxels<LPE, R>(c, lane, t0, nt, prim, pf + TSIM_PF_SHAPE, sf, P, wtac, g);
      if (!any_live) continue;
      seg_sum_many<LPE, 12>(g);
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

#ifdef TS_BWD_TWO_WAVES      // A/B builds only: the adjoint kernel may share a SIMD with a second wavefront (two-environment wavefronts, TSIM_BWD_LPE=32)
#define TS_KLB_BWD __launch_bounds__(TS_WAVE)
#else
#define TS_KLB_BWD TS_KLB
#endif
template <class R, int NRM, bool EXPJ, int LPE, bool POLICY = false, class MS = void>
__global__ void TS_KLB_BWD k_backward(BwdArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  constexpr int NS = TS_WAVE / LPE;
  const int slot = threadIdx.x / LPE, lane = threadIdx.x % LPE;
  const bool valid = (int)blockIdx.x * NS + slot < a.B;
  const int env = min((int)blockIdx.x * NS + slot, a.B - 1);
  // (the context's sizes stay run-time values HERE: with them folded the adjoint kernel is 10 % shorter and 18 % SLOWER — 0.82 -> 0.96 ms per 20-step
  // launch, round 6; the forward kernel gains 5 % from the same fold)
  Ctx<R> c; ctx_init<R, MS>(c, a.I, a.F, lds, NS, slot, lane, LPE, a.stage_cpt != 0, a.Fenv ? a.Fenv + (size_t)env * a.fstride : nullptr);
  c.cull = a.cull;
  const int nr = c.nr, nu = c.nu, REC = ts_rec(nr, nu, (int)sizeof(R));
  const int nvar3 = 3 * c.nvar, ntac3 = 3 * c.ntax;
  R* H2 = c.H2;    // taped Newton matrix of the sub-step
  init_world(c, lane, LPE);
  if (a.cyc && blockIdx.x == 0) c.stamps = a.cyc;
  if (lane < nr) { c.lamq[lane] = a.lamq[(size_t)env * nr + lane]; c.lamv[lane] = a.lamv[(size_t)env * nr + lane]; }
  // BDF2 models: taped sub-step t >= 2 is a BDF2 step (the first one after a reset is the BDF1 start-up, k_forward).  Its new state
  // depends on the TWO states before it, so next to the adjoint of the state one step back (lamq, lamv) the kernel carries what later
  // sub-steps already contributed to the state two steps back (lq1, lv1: one value per lane, in registers; second half of the buffers).
  const bool bdf2_model = ts_integrator<MS>(c) == 2;      // (a compile-time constant for a compiled-in model: its BDF2 branches fold away)
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
    // (the loads above must be ISSUED here, a whole sub-step ahead of their use: with the model's sizes as compile-time constants the scheduler
    // otherwise sinks them to the top of the next iteration and a lone wavefront waits ~2 us of HBM latency per sub-step — measured in round 6:
    // k_backward 0.82 -> 0.96 ms per 20-step launch with 10 % FEWER instructions)
    __builtin_amdgcn_sched_barrier(0);
    TS_SYNC();
    TS_STAMP(c);
    // A statically known model (tsim_static_eval.h): the evaluation at the taped state is one register-resident pass AFTER the adjoint solve
    // and returns this lane's (H^T z, M z) — no records, no c.H.  Only a sub-step that carries a loss seed needs link records in LDS
    // (output_vjp reads them): it runs the link sweep alone first.
    constexpr bool kFused = ts_static_fused<MS, R>();
    const bool seeded = (j + 1) % a.seed_stride == 0;
    if constexpr (kFused) { }      // (a seeded sub-step runs its own link sweep inside ts_static_output_vjp)
    else if constexpr (std::is_void<MS>::value) phase1<R, true, EXPJ>(c, lane, R(1), R(0), R(0));
    else phase1_static_levels<R, MS, true>(c, lane, R(1), R(0), R(0));
    TS_STAMP(c);
    // direct partials of the loss w.r.t. this sub-step's outputs
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
      if constexpr (kFused) ts_static_output_vjp<R, LPE, MS>(c, lane, (a.df_dvar && nvar3) ? a.df_dvar + so * nvar3 : nullptr, wtac_);
      else output_vjp<LPE>(c, lane, (a.df_dvar && nvar3) ? a.df_dvar + so * nvar3 : nullptr, wtac_);
    }
    TS_STAMP(c);
#ifdef TS_BWD_REEVAL      // A/B builds only (profiles/r06_tape_ab.md): what a tape WITHOUT the Newton matrix would cost — the matrix of the taped point is evaluated
                          // again here (the forward kernel's evaluation with tangents) and the adjoint solve uses it instead of the taped one
    if (lane < nr) {
      c.qp[lane] = c.q0[lane] + c.h * c.qd0[lane]; c.qdp[lane] = c.qd0[lane];
      c.dl[lane] = (R)(c.qD[lane] - (double)c.qp[lane]); c.qpD[lane] = c.qD[lane] - (double)c.dl[lane];
    }
    TS_SYNC();
    evaluate<R, NRM, EXPJ, LPE, MS>(c, lane, R(1), c.cv, c.ca, true);
    for (int e = lane; e < nr * nr; e += LPE) H2[e] = c.H[e];
    TS_SYNC();
#endif
    if (lane < nr) c.rhs[lane] = c.lamq[lane] + c.cv * c.lamv[lane];      // d qd1 / d q1 = cv
    TS_SYNC();
    solve_newton<R, NRM, LPE, double, ts_static_nr<MS>()>(H2, c.rhs, c.z, nr, true, lane);
    TS_STAMP(c);
    R ym, yq = R(0);
    if constexpr (!kFused) {
      phase2<R, NRM, LPE, MS>(c, lane, R(1));
      TS_STAMP(c);
      phase3<R, EXPJ, LPE>(c, lane, R(1), R(0));       // c.H = h^2 dr/dq
      TS_STAMP(c);
      ym = mass_times_z<LPE>(c, lane);
      if (lane < nr) for (int i = 0; i < nr; ++i) yq += c.z[i] * c.H[i * nr + lane];
    } else {
      TS_STAMP(c);
      evaluate_static_fused_adjoint<R, NRM, LPE, MS>(c, lane, yq, ym);
      TS_STAMP(c);
    }
    TS_STAMP(c);
    if (lane < nr) {
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


// ---- output_vjp for a statically known model with the fused evaluation (tsim_static_eval.h): the link sweep's states stay in registers, so
// the pose of each (sensor, primitive) combination, this lane's 12-vector of it and the end-effector Jacobians need no record in LDS;
// primitive type, shape and the sensor's penalty record are constants.  Adds to lam_q / lam_v exactly what output_vjp adds.
template <class R, int LPE, class MS, int SN, int J>
__device__ __forceinline__ void ts_vjp_sprim(const Ctx<R>& c, int lane, const TsLinkState<R>* st, const S6<R>& Wk, const R* wtac, R& dlq, R& dlv) {
  using T = TsTopo<MS>;
  constexpr int so = MS::Iv(TSIM_IH_OFF_SENSOR) + SN * TSIM_SI_SIZE, nsp = MS::Iv(so + TSIM_SI_NSPRIM);
  if constexpr (J < nsp) {
    constexpr int pk = MS::Iv(MS::Iv(TSIM_IH_OFF_SPRIM) + MS::Iv(so + TSIM_SI_SPRIM0) + J);
    constexpr int o = MS::Iv(TSIM_IH_OFF_PAIR) + pk * TSIM_PI_SIZE, prim = MS::Iv(o + TSIM_PI_PRIM), la = MS::Iv(o + TSIM_PI_LINKA), lb = MS::Iv(o + TSIM_PI_LINKB);
    const int t0 = ts_u(c.I[c.off_sensor + SN * TSIM_SI_SIZE + TSIM_SI_TAX0]), nt = ts_u(c.I[c.off_sensor + SN * TSIM_SI_SIZE + TSIM_SI_NTAX]);      // the taxel layout is the batch's own (blob_equals_static)
    R pf[TSIM_PF_SIZE], sf[TSIM_SF_SIZE];
#pragma unroll
    for (int e = 0; e < TSIM_SF_SIZE; ++e) sf[e] = ts_F<R, MS>(c, MS::Iv(TSIM_IH_FOFF_SENSOR) + SN * TSIM_SF_SIZE + e);
    M3<R> RP; V3<R> pP; PairPose<R> P; S6<R> Vrel, dVA, dVB;
    ts_fused_pair_pose<R, MS, pk>(c, st, pf, RP, pP, P, Vrel, dVA, dVB);
    R g[12];
    const bool any_live = vjp_taxels<LPE, R, prim>(c, lane, t0, nt, prim, pf + TSIM_PF_SHAPE, sf, P, wtac, g);
    if (any_live) {
      seg_sum_many<LPE, 12>(g);
      // lanes = directions: this lane's 12-vector of the pair (pair_stage_tangent, vmode 0) and its twist column alone (vmode 1)
      const int k = lane;
      constexpr int ancA = la != 0 ? T::li(la == 0 ? 1 : la, TSIM_LI_ANCMASK) : 0, ancB = lb != 0 ? T::li(lb == 0 ? 1 : lb, TSIM_LI_ANCMASK) : 0;
      const R inA = ((ancA >> k) & 1) ? R(1) : R(0), inB = ((ancB >> k) & 1) ? R(1) : R(0);
      const S6<R> dxiP = to_frame(RP, pP, Wk * (inA - inB));
      const S6<R> dxiB = to_frame(RP, pP, Wk * inB);
      const S6<R> dVrel = to_frame(RP, pP, dVA - dVB) - crm(dxiB, Vrel);
      dlq += g[0] * dxiP.a.x + g[1] * dxiP.a.y + g[2] * dxiP.a.z + g[3] * dxiP.l.x + g[4] * dxiP.l.y + g[5] * dxiP.l.z
           + g[6] * dVrel.a.x + g[7] * dVrel.a.y + g[8] * dVrel.a.z + g[9] * dVrel.l.x + g[10] * dVrel.l.y + g[11] * dVrel.l.z;
      dlv += g[6] * dxiP.a.x + g[7] * dxiP.a.y + g[8] * dxiP.a.z + g[9] * dxiP.l.x + g[10] * dxiP.l.y + g[11] * dxiP.l.z;      // d(relative twist) / d qd_k = the same frame change of W_k
    }
    ts_vjp_sprim<R, LPE, MS, SN, J + 1>(c, lane, st, Wk, wtac, dlq, dlv);
  }
}
template <class R, int LPE, class MS, int SN>
__device__ __forceinline__ void ts_vjp_sensors(const Ctx<R>& c, int lane, const TsLinkState<R>* st, const S6<R>& Wk, const R* wtac, R& dlq, R& dlv) {
  if constexpr (SN < MS::Iv(TSIM_IH_NSENSOR)) {
    ts_vjp_sprim<R, LPE, MS, SN, 0>(c, lane, st, Wk, wtac, dlq, dlv);
    ts_vjp_sensors<R, LPE, MS, SN + 1>(c, lane, st, Wk, wtac, dlq, dlv);
  }
}
template <class R, class MS, int E>
__device__ __forceinline__ void ts_vjp_vars(const Ctx<R>& c, int lane, const TsLinkTmp<R>* tmp, const S6<R>& Wk, const R* wvar, R& dlq) {
  using T = TsTopo<MS>;
  if constexpr (E < MS::Iv(TSIM_IH_NVAR)) {
    constexpr int l = MS::Iv(MS::Iv(TSIM_IH_OFF_VAR) + E * TSIM_VI_SIZE + TSIM_VI_LINK), fo = MS::Iv(TSIM_IH_FOFF_VAR) + E * TSIM_VF_SIZE;
    if constexpr (l != 0) {
      constexpr int anc = T::li(l == 0 ? 1 : l, TSIM_LI_ANCMASK);
      const R mv = ((anc >> lane) & 1) ? R(1) : R(0);
      const V3<R> x = mulMv(tmp[l].XR, mk3<R>(ts_F<R, MS>(c, fo), ts_F<R, MS>(c, fo + 1), ts_F<R, MS>(c, fo + 2))) + tmp[l].Xp;
      const V3<R> Jv_ = cross3(Wk.a, x) + Wk.l;
      dlq += mv * (wvar[3 * E] * Jv_.x + wvar[3 * E + 1] * Jv_.y + wvar[3 * E + 2] * Jv_.z);
    }
    ts_vjp_vars<R, MS, E + 1>(c, lane, tmp, Wk, wvar, dlq);
  }
}
template <class R, int LPE, class MS>
__device__ __forceinline__ void ts_static_output_vjp(const Ctx<R>& c, int lane, const R* wvar, const R* wtac) {
  using T = TsTopo<MS>;
  TS_SYNC();
  TsLinkState<R> st[T::NL + 1];
  TsLinkTmp<R> tmp[T::NL + 1];
  S6<R> Wk = zero6<R>(), dFl[T::NL + 1];
  ts_l_level<R, MS, true, false, 0, 0>(c, lane, R(1), R(0), R(0), st, tmp, Wk, dFl);      // values + this lane's twist tangents (seeds (1, 0, 0)), nothing stored
  R dlq = R(0), dlv = R(0);
  if (wvar) ts_vjp_vars<R, MS, 0>(c, lane, tmp, Wk, wvar, dlq);
  if (wtac) ts_vjp_sensors<R, LPE, MS, 0>(c, lane, st, Wk, wtac, dlq, dlv);
  if (lane < T::NR) { c.lamq[lane] += dlq; c.lamv[lane] += dlv; }
  TS_SYNC();
}

// ================================================================================================ debug evaluation
template <class R> struct DbgArgs { const int* I; const R* F; const R* Fenv; int fstride; int B; const R *q1, *q0, *qd0, *u; R *g, *H; long long* cyc; int stage_cpt; int cull; };

template <class R, int LPE, class MS = void>
__global__ void __launch_bounds__(TS_WAVE) k_debug_eval(DbgArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  R* lds = reinterpret_cast<R*>(smem_raw);
  constexpr int NS = TS_WAVE / LPE;
  const int slot = threadIdx.x / LPE, lane = threadIdx.x % LPE;
  const bool valid = (int)blockIdx.x * NS + slot < a.B;
  const int env = min((int)blockIdx.x * NS + slot, a.B - 1);
  Ctx<R> c; ctx_init<R, MS>(c, a.I, a.F, lds, NS, slot, lane, LPE, a.stage_cpt != 0, a.Fenv ? a.Fenv + (size_t)env * a.fstride : nullptr);
  c.cull = a.cull;
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
    evaluate<R, 8, false, LPE, MS>(c, lane, R(1), R(1) / c.h, R(1) / (c.h * c.h));
    if (lane < nr) c.rhs[lane] = -c.g[lane];
    TS_SYNC();
    solve_newton<R, 8, LPE>(c.H, c.rhs, c.dq, nr, false, lane);
    TS_STAMP(c);
    if (lane == 0 && valid) for (int i = (slot == 0 ? c.nstamp : 0); i < 32; ++i) c.stamps[i] = 0;
  } else if constexpr (std::is_void<MS>::value) {
    evaluate<R, 16, true, LPE>(c, lane, R(1), R(1) / c.h, R(1) / (c.h * c.h));
  } else {
    evaluate<R, 8, false, LPE, MS>(c, lane, R(1), R(1) / c.h, R(1) / (c.h * c.h));
  }
  if (lane < nr && valid) a.g[(size_t)env * nr + lane] = c.g[lane];
  if (valid) for (int e = lane; e < nr * nr; e += LPE) a.H[(size_t)env * nr * nr + e] = c.H[e];
}

