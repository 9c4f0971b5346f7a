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
#include <cmath>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/tsim.h"
#include "tsim_eval.h"
#include "tsim_policy_push.h"
#include "tsim_static_pusher.h"


#include "tsim_kernels.h"

// ================================================================================================ LPT ordering
// One block: counting sort of the environments by their residual-evaluation count of the last launch, descending
// (64 bins; order inside a bin is irrelevant), dealt out to the wavefronts like cards: rank r goes to slot r / nwaves of
// wavefront r % nwaves (slot 0) or nwaves - 1 - r % nwaves (the other slots: see `deal` below).  The expensive environments start first AND sit in different wavefronts — the slots of a
// wavefront are sub-step-synchronous, so two expensive environments in one wavefront cost the sum of their per-sub-step
// maxima (measured: a batch sorted by work runs 22 % slower than the unsorted one, profiles/r01_imbalance_exp.json).
// Runs on the same stream right after k_forward.
// Episode launches (nframes > 1) are ordered by the per-environment TOTALS of the previous episode launch of the same length — useful
// when consecutive episodes resemble each other (the bench replays one table; a GD epoch with fixed start states and a slowly
// changing policy), harmless when they do not (any order is a valid one).  bin = (evals - lo) * 64 / span maps the totals onto the
// 64 bins (lo = 2 evaluations per sub-step, the minimum of a converging Newton loop; span = 2.5 per sub-step); per-step launches keep
// bin = evals (lo 0, span 64).
__global__ void __launch_bounds__(1024) k_order_by_evals(const int* evals, int* order, int B, int ns, int lo, int span, int deal) {
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
    int w = r % nwaves;                                         // slot r / nwaves of wavefront r % nwaves (B % ns == 0)
    const int sl = r / nwaves;
    // deal 2 (round 6, the default): every slot but the first is dealt BACKWARDS — the most expensive environments then share their wavefronts with
    // the cheapest of each quantile, whose slots finish first and become their line-search helpers earliest (D'Claw at B = 2048, two environments per
    // wavefront: 34.9 -> 32.4 ms per 50-step launch; TactilePush 2.766 -> 2.747 ms; TactileInsertion unchanged; profiles/r06_lpt_deal_ab.json).
    // A/B at creation, TSIM_LPT_DEAL: 0 = every slot dealt forwards (rounds 1 - 5), 1 = every other slot backwards (snake: no better than 0).
    if ((deal == 1 && (sl & 1)) || (deal == 2 && sl > 0)) w = nwaves - 1 - w;
    order[w * ns + sl] = e;
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
  // blockIdx.x = frame * B + environment: the pose records are [frames][B][nspt] (one frame for tsim_readout; all frames of an episode
  // launch whose read-out was deferred, launch_forward); frame f goes to slot tac_slot[f] of tac_out (null: slot f; < 0: not read out)
  int B; const int* tac_slot;
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
// 4096 environments 4.4 - 4.8 -> 4.7 - 5.0 TB/s  (Ordinary stores, the A/B's losing side: profiles/r04_readout_hbm.md.)
template <class R> __device__ __forceinline__ void ts_store3(R* o, R a, R b, R c) {
  __builtin_nontemporal_store(a, o); __builtin_nontemporal_store(b, o + 1); __builtin_nontemporal_store(c, o + 2);
}
// One taxel against the staged tables of its environment (LDS): per sensor the end of its taxel range, its first (sensor, primitive)
// record and their number, its penalty parameters; per record the primitive type, its shape, the pose record (R part, double part) and
// the bounding sphere in the sensor-link frame.  The three outputs go out as ONE 12-byte (fp64: 24-byte) store per lane.
template <class R>
__device__ __forceinline__ void taxel_eval(int t, V3<R> xa, int nsensor, const int* sEnd, const int* sKb, const int* sNsp, const int* sPrim, const R* sSf,
                                           const R* sShape, const R* sP, const double* sD, const R* sC, const R* tax, int ntax, R* out) {
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
  ts_store3(out + 3 * t, o0, o1, o2);
}
// bounding sphere of a (sensor, primitive) record in the sensor-link frame: centre c_A = -R_PA^T p_PA and (radius + margin)^2, < 0 for planes
template <class R> __device__ __forceinline__ void taxel_bound(const R* P, int prim, const R* sh, R* C) {
  const V3<R> cA = mulMtv(ldm(P), ldv(P + 9)) * R(-1);
  R rb = R(-1);
  if (prim == TSIM_P_SPHERE) rb = sh[0];
  else if (prim == TSIM_P_CUBOID) rb = t_sqrt(sh[0] * sh[0] + sh[1] * sh[1] + sh[2] * sh[2]);
  else if (prim == TSIM_P_CYLINDER) rb = t_sqrt(sh[0] * sh[0] + sh[1] * sh[1]);
  C[0] = cA.x; C[1] = cA.y; C[2] = cA.z;
  C[3] = rb < R(0) ? R(-1) : (rb + R(TS_FAR_MARGIN)) * (rb + R(TS_FAR_MARGIN));
}
enum { TS_TAX_UNROLL = 2 };      // taxels per thread and loop iteration: their loads are in flight together (1 / 2 / 4 measured: profiles/r03_readout_ab.md)
template <class R>
__global__ void __launch_bounds__(256) k_taxels(TaxArgs<R> a) {
  const int rec_env = blockIdx.x, env = rec_env % a.B, frame = rec_env / a.B;
  const int tslot = a.tac_slot ? a.tac_slot[frame] : frame;
  if (tslot < 0) return;                                                // block-uniform
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
    for (int i = threadIdx.x; i < a.nspt * TP_R_SIZE; i += 256) sP[i] = a.poseR[(size_t)rec_env * a.nspt * TP_R_SIZE + i];
    for (int i = threadIdx.x; i < a.nspt * TP_D_SIZE; i += 256) sD[i] = a.poseD[(size_t)rec_env * a.nspt * TP_D_SIZE + i];
    __syncthreads();
    // "Certainly outside" in 7 instructions: a taxel at x_A (sensor-link frame) cannot touch a primitive whose bounding sphere (centre
    // c_A = -R_PA^T p_PA, radius r_b) it is farther from than r_b + TS_FAR_MARGIN.  Most taxels of a large pad are nowhere near the
    // primitive, and for them this replaces the rotation into the primitive's frame (12 LDS reads, ~25 instructions); the margin is
    // far above the rounding of the test, so the set of taxels contact_law accepts is unchanged.  Planes have no bound (radius < 0).
    for (int k = threadIdx.x; k < a.nspt; k += 256) taxel_bound<R>(sP + k * TP_R_SIZE, sPrim[k], sShape + k * 4, sC + 4 * k);
    __syncthreads();
  }
  const R* tax = a.F + a.foff_taxel;                          // SoA planes: position (3), axis0, axis1, normal (9); shared
  R* out = a.tac_out + ((size_t)tslot * a.B + env) * 3 * ntax;
  const int te = min(ntax, ((int)blockIdx.y + 1) * a.slice);
  // One taxel: its three outputs go out as ONE 12-byte (fp64: 24-byte) store per lane — consecutive lanes, consecutive addresses: a
  // wavefront's store instruction covers 768 contiguous bytes (global_store_dwordx3 in the ISA).  Routing them through LDS for 16-byte
  // vectors instead was measured twice and is slower both ways (block-wide with barriers: round 2; wave-private without: round 3,
  // 2.45 -> 2.12 TB/s on the bench leg, profiles/r03_readout_ab.md).
  auto taxel = [&](int t, V3<R> xa) { taxel_eval<R>(t, xa, nsensor, sEnd, sKb, sNsp, sPrim, sSf, sShape, sP, sD, sC, tax, ntax, out); };
  if (nsensor == 1 && a.nspt == 1 && sC[3] >= R(0)) {
    // One sensor against one bounded primitive (RollingBall's pad and ball, TactilePush's pad and box), in two passes per 1024 taxels:
    //   pass 1  lanes = taxels: 3 loads, the bounding-sphere test against centre / radius^2 held in registers (7 instructions), zeros
    //           stored for the taxels outside; the few inside are appended to a list in LDS;
    //   pass 2  lanes = LISTED taxels: the penalty law, with full wavefronts.
    // Without the list a wavefront runs the ~1000-instruction law whenever ONE of its 64 taxels is near the primitive — 15 % of the
    // wavefronts of the RollingBall pad for 5 % of its taxels, and that was the kernel's time (12.4 M vector instructions for
    // 10.2 M taxels; profiles/r03_readout_ab.md).  A taxel's result does not depend on the lane that computes it: same bits as before.
    enum { CH = 8 };
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

// Small pads, many records (the frames of an episode launch: 81 920 records of 130 taxels on BASELINE's headline batch): one block per
// record is all prologue (k_taxels above: ~3 us of dependent loads and two barriers for 130 taxels).  Here a block stages TXS_RPB records
// at once — one cooperative copy, one barrier — and its threads then run over (record, taxel) items; consecutive items are consecutive
// taxels of one record, so loads and the 12-byte stores stay contiguous.  Same taxel_eval, same tables: the same bits as k_taxels.
enum { TXS_RPB = 16, TXS_MAXK = 8, TXS_MAXS = 4, TXS_MAX_TAXELS = 1024 };
template <class R>
__global__ void __launch_bounds__(256) k_taxels_small(TaxArgs<R> a, int nrec) {
  const int rec0 = blockIdx.x * TXS_RPB, nr_ = min((int)TXS_RPB, nrec - rec0);
  const int ntax = a.ntax, nsensor = a.nsensor, nspt = a.nspt;
  __shared__ int sEnd[TXS_MAXS], sKb[TXS_MAXS], sNsp[TXS_MAXS], sPrim[TXS_MAXK], sPair[TXS_MAXK], sSlot[TXS_RPB];
  __shared__ R sSf[TXS_RPB][TXS_MAXS * TSIM_SF_SIZE], sShape[TXS_RPB][TXS_MAXK * 4], sP[TXS_RPB][TXS_MAXK * TP_R_SIZE];
  __shared__ double sD[TXS_RPB][TXS_MAXK * TP_D_SIZE];
  __shared__ __attribute__((aligned(16))) R sC[TXS_RPB][TXS_MAXK * 4];
  __shared__ R sX[3 * TXS_MAX_TAXELS];                                  // the taxels' positions (shared by all records): no global load left in the item loop
  const R* tax = a.F + a.foff_taxel;
  for (int i = threadIdx.x; i < 3 * ntax; i += 256) sX[i] = tax[i];
  const int* TT = a.I + a.tt_off;
  for (int i = threadIdx.x; i < nsensor; i += 256) { sEnd[i] = TT[3 * i]; sKb[i] = TT[3 * i + 1]; sNsp[i] = TT[3 * i + 2]; }
  for (int i = threadIdx.x; i < nspt; i += 256) { sPrim[i] = TT[3 * nsensor + 2 * i]; sPair[i] = TT[3 * nsensor + 2 * i + 1]; }
  for (int i = threadIdx.x; i < nr_; i += 256) { const int fr = (rec0 + i) / a.B; sSlot[i] = a.tac_slot ? a.tac_slot[fr] : fr; }
  for (int i = threadIdx.x; i < nr_ * nspt * TP_R_SIZE; i += 256) { const int r = i / (nspt * TP_R_SIZE), e = i % (nspt * TP_R_SIZE); sP[r][e] = a.poseR[(size_t)(rec0 + r) * nspt * TP_R_SIZE + e]; }
  for (int i = threadIdx.x; i < nr_ * nspt * TP_D_SIZE; i += 256) { const int r = i / (nspt * TP_D_SIZE), e = i % (nspt * TP_D_SIZE); sD[r][e] = a.poseD[(size_t)(rec0 + r) * nspt * TP_D_SIZE + e]; }
  for (int i = threadIdx.x; i < nr_ * nsensor * TSIM_SF_SIZE; i += 256) {
    const int r = i / (nsensor * TSIM_SF_SIZE), e = i % (nsensor * TSIM_SF_SIZE);
    const R* F = a.Fenv ? a.Fenv + (size_t)((rec0 + r) % a.B) * a.fstride : a.F;
    sSf[r][e] = F[a.foff_sensor + e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nr_ * nspt * 4; i += 256) {
    const int r = i / (nspt * 4), e = i % (nspt * 4);
    const R* F = a.Fenv ? a.Fenv + (size_t)((rec0 + r) % a.B) * a.fstride : a.F;
    sShape[r][e] = F[a.foff_pair + sPair[e >> 2] * TSIM_PF_SIZE + TSIM_PF_SHAPE + (e & 3)];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nr_ * nspt; i += 256) { const int r = i / nspt, k = i % nspt; taxel_bound<R>(sP[r] + k * TP_R_SIZE, sPrim[k], sShape[r] + k * 4, sC[r] + 4 * k); }
  __syncthreads();
  for (int item = threadIdx.x; item < nr_ * ntax; item += 256) {
    const int r = item / ntax, t = item - r * ntax;
    const int tslot = sSlot[r];
    if (tslot < 0) continue;
    R* out = a.tac_out + ((size_t)tslot * a.B + (rec0 + r) % a.B) * 3 * ntax;
    taxel_eval<R>(t, mk3<R>(sX[t], sX[t + ntax], sX[t + 2 * ntax]), nsensor, sEnd, sKb, sNsp, sPrim, sSf[r], sShape[r], sP[r], sD[r], sC[r], tax, ntax, out);
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

// ================================================================================================ host side
static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
int tsim_fail_(const std::string& m) { return fail(m); }      // for the library's other translation units (tsim_model.cpp)
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
struct KtPair { hipEvent_t a, b; int kind; };
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
  int* helped = nullptr;         // ... and how many of its line-search trials a helper slot evaluated (k_forward; tsim_last_helper_trials)
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
  int value_trials = 2;          // line-search trials after this many rejected ones evaluate the residual only (0: off; tsim_set_option TSIM_OPT_VALUE_TRIALS; TSIM_VALUE_TRIALS=n at creation)
  int trial_helpers = 1;        // finished slots of a wavefront evaluate the next line-search trials of a slot that is still in one (tsim_set_option TSIM_OPT_TRIAL_HELPERS; TSIM_NO_TRIAL_HELPERS=1 at creation: off)
  int value_first = 1;           // launches without a tape: the first trial after a Newton step evaluates the residual only where the previous sub-step converged in one step (tsim_set_option TSIM_OPT_VALUE_FIRST; TSIM_NO_VALUE_FIRST=1 at creation: off)
  // A/B switches of the environment, read ONCE at creation (launches are on the host-bound path of the per-step collectors):
  // TSIM_NO_EPISODE_LPT, TSIM_INKERNEL_READOUT, TSIM_NO_FREE_RUN, TSIM_LOCKSTEP, TSIM_TAXELS_PER_RECORD, TSIM_NO_ENVTAB_CPT
  bool ab_no_episode_lpt = false, ab_inkernel_readout = false, ab_no_free_run = false, ab_lockstep = false, ab_taxels_per_record = false, ab_no_envtab_cpt = false, ab_no_default_opts = false; int ab_bwd_lpe = 0, ab_lpt_deal = 2;
  int pair_cull = 1;             // phase 2 skips contact pairs out of reach of their primitive (tsim_set_option TSIM_OPT_PAIR_CULL; TSIM_NO_PAIR_CULL=1 at creation: off)
  // Compiled-in models (csrc/tsim_static.h).  static_id: the model whose STRUCTURE the batch's blob has (ints + the structural floats: 1 TactilePush);
  // static_exact: every float record equals the compiled asset's bit for bit as well (the fully static instantiation); env_struct_ok: the
  // per-environment tables keep the structural floats (checked on the device by tsim_set_env_tables).  TSIM_NO_STATIC=1 at creation /
  // tsim_set_static(0) keep the generic kernels.
  int static_id = 0, static_exact = 0, env_struct_ok = 0, no_static = 0;
  unsigned char* dKmask = nullptr; int* dFlag = nullptr;      // structural-float mask of static_id on the device, one-int result of the table check
  void* fposeR = nullptr; double* fposeD = nullptr; int fpose_frames = 0;   // pose records per frame [fpose_frames][B][nspt] of episode launches with a deferred read-out
  int pose_valid = 0;            // the pose records are those of the current state (left by the last forward launch)
  int pose_off = 0;              // a launch of this batch was captured in a HIP graph: replays change the state behind the host's back, no reuse
  int tt_off = 0;                // offset of the taxel staging table in the device int blob (I[NI] + S[TS_SCHED_TAXTAB])
  int tax_slots = 0;             // blocks of k_taxels the device holds at once (occupancy x CUs), queried on first use
  size_t esz;
  std::vector<CacheEntry> cache;   // saved tapes, newest last
  std::vector<void*> pool;          // spare tape buffers
  std::vector<void*> retired;       // per-frame pose records replaced by larger ones while a captured graph may still name them (launch_forward)
  long long* bwd_stamps = nullptr;  // diagnostics (tsim_debug_stamps)
  // tsim_kernel_timing: HIP events around every launch of the simulation kernels, on the stream they are launched on
  int kt_on = 0;
  std::vector<KtPair> kt;           // pairs recorded since the last tsim_kernel_times
  std::vector<hipEvent_t> kt_free;  // events to reuse
};
// One timed launch: start event in the constructor, stop event in the destructor (nothing under stream capture: no events inside a graph).
struct KtScope {
  tsim_batch* b; hipStream_t st; KtPair p; bool on = false;
  KtScope(tsim_batch* b_, int kind, hipStream_t st_) : b(b_), st(st_) {
    if (!b->kt_on || b->kt.size() >= (size_t)1 << 16) return;      // (a caller that never reads the times back does not grow the list without bound)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
    hipEvent_t e[2];
    for (int i = 0; i < 2; ++i) {
      if (!b->kt_free.empty()) { e[i] = b->kt_free.back(); b->kt_free.pop_back(); }
      else if (hipEventCreate(&e[i]) != hipSuccess) { (void)hipGetLastError(); if (i) b->kt_free.push_back(e[0]); return; }
    }
    p.a = e[0]; p.b = e[1]; p.kind = kind;
    on = hipEventRecord(p.a, st) == hipSuccess;
    if (!on) { b->kt_free.push_back(e[0]); b->kt_free.push_back(e[1]); }
  }
  ~KtScope() { if (on) { (void)hipEventRecord(p.b, st); b->kt.push_back(p); } }
};
// The state changed other than by a forward launch (or is about to, in a captured graph): tsim_readout recomputes the kinematics.
static void pose_invalidate(tsim_batch* b, hipStream_t st) {
  b->pose_valid = 0;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) b->pose_off = 1;
}


// sweep schedule of the link tree (layout: ts_sched in tsim_device.h)
static std::vector<int32_t> build_sched(const std::vector<int32_t>& I, const std::vector<double>& F) {
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
  {   // bounding sphere of each contact pair's points, link-A frame (ts_pair_bound): centre of the points' box, radius rounded UP
    S[TS_SCHED_PBOUND] = (int32_t)S.size();
    const int ncpt = I[TSIM_IH_NCPT], fc = I[TSIM_IH_FOFF_CPT];
    for (int pk = 0; pk < npair; ++pk) {
      const int32_t* pi = &I[op + pk * TSIM_PI_SIZE];
      const int p0_ = pi[TSIM_PI_PT0], n_ = pi[TSIM_PI_NPT];
      float bd[4] = {0.f, 0.f, 0.f, -1.f};
      if (n_ > 0 && !(pi[TSIM_PI_FLAGS] & 2)) {
        double lo[3], hi[3], cc[3];
        for (int a_ = 0; a_ < 3; ++a_) { lo[a_] = 1e300; hi[a_] = -1e300; }
        for (int i = 0; i < n_; ++i) for (int a_ = 0; a_ < 3; ++a_) { const double v = F[fc + a_ * ncpt + p0_ + i]; lo[a_] = std::min(lo[a_], v); hi[a_] = std::max(hi[a_], v); }
        for (int a_ = 0; a_ < 3; ++a_) { bd[a_] = (float)(0.5 * (lo[a_] + hi[a_])); cc[a_] = (double)bd[a_]; }      // the centre the device will use
        double r2 = 0.0;
        for (int i = 0; i < n_; ++i) { double d2 = 0.0; for (int a_ = 0; a_ < 3; ++a_) { const double d = F[fc + a_ * ncpt + p0_ + i] - cc[a_]; d2 += d * d; } r2 = std::max(r2, d2); }
        bd[3] = (float)(std::sqrt(r2) * (1.0 + 1e-6) + 1e-7);
      }
      for (int e = 0; e < 4; ++e) { int32_t w; std::memcpy(&w, &bd[e], 4); S.push_back(w); }
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

// Does the batch's model equal a compiled-in static model (int blob and float records, bit for bit)?  Then its launches may use the
// instantiation that has the model's structure folded in (tsim_static.h); any edit of a record (tsim_update_model) drops back to the
// generic kernels, and so do per-environment tables.
// Is the batch's model the compiled-in one in everything the static kernels take from it?  All int records and all float records in front of
// the per-point arrays, bit for bit — EXCEPT the taxel layout (number of taxels, a sensor's taxel range and image shape): the static code reads
// none of it (the read-out kernels and the tactile adjoint take taxels from the batch's own blob), so a TactilePush pad re-gridded to
// 13 x 13 taxels (BASELINE configs[1]) runs on the same instantiation as the XML's 13 x 10.
template <class MS> static bool blob_equals_static(const tsim_batch* b) {
  if ((int)b->I.size() != MS::NI || b->I[TSIM_IH_FOFF_CPT] != MS::NFREC) return false;
  auto taxel_layout = [&](int i) {
    if (i == TSIM_IH_NTAXEL || i == TSIM_IH_NF || i == TSIM_IH_NDOF_TACTILE) return true;
    const int os = MS::Iv(TSIM_IH_OFF_SENSOR), ns = MS::Iv(TSIM_IH_NSENSOR);
    if (i >= os && i < os + ns * TSIM_SI_SIZE) { const int f = (i - os) % TSIM_SI_SIZE; return f == TSIM_SI_TAX0 || f == TSIM_SI_NTAX || f == TSIM_SI_ROWS || f == TSIM_SI_COLS; }
    return false;
  };
  for (int i = 0; i < MS::NI; ++i) if (b->I[i] != MS::Iv(i) && !taxel_layout(i)) return false;
  // fp64 batches: the doubles, bit for bit.  fp32 batches: the FLOATS, bit for bit — the fp32 kernels see a model's reals only after their conversion to
  // float (upload_model; ts_F casts the compiled-in constant the same way), so a blob that differs from the asset below float resolution (another host's
  // BLAS in the Python compiler, the native loader of tsim_model.cpp: last bits of mesh-derived mass properties) IS the compiled-in model to them.
  for (int i = 0; i < MS::NFREC; ++i) {
    if (b->dtype == TSIM_F32) { const float x = (float)b->F[i], y = (float)MS::Fv(i); if (std::memcmp(&x, &y, sizeof(float)) != 0) return false; }
    else if (std::memcmp(&b->F[i], (const double[]){MS::Fv(i)}, sizeof(double)) != 0) return false;
  }
  return true;
}
// ... or in its STRUCTURE only: all ints (but the taxel layout) and the structural floats of the compiled asset (TsParam::Fk: exact 0, 1, -1)
template <class MS> static bool blob_has_structure(const tsim_batch* b) {
  if ((int)b->I.size() != MS::NI || b->I[TSIM_IH_FOFF_CPT] != MS::NFREC) return false;
  auto taxel_layout = [&](int i) {
    if (i == TSIM_IH_NTAXEL || i == TSIM_IH_NF || i == TSIM_IH_NDOF_TACTILE) return true;
    const int os = MS::Iv(TSIM_IH_OFF_SENSOR), ns = MS::Iv(TSIM_IH_NSENSOR);
    if (i >= os && i < os + ns * TSIM_SI_SIZE) { const int f = (i - os) % TSIM_SI_SIZE; return f == TSIM_SI_TAX0 || f == TSIM_SI_NTAX || f == TSIM_SI_ROWS || f == TSIM_SI_COLS; }
    return false;
  };
  for (int i = 0; i < MS::NI; ++i) if (b->I[i] != MS::Iv(i) && !taxel_layout(i)) return false;
  for (int i = 0; i < MS::NFREC; ++i) if (MS::Fk(i) && !(b->F[i] == MS::Fv(i))) return false;
  return true;
}
static void detect_static_model(tsim_batch* b) {
  b->static_id = blob_has_structure<TsParam<TsStaticPusher>>(b) ? 1 : 0;
  b->static_exact = b->static_id == 1 && blob_equals_static<TsStaticPusher>(b);
}
// which instantiation the next launch of the simulation kernels uses: 0 generic, 1 fully static, 2 structure-static (parameters at run time)
enum { TS_KM_GENERIC = 0, TS_KM_STATIC = 1, TS_KM_PARAM = 2 };
static int kernel_mode(const tsim_batch* b) {
  if (b->static_id == 0 || b->no_static) return TS_KM_GENERIC;
  if (b->dtype == TSIM_F64 && b->lpe_forced == 16) return TS_KM_GENERIC;    // fp64: no compiled-in instantiation with four environments per wavefront (16 lanes only when forced: its LDS is over the automatic cap)
  if (b->dFenv) return b->env_struct_ok ? TS_KM_PARAM : TS_KM_GENERIC;      // (the table check is fp32 only: fp64 batches with per-environment tables stay generic)
  return b->static_exact ? TS_KM_STATIC : TS_KM_PARAM;
}
// every solver / scheduling option of the batch at its default: the forward launch of a compiled-in model may use the TsDefaultOpts<> instantiation
// (tsim_static.h), which has them as constants.  (TSIM_NO_DEFAULT_OPTS=1 at creation keeps the run-time-option kernel: A/B.)
static bool default_options(const tsim_batch* b) {
  return b->cross_kinks == 1 && b->eval_budget == 0 && b->value_trials == 2 && b->trial_helpers == 1 && b->value_first == 1 && !b->ab_lockstep && !b->ab_no_default_opts;
}
static int upload_model(tsim_batch* b, hipStream_t st) {
  detect_static_model(b);
  HIPCHK(hipMemcpyAsync(b->dI, b->I.data(), b->I.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
  {
    std::vector<int32_t> S = build_sched(b->I, b->F);              // appended to the device copy at I[TSIM_IH_NI]
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

// do the per-environment tables [B][nfrec] keep the structural floats of the compiled-in model (mask km: 1 = structural; ref: the batch's
// shared records, which do)?  flag != 0: some environment does not
__global__ void k_check_structure(const float* tables, const float* ref, const unsigned char* km, int nfrec, size_t n, int* flag) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = (int)(i % (size_t)nfrec);
  if (km[f] && !(tables[i] == ref[f])) atomicOr(flag, 1);
}
// the float header of every per-environment row (time step, gravity, Newton tolerance) is the shared model's: see tsim_set_env_tables
template <class R> __global__ void k_env_header(R* tables, const R* ref, int nfrec, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * TSIM_FH_SIZE) tables[(size_t)(i / TSIM_FH_SIZE) * nfrec + i % TSIM_FH_SIZE] = ref[i % TSIM_FH_SIZE];
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
    // (the fused static kernels: 1 : 1.13 : 1.24 — their rounds get cheaper with more lanes per environment, so a batch that fills the SIMDs
    // with one environment per wavefront takes that shape: TactilePush 13 x 13 at B = 1024, forward only: 9.0 / 8.0 / 7.3 M env-steps/s)
    const bool fused_static = kernel_mode(b) != TS_KM_GENERIC;
    const double lat_generic[3] = {1.0, 0.93, 1.02}, lat_static[3] = {1.0, 1.13, 1.24};
    const double* lat = fused_static ? lat_static : lat_generic;
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
  if (b->dFenv && b->ab_no_envtab_cpt) return;      // A/B: the round-3 behaviour (contact points from global memory next to per-environment tables)
  const int lpe0 = launch_shape(b).lpe;
  b->stage_cpt = 1;
  if (launch_shape(b).lpe != lpe0 || lds_bytes_for(b, 1) > 64 * 1024) b->stage_cpt = 0;
}
// launchers of the statically specialised instantiations (tsim_static_pusher.hip)
void ts_static_pusher_launch(const FwdArgs<float>& a, int lpe, unsigned grid, size_t lds, hipStream_t st);
void ts_static_pusher_launch(const BwdArgs<float>& a, int lpe, unsigned grid, size_t lds, hipStream_t st);
void ts_static_pusher_launch_policy(const FwdArgs<float>& a, unsigned grid, size_t lds, hipStream_t st);      // ... with the policy between the frames
void ts_static_pusher_launch_policy(const BwdArgs<float>& a, unsigned grid, size_t lds, hipStream_t st);
void ts_param_pusher_launch_policy(const FwdArgs<float>& a, unsigned grid, size_t lds, hipStream_t st);       // ... on the structure-static kernels (tsim_param_pusher_policy.hip)
void ts_param_pusher_launch_policy(const BwdArgs<float>& a, unsigned grid, size_t lds, hipStream_t st);
void ts_static_pusher_launch_debug(const DbgArgs<float>& a, unsigned grid, size_t lds, hipStream_t st);
// ... and of the structure-static ones (tsim_param_pusher.hip): the same kernels with the model's parameters read from the float records
void ts_param_pusher_launch(const FwdArgs<float>& a, int lpe, unsigned grid, size_t lds, hipStream_t st);
void ts_param_pusher_launch(const BwdArgs<float>& a, int lpe, unsigned grid, size_t lds, hipStream_t st);
void ts_param_pusher_launch_debug(const DbgArgs<float>& a, unsigned grid, size_t lds, hipStream_t st);
// ... both in fp64 (two or one environments per wavefront)
void ts_static_pusher_launch(const FwdArgs<double>& a, int lpe, unsigned grid, size_t lds, hipStream_t st);
void ts_static_pusher_launch(const BwdArgs<double>& a, int lpe, unsigned grid, size_t lds, hipStream_t st);
void ts_param_pusher_launch(const FwdArgs<double>& a, int lpe, unsigned grid, size_t lds, hipStream_t st);
void ts_param_pusher_launch(const BwdArgs<double>& a, int lpe, unsigned grid, size_t lds, hipStream_t st);
// kernel variants: NRM = 8 / 16 rows in the register solve; EXPJ = model has a rotation-vector joint (its code is
// compiled out otherwise: it costs registers in every evaluation); LPE as above
#define TS_LAUNCH_L(KERNEL, R, NRM, L, st, a) do {                                                                       \
    if (L.lpe == 64) hipLaunchKernelGGL((KERNEL<R, NRM, false, 64>), dim3(L.grid), dim3(TS_WAVE), L.lds, st, a);         \
    else if (L.lpe == 32) hipLaunchKernelGGL((KERNEL<R, NRM, false, 32>), dim3(L.grid), dim3(TS_WAVE), L.lds, st, a);    \
    else hipLaunchKernelGGL((KERNEL<R, NRM, false, 16>), dim3(L.grid), dim3(TS_WAVE), L.lds, st, a);                     \
  } while (0)
#define TS_LAUNCH(KERNEL, R, b, st, a) do {                                                                              \
    const LaunchShape L = launch_shape(b);                                                                               \
    if (sizeof(R) == 4 || L.lpe != 16) {   /* a statically known model (tsim_static.h): instantiated in its own translation unit (fp64: not four environments per wavefront) */ \
      const int km_ = kernel_mode(b);                                                                                     \
      if (km_ == TS_KM_STATIC) { ts_static_pusher_launch(a, L.lpe, L.grid, L.lds, st); break; }                           \
      if (km_ == TS_KM_PARAM) { ts_param_pusher_launch(a, L.lpe, L.grid, L.lds, st); break; }                             \
    }                                                                                                                     \
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

// k_taxels over the pose records [frames][B][nspt]: frame f -> slot tac_slot[f] (device array; null: slot f) of tac_out.
// Taxels per block.  A block's prologue (staging the pose records and the model tables of its environment, two barriers)
// is a chain of dependent global loads, ~3 us whatever the slice; with many short blocks per SIMD slot the kernel WAS that prologue
// (8192 blocks of 5 taxels per thread: 4.6 rounds of ~7 us for 256 x 40 000 taxels).  So: ONE round — as many blocks as the device
// holds at once (occupancy x CUs), each with a slice long enough to cover the batch; never less than 256 taxels per block, so the
// single environment of test_sim_speed.py still spreads its 40 000 taxels over 157 blocks.
static bool taxels_supported(const tsim_batch* b) { return b->ntax > 0 && b->nspt > 0 && b->nspt <= TX_MAXK && b->I[TSIM_IH_NSENSOR] <= TX_MAXS; }
template <class R>
static int launch_taxels(tsim_batch* b, const void* poseR, const double* poseD, int frames, const int32_t* tac_slot, void* tac_out, hipStream_t st) {
  if (b->tax_slots == 0) {
    int per_cu = 0;
    const hipError_t e_ = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_taxels<R>, 256, 0);
    b->tax_slots = (e_ == hipSuccess && per_cu > 0 ? per_cu : 4) * (b->n_simd / 4);
  }
  const long long nrec = (long long)frames * b->B;
  if (nrec >= 4 * TXS_RPB && b->ntax <= TXS_MAX_TAXELS && b->nspt <= TXS_MAXK && b->I[TSIM_IH_NSENSOR] <= TXS_MAXS && !b->ab_taxels_per_record) {
    TaxArgs<R> t{b->dI, (const R*)b->dF, (const R*)b->dFenv, b->nfrec, (const R*)poseR, poseD, b->nspt, (R*)tac_out, 0, b->B, tac_slot,
                 b->ntax, b->I[TSIM_IH_NSENSOR], b->I[TSIM_IH_FOFF_SENSOR], b->I[TSIM_IH_FOFF_PAIR], b->I[TSIM_IH_FOFF_TAXEL], b->tt_off};
    hipLaunchKernelGGL(k_taxels_small<R>, dim3((unsigned)((nrec + TXS_RPB - 1) / TXS_RPB)), dim3(256), 0, st, t, (int)nrec);
    HIPCHK(hipGetLastError());
    return 0;
  }
  const int per_env = (int)std::max<long long>(1, std::min<long long>(b->tax_slots / nrec, (b->ntax + 255) / 256));
  const int slice = ((b->ntax + per_env - 1) / per_env + 255) / 256 * 256;
  const dim3 tgrid((unsigned)nrec, (b->ntax + slice - 1) / slice);
  TaxArgs<R> t{b->dI, (const R*)b->dF, (const R*)b->dFenv, b->nfrec, (const R*)poseR, poseD, b->nspt, (R*)tac_out, slice, b->B, tac_slot,
               b->ntax, b->I[TSIM_IH_NSENSOR], b->I[TSIM_IH_FOFF_SENSOR], b->I[TSIM_IH_FOFF_PAIR], b->I[TSIM_IH_FOFF_TAXEL], b->tt_off};
  hipLaunchKernelGGL(k_taxels<R>, tgrid, dim3(256), 0, st, t);
  HIPCHK(hipGetLastError());
  return 0;
}

template <class R>
static int launch_forward(tsim_batch* b, const void* u, int nframes, const int32_t* tac_slot, int nsub, void* q_out, void* qd_out, void* var_out, void* tac_out, int32_t* status, hipStream_t st) {
  FwdArgs<R> a;
  a.I = b->dI; a.F = (const R*)b->dF; a.Fenv = (const R*)b->dFenv; a.fstride = b->nfrec; a.B = b->B; a.nsub = nsub; a.record = b->record; a.t0 = b->t_cur; a.nframes = nframes; a.tac_slot = tac_slot;
  a.tape = (R*)b->tape; a.u = (const R*)u;
  a.q_out = (R*)q_out; a.qd_out = (R*)qd_out; a.var_out = (R*)var_out; a.tac_out = (R*)tac_out; a.status = status; a.evals = b->evals; a.order = (b->B >= 256 && nframes == 1 && b->order_valid) ? b->order : (b->B >= 256 && nframes > 1 && b->order_ep_n == nframes * nsub && !b->ab_no_episode_lpt) ? b->order_ep : nullptr;
  a.prev = (double*)b->prev; a.has_prev = b->has_prev; a.stage_cpt = b->stage_cpt;
  a.cross_kinks = b->cross_kinks; a.eval_budget = b->eval_budget; a.gnorm = b->gnorm; a.cull = b->pair_cull; a.vo_ls = b->value_trials; a.vo_first = b->value_first; a.helpers = b->trial_helpers; a.helped = b->helped;
  const bool emit = pose_emit(b, st);
  a.nspt = b->nspt;
  if (emit) { a.poseR = (R*)b->poseR; a.poseD = b->poseD; }
  // The tactile frames of the launch: by k_taxels afterwards, from the pose records the launch leaves per frame — which is what lets the
  // slots of a wavefront run their frames independently (k_forward, main loop).  TSIM_INKERNEL_READOUT=1 / TSIM_NO_FREE_RUN=1: A/B.
  // (not when the per-frame pose records would take more than 1 GiB — 180-frame grasp episodes with 22 (sensor, primitive) combinations —: those
  // launches keep the in-kernel read-out)
  const size_t fpose_bytes = (size_t)nframes * b->B * (size_t)std::max(b->nspt, 1) * (TP_R_SIZE * b->esz + TP_D_SIZE * sizeof(double));
  bool defer = tac_out && taxels_supported(b) && fpose_bytes <= ((size_t)1 << 30) && !b->ab_inkernel_readout;
  if (defer && b->fpose_frames < nframes) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) defer = false;      // no allocation inside a capture: the in-kernel read-out
    else {
      HIPCHK(hipStreamSynchronize(st));          // an earlier launch may still be writing the old records
      // A HIP graph captured from this batch (pose_off) holds the OLD buffers' addresses in its kernel arguments — a replay after this
      // point would write freed memory: such buffers are retired (freed with the batch), not freed
      if (b->pose_off) { if (b->fposeR) b->retired.push_back(b->fposeR); if (b->fposeD) b->retired.push_back(b->fposeD); }
      else { (void)hipFree(b->fposeR); (void)hipFree(b->fposeD); }
      b->fposeR = nullptr; b->fposeD = nullptr; b->fpose_frames = 0;
      if (hipMalloc(&b->fposeR, (size_t)nframes * b->B * b->nspt * TP_R_SIZE * b->esz) != hipSuccess ||
          hipMalloc((void**)&b->fposeD, (size_t)nframes * b->B * b->nspt * TP_D_SIZE * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(b->fposeR); b->fposeR = nullptr; b->fposeD = nullptr; defer = false;
      } else b->fpose_frames = nframes;
    }
  }
  if (defer) { a.fposeR = (R*)b->fposeR; a.fposeD = b->fposeD; }
  a.free_run = (defer || !tac_out || b->ntax == 0) && !b->ab_no_free_run;
  a.lockstep = b->ab_lockstep ? 1 : 0;
  if (a.lockstep) a.free_run = 0;
  a.default_opts = default_options(b) ? 1 : 0;
  { KtScope kt_(b, TSIM_KT_FORWARD, st); TS_LAUNCH(k_forward, R, b, st, a); }
  HIPCHK(hipGetLastError());
  if (defer) { KtScope kt_(b, TSIM_KT_TAXELS, st); if (launch_taxels<R>(b, b->fposeR, b->fposeD, nframes, tac_slot, tac_out, st)) return 1; }
  b->pose_valid = emit ? 1 : 0;
  if (b->B >= 256) {
    const int ns = TS_WAVE / launch_shape(b).lpe, nsv = (b->B % ns == 0) ? ns : 1;
    if (nframes > 1) {          // episode totals: the order of the next episode launch of this length (kept across resets)
      const int n = nframes * nsub;
      hipLaunchKernelGGL(k_order_by_evals, dim3(1), dim3(1024), 0, st, (const int*)b->evals, b->order_ep, b->B, nsv, 2 * n, std::max(1, 5 * n / 2), b->ab_lpt_deal);
      b->order_ep_n = n; b->order_valid = 0;
    } else {
      hipLaunchKernelGGL(k_order_by_evals, dim3(1), dim3(1024), 0, st, (const int*)b->evals, b->order, b->B, nsv, 0, 64, b->ab_lpt_deal);
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
  a.lamq = (R*)b->lamq; a.lamv = (R*)b->lamv; a.df_du = (R*)df_du; a.stage_cpt = b->stage_cpt; a.cyc = b->bwd_stamps; a.cull = b->pair_cull;
  {
    KtScope kt_(b, TSIM_KT_BACKWARD, st);
    const int keep_ = b->lpe_forced;
    if (b->ab_bwd_lpe) b->lpe_forced = b->ab_bwd_lpe;      // A/B (TSIM_BWD_LPE at creation): another launch shape for the adjoint kernel (the tape does not depend on it)
    TS_LAUNCH(k_backward, R, b, st, a);
    b->lpe_forced = keep_;
  }
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
  { const std::vector<int32_t> S_ = build_sched(b->I, b->F); b->nsched = (int)S_.size(); b->tt_off = b->I[TSIM_IH_NI] + S_[TS_SCHED_TAXTAB]; }
  b->pair_cull = getenv("TSIM_NO_PAIR_CULL") ? 0 : 1;
  b->no_static = getenv("TSIM_NO_STATIC") != nullptr;
  b->trial_helpers = getenv("TSIM_NO_TRIAL_HELPERS") ? 0 : 1;
  b->ab_no_episode_lpt = getenv("TSIM_NO_EPISODE_LPT") != nullptr; b->ab_inkernel_readout = getenv("TSIM_INKERNEL_READOUT") != nullptr;
  b->ab_no_free_run = getenv("TSIM_NO_FREE_RUN") != nullptr; b->ab_lockstep = getenv("TSIM_LOCKSTEP") != nullptr; b->ab_no_default_opts = getenv("TSIM_NO_DEFAULT_OPTS") != nullptr;
  if (const char* e = getenv("TSIM_LPT_DEAL")) b->ab_lpt_deal = atoi(e);
  if (const char* e = getenv("TSIM_BWD_LPE")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) b->ab_bwd_lpe = v; }
  b->ab_taxels_per_record = getenv("TSIM_TAXELS_PER_RECORD") != nullptr; b->ab_no_envtab_cpt = getenv("TSIM_NO_ENVTAB_CPT") != nullptr;
  b->value_first = getenv("TSIM_NO_VALUE_FIRST") ? 0 : 1;
  if (const char* e = getenv("TSIM_VALUE_TRIALS")) b->value_trials = std::max(0, atoi(e));
  if (const char* e = getenv("TSIM_LPE")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) b->lpe_forced = v; }
  b->cross_kinks = dtype == TSIM_F32 ? 1 : 0;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    b->n_simd = 4 * cus;
  }
  b->stage_cpt = 0;
  detect_static_model(b);      // before the staging decision: the launch shape depends on which kernels run
  if (lds_bytes_for(b, 1) > 64 * 1024) { delete b; return fail("model needs more than 64 KiB of LDS per environment"); }
  decide_stage_cpt(b);
  b->dI = nullptr; b->dF = nullptr; b->tape = nullptr; b->lamq = nullptr; b->lamv = nullptr; b->evals = nullptr; b->order = nullptr; b->order_valid = 0; b->prev = nullptr; b->has_prev = 0; b->poseR = nullptr; b->poseD = nullptr; b->nspt = b->I[TSIM_IH_NSPRIM];
  size_t tape_bytes = (size_t)(tape_capacity + 1) * B * b->rec * b->esz;
  if (hipMalloc(&b->dI, (b->I.size() + b->nsched) * sizeof(int32_t)) != hipSuccess || hipMalloc(&b->dF, b->F.size() * b->esz) != hipSuccess ||
      hipMalloc(&b->tape, tape_bytes) != hipSuccess || hipMalloc(&b->lamq, (size_t)2 * B * nr * b->esz) != hipSuccess ||
      hipMalloc(&b->lamv, (size_t)2 * B * nr * b->esz) != hipSuccess || hipMalloc(&b->evals, (size_t)B * sizeof(int)) != hipSuccess || hipMalloc((void**)&b->helped, (size_t)B * sizeof(int)) != hipSuccess || hipMemset(b->helped, 0, (size_t)B * sizeof(int)) != hipSuccess || hipMalloc((void**)&b->gnorm, (size_t)B * sizeof(float)) != hipSuccess || hipMalloc(&b->order, (size_t)B * sizeof(int)) != hipSuccess || hipMalloc((void**)&b->order_ep, (size_t)B * sizeof(int)) != hipSuccess || hipMalloc(&b->prev, (size_t)B * 2 * nr * sizeof(double)) != hipSuccess ||
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
  for (void* p : b->retired) (void)hipFree(p);
  for (auto& k : b->kt) { (void)hipEventDestroy(k.a); (void)hipEventDestroy(k.b); }
  for (hipEvent_t e : b->kt_free) (void)hipEventDestroy(e);
  (void)hipFree(b->dFenv); (void)hipFree(b->dI); (void)hipFree(b->dF); (void)hipFree(b->tape); (void)hipFree(b->lamq); (void)hipFree(b->lamv); (void)hipFree(b->evals); (void)hipFree(b->helped); (void)hipFree(b->gnorm); (void)hipFree(b->order); (void)hipFree(b->order_ep); (void)hipFree(b->prev); (void)hipFree(b->poseR); (void)hipFree(b->poseD); (void)hipFree(b->fposeR); (void)hipFree(b->fposeD); (void)hipFree(b->dKmask); (void)hipFree(b->dFlag);
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
int tsim_kernel_timing(tsim_batch* b, int enable) { b->kt_on = enable ? 1 : 0; return 0; }
int tsim_kernel_times(tsim_batch* b, double* ms_sum, int32_t* launches) {
  if (!ms_sum || !launches) return fail("kernel_times: null argument");
  TS_DEVICE(b);
  for (int k = 0; k < TSIM_KT_COUNT; ++k) { ms_sum[k] = 0.0; launches[k] = 0; }
  for (auto& p : b->kt) {
    float ms = 0.f;
    HIPCHK(hipEventSynchronize(p.b));
    HIPCHK(hipEventElapsedTime(&ms, p.a, p.b));
    ms_sum[p.kind] += ms; launches[p.kind] += 1;
    b->kt_free.push_back(p.a); b->kt_free.push_back(p.b);
  }
  b->kt.clear();
  return 0;
}
int tsim_set_lanes_per_env(tsim_batch* b, int lanes) {
  if (lanes != 0 && lanes != 16 && lanes != 32 && lanes != 64) return fail("set_lanes_per_env: 0 (automatic), 16, 32 or 64");
  b->lpe_forced = lanes;
  b->order_valid = 0; b->order_ep_n = 0;
  decide_stage_cpt(b);      // depends on the shape; the flag travels with every launch as a kernel argument: nothing on the device to update
  return 0;
}
int tsim_static_model(const tsim_batch* b) { return kernel_mode(b) != TS_KM_GENERIC ? b->static_id : 0; }
const char* tsim_kernel_variant(const tsim_batch* b) {
  const int km = kernel_mode(b);
  if (km == TS_KM_STATIC) return "static:pusher";
  if (km == TS_KM_PARAM) return "param:pusher";
  return "generic";
}
int tsim_set_static(tsim_batch* b, int allow) {
  b->no_static = allow ? 0 : 1;
  b->order_valid = 0; b->order_ep_n = 0;
  decide_stage_cpt(b);      // the launch shape depends on which kernels run
  return 0;
}
int tsim_set_option(tsim_batch* b, int option, int value) {
  if (option == TSIM_OPT_PAIR_CULL) { b->pair_cull = value != 0; return 0; }
  if (option == TSIM_OPT_VALUE_TRIALS) { if (value < 0) return fail("set_option: TSIM_OPT_VALUE_TRIALS >= 0"); b->value_trials = value; return 0; }
  if (option == TSIM_OPT_TRIAL_HELPERS) { b->trial_helpers = value != 0; return 0; }
  if (option == TSIM_OPT_VALUE_FIRST) { b->value_first = value != 0; return 0; }
  return fail("set_option: unknown option " + std::to_string(option));
}
int tsim_get_option(const tsim_batch* b, int option) {
  if (option == TSIM_OPT_PAIR_CULL) return b->pair_cull;
  if (option == TSIM_OPT_VALUE_TRIALS) return b->value_trials;
  if (option == TSIM_OPT_TRIAL_HELPERS) return b->trial_helpers;
  if (option == TSIM_OPT_VALUE_FIRST) return b->value_first;
  if (option == TSIM_OPT_CROSS_KINKS) return b->cross_kinks;
  if (option == TSIM_OPT_EVAL_BUDGET) return b->eval_budget;
  if (option == TSIM_OPT_ALL_DEFAULT) return default_options(b) ? 1 : 0;
  return -1;
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
int tsim_last_helper_trials(tsim_batch* b, int32_t* host_out) {
  TS_DEVICE(b);
  if (hipMemcpy(host_out, b->helped, (size_t)b->B * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail("last_helper_trials: copy failed");
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
  if (b->dFenv) { HIPCHK(hipFree(b->dFenv)); b->dFenv = nullptr; b->env_struct_ok = 0; }     // per-environment tables refer to the old model
  if (int rc = upload_model(b, (hipStream_t)stream)) return rc;
  b->order_valid = 0; b->order_ep_n = 0;
  decide_stage_cpt(b);      // the edit may have moved the batch between the compiled-in and the generic kernels: another launch shape
  return 0;
}

int tsim_set_env_tables(tsim_batch* b, const void* tables, void* stream) {
  TS_DEVICE(b);
  pose_invalidate(b, (hipStream_t)stream);
  if (!tables) { if (b->dFenv) { HIPCHK(hipFree(b->dFenv)); b->dFenv = nullptr; b->env_struct_ok = 0; decide_stage_cpt(b); } return 0; }
  size_t bytes = (size_t)b->B * b->nfrec * b->esz;
  const bool fresh = !b->dFenv;
  if (fresh) HIPCHK(hipMalloc(&b->dFenv, bytes));
  HIPCHK(hipMemcpyAsync(b->dFenv, tables, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  // Time step, gravity and Newton tolerance are properties of the batch, not of an environment (the reference's randomisers never touch them, and
  // the kernels cache them per wavefront: a helper slot evaluating another environment's trial point would use its own): the header of every row
  // is overwritten with the shared model's.
  {
    const unsigned nh = (unsigned)((b->B * TSIM_FH_SIZE + 255) / 256);
    if (b->dtype == TSIM_F32) hipLaunchKernelGGL(k_env_header<float>, dim3(nh), dim3(256), 0, (hipStream_t)stream, (float*)b->dFenv, (const float*)b->dF, b->nfrec, b->B);
    else hipLaunchKernelGGL(k_env_header<double>, dim3(nh), dim3(256), 0, (hipStream_t)stream, (double*)b->dFenv, (const double*)b->dF, b->nfrec, b->B);
    HIPCHK(hipGetLastError());
  }
  // A batch whose model has a compiled-in structure stays on that instantiation if every environment's table keeps the structural floats
  // (the exact 0 / 1 / -1 entries the instantiation has folded away): checked here, on the device, once per call — one small kernel and a
  // 4-byte read-back (this call synchronises then; inside a stream capture the check is skipped and the batch takes the generic kernels).
  const int ok_before = b->env_struct_ok;
  b->env_struct_ok = 0;
  if (b->static_id != 0 && b->dtype == TSIM_F32) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
      if (!b->dKmask) {
        std::vector<unsigned char> km(b->nfrec);
        for (int i = 0; i < b->nfrec; ++i) km[i] = TsParam<TsStaticPusher>::Fk(i) ? 1 : 0;
        HIPCHK(hipMalloc((void**)&b->dKmask, km.size())); HIPCHK(hipMalloc((void**)&b->dFlag, sizeof(int)));
        HIPCHK(hipMemcpy(b->dKmask, km.data(), km.size(), hipMemcpyHostToDevice));
      }
      int flag = 0;
      HIPCHK(hipMemsetAsync(b->dFlag, 0, sizeof(int), (hipStream_t)stream));
      const size_t n = (size_t)b->B * b->nfrec;
      hipLaunchKernelGGL(k_check_structure, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)b->dFenv, (const float*)b->dF, b->dKmask, b->nfrec, n, b->dFlag);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(&flag, b->dFlag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
      HIPCHK(hipStreamSynchronize((hipStream_t)stream));
      b->env_struct_ok = flag == 0;
    }
  }
  if (fresh || ok_before != b->env_struct_ok) { b->order_valid = 0; b->order_ep_n = 0; decide_stage_cpt(b); }      // the block's LDS layout / the kernels change
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
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { b->pose_off = 1; b->pose_valid = 0; }
  }
  const bool fk = var_out || !tac || !b->pose_valid;      // the forward launch left the pose records of this state: k_taxels alone
  if (b->dtype == TSIM_F32) {
    ReadArgs<float> a{b->dI, (const float*)b->dF, (const float*)b->dFenv, b->nfrec, b->B, b->t_cur, (const float*)b->tape, (float*)var_out, tac ? (float*)b->poseR : nullptr, b->poseD, b->nspt, b->stage_cpt};
    if (fk) hipLaunchKernelGGL(k_readout<float>, dim3(b->B), dim3(TS_WAVE), lds_bytes_for(b, 1), st, a);
    if (tac) { KtScope kt_(b, TSIM_KT_TAXELS, st); if (launch_taxels<float>(b, b->poseR, b->poseD, 1, nullptr, tac_out, st)) return 1; }
  } else {
    ReadArgs<double> a{b->dI, (const double*)b->dF, (const double*)b->dFenv, b->nfrec, b->B, b->t_cur, (const double*)b->tape, (double*)var_out, tac ? (double*)b->poseR : nullptr, b->poseD, b->nspt, b->stage_cpt};
    if (fk) hipLaunchKernelGGL(k_readout<double>, dim3(b->B), dim3(TS_WAVE), lds_bytes_for(b, 1), st, a);
    if (tac) { KtScope kt_(b, TSIM_KT_TAXELS, st); if (launch_taxels<double>(b, b->poseR, b->poseD, 1, nullptr, tac_out, st)) return 1; }
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
    DbgArgs<float> a{b->dI, (const float*)b->dF, (const float*)b->dFenv, b->nfrec, b->B, (const float*)q1, (const float*)q0, (const float*)qd0, (const float*)u, (float*)g_out, (float*)H_out, cycles, b->stage_cpt, b->pair_cull};
    if (lpe == 16 && kernel_mode(b) == TS_KM_STATIC) ts_static_pusher_launch_debug(a, grid.x, lds, (hipStream_t)stream);      // the static sweep's g and H
    else if (lpe == 16 && kernel_mode(b) == TS_KM_PARAM) ts_param_pusher_launch_debug(a, grid.x, lds, (hipStream_t)stream);
    else if (lpe == 16) hipLaunchKernelGGL((k_debug_eval<float, 16>), grid, blk, lds, (hipStream_t)stream, a);
    else if (lpe == 32) hipLaunchKernelGGL((k_debug_eval<float, 32>), grid, blk, lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_debug_eval<float, 64>), grid, blk, lds, (hipStream_t)stream, a);
  } else {
    DbgArgs<double> a{b->dI, (const double*)b->dF, (const double*)b->dFenv, b->nfrec, b->B, (const double*)q1, (const double*)q0, (const double*)qd0, (const double*)u, (double*)g_out, (double*)H_out, cycles, b->stage_cpt, b->pair_cull};
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
