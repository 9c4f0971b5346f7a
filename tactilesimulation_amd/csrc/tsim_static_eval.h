// tsim_static_eval.h — the residual evaluation of a statically known model as ONE register-resident pass (included by tsim_eval.h).
//
// The generic evaluation is three phases that talk through LDS records: the link sweep leaves a value record per link and a tangent
// record per (link, direction); phase 2 stages each contact pair's pose (lanes = pairs) and per-direction 12-vectors (lanes = (pair,
// direction)) in LDS, runs the point loops, leaves the pairs' wrenches in LDS again and folds them into the links' records (lanes =
// directions); phase 3 reads those back leaf -> root.  Every hand-over is a store, a wait and a dependent load for a lone wavefront,
// and every phase re-derives who it is from the schedule.
// With the model's blob a compile-time constant (tsim_static.h) the level-parallel sweep already has every link's state in EVERY lane's
// registers (all lanes compute all values; lane k computes the tangent w.r.t. dof k).  From there on nothing needs LDS:
//   * a pair's pose in its primitive's frame is a function of two link states the lane holds — every lane computes it (the pair's
//     primitive frame, its links, its point range are constants; a world-fixed link folds to the identity);
//   * the 12-vector of (pair, direction k) is a function of that pose and of the lane's own tangents;
//   * the point loop (pair_points_matrix, the same code as the generic kernels') reduces to the 6 x 12 matrix in every lane; lane k
//     applies it to ITS 12-vector, brings the wrench and its tangent to the world frame and adds them to its per-link accumulators;
//   * the leaf -> root pass is a compile-time loop over the levels on those accumulators; the parents are constants.
// What still goes to LDS is the result — g and H in the forward kernel (the solve and the tape read them), nothing in the adjoint kernel
// (it takes H^T z and M z back in registers) — and, once per frame, the link value records the read-out reads (ts_static_value_records).
// An evaluation round of four environments (fine stamps, profiles/r04_static_model.md): 37.7 k -> 17.1 k cycles.
#pragma once

// pose of link A of pair PK in the pair's primitive frame (pair_stage_value) from the link states in registers; link 0 is the world.  Also
// returns the pair's float record as constants (rounded to R first: the device blob holds R), the primitive frame in the world (RP, pP), the
// relative twist there and this lane's twist tangents of the two links
template <class R, class MS, int PK>
__device__ __forceinline__ void ts_fused_pair_pose(const Ctx<R>& c, const TsLinkState<R>* st, R (&pf)[TSIM_PF_SIZE], M3<R>& RP, V3<R>& pP, PairPose<R>& P, S6<R>& Vrel, S6<R>& dVA, S6<R>& dVB) {
  constexpr int o = MS::Iv(TSIM_IH_OFF_PAIR) + PK * TSIM_PI_SIZE, fo = MS::Iv(TSIM_IH_FOFF_PAIR) + PK * TSIM_PF_SIZE;
  constexpr int la = MS::Iv(o + TSIM_PI_LINKA), lb = MS::Iv(o + TSIM_PI_LINKB);
#pragma unroll
  for (int e = 0; e < TSIM_PF_SIZE; ++e) pf[e] = ts_F<R, MS>(c, fo + e);
  M3<double> Rprim; V3<double> pprim;
#pragma unroll
  for (int e = 0; e < 9; ++e) Rprim.m[e] = (double)pf[TSIM_PF_R + e];
  pprim = mk3<double>((double)pf[TSIM_PF_P], (double)pf[TSIM_PF_P + 1], (double)pf[TSIM_PF_P + 2]);
  M3<double> RPd; V3<double> pPd;
  if constexpr (lb == 0) { RPd = Rprim; pPd = pprim; }
  else { RPd = mulMM(st[lb].Rd, Rprim); pPd = mulMv(st[lb].Rd, pprim) + st[lb].pd; }
  if constexpr (la == 0) { P.RPAd = mulMtM(RPd, M3<double>{{1, 0, 0, 0, 1, 0, 0, 0, 1}}); P.pPAd = mulMtv(RPd, zero3<double>() - pPd); }
  else { P.RPAd = mulMtM(RPd, st[la].Rd); P.pPAd = mulMtv(RPd, st[la].pd - pPd); }
  RP = cvtm<R>(RPd);
  pP = cvt3<R>(pPd);
  S6<R> VA = zero6<R>(), VB = zero6<R>();
  dVA = zero6<R>(); dVB = zero6<R>();
  if constexpr (la != 0) { VA = st[la].V; dVA = st[la].dV; }
  if constexpr (lb != 0) { VB = st[lb].V; dVB = st[lb].dV; }
  Vrel = to_frame(RP, pP, VA - VB);
  P.RPA = cvtm<R>(P.RPAd); P.pPA = cvt3<R>(P.pPAd); P.wrel = Vrel.a; P.vrel = Vrel.l;
}

// pair PK of the static model: pose, this lane's 12-vector, the point loop, the fold into the lane's per-link wrench accumulators
template <class R, int NRM, int LPE, class MS, int PK>
__device__ __forceinline__ void ts_fused_pair(const Ctx<R>& c, int lane, R sq, const TsLinkState<R>* st, const S6<R>& Wk, S6<R>* Fl, S6<R>* dFl, bool tang = true) {
  using T = TsTopo<MS>;
  constexpr int NP = MS::Iv(TSIM_IH_NPAIR);
  if constexpr (PK < NP) {
    constexpr int o = MS::Iv(TSIM_IH_OFF_PAIR) + PK * TSIM_PI_SIZE;
    constexpr int flags = MS::Iv(o + TSIM_PI_FLAGS), prim = MS::Iv(o + TSIM_PI_PRIM), npt = MS::Iv(o + TSIM_PI_NPT), pt0 = MS::Iv(o + TSIM_PI_PT0);
    constexpr int la = MS::Iv(o + TSIM_PI_LINKA), lb = MS::Iv(o + TSIM_PI_LINKB);
    if constexpr ((flags & 1) != 0) {
      const int k = lane;
      R pf[TSIM_PF_SIZE];
      M3<R> RP; V3<R> pP; PairPose<R> P; S6<R> Vrel, dVA, dVB;
      ts_fused_pair_pose<R, MS, PK>(c, st, pf, RP, pP, P, Vrel, dVA, dVB);
      // ---- this lane's direction (pair_stage_tangent, vmode 0): relative displacement and d(relative twist) in the primitive's frame
      constexpr int ancA = la != 0 ? T::li(la == 0 ? 1 : la, TSIM_LI_ANCMASK) : 0, ancB = lb != 0 ? T::li(lb == 0 ? 1 : lb, TSIM_LI_ANCMASK) : 0;
      const R inA = ((ancA >> k) & 1) ? R(1) : R(0), inB = ((ancB >> k) & 1) ? R(1) : R(0);
      // ---- lanes = contact points
      R w0[6], M[6][12];
      const bool any_hit = pair_points_matrix<R, LPE, prim, flags>(c, pt0, npt, prim, (flags & 2) != 0, pf, P, lane, w0, M, tang);
      TS_STAMP2(c);
      if (any_hit) {
        constexpr bool kHalfRow = npt <= 8 && NRM <= 8;      // all points (and all directions) in the first 8 lanes of the slot
        if constexpr (kHalfRow) {
#pragma unroll
          for (int e = 0; e < 6; ++e) w0[e] = half_row_sum(w0[e]);
        } else seg_sum_many<LPE, 6>(w0);
        const S6<R> Ww = wrench_to_world(RP, pP, mk6<R>(mk3<R>(w0[0], w0[1], w0[2]), mk3<R>(w0[3], w0[4], w0[5])));
        if constexpr (la != 0) Fl[la] = Fl[la] - Ww;      // link 0 (world-fixed general bodies) takes no wrench
        if constexpr (lb != 0) Fl[lb] = Fl[lb] + Ww;
        if (tang) {
          if constexpr (kHalfRow) {
#pragma unroll
            for (int e = 0; e < 6; ++e)
#pragma unroll
              for (int j = 0; j < 12; ++j) M[e][j] = half_row_sum(M[e][j]);
          } else {
#pragma unroll
            for (int e = 0; e < 6; ++e) seg_sum_many<LPE, 12>(M[e]);
          }
          TS_STAMP2(c);
          // ---- lanes = directions: (dn; dF) = M t, to the world frame, into the links (pair_fold)
          const S6<R> dxiP = to_frame(RP, pP, Wk * (sq * (inA - inB)));
          const S6<R> dxiB = to_frame(RP, pP, Wk * (sq * inB));
          const S6<R> dVrel = to_frame(RP, pP, dVA - dVB) - crm(dxiB, Vrel);
          const R t[12] = {dxiP.a.x, dxiP.a.y, dxiP.a.z, dxiP.l.x, dxiP.l.y, dxiP.l.z, dVrel.a.x, dVrel.a.y, dVrel.a.z, dVrel.l.x, dVrel.l.y, dVrel.l.z};
          R acc[6];
#pragma unroll
          for (int e = 0; e < 6; ++e) {
            R s_ = R(0);
#pragma unroll
            for (int j = 0; j < 12; ++j) s_ += M[e][j] * t[j];
            acc[e] = s_;
          }
          const S6<R> dWw = wrench_to_world(RP, pP, mk6<R>(mk3<R>(acc[0], acc[1], acc[2]), mk3<R>(acc[3], acc[4], acc[5]))) + crf(Wk * (sq * inB), Ww);
          if constexpr (la != 0) dFl[la] = dFl[la] - dWw;
          if constexpr (lb != 0) dFl[lb] = dFl[lb] + dWw;
        }
      } else TS_STAMP2(c);
      TS_STAMP2(c);
    }
    ts_fused_pair<R, NRM, LPE, MS, PK + 1>(c, lane, sq, st, Wk, Fl, dFl, tang);
  }
}

// joint-space forces of dof J (phase3_joint_space, with the dof's damping, limit and motor records as constants): adds to the value gj and
// returns the joint-space part of H[J][J]; every lane computes it (q, qd, u are broadcast reads)
template <class R, class MS, int J, int M>
__device__ __forceinline__ void ts_fused_motors(const Ctx<R>& c, R sq, R sv, R& gj, R& hjj) {
  if constexpr (M < MS::Iv(TSIM_IH_NU)) {
    constexpr int mo = MS::Iv(TSIM_IH_OFF_MOTOR) + M * TSIM_MI_SIZE, mfo = MS::Iv(TSIM_IH_FOFF_MOTOR) + M * TSIM_MF_SIZE;
    if constexpr (MS::Iv(mo + TSIM_MI_DOF) == J) {
      if constexpr (MS::Iv(mo + TSIM_MI_CTRL) == 0) {
        const R uc = fmin(fmax(c.u[M], R(-1)), R(1));
        gj -= ts_F<R, MS>(c, mfo + TSIM_MF_LO) + (uc + R(1)) * (R(0.5) * (ts_F<R, MS>(c, mfo + TSIM_MF_HI) - ts_F<R, MS>(c, mfo + TSIM_MF_LO)));
      } else {
        gj -= ts_F<R, MS>(c, mfo + TSIM_MF_P) * (c.u[M] - c.q[J]) - ts_F<R, MS>(c, mfo + TSIM_MF_D) * c.qd[J];
        hjj += ts_F<R, MS>(c, mfo + TSIM_MF_P) * sq + ts_F<R, MS>(c, mfo + TSIM_MF_D) * sv;
      }
    }
    ts_fused_motors<R, MS, J, M + 1>(c, sq, sv, gj, hjj);
  }
}
template <class R, class MS, int J>
__device__ __forceinline__ R ts_fused_joint_space(const Ctx<R>& c, R sq, R sv, R& gj) {
  constexpr int dfo = MS::Iv(TSIM_IH_FOFF_DOF) + J * TSIM_DF_SIZE;
  const R damping = ts_F<R, MS>(c, dfo + TSIM_DF_DAMPING), lk = ts_F<R, MS>(c, dfo + TSIM_DF_LIM_K), lo = ts_F<R, MS>(c, dfo + TSIM_DF_LIM_LO), hi = ts_F<R, MS>(c, dfo + TSIM_DF_LIM_HI);
  R hjj = R(0);
  gj += damping * c.qd[J]; hjj += damping * sv;
  constexpr bool lim_const = ts_is_const<MS>(dfo + TSIM_DF_LIM_K);      // a structure-static model with a limit stiffness in its records: a run-time test
  if constexpr (!lim_const || MS::Fv(dfo + TSIM_DF_LIM_K) > 0.0) {
    if (lim_const || lk > R(0)) {
      if (c.q[J] < lo) { gj -= lk * (lo - c.q[J]); hjj += lk * sq; }
      else if (c.q[J] > hi) { gj += lk * (c.q[J] - hi); hjj += lk * sq; }
    }
  }
  ts_fused_motors<R, MS, J, 0>(c, sq, sv, gj, hjj);
  return hjj;
}

// dof JJ of link LINK in the leaf -> root pass: tau_j = W_j . F, its tangent (column k of H, lane k), the joint-space forces
// ADJ: the adjoint kernel's use — nothing is stored; lane k accumulates yq_k = sum_j z_j H[j][k] over its column as it is produced
template <class R, class MS, int LINK, int JJ, bool ADJ>
__device__ __forceinline__ void ts_fused_dof(const Ctx<R>& c, int k, R sq, R sv, R h2, R mv, const S6<R>& Wj, const S6<R>& Wk, const S6<R>& F, const S6<R>& dF, const R* zr, R& yq, bool tang) {
  using T = TsTopo<MS>;
  constexpr int k0 = T::li(LINK, TSIM_LI_DOF0), ndj = T::li(LINK, TSIM_LI_NDOF), nr = T::NR;
  if constexpr (JJ < ndj) {
    constexpr int j = k0 + JJ;
    R gj = dot6(Wj, F);
    const R hjj = ts_fused_joint_space<R, MS, j>(c, sq, sv, gj);      // damping, limits, motors of this dof
    if (ADJ || tang) {
      const R dtau = dot6(Wj, dF) + mv * dot6(crm(Wk, Wj), F);
      const R Hjk = dtau * h2 + (k == j ? hjj : R(0)) * h2;      // columns are scaled by 1 / ca (g = r / ca)
      if constexpr (ADJ) yq += zr[j] * Hjk;
      else if (k < nr) c.H[j * nr + k] = Hjk;
    }
    if constexpr (!ADJ) { if (k == 0) c.g[j] = gj * h2; }
  }
}

// leaf -> root over the lane's accumulators: tau_j = W_j . F_subtree(link(j)), column k of H, the value g (lane 0), a link's subtree
// wrench into its parent's — links of level LEVEL, then the levels above.  The joint-space forces (damping, limits, motors: constants of the
// dof) go in right here: every lane has the value g_j, lane j adds its diagonal entry
template <class R, class MS, bool ADJ, int LEVEL, int LINK>
__device__ __forceinline__ void ts_fused_up_links(const Ctx<R>& c, int lane, R sq, R sv, R h2, const TsLinkTmp<R>* tmp, const S6<R>& Wk, S6<R>* Fl, S6<R>* dFl, const R* zr, R& yq, bool tang) {
  using T = TsTopo<MS>;
  if constexpr (LINK <= T::NL) {
    if constexpr (TsLevels<MS>::depth(LINK) == LEVEL) {
      constexpr int i = LINK, par = T::li(i, TSIM_LI_PARENT), ancm = T::li(i, TSIM_LI_ANCMASK);
      const int k = lane;
      const S6<R> F = Fl[i];
      S6<R> dF = zero6<R>();
      if (tang) dF = dFl[i];
      const S6<R> Wj[3] = {tmp[i].Wj0, tmp[i].Wj1, tmp[i].Wj2};
      const R mv = ((ancm >> k) & 1) ? sq : R(0);          // does dof k move link i (its own joint's dofs included)
      ts_fused_dof<R, MS, LINK, 0, ADJ>(c, k, sq, sv, h2, mv, Wj[0], Wk, F, dF, zr, yq, tang);
      ts_fused_dof<R, MS, LINK, 1, ADJ>(c, k, sq, sv, h2, mv, Wj[1], Wk, F, dF, zr, yq, tang);
      ts_fused_dof<R, MS, LINK, 2, ADJ>(c, k, sq, sv, h2, mv, Wj[2], Wk, F, dF, zr, yq, tang);
      if constexpr (par != 0) { Fl[par] = Fl[par] + F; if (tang) dFl[par] = dFl[par] + dF; }
    }
    ts_fused_up_links<R, MS, ADJ, LEVEL, LINK + 1>(c, lane, sq, sv, h2, tmp, Wk, Fl, dFl, zr, yq, tang);
  }
}
template <class R, class MS, bool ADJ, int LEVEL>
__device__ __forceinline__ void ts_fused_up(const Ctx<R>& c, int lane, R sq, R sv, R h2, const TsLinkTmp<R>* tmp, const S6<R>& Wk, S6<R>* Fl, S6<R>* dFl, const R* zr, R& yq, bool tang = true) {
  if constexpr (LEVEL >= 0) {
    ts_fused_up_links<R, MS, ADJ, LEVEL, 1>(c, lane, sq, sv, h2, tmp, Wk, Fl, dFl, zr, yq, tang);
    ts_fused_up<R, MS, ADJ, LEVEL - 1>(c, lane, sq, sv, h2, tmp, Wk, Fl, dFl, zr, yq, tang);
  }
}
template <class R, class MS, int LINK>
__device__ __forceinline__ void ts_fused_link_wrenches(const TsLinkState<R>* st, const TsLinkTmp<R>* tmp, S6<R>* Fl) {
  if constexpr (LINK <= TsTopo<MS>::NL) {
    Fl[LINK] = tmp[LINK].IA + crf(st[LINK].V, tmp[LINK].h);      // the value record's LK_FN (ts_l_store_values)
    ts_fused_link_wrenches<R, MS, LINK + 1>(st, tmp, Fl);
  }
}

// phases 1 - 3 of one evaluation (what evaluate() runs between setting q / qd / qa and returning g, H).  RECORDS: the link value records
// (with the COM / inertia entries), the joint screws and the twist tangents are left in LDS as well, for code that reads them there (A/B
// only: the forward kernel's read-out needs the value records of a frame's FINAL state only and writes them then, ts_static_value_records —
// 105 LDS store instructions less in every evaluation round; the adjoint kernel has its own pass, evaluate_static_fused_adjoint, and
// ts_static_output_vjp for the seeded sub-steps: no records at all).
template <class R, int NRM, int LPE, class MS, bool RECORDS>
__device__ __forceinline__ void evaluate_static_fused(const Ctx<R>& c, int lane, R sq, R sv, R sa, bool tang = true) {
  using T = TsTopo<MS>;
  static_assert(!T::has_exp() && T::NR <= 16, "static sweep: no rotation-vector joint, at most 16 dofs");
  static_assert(MS::Iv(TSIM_IH_NPAIR) <= 8, "fused static evaluation: the pairs are unrolled");
  TS_SYNC();
  TsLinkState<R> st[T::NL + 1];
  TsLinkTmp<R> tmp[T::NL + 1];
  S6<R> Wk = zero6<R>(), dFl[T::NL + 1], Fl[T::NL + 1];
  ts_l_level<R, MS, true, RECORDS, RECORDS ? 1 : 0, 0>(c, lane, sq, sv, sa, st, tmp, Wk, dFl, tang);
  ts_fused_link_wrenches<R, MS, 1>(st, tmp, Fl);
  TS_STAMP(c);
  ts_fused_pair<R, NRM, LPE, MS, 0>(c, lane, sq, st, Wk, Fl, dFl, tang);
  TS_STAMP(c);
  const R h2 = R(1) / c.ca;      // g = r / ca  (BDF1: h^2 r)
  R yq_unused = R(0);
  ts_fused_up<R, MS, false, TsLevels<MS>::max_depth()>(c, lane, sq, sv, h2, tmp, Wk, Fl, dFl, nullptr, yq_unused, tang);      // ... with the joint-space forces of each dof
  TS_SYNC();
  TS_STAMP2(c);
}

// (M z)_j = sum_i J_i^T I_i J_i z in registers (the generic kernels' mass_times_z goes through LDS twice): root -> leaf the motion
// A_i = sum over the dofs above link i of W_k z_k and the link's f_i = I_i A_i; leaf -> root the subtree sums and tau_j = W_j . f_subtree(link(j))
template <class R, class MS, int LEVEL, int LINK>
__device__ __forceinline__ void ts_mz_down_links(const TsLinkTmp<R>* tmp, const R* zr, S6<R>* A, S6<R>* f) {
  using T = TsTopo<MS>;
  if constexpr (LINK <= T::NL) {
    if constexpr (TsLevels<MS>::depth(LINK) == LEVEL) {
      constexpr int i = LINK, par = T::li(i, TSIM_LI_PARENT), k0 = T::li(i, TSIM_LI_DOF0), ndj = T::li(i, TSIM_LI_NDOF);
      S6<R> Ai = zero6<R>();
      if constexpr (par != 0) Ai = A[par];
      if constexpr (ndj > 0) Ai = Ai + tmp[i].Wj0 * zr[k0];
      if constexpr (ndj > 1) Ai = Ai + tmp[i].Wj1 * zr[k0 + 1];
      if constexpr (ndj > 2) Ai = Ai + tmp[i].Wj2 * zr[k0 + 2];
      A[i] = Ai;
      f[i] = imul(tmp[i].mass, tmp[i].cw, tmp[i].Ic, Ai);
    }
    ts_mz_down_links<R, MS, LEVEL, LINK + 1>(tmp, zr, A, f);
  }
}
template <class R, class MS, int LEVEL>
__device__ __forceinline__ void ts_mz_down(const TsLinkTmp<R>* tmp, const R* zr, S6<R>* A, S6<R>* f) {
  if constexpr (LEVEL <= TsLevels<MS>::max_depth()) { ts_mz_down_links<R, MS, LEVEL, 1>(tmp, zr, A, f); ts_mz_down<R, MS, LEVEL + 1>(tmp, zr, A, f); }
}
template <class R, class MS, int LEVEL, int LINK>
__device__ __forceinline__ void ts_mz_up_links(int lane, const TsLinkTmp<R>* tmp, S6<R>* f, R& ym) {
  using T = TsTopo<MS>;
  if constexpr (LINK <= T::NL) {
    if constexpr (TsLevels<MS>::depth(LINK) == LEVEL) {
      constexpr int i = LINK, par = T::li(i, TSIM_LI_PARENT), k0 = T::li(i, TSIM_LI_DOF0), ndj = T::li(i, TSIM_LI_NDOF);
      if constexpr (ndj > 0) { const R t_ = dot6(tmp[i].Wj0, f[i]); ym = lane == k0 ? t_ : ym; }
      if constexpr (ndj > 1) { const R t_ = dot6(tmp[i].Wj1, f[i]); ym = lane == k0 + 1 ? t_ : ym; }
      if constexpr (ndj > 2) { const R t_ = dot6(tmp[i].Wj2, f[i]); ym = lane == k0 + 2 ? t_ : ym; }
      if constexpr (par != 0) f[par] = f[par] + f[i];
    }
    ts_mz_up_links<R, MS, LEVEL, LINK + 1>(lane, tmp, f, ym);
  }
}
template <class R, class MS, int LEVEL>
__device__ __forceinline__ void ts_mz_up(int lane, const TsLinkTmp<R>* tmp, S6<R>* f, R& ym) {
  if constexpr (LEVEL >= 0) { ts_mz_up_links<R, MS, LEVEL, 1>(lane, tmp, f, ym); ts_mz_up<R, MS, LEVEL - 1>(lane, tmp, f, ym); }
}

// The adjoint kernel's evaluation at a taped state (seeds (1, 0, 0): H = dr/dq / ca) with c.z known: nothing is stored — lane k returns
//   yq = (H^T z)_k  and  ym = (M z)_k ,  which is all the kernel uses of it.
template <class R, int NRM, int LPE, class MS>
__device__ __forceinline__ void evaluate_static_fused_adjoint(const Ctx<R>& c, int lane, R& yq, R& ym) {
  using T = TsTopo<MS>;
  TS_SYNC();
  TsLinkState<R> st[T::NL + 1];
  TsLinkTmp<R> tmp[T::NL + 1];
  S6<R> Wk = zero6<R>(), dFl[T::NL + 1], Fl[T::NL + 1];
  ts_l_level<R, MS, true, false, 0, 0>(c, lane, R(1), R(0), R(0), st, tmp, Wk, dFl);
  ts_fused_link_wrenches<R, MS, 1>(st, tmp, Fl);
  TS_STAMP(c);
  ts_fused_pair<R, NRM, LPE, MS, 0>(c, lane, R(1), st, Wk, Fl, dFl);
  TS_STAMP(c);
  R zr[T::NR];
#pragma unroll
  for (int j = 0; j < T::NR; ++j) zr[j] = c.z[j];
  const R h2 = R(1) / c.ca;
  yq = R(0);
  ts_fused_up<R, MS, true, TsLevels<MS>::max_depth()>(c, lane, R(1), R(0), h2, tmp, Wk, Fl, dFl, zr, yq);
  TS_STAMP(c);
  S6<R> A[T::NL + 1], f[T::NL + 1];
  ym = R(0);
  ts_mz_down<R, MS, 0>(tmp, zr, A, f);
  ts_mz_up<R, MS, TsLevels<MS>::max_depth()>(lane, tmp, f, ym);
  TS_SYNC();
}

// the link value records and joint screws of the state in c.q / c.qd / c.qa (the last evaluation's), for the code that reads them from LDS:
// the forward kernel's read-out at the end of a frame
template <class R, class MS>
__device__ __forceinline__ void ts_static_value_records(const Ctx<R>& c, int lane) {
  using T = TsTopo<MS>;
  TS_SYNC();
  TsLinkState<R> st[T::NL + 1];
  TsLinkTmp<R> tmp[T::NL + 1];
  S6<R> Wk = zero6<R>(), dFl[T::NL + 1];
  ts_l_level<R, MS, false, false, 1, 0>(c, lane, R(0), R(0), R(0), st, tmp, Wk, dFl);
  TS_SYNC();
}
