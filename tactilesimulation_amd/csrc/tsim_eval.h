// tsim_eval.h — one residual evaluation g(q1) with its exact tangents, and the small dense solve.
// See tsim_device.h for the execution model.  Reference counterpart: the per-sub-step Newton body behind
// `sim.forward()` (envs/redmax_torch_functions.py:132); formulation in DESIGN.md §1.
//
// Tangents are propagated analytically with world-frame spatial algebra (constant joint-frame twists S):
//   d(pose of link i)/d q_k   = W_k                        (as a spatial displacement)  if dof k is at or above link i
//   d W_j / d q_k             = W_k x W_j                  (spatial cross product)       if dof k is at or above link(j)
//   d V_i, d A_i              by the recursion that defines V_i, A_i
//   d(I_i) m                  = dxi x* (I m) - I (dxi x m)
// and are verified against the oracle's dual-number Jacobian to round-off (tests/test_gpu_parity.py).
#pragma once
#include "tsim_device.h"
#include "tsim_static.h"
#include <type_traits>

// ================================================================================================ phase 1 (+ 1t)
// One root -> leaf sweep does both the values and, on lanes k < nr, the tangents w.r.t. dof k (seeds: q_k += eps*sq,
// qd_k += eps*sv, qdd_k += eps*sa).  Lane k walks the links of the root branch of its dof k only (ts_sched in
// tsim_device.h): the branches advance together, one link per step; every lane of a branch computes that link's values
// (the branch's leader lane stores them), its own tangent, and carries the parent's state in registers when the parent
// is the link it processed in the previous step (chains), so the common case has no LDS round trip.
template <class R, bool TANGENT, bool EXPJ>
__device__ __forceinline__ void phase1(const Ctx<R>& c, int lane, R sq, R sv, R sa, bool tang = true) {
  const int k = lane, nd = c.nd;
  const bool act = TANGENT && tang && lane < c.nr;      // tang (wave-uniform): false = values only, see evaluate()
  const int* S = c.LI;
  const int nsteps = ts_u(S[1]), rec0 = TS_SCHED_ENT + nsteps * 16;
  const int l16 = lane & 15;
  const int mybranch = lane < 16 ? S[TS_SCHED_BRANCH + l16] : -1;
  if (act) {                                         // wrench tangents of the links outside this lane's branch start at zero
    for (int i = 1; i <= c.nl; ++i)                  // (contacts couple branches: phase 2 adds to them)
      if (S[rec0 + (i - 1) * TS_LR_SIZE + TS_LR_BRANCH] != mybranch) st6(c.DT + (i * nd + k) * DT_SIZE + DT_FN, zero6<R>());
  }
  M3<double> pRd; V3<double> ppd;                    // pose of the link this lane processed in the previous step (double)
  S6<R> pV, pA, pdV, pdA;                            // ... its twist, acceleration and their tangents
  S6<R> Wk = zero6<R>();                             // twist column of this lane's own dof, once its link has been swept
  pRd = ldm(c.LPd); ppd = zero3<double>(); pV = zero6<R>(); pA = zero6<R>(); pdV = zero6<R>(); pdA = zero6<R>();
  int prev = 0;
  for (int st = 0; st < nsteps; ++st) {
    TS_SYNC();                                 // value records stored by the leaders in the previous step
    const int ent = lane < 16 ? S[TS_SCHED_ENT + st * 16 + l16] : 0;
    const int i = ent & 0xff;
    const bool leader = (ent >> 8) != 0;
    if (i == 0) continue;
    const int* li = S + rec0 + (i - 1) * TS_LR_SIZE;
    const R* lf = c.F + c.foff_link + (i - 1) * TSIM_LF_SIZE;
    const int par = li[TS_LR_PARENT], jt = li[TS_LR_JTYPE], k0 = li[TS_LR_DOF0], ndj = li[TS_LR_NDOF], ancm = li[TS_LR_ANCMASK];
    R* X = c.LP + i * LK_SIZE;
    M3<double> PRd; V3<double> Ppd; S6<R> PV, PA, PdV = zero6<R>(), PdA = zero6<R>();
    if (par == prev && par > 0) { PRd = pRd; Ppd = ppd; PV = pV; PA = pA; PdV = pdV; PdA = pdA; }
    else {
      const R* P = c.LP + par * LK_SIZE;
      PRd = ldm(c.LPd + par * 12); Ppd = ldv(c.LPd + par * 12 + 9); PV = ld6(P + LK_W); PA = ld6(P + LK_AW);
      if (act && par > 0) { const R* Dp = c.DT + (par * nd + k) * DT_SIZE; PdV = ld6(Dp + DT_VW); PdA = ld6(Dp + DT_AW); }
    }
    TS_STAMP2(c);
    // pose chain in double (Ctx): R0d = joint frame before the joint motion, (XRd, Xpd) = link frame
    const M3<double> R0d = mulMM(PRd, ldm_as<double>(lf + TSIM_LF_R));
    V3<double> Xpd = mulMv(PRd, ldv_as<double>(lf + TSIM_LF_P)) + Ppd;
    M3<double> XRd = R0d;
    const M3<R> R0 = cvtm<R>(R0d);
    V3<R> Xp = cvt3<R>(Xpd);
    M3<R> XR = R0;
    const R* ax = lf + TSIM_LF_AXES;
    S6<R> VJ = zero6<R>(), AJ = zero6<R>();
    S6<R> Wj[3];                                     // twist columns of this joint (<= 3 dofs on the HIP path)
    // rotation-vector joint only: Bvec = (b; p x b), b = R0 (d/dt J_l) thdot, and what lane k needs for its tangents
    const bool is_exp = EXPJ && jt == TSIM_J_SPHERICAL_EXP;      // EXPJ = false compiles the whole branch away
    S6<R> Bvec = zero6<R>(), dWexp[3], dBexp = zero6<R>();
    dWexp[0] = dWexp[1] = dWexp[2] = zero6<R>();
    if (is_exp) {
      typedef Jet<R, 1> J1;
      typedef Jet<J1, 3> J3;
      J3 t[3], s1, s2, c2, JL[9];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        t[m].v.v = c.q[k0 + m]; t[m].v.d[0] = c.qd[k0 + m];
#pragma unroll
        for (int n = 0; n < 3; ++n) { t[m].d[n].v = (m == n) ? R(1) : R(0); t[m].d[n].d[0] = R(0); }
      }
      so3_coeffs<J3>(t[0], t[1], t[2], s1, s2, c2);
      so3_mat<J3>(t[0], t[1], t[2], s2, c2, JL);             // J_l with d/dtheta_n (outer) and d/ds along thdot (inner)
      R E[9];
      so3_mat<R>(c.q[k0], c.q[k0 + 1], c.q[k0 + 2], s1.v.v, s2.v.v, E);
      M3<R> Em;
#pragma unroll
      for (int e = 0; e < 9; ++e) Em.m[e] = E[e];
      XRd = mulMM(R0d, cvtm<double>(Em));            // the rotation-vector exponential itself is evaluated in R
      XR = cvtm<R>(XRd);
      const V3<R> thd = mk3<R>(c.qd[k0], c.qd[k0 + 1], c.qd[k0 + 2]);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const V3<R> a = mulMv(R0, mk3<R>(JL[m].v.v, JL[3 + m].v.v, JL[6 + m].v.v));
        Wj[m] = mk6<R>(a, cross3(Xp, a));
        VJ = VJ + Wj[m] * c.qd[k0 + m]; AJ = AJ + Wj[m] * c.qa[k0 + m];
      }
      {
        const V3<R> b0 = mk3<R>(JL[0].v.d[0] * thd.x + JL[1].v.d[0] * thd.y + JL[2].v.d[0] * thd.z,
                                JL[3].v.d[0] * thd.x + JL[4].v.d[0] * thd.y + JL[5].v.d[0] * thd.z,
                                JL[6].v.d[0] * thd.x + JL[7].v.d[0] * thd.y + JL[8].v.d[0] * thd.z);
        const V3<R> b = mulMv(R0, b0);
        Bvec = mk6<R>(b, cross3(Xp, b));
      }
      // d W_m / d theta_kk for all (m, kk): stored for phase 3 by lane 0, kept in registers by the lane that owns kk
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        S6<R> dWm[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const V3<R> bm = mulMv(R0, mk3<R>(JL[m].d[kk].v, JL[3 + m].d[kk].v, JL[6 + m].d[kk].v));
          dWm[m] = mk6<R>(bm, cross3(Xp, bm));
          if (leader) st6(c.expw + (m * 3 + kk) * 6, dWm[m]);
        }
        if (k == k0 + kk) {
          dWexp[0] = dWm[0]; dWexp[1] = dWm[1]; dWexp[2] = dWm[2];
          // d b0 / d theta_kk (thdot fixed) and d b0 / d thdot_kk
          const V3<R> Cq = mk3<R>(JL[0].d[kk].d[0] * thd.x + JL[1].d[kk].d[0] * thd.y + JL[2].d[kk].d[0] * thd.z,
                                  JL[3].d[kk].d[0] * thd.x + JL[4].d[kk].d[0] * thd.y + JL[5].d[kk].d[0] * thd.z,
                                  JL[6].d[kk].d[0] * thd.x + JL[7].d[kk].d[0] * thd.y + JL[8].d[kk].d[0] * thd.z);
          const V3<R> Cv = mk3<R>(JL[0].d[kk].v * thd.x + JL[1].d[kk].v * thd.y + JL[2].d[kk].v * thd.z + JL[kk].v.d[0],
                                  JL[3].d[kk].v * thd.x + JL[4].d[kk].v * thd.y + JL[5].d[kk].v * thd.z + JL[3 + kk].v.d[0],
                                  JL[6].d[kk].v * thd.x + JL[7].d[kk].v * thd.y + JL[8].d[kk].v * thd.z + JL[6 + kk].v.d[0]);
          const V3<R> db = mulMv(R0, Cq * sq + Cv * sv);
          dBexp = mk6<R>(db, cross3(Xp, db));
        }
      }
    } else if (jt == TSIM_J_REVOLUTE) {
      double s, co; t_sincos_d(c.qD[k0], s, co);
      const double t = 1.0 - co;
      const double a0 = (double)ax[0], a1 = (double)ax[1], a2 = (double)ax[2];
      M3<double> Q;
      Q.m[0] = t * a0 * a0 + co;       Q.m[1] = t * a0 * a1 - s * a2;  Q.m[2] = t * a0 * a2 + s * a1;
      Q.m[3] = t * a0 * a1 + s * a2;   Q.m[4] = t * a1 * a1 + co;      Q.m[5] = t * a1 * a2 - s * a0;
      Q.m[6] = t * a0 * a2 - s * a1;   Q.m[7] = t * a1 * a2 + s * a0;  Q.m[8] = t * a2 * a2 + co;
      XRd = mulMM(R0d, Q);
      XR = cvtm<R>(XRd);
      const V3<R> a = mulMv(R0, ldv(ax));          // the axis is invariant under its own rotation
      Wj[0] = mk6<R>(a, cross3(Xp, a)); Wj[1] = zero6<R>(); Wj[2] = zero6<R>();
      VJ = Wj[0] * c.qd[k0]; AJ = Wj[0] * c.qa[k0];
    } else {   // prismatic / planar / translational
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        Wj[kk] = zero6<R>();
        if (kk < ndj) {
          const R e[3] = {kk == 0 ? R(1) : R(0), kk == 1 ? R(1) : R(0), kk == 2 ? R(1) : R(0)};
          const V3<R> a = mulMv(R0, ldv(jt == TSIM_J_TRANSLATIONAL ? e : ax + 3 * kk));
          Xpd = Xpd + mulMv(R0d, ldv_as<double>(jt == TSIM_J_TRANSLATIONAL ? e : ax + 3 * kk)) * c.qD[k0 + kk];
          Xp = cvt3<R>(Xpd);
          Wj[kk] = mk6<R>(zero3<R>(), a);
          VJ = VJ + Wj[kk] * c.qd[k0 + kk]; AJ = AJ + Wj[kk] * c.qa[k0 + kk];
        }
      }
    }
    TS_STAMP2(c);
    const S6<R> V = PV + VJ;
    const S6<R> A = PA + AJ + crm(V, VJ) + Bvec;
    const V3<R> cw = mulMv(XR, ldv(lf + TSIM_LF_COM)) + Xp;
    // world rotational inertia  Ic = XR Il XR^T  (symmetric, 6 entries)
    const R* il = lf + TSIM_LF_INERTIA;
    M3<R> T;        // T = XR * Il
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      T.m[3 * r + 0] = XR.m[3 * r] * il[0] + XR.m[3 * r + 1] * il[3] + XR.m[3 * r + 2] * il[4];
      T.m[3 * r + 1] = XR.m[3 * r] * il[3] + XR.m[3 * r + 1] * il[1] + XR.m[3 * r + 2] * il[5];
      T.m[3 * r + 2] = XR.m[3 * r] * il[4] + XR.m[3 * r + 1] * il[5] + XR.m[3 * r + 2] * il[2];
    }
    R Ic[6];
    Ic[0] = T.m[0] * XR.m[0] + T.m[1] * XR.m[1] + T.m[2] * XR.m[2];
    Ic[1] = T.m[3] * XR.m[3] + T.m[4] * XR.m[4] + T.m[5] * XR.m[5];
    Ic[2] = T.m[6] * XR.m[6] + T.m[7] * XR.m[7] + T.m[8] * XR.m[8];
    Ic[3] = T.m[0] * XR.m[3] + T.m[1] * XR.m[4] + T.m[2] * XR.m[5];
    Ic[4] = T.m[0] * XR.m[6] + T.m[1] * XR.m[7] + T.m[2] * XR.m[8];
    Ic[5] = T.m[3] * XR.m[6] + T.m[4] * XR.m[7] + T.m[5] * XR.m[8];
    const R mass = lf[TSIM_LF_MASS];
    const S6<R> h = imul(mass, cw, Ic, V), IA = imul(mass, cw, Ic, A);
    TS_STAMP2(c);
    if (leader) {
      stm(c.LPd + i * 12, XRd); stv(c.LPd + i * 12 + 9, Xpd);
      stm(X + LK_R, XR); stv(X + LK_P, Xp); st6(X + LK_W, V); st6(X + LK_AW, A); st6(X + LK_FN, IA + crf(V, h));
      stv(X + LK_C, cw);
#pragma unroll
      for (int e = 0; e < 6; ++e) X[LK_IC + e] = Ic[e];
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) if (kk < ndj) st6(c.WP + (k0 + kk) * 6, Wj[kk]);
    }
    TS_STAMP2(c);
    S6<R> dV = zero6<R>(), dA = zero6<R>();
    if (act) {
      R* D = c.DT + (i * nd + k) * DT_SIZE;
      S6<R> dF = zero6<R>();
      if ((ancm >> k) & 1) {                         // dof k moves link i
        // W_k: this joint's column, or an ancestor's column saved in this lane's registers when its link was swept
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) if (k == k0 + kk && kk < ndj) Wk = Wj[kk];
        const S6<R> dxi = Wk * sq;                                  // displacement of link i
        // joint part: dW_j = dxi x W_j ;  d(VJ) = sum dW_j qd_j + W_k sv ;  d(AJ) = sum dW_j qdd_j + W_k sa
        S6<R> dVJ = zero6<R>(), dAJ = zero6<R>();
        const bool own = k >= k0 && k < k0 + ndj;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          if (kk < ndj) {
            const S6<R> dW = (is_exp && own) ? dWexp[kk] * sq : crm(dxi, Wj[kk]);
            dVJ = dVJ + dW * c.qd[k0 + kk]; dAJ = dAJ + dW * c.qa[k0 + kk];
          }
        }
        if (own) { dVJ = dVJ + Wk * sv; dAJ = dAJ + Wk * sa; }
        dV = PdV + dVJ;
        dA = PdA + dAJ + crm(dV, VJ) + crm(V, dVJ);
        if (is_exp) dA = dA + (own ? dBexp : crm(dxi, Bvec));
        // inertial wrench  F = I A + V x* (I V);  d(I m) = dxi x* (I m) - I (dxi x m) + I dm
        const S6<R> dh = crf(dxi, h) + imul(mass, cw, Ic, dV - crm(dxi, V));
        dF = crf(dxi, IA) + imul(mass, cw, Ic, dA - crm(dxi, A)) + crf(dV, h) + crf(V, dh);
      }
      st6(D + DT_VW, dV); st6(D + DT_AW, dA); st6(D + DT_FN, dF);
    }
    pRd = XRd; ppd = Xpd; pV = V; pA = A; pdV = dV; pdA = dA; prev = i;
    TS_STAMP2(c);
  }
  TS_SYNC();
}

// ================================================================================================ staged pairs
// value record of pair pk in slot: pose of A in the primitive frame, relative twist (A w.r.t. B) in that frame.
// pk may differ between lanes (lanes = pairs in phase 2); `store`: this lane writes the record.
// Returns whether some contact point of the pair MAY touch the primitive: false only if the bounding sphere of the pair's points (link-A
// frame, ts_pair_bound) is farther from the primitive than its radius + TS_FAR_MARGIN — prim_distance is a lower bound of the Euclidean
// distance outside every primitive, and the margin is far above the rounding of an R-precision pose product, so no point contact_law would
// accept is ever behind a `false`.
template <class R>
__device__ __forceinline__ bool pair_stage_value(const Ctx<R>& c, int pk, int slot, bool store) {
  const int* pi = ts_pair_rec(c, pk);
  const R* pf = c.F + c.foff_pair + pk * TSIM_PF_SIZE;
  const int la = pi[TSIM_PI_LINKA], lb = pi[TSIM_PI_LINKB];
  const R* A = c.LP + la * LK_SIZE;
  const R* B = c.LP + lb * LK_SIZE;
  // relative pose in double: it is what the penetration depths are computed from
  const M3<double> RBd = ldm(c.LPd + lb * 12);
  const M3<double> RPd = mulMM(RBd, ldm_as<double>(pf + TSIM_PF_R));
  const V3<double> pPd = mulMv(RBd, ldv_as<double>(pf + TSIM_PF_P)) + ldv(c.LPd + lb * 12 + 9);
  const M3<double> RPAd = mulMtM(RPd, ldm(c.LPd + la * 12));
  const V3<double> pPAd = mulMtv(RPd, ldv(c.LPd + la * 12 + 9) - pPd);
  const M3<R> RP = cvtm<R>(RPd);
  const V3<R> pP = cvt3<R>(pPd);
  const S6<R> Vrel = to_frame(RP, pP, ld6(A + LK_W) - ld6(B + LK_W));
  if (store) {
    R* S = c.PP + slot * PP_SIZE;
    stm(c.PPd + slot * 12, RPAd); stv(c.PPd + slot * 12 + 9, pPAd);
    stm(S + PP_RPA, cvtm<R>(RPAd)); stv(S + PP_PPA, cvt3<R>(pPAd)); st6(S + PP_WREL, Vrel); stm(S + PP_RP, RP); stv(S + PP_PP, pP);
    st6(S + PP_WN, zero6<R>());
  }
  const float* bd = ts_pair_bound(c, pk);
  if (!(bd[3] >= 0.0f)) return true;
  const V3<R> xc = mulMv(cvtm<R>(RPAd), mk3<R>((R)bd[0], (R)bd[1], (R)bd[2])) + cvt3<R>(pPAd);
  return prim_distance<R>(pi[TSIM_PI_PRIM], pf + TSIM_PF_SHAPE, xc) < (R)bd[3] + R(TS_FAR_MARGIN);
}
// per-direction record of (pair pk, direction k): relative displacement and d(relative twist), both in the primitive
// frame.  pk and k may differ between lanes (lanes = (pair, direction) in phase 2).
// vmode 0: tangents of the current seeds (link records DT);  vmode 1: d/d(qd_k) only (d twist = W_k, poses fixed).
template <class R>
__device__ __forceinline__ void pair_stage_tangent(const Ctx<R>& c, int pk, int slot, int k, R sq, int vmode) {
  const int nd = c.nd;
  if (k >= c.nr) return;
  const int* pi = ts_pair_rec(c, pk);
  const int la = pi[TSIM_PI_LINKA], lb = pi[TSIM_PI_LINKB];
  const R* S = c.PP + slot * PP_SIZE;
  R* T = c.PT + (slot * nd + k) * PT_SIZE;
  const int* LR = c.LI + ts_sched_rec(c.LI);
  const R inA = (la > 0 && ((LR[(la - 1) * TS_LR_SIZE + TS_LR_ANCMASK] >> k) & 1)) ? R(1) : R(0);
  const R inB = (lb > 0 && ((LR[(lb - 1) * TS_LR_SIZE + TS_LR_ANCMASK] >> k) & 1)) ? R(1) : R(0);
  const M3<R> RP = ldm(S + PP_RP);
  const V3<R> pP = ldv(S + PP_PP);
  const S6<R> Wk = ld6(c.WP + k * 6);
  S6<R> dxiP = zero6<R>(), dVrel;
  if (vmode == 0) {
    dxiP = to_frame(RP, pP, Wk * (sq * (inA - inB)));
    const S6<R> dxiB = to_frame(RP, pP, Wk * (sq * inB));
    dVrel = to_frame(RP, pP, ld6(c.DT + (la * nd + k) * DT_SIZE + DT_VW) - ld6(c.DT + (lb * nd + k) * DT_SIZE + DT_VW))
            - crm(dxiB, ld6(S + PP_WREL));
  } else {
    dVrel = to_frame(RP, pP, Wk * (inA - inB));
  }
  st6(T + PT_DTH, dxiP); st6(T + PT_DW, dVrel); st6(T + PT_WN, zero6<R>());
}

// ================================================================================================ phase 2
// lanes = contact points of the staged pairs.  Everything is linear in the 12 numbers that describe a direction for the
// pair (relative displacement dth, drho and d(relative twist) dw, dv, primitive frame):
//     dx  = dth x c + drho                      (c: material point, x: contact point — they differ for sphere-on-plane)
//     dxd = dv + dw x x + wrel x dx
//     dF  = Jx dx + Jv dxd ,   dn = dx x F + x x dF
// so a lane accumulates, over its points, the value wrench (6) and the 6 x 12 matrix M with (dn; dF) = M (dth, drho, dw,
// dv) — independent of the number of directions; one segmented reduction per pair; then lanes = directions apply M to
// their own 12-vector.  (The first version evaluated dF, dn per point AND per direction: 7 x 54 FMA per point on
// TactilePush against ~170 here.)
// PRIMC / FLAGSC >= 0: the pair's primitive type and flags as compile-time constants (a statically known model, tsim_static.h): the type switch
// of contact_law disappears and, for the flat-faced primitives (plane, cuboid: no curvature term), so does every product with it
// NPTC >= 0: the pair's point count, also static — used only to narrow the reduction when all points sit in the first 8 lanes of the slot
// the staged pose of a pair as VALUES: pose of link A in the primitive's frame (double and R precision), relative twist there
template <class R> struct PairPose { M3<double> RPAd; V3<double> pPAd; M3<R> RPA; V3<R> pPA; V3<R> wrel, vrel; };
// The point loop of the matrix form: lanes = contact points; accumulates, per lane, the value wrench w0 and the 6 x 12 matrix M of its points.
// Returns whether any lane of the WAVEFRONT had a penetrating point (wave-uniform).  pf: the pair's float record (shape at TSIM_PF_SHAPE,
// penalty parameters from TSIM_PF_KN) — LDS for the generic kernels, an array of compile-time constants for a static model.
template <class R, int LPE, int PRIMC, int FLAGSC>
__device__ __forceinline__ bool pair_points_matrix(const Ctx<R>& c, int pt0, int npt, int prim, bool sphere_plane, const R* pf, const PairPose<R>& P, int lane, R (&w0)[6], R (&M)[6][12], bool tang = true) {
  const M3<double> RPAd = P.RPAd; const V3<double> pPAd = P.pPAd;
  const M3<R> RPA = P.RPA; const V3<R> pPA = P.pPA, wrel = P.wrel, vrel = P.vrel;
#pragma unroll
  for (int e = 0; e < 6; ++e) {
    w0[e] = R(0);
#pragma unroll
    for (int j = 0; j < 12; ++j) M[e][j] = R(0);
  }
  bool any_hit = false;
  for (int base = 0; base < npt; base += LPE) {
    const int pidx = base + lane;
    bool hit = false;
    V3<R> cP = zero3<R>(), xP = cP, F = cP;
    M3<R> Jx, Jv;
    if (pidx < npt) {
      const V3<R> cp = ld_cpt(c, pt0 + pidx);                // SoA: consecutive lanes -> consecutive addresses
      // "certainly outside" in the kernel's own precision first (TS_FAR_MARGIN is far above the rounding of an R-precision pose product, so the
      // set of points contact_law accepts is unchanged): the far cap of a pad never gets to the double-precision part.  Static models always;
      // the generic fp32 kernels with the batch's pair-cull option (round 5: the primitive type is then a wave-uniform run-time value)
      bool near_ = true;
      if ((PRIMC >= 0 || c.cull) && sizeof(R) == 4 && !sphere_plane) near_ = prim_distance<R>(prim, pf + TSIM_PF_SHAPE, mulMv(RPA, cp) + pPA) < R(TS_FAR_MARGIN);
      if (near_) {
        V3<double> xPd = mulMv(RPAd, cvt3<double>(cp)) + pPAd;
        cP = cvt3<R>(xPd);
        if (sphere_plane) xPd.z -= (double)pf[TSIM_PF_SHAPE]; // lowest point of the sphere (plane normal = +z of P)
        xP = cvt3<R>(xPd);
        hit = contact_law<R, true>(prim, pf + TSIM_PF_SHAPE, pf + TSIM_PF_KN, xP, vrel + cross3(wrel, xP), F, Jx, Jv, xPd, nullptr, tang);
      }
    }
    if (!__any(hit)) continue;
    any_hit = true;
    if (hit) {
      const V3<R> n0 = cross3(xP, F);
      w0[0] += n0.x; w0[1] += n0.y; w0[2] += n0.z; w0[3] += F.x; w0[4] += F.y; w0[5] += F.z;
    }
    if (hit && tang) {
      // columns of dF:  d/d(drho) = K = Jx + Jv [wrel]x ;  d/d(dth) = -K [c]x ;  d/d(dw) = -Jv [x]x ;  d/d(dv) = Jv
      // (row r of A [w]x is (A_r x w)^T)
      V3<R> Kr[3], Ath[3], Aw[3], Jvr[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const V3<R> jv = mk3<R>(Jv.m[3 * r], Jv.m[3 * r + 1], Jv.m[3 * r + 2]);
        Jvr[r] = jv;
        Kr[r] = mk3<R>(Jx.m[3 * r], Jx.m[3 * r + 1], Jx.m[3 * r + 2]) + cross3(jv, wrel);
        Ath[r] = cross3(cP, Kr[r]);                        // -(K_r x c)
        Aw[r] = cross3(xP, jv);                            // -(Jv_r x x)
      }
      // dF rows as 12-vectors
      R dFm[3][12];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        dFm[r][0] = Ath[r].x; dFm[r][1] = Ath[r].y; dFm[r][2] = Ath[r].z;
        dFm[r][3] = Kr[r].x;  dFm[r][4] = Kr[r].y;  dFm[r][5] = Kr[r].z;
        dFm[r][6] = Aw[r].x;  dFm[r][7] = Aw[r].y;  dFm[r][8] = Aw[r].z;
        dFm[r][9] = Jvr[r].x; dFm[r][10] = Jvr[r].y; dFm[r][11] = Jvr[r].z;
      }
      // dn = x x dF - F x dx ,  dx = -[c]x dth + drho :   -F x dx has columns  F x (c x e_j)  (dth)  and  -(F x e_j)  (drho)
      const R xx[3] = {xP.x, xP.y, xP.z};
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const R a0 = dFm[0][j], a1 = dFm[1][j], a2 = dFm[2][j];
        M[0][j] += xx[1] * a2 - xx[2] * a1;
        M[1][j] += xx[2] * a0 - xx[0] * a2;
        M[2][j] += xx[0] * a1 - xx[1] * a0;
        M[3][j] += a0; M[4][j] += a1; M[5][j] += a2;
      }
      // F x (c x e_j) = c (F.e_j) - e_j (F.c)
      const R Fc = dot3(F, cP);
      M[0][0] += cP.x * F.x - Fc; M[1][0] += cP.y * F.x;      M[2][0] += cP.z * F.x;
      M[0][1] += cP.x * F.y;      M[1][1] += cP.y * F.y - Fc; M[2][1] += cP.z * F.y;
      M[0][2] += cP.x * F.z;      M[1][2] += cP.y * F.z;      M[2][2] += cP.z * F.z - Fc;
      // -(F x e_j):  j = x: (0, -Fz, Fy) ; y: (Fz, 0, -Fx) ; z: (-Fy, Fx, 0)
      M[1][3] -= F.z; M[2][3] += F.y;
      M[0][4] += F.z; M[2][4] -= F.x;
      M[0][5] -= F.y; M[1][5] += F.x;
    }
  }
  return any_hit;
}

// Returns whether any environment of the wavefront has a penetrating point of this pair (wave-uniform): if none has, the pair's staged wrench and
// wrench tangents are the zeros they were staged with, and the fold has nothing to add (phase2).
template <class R, int NRM, int LPE, int PRIMC = -1, int FLAGSC = -1, int NPTC = -1>
__device__ __forceinline__ bool pair_contacts_matrix(const Ctx<R>& c, int pk, int slot, int lane, bool tang = true) {
  const int nd = c.nd;
  const int* pi = c.I + c.off_pair + pk * TSIM_PI_SIZE;
  const R* pf = c.F + c.foff_pair + pk * TSIM_PF_SIZE;
  const int pt0 = ts_u(pi[TSIM_PI_PT0]), npt = ts_u(pi[TSIM_PI_NPT]);
  const int prim = PRIMC >= 0 ? PRIMC : ts_u(pi[TSIM_PI_PRIM]);
  const bool sphere_plane = ((FLAGSC >= 0 ? FLAGSC : ts_u(pi[TSIM_PI_FLAGS])) & 2) != 0;
  R* S = c.PP + slot * PP_SIZE;
  PairPose<R> P;
  P.RPAd = ldm(c.PPd + slot * 12); P.pPAd = ldv(c.PPd + slot * 12 + 9);
  P.wrel = ldv(S + PP_WREL); P.vrel = ldv(S + PP_VREL);
  P.RPA = ldm(S + PP_RPA); P.pPA = ldv(S + PP_PPA);      // R-precision copy of the staged pose (the far test of static models)
  R w0[6], M[6][12];               // value wrench (n; F) and d(n; F) / d(dth, drho, dw, dv)
  const bool any_hit = pair_points_matrix<R, LPE, PRIMC, FLAGSC>(c, pt0, npt, prim, sphere_plane, pf, P, lane, w0, M, tang);
  TS_STAMP2(c);
  if (!any_hit) return false;
  constexpr bool kHalfRow = NPTC >= 0 && NPTC <= 8 && NRM <= 8;      // all points (and all directions) in the first 8 lanes of the slot
  if constexpr (kHalfRow) {
#pragma unroll
    for (int e = 0; e < 6; ++e) w0[e] = half_row_sum(w0[e]);
    if (tang) {
#pragma unroll
      for (int e = 0; e < 6; ++e)
#pragma unroll
        for (int j = 0; j < 12; ++j) M[e][j] = half_row_sum(M[e][j]);
    }
  } else {
    seg_sum_many<LPE, 6>(w0);
    if (tang) {
#pragma unroll
      for (int e = 0; e < 6; ++e) seg_sum_many<LPE, 12>(M[e]);
    }
  }
  TS_STAMP2(c);
  if (lane == 0) {
#pragma unroll
    for (int e = 0; e < 6; ++e) S[PP_WN + e] = w0[e];
  }
  if (tang && lane < nd) {                 // lanes = directions: (dn; dF) = M t, t = this direction's staged 12-vector
    R* T = c.PT + (slot * nd + lane) * PT_SIZE;
    R t[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) t[j] = T[j];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      R acc = R(0);
#pragma unroll
      for (int j = 0; j < 12; ++j) acc += M[e][j] * t[j];
      T[PT_WN + e] = acc;
    }
  }
  return true;
}

// The per-direction form: dF, dn evaluated per point and per relevant direction (54 FMA each), 6 (1 + directions)
// accumulators.  Kept for fp64, where the 78 accumulators of the matrix form cost 156 registers and the kernel loses more
// to spills than it gains (measured: 3.07 M vs 2.84 M env-steps/s at two environments per wavefront).
template <class R, int NRM, int LPE>
__device__ __forceinline__ bool pair_contacts_per_direction(const Ctx<R>& c, int pk, int slot, int lane, bool tang = true) {
  const int nd = c.nd;
  const int* pi = c.I + c.off_pair + pk * TSIM_PI_SIZE;
  const R* pf = c.F + c.foff_pair + pk * TSIM_PF_SIZE;
  const int la = ts_u(pi[TSIM_PI_LINKA]), lb = ts_u(pi[TSIM_PI_LINKB]), pt0 = ts_u(pi[TSIM_PI_PT0]), npt = ts_u(pi[TSIM_PI_NPT]);
  const int prim = ts_u(pi[TSIM_PI_PRIM]);
  const bool sphere_plane = (ts_u(pi[TSIM_PI_FLAGS]) & 2) != 0;
  const int anc = ts_u(anc_of(c.I, c.off_link, la) | anc_of(c.I, c.off_link, lb));
  R* S = c.PP + slot * PP_SIZE;
  const M3<double> RPAd = ldm(c.PPd + slot * 12);
  const V3<double> pPAd = ldv(c.PPd + slot * 12 + 9);
  const V3<R> wrel = ldv(S + PP_WREL), vrel = ldv(S + PP_VREL);
  R acc[NRM + 1][6];
#pragma unroll
  for (int d = 0; d <= NRM; ++d)
#pragma unroll
    for (int e = 0; e < 6; ++e) acc[d][e] = R(0);
  bool any_hit = false;
  for (int base = 0; base < npt; base += LPE) {
    const int pidx = base + lane;
    bool hit = false;
    V3<R> cP = zero3<R>(), xP = cP, F = cP;
    M3<R> Jx, Jv;
    if (pidx < npt) {
      const V3<R> cp = ld_cpt(c, pt0 + pidx);                // SoA: consecutive lanes -> consecutive addresses
      V3<double> xPd = mulMv(RPAd, cvt3<double>(cp)) + pPAd;
      cP = cvt3<R>(xPd);
      if (sphere_plane) xPd.z -= (double)pf[TSIM_PF_SHAPE]; // lowest point of the sphere (plane normal = +z of P)
      xP = cvt3<R>(xPd);
      hit = contact_law<R, true>(prim, pf + TSIM_PF_SHAPE, pf + TSIM_PF_KN, xP, vrel + cross3(wrel, xP), F, Jx, Jv, xPd, nullptr, tang);
    }
    if (!__any(hit)) continue;
    any_hit = true;
    if (hit) {
      const V3<R> n0 = cross3(xP, F);
      acc[0][0] += n0.x; acc[0][1] += n0.y; acc[0][2] += n0.z; acc[0][3] += F.x; acc[0][4] += F.y; acc[0][5] += F.z;
    }
    if (!tang) continue;
#pragma unroll
    for (int d = 0; d < NRM; ++d) {
      if (d < nd && ((anc >> d) & 1)) {
        const R* T = c.PT + (slot * nd + d) * PT_SIZE;                       // LDS broadcast, 12 reals
        const V3<R> dth = ldv(T + PT_DTH), drho = ldv(T + PT_DRHO), dw = ldv(T + PT_DW), dv = ldv(T + PT_DV);
        if (hit) {
          const V3<R> dx = cross3(dth, cP) + drho;                           // displacement of the material point
          const V3<R> dxd = dv + cross3(dw, xP) + cross3(wrel, dx);
          const V3<R> dF = mulMv(Jx, dx) + mulMv(Jv, dxd);
          const V3<R> dn = cross3(dx, F) + cross3(xP, dF);
          acc[d + 1][0] += dn.x; acc[d + 1][1] += dn.y; acc[d + 1][2] += dn.z;
          acc[d + 1][3] += dF.x; acc[d + 1][4] += dF.y; acc[d + 1][5] += dF.z;
        }
      }
    }
  }
  if (!any_hit) return false;
  {
    R s[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) s[e] = seg_sum<LPE>(acc[0][e]);
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < 6; ++e) S[PP_WN + e] = s[e];
    }
  }
  if (!tang) return true;
#pragma unroll
  for (int d = 0; d < NRM; ++d) {
    if (d < nd && ((anc >> d) & 1)) {
      R s[6];
#pragma unroll
      for (int e = 0; e < 6; ++e) s[e] = seg_sum<LPE>(acc[d + 1][e]);
      if (lane == 0) {
        R* T = c.PT + (slot * nd + d) * PT_SIZE;
#pragma unroll
        for (int e = 0; e < 6; ++e) T[PT_WN + e] = s[e];
      }
    }
  }
  return true;
}

template <class R, int NRM, int LPE, int PRIMC = -1, int FLAGSC = -1, int NPTC = -1>
__device__ __forceinline__ bool pair_contacts(const Ctx<R>& c, int pk, int slot, int lane, bool tang = true) {
  if (sizeof(R) == 4) return pair_contacts_matrix<R, NRM, LPE, PRIMC, FLAGSC, NPTC>(c, pk, slot, lane, tang);
  else return pair_contacts_per_direction<R, NRM, LPE>(c, pk, slot, lane, tang);
}
// the contact loops of a group of pairs with each pair's primitive type / flags taken from the static model (template recursion over the pairs)
template <class R, int NRM, int LPE, class MS, int PK>
__device__ __forceinline__ unsigned pair_contacts_static(const Ctx<R>& c, int p0, int pe, unsigned act, int lane, bool tang = true) {
  constexpr int NP = MS::Iv(TSIM_IH_NPAIR);
  unsigned hit = 0;
  if constexpr (PK < NP) {
    constexpr int o = MS::Iv(TSIM_IH_OFF_PAIR) + PK * TSIM_PI_SIZE, flags = MS::Iv(o + TSIM_PI_FLAGS), prim = MS::Iv(o + TSIM_PI_PRIM), npt = MS::Iv(o + TSIM_PI_NPT);
    if constexpr ((flags & 1) != 0) { if (PK >= p0 && PK < pe && ((act >> (PK - p0)) & 1u)) { if (pair_contacts<R, NRM, LPE, prim, flags, npt>(c, PK, PK - p0, lane, tang)) hit |= 1u << (PK - p0); } }
    hit |= pair_contacts_static<R, NRM, LPE, MS, PK + 1>(c, p0, pe, act, lane, tang);
  }
  return hit;
}

// lanes = directions: bring the staged pair's wrench (value + tangent k) to the world frame and fold it into the links
template <class R>
__device__ __forceinline__ void pair_fold(const Ctx<R>& c, int pk, int slot, int lane, R sq, bool tang = true) {
  const int k = lane, nd = c.nd;
  if (k >= c.nr) return;
  const int* pi = c.I + c.off_pair + pk * TSIM_PI_SIZE;
  const int la = ts_u(pi[TSIM_PI_LINKA]), lb = ts_u(pi[TSIM_PI_LINKB]);
  const R* S = c.PP + slot * PP_SIZE;
  const R* T = c.PT + (slot * nd + k) * PT_SIZE;
  const M3<R> RP = ldm(S + PP_RP);
  const V3<R> pP = ldv(S + PP_PP);
  const S6<R> Ww = wrench_to_world(RP, pP, ld6(S + PP_WN));
  if (tang) {
    const R inB = ((ts_u(anc_of(c.I, c.off_link, lb)) >> k) & 1) ? R(1) : R(0);
    const S6<R> dWw = wrench_to_world(RP, pP, ld6(T + PT_WN)) + crf(ld6(c.WP + k * 6) * (sq * inB), Ww);
    if (la > 0) {         // link 0 (world-fixed general bodies) takes no wrench
      acc6(c.DT + (la * nd + k) * DT_SIZE + DT_FN, dWw, R(-1));
    }
    if (lb > 0) acc6(c.DT + (lb * nd + k) * DT_SIZE + DT_FN, dWw, R(1));
  }
  if (k == 0) {
    if (la > 0) acc6(c.LP + la * LK_SIZE + LK_FN, Ww, R(-1));
    if (lb > 0) acc6(c.LP + lb * LK_SIZE + LK_FN, Ww, R(1));
  }
}

template <class R, int NRM, int LPE, class MS = void>
__device__ __forceinline__ void phase2(const Ctx<R>& c, int lane, R sq, bool tang = true) {
  for (int p0 = 0; p0 < c.npair; p0 += TS_PAIR_GROUP) {
    const int pe = min(p0 + TS_PAIR_GROUP, c.npair), np = pe - p0;
    // lanes = pairs of the group: value records, and whether the pair can be in contact at all (its points' bounding sphere against the
    // primitive).  A pair that cannot, in ANY environment of the wavefront, is skipped from here on — its staged wrench and wrench tangents
    // would be exact zeros, and what the fold adds for it is x + 0 (round 5: TactileInsertion's hole walls and the ground are out of reach for
    // most of an attempt, D'Claw's fingertips touch the cap 15 % of the time; c.cull == 0 keeps every pair live)
    bool near_ = false;
    if (lane < np && (ts_pair_rec(c, p0 + lane)[TSIM_PI_FLAGS] & 1)) near_ = pair_stage_value(c, p0 + lane, lane, true) || !c.cull;
    const unsigned long long nb = __ballot(near_);
    unsigned act = 0;                          // bit j: pair p0 + j is live in some slot of the wavefront (wave-uniform)
#pragma unroll
    for (int s_ = 0; s_ < TS_WAVE / LPE; ++s_) act |= (unsigned)(nb >> (s_ * LPE)) & ((1u << TS_PAIR_GROUP) - 1u);
    TS_SYNC();
    TS_STAMP(c);
    if (act == 0) { TS_STAMP(c); TS_STAMP(c); continue; }
    // lanes = (pair, direction): per-direction records
    if (tang) {
      const int ntask = np * c.nr;
      // wave-uniform trip count with the idle lanes masked inside: the lane-strided form (t = lane; t < ntask; t += LPE)
      // of this loop produced wrong adjoints in the 32-lane shape only (toolchain issue with the divergent loop, not
      // understood further; the whole GPU suite runs under TSIM_LPE = 64 / 32 / 16 because of it)
      for (int t0 = 0; t0 < ntask; t0 += LPE) {
        const int t = t0 + lane;
        if (t < ntask) {
          const int p = t / c.nr, k = t - p * c.nr;
          if ((ts_pair_rec(c, p0 + p)[TSIM_PI_FLAGS] & 1) && ((act >> p) & 1u)) pair_stage_tangent(c, p0 + p, p, k, sq, 0);
        }
      }
    }
    TS_SYNC();
    TS_STAMP(c);
    // bit j of `hit`: some environment of the wavefront has a penetrating point of pair p0 + j (wave-uniform).  A near pair without one — most of
    // TactileInsertion's eleven during an attempt: the bounding spheres of the hole's walls and of the object's faces overlap long before a sampled
    // point is inside the other body — keeps the zero wrench and zero wrench tangents it was staged with: the fold would add x + 0 (round 5; with c.cull
    // == 0 every live pair is folded, as before)
    unsigned hit = 0;
    if constexpr (std::is_void<MS>::value) {
      for (int pk = p0; pk < pe; ++pk)        // lanes = contact points
        if ((ts_u(c.I[c.off_pair + pk * TSIM_PI_SIZE + TSIM_PI_FLAGS]) & 1) && ((act >> (pk - p0)) & 1u)) { if (pair_contacts<R, NRM, LPE>(c, pk, pk - p0, lane, tang)) hit |= 1u << (pk - p0); }
    } else hit = pair_contacts_static<R, NRM, LPE, MS, 0>(c, p0, pe, act, lane, tang);
    if (!c.cull) hit = act;
    TS_SYNC();
    TS_STAMP(c);
    for (int pk = p0; pk < pe; ++pk)        // lanes = directions; serial over pairs: two pairs may touch the same link
      if ((ts_u(c.I[c.off_pair + pk * TSIM_PI_SIZE + TSIM_PI_FLAGS]) & 1) && ((hit >> (pk - p0)) & 1u)) pair_fold(c, pk, pk - p0, lane, sq, tang);
    TS_SYNC();
  }
}

// joint-space forces: damping, limits, motor (lanes = dofs; the motor of a dof comes from the schedule in LDS), and the
// 1 / ca scaling of g and H (each lane scales its own column)
template <class R>
__device__ __forceinline__ void phase3_joint_space(const Ctx<R>& c, int lane, R sq, R sv, R h2, bool tang = true) {
  const int nr = c.nr;
  const bool act = lane < nr;
  if (act) {
    const int j = lane;
    const R* df = c.F + c.foff_dof + j * TSIM_DF_SIZE;
    R gj = c.g[j], hjj = R(0);                       // hjj: joint-space part of H[j][j], added (scaled) at the end
    gj += df[TSIM_DF_DAMPING] * c.qd[j]; hjj += df[TSIM_DF_DAMPING] * sv;
    if (df[TSIM_DF_LIM_K] > R(0)) {
      if (c.q[j] < df[TSIM_DF_LIM_LO]) { gj -= df[TSIM_DF_LIM_K] * (df[TSIM_DF_LIM_LO] - c.q[j]); hjj += df[TSIM_DF_LIM_K] * sq; }
      else if (c.q[j] > df[TSIM_DF_LIM_HI]) { gj += df[TSIM_DF_LIM_K] * (c.q[j] - df[TSIM_DF_LIM_HI]); hjj += df[TSIM_DF_LIM_K] * sq; }
    }
    const int dm = ts_dof_motor(c)[j];
    for (int m = (dm == -2 ? 0 : dm); m >= 0 && m < c.nu; ++m) {
      const int* mi = ts_motor_rec(c, m);
      if (mi[TSIM_MI_DOF] == j) {
        const R* mf = c.F + c.foff_motor + m * TSIM_MF_SIZE;
        if (mi[TSIM_MI_CTRL] == 0) {
          R uc = fmin(fmax(c.u[m], R(-1)), R(1));
          gj -= mf[TSIM_MF_LO] + (uc + R(1)) * (R(0.5) * (mf[TSIM_MF_HI] - mf[TSIM_MF_LO]));
        } else {
          gj -= mf[TSIM_MF_P] * (c.u[m] - c.q[j]) - mf[TSIM_MF_D] * c.qd[j];
          hjj += mf[TSIM_MF_P] * sq + mf[TSIM_MF_D] * sv;
        }
      }
      if (dm != -2) break;             // the usual case: exactly one motor on this dof
    }
    c.g[j] = gj * h2;
    if (tang) c.H[j * nr + j] += hjj * h2;
  }
}

// ================================================================================================ phase 3
// lanes = directions.  Leaf -> root: tau_j = W_j . F_subtree(link(j)); fold each link's wrench into its
// parent; then the joint-space forces.  Result: g (value, LDS) and H[j][k] = d g_j / d dir_k (lane k owns
// column k).  Both are scaled by h^2.
template <class R, bool EXPJ, int LPE>
__device__ __forceinline__ void phase3(const Ctx<R>& c, int lane, R sq, R sv, bool tang = true) {
  const int nd = c.nd, nr = c.nr;
  const bool act = lane < nr;
  const R h2 = R(1) / c.ca;      // g = r / ca  (BDF1: h^2 r)
  // One lane per (direction k, root branch b): it sweeps the links of branch b leaf -> root for direction k (the
  // branches are independent here too; contacts coupled them in phase 2 already).  A child's subtree wrench is handed to
  // its parent in registers when the parent is the next link of the lane's sweep (chains); only branching parents go
  // through LDS.
  const int* S = c.LI;
  const int nsteps = ts_u(S[1]), rec0 = TS_SCHED_ENT + nsteps * 16, ntask = ts_u(S[TS_SCHED_NB]) * nr;
  for (int t0 = 0; t0 < ntask; t0 += LPE) {
    const int t = t0 + lane;
    const bool has = t < ntask && (tang || t % nr == 0);      // values only: direction 0's lanes carry g, the others have nothing to do
    const int b = has ? t / nr : 0, k = has ? t - b * nr : 0;
    const int col = S[TS_SCHED_LEADER + b];
    const S6<R> Wk = ld6(c.WP + k * 6);
    S6<R> cF = zero6<R>(), cdF = zero6<R>();
    int carry_to = -1;
    for (int st = nsteps - 1; st >= 0; --st) {
      TS_SYNC();                               // wrenches folded into branching parents in the previous step
      const int i = has ? (S[TS_SCHED_ENT + st * 16 + col] & 0xff) : 0;
      if (i == 0) continue;
      const int* li = S + rec0 + (i - 1) * TS_LR_SIZE;
      const int par = li[TS_LR_PARENT], k0 = li[TS_LR_DOF0], ndj = li[TS_LR_NDOF];
      S6<R> F = ld6(c.LP + i * LK_SIZE + LK_FN);
      S6<R> dF = zero6<R>();
      if (tang) dF = ld6(c.DT + (i * nd + k) * DT_SIZE + DT_FN);
      if (carry_to == i) { F = F + cF; dF = dF + cdF; }
      const bool moves = (li[TS_LR_ANCMASK] >> k) & 1;
      for (int j = k0; j < k0 + ndj; ++j) {
        const S6<R> Wj = ld6(c.WP + j * 6);
        if (tang) {
          R dtau = dot6(Wj, dF);
          if (moves) {
            const bool same_exp = EXPJ && li[TS_LR_JTYPE] == TSIM_J_SPHERICAL_EXP && k >= k0 && k < k0 + ndj;
            const S6<R> dW = same_exp ? ld6(c.expw + ((j - k0) * 3 + (k - k0)) * 6) : crm(Wk, Wj);
            dtau += sq * dot6(dW, F);
          }
          c.H[j * nr + k] = dtau * h2;                  // columns are stored scaled by 1 / ca (g = r / ca)
        }
        if (k == 0) c.g[j] = dot6(Wj, F);
      }
      if (par > 0) {
        const int nxt = st > 0 ? (S[TS_SCHED_ENT + (st - 1) * 16 + col] & 0xff) : 0;
        if (par == nxt) { cF = F; cdF = dF; carry_to = par; }
        else {
          if (tang) acc6(c.DT + (par * nd + k) * DT_SIZE + DT_FN, dF, R(1));
          if (k == 0) acc6(c.LP + par * LK_SIZE + LK_FN, F, R(1));
        }
      }
    }
  }
  TS_SYNC();
  TS_STAMP2(c);
  phase3_joint_space<R>(c, lane, sq, sv, h2, tang);
  TS_SYNC();
}

#include "tsim_static_eval.h"
// is the evaluation of model MS the fused register-resident pass?  (a static model that asks for it; both precisions since round 5: the fp64
// instantiation for TactilePush fits 512 registers with 8 spilled, against 90 in the generic fp64 kernel)
template <class MS, class R> constexpr bool ts_static_fused() {
  if constexpr (std::is_void<MS>::value) return false; else return MS::FUSED;
}

// full evaluation at the trial increment held in c.dl (with c.q0, c.qd0, c.u): fills c.q, c.qd, c.qa, link state, g, H.
// The Newton unknown is the increment  dl = q1 - qp  over the force-free predictor qp (BDF1: q0 + h qd0; BDF2:
// 4/3 q0 - 1/3 q_1 + 8/9 h qd0 - 2/9 h qd_1), not q1 itself: qd1 = qdp + cv dl, qdd1 = ca dl keep full relative
// precision in fp32 (no q1 - q0 cancellation).  forward seeds: (1, cv, ca) -> H = dg/dq1 ;  adjoint seeds (1, 0, 0)
// -> H = (1/ca) dr/dq  (BDF1: h^2 dr/dq).
// tang (wave-uniform, default true): false = the VALUES only — q, qd, qa, the link state and g; no tangent, no H (c.H keeps what it held).  Every
// number it does compute is computed exactly as the full evaluation computes it.  k_forward asks for it when no environment of the wavefront
// needs a Newton matrix from this round: line-search trials deep in a backtracking (only ||g|| decides them), finished slots.
template <class R, int NRM, bool EXPJ, int LPE, class MS = void>
__device__ __forceinline__ void evaluate(const Ctx<R>& c, int lane, R sq, R sv, R sa, bool tang = true) {
  if (lane < c.nr) {
    const R d = c.dl[lane];
    c.qd[lane] = c.qdp[lane] + c.cv * d;
    c.qa[lane] = c.ca * d;
    c.q[lane] = c.qp[lane] + d;
    c.qD[lane] = c.qpD[lane] + (double)d;
  }
  TS_SYNC();
  TS_STAMP(c);
  if constexpr (ts_static_fused<MS, R>()) {
    evaluate_static_fused<R, NRM, LPE, MS, false>(c, lane, sq, sv, sa, tang);      // a statically known model: one register-resident pass (tsim_static_eval.h)
    TS_STAMP(c);
    return;
  }
  if constexpr (std::is_void<MS>::value) phase1<R, true, EXPJ>(c, lane, sq, sv, sa, tang);
  else phase1_static_levels<R, MS, true, false>(c, lane, sq, sv, sa);      // a statically known model whose evaluation is not fused (tsim_static.h; forward: no COM / inertia records)
  TS_STAMP(c);
  phase2<R, NRM, LPE, MS>(c, lane, sq, tang);
  TS_STAMP(c);
  phase3<R, EXPJ, LPE>(c, lane, sq, sv, tang);
  TS_STAMP(c);
}

// ================================================================================================ dense solve
// Gauss-Jordan with partial pivoting, one matrix row per lane (lanes 0..n-1 of the slot) held in registers, in
// precision S: fp64 for the adjoint solves (their error goes straight into the gradient) and for the Newton steps.
// Pivot search: 4 DPP steps inside the slot's first 16-lane row; pivot row broadcast: v_readlane (one slot per
// wavefront) or the LDS crossbar (several).  No LDS memory traffic.
// Solves A x = b (or A^T x = b), n <= NRM <= 16; x is written only where `write` holds (a per-slot predicate).
__device__ __forceinline__ double fast_rcp(double x) {          // full double accuracy but for the last bit or two
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
template <class R, int NRM, int LPE, class S = double>
__device__ __forceinline__ void solve_lanes(const R* A, const R* b, R* x, int n, bool transpose, int lane, bool write = true) {
  // branch-free set-up: every lane loads a valid element (row min(lane, n - 1)) and lanes / columns beyond n are turned into
  // identity rows by selects
  // broadcasts inside the slot go through the LDS crossbar in every shape (per-lane source = slot base + row): the
  // v_readlane form of the one-environment-per-wavefront shape keeps the pivot row in SGPRs, which this kernel has none to spare
  const int sbase = (int)threadIdx.x & ~(LPE - 1);
  auto sbc = [sbase](auto v, int src) { return lane_gather(v, sbase + src); };
  S a[NRM], rb;
  const bool row = lane < n;
  const int r = min(lane, n - 1);
#pragma unroll
  for (int j = 0; j < NRM; ++j) {
    const int jj = min(j, n - 1);
    const S v = (S)(transpose ? A[jj * n + r] : A[r * n + jj]);
    a[j] = (row && j < n) ? v : ((j == lane) ? S(1) : S(0));
  }
  rb = row ? (S)b[r] : S(0);
  // Pivot search: one unsigned key per lane = the bit pattern of |a[col]| (monotonic for non-negative floats) with its low four
  // bits replaced by 15 - lane, so that a 4-step DPP maximum over the slot's first row yields the largest magnitude and, among
  // magnitudes equal to within 16 ulp, the lowest lane.  Rows already used as pivots (and lanes >= n) carry key 0.
  bool done = !row; int mycol = -1; S mypiv = S(1);
#pragma unroll
  for (int col = 0; col < NRM; ++col) {
    if (col < n) {
      unsigned key = done ? 0u : ((__builtin_bit_cast(unsigned, fabsf((float)a[col])) & ~0xFu) | (unsigned)(15 - (lane & 15)) | 0x10u);
      key = max(key, (unsigned)dpp_i<0xB1, 0xf>((int)key));
      key = max(key, (unsigned)dpp_i<0x4E, 0xf>((int)key));
      key = max(key, (unsigned)dpp_i<0x141, 0xf>((int)key));
      key = max(key, (unsigned)dpp_i<0x140, 0xf>((int)key));
      const int p = 15 - (int)(sbc((int)key, 0) & 0xFu);      // the slot's first row holds the matrix rows
      // the pivot row (entries col.., rhs) travels in ONE batch of broadcasts; the multiplier comes from a reciprocal
      S prow[NRM];
#pragma unroll
      for (int j = 0; j < NRM; ++j) if (j >= col) prow[j] = sbc(a[j], p);
      const S pb = sbc(rb, p);
      const S piv = prow[col];
      const bool isp = lane == p;
      const S f = isp ? S(0) : a[col] * fast_rcp(piv);
#pragma unroll
      for (int j = 0; j < NRM; ++j) if (j >= col) a[j] -= f * prow[j];
      rb -= f * pb;
      done = done || isp; mycol = isp ? col : mycol; mypiv = isp ? piv : mypiv;
    }
  }
  if (write && row && mycol >= 0) x[mycol] = (R)(rb * fast_rcp(mypiv));
  TS_SYNC();
}

// ---- the same system without pivoting and without the LDS crossbar (round 4; fp32 kernels) ----------------------------------------
// With the pivot of column c fixed to row c, the source lane of every broadcast is a compile-time lane of the slot's first 16-lane row, and
// gfx90a+ has a DPP control for exactly that: row_newbcast:c — one VALU instruction per dword instead of a ds_bpermute round trip, and
// no pivot search (4 DPP maxima + one more round trip per column).  solve_lanes spends ~3.9 k of an evaluation round's 50.6 k cycles, most
// of it waiting for those round trips (two dependent ones per column); this form is ~45 instructions per column.
// The Newton matrix H = M + h D + h^2 K has the mass matrix on its diagonal and is solved in double, so elimination in the natural order is
// normally fine; it is CHECKED, not assumed: a multiplier that is not finite or exceeds 1e6 flags the lane, and the caller then repeats the
// solve with partial pivoting (solve_lanes) for the whole wavefront.  The fp64 kernels keep the pivoted solve — they are the ones that walk
// the oracle's iterates to round-off, and the oracle pivots.
template <int L> __device__ __forceinline__ double row_bcast(double x) {            // value of lane L of the caller's 16-lane row
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = dpp_i<0x150 + L, 0xf>((int)(b & 0xffffffffll)), hi = dpp_i<0x150 + L, 0xf>((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int L> __device__ __forceinline__ float row_bcast(float x) { return __builtin_bit_cast(float, dpp_i<0x150 + L, 0xf>(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ float fast_rcp_s(float x) { const float r = __builtin_amdgcn_rcpf(x); return fmaf(fmaf(-x, r, 1.0f), r, r); }
__device__ __forceinline__ double fast_rcp_s(double x) { return fast_rcp(x); }
// NC > 0: the system's size as a compile-time constant (static models): columns >= NC do not exist
template <int COL, int NRM, class S, int NC = 0>
__device__ __forceinline__ void gj_step_nopivot(S (&a)[NRM], S& rb, S& mypiv, bool& bad, int n, bool row, int lane) {
  constexpr int NCOL = NC > 0 ? NC : NRM;
  if (COL < n) {                                   // wave-uniform
    S prow[NRM];
#pragma unroll
    for (int j = COL; j < NCOL; ++j) prow[j] = row_bcast<COL>(a[j]);
    const S pb = row_bcast<COL>(rb);
    const S piv = prow[COL];
    const bool elim = row && lane != COL;          // identity rows (lanes >= n, other 16-lane rows of a wide slot) take no part
    const S f = elim ? a[COL] * fast_rcp_s(piv) : S(0);
    bad = bad || (elim && (!ts_finite(f) || t_abs(f) >= S(1e6)));      // also catches a zero / non-finite pivot
#pragma unroll
    for (int j = COL; j < NCOL; ++j) a[j] -= f * prow[j];
    rb -= f * pb;
    if (lane == COL) mypiv = piv;
  }
  if constexpr (COL + 1 < NCOL) gj_step_nopivot<COL + 1, NRM, S, NC>(a, rb, mypiv, bad, n, row, lane);
}
// Solves A x = b (or A^T x = b) as solve_lanes does; returns false in the lanes of a slot whose elimination met a bad multiplier (x is then
// NOT written for that slot: the caller falls back to solve_lanes).
template <class R, int NRM, int LPE, class S = double, int NC = 0>
__device__ __forceinline__ bool solve_lanes_nopivot(const R* A, const R* b, R* x, int n_, bool transpose, int lane, bool write = true) {
  const int n = NC > 0 ? NC : n_;
  S a[NRM], rb, mypiv = S(1);
  const bool row = lane < n;
  const int r = min(lane, n - 1);
#pragma unroll
  for (int j = 0; j < NRM; ++j) {
    const int jj = min(j, n - 1);
    const S v = (S)(transpose ? A[jj * n + r] : A[r * n + jj]);
    a[j] = (row && j < n) ? v : ((j == lane) ? S(1) : S(0));
  }
  rb = row ? (S)b[r] : S(0);
  bool bad = false;
  gj_step_nopivot<0, NRM, S, NC>(a, rb, mypiv, bad, n, row, lane);
  const bool slot_bad = seg_max<LPE>(bad ? 1.0f : 0.0f) > 0.0f;
  if (write && row && !slot_bad) x[lane] = (R)(rb * fast_rcp_s(mypiv));
  TS_SYNC();
  return !slot_bad;
}
// the solve the kernels call: fp32 kernels try the pivot-free DPP form first (-DTS_SOLVE_PIVOT_ONLY: A/B), fp64 kernels always pivot
// S: the precision of the elimination.  double for the adjoint solves (their error goes straight into the gradient); the NEWTON steps of the
// fp32 kernels may take R: the matrix they eliminate is an fp32 rounding already, and the step only has to reduce ||g|| (the residual decides)
template <class R, int NRM, int LPE, class S = double, int NC = 0>
__device__ __forceinline__ void solve_newton(const R* A, const R* b, R* x, int n, bool transpose, int lane, bool write = true) {
#ifndef TS_SOLVE_PIVOT_ONLY
  if (sizeof(R) == 4) {
    const bool ok = solve_lanes_nopivot<R, NRM, LPE, S, NC>(A, b, x, n, transpose, lane, write);
    if (__all(ok || !write)) return;
  }
#endif
  solve_lanes<R, NRM, LPE, double>(A, b, x, n, transpose, lane, write);
}

template <int LPE, class R> __device__ __forceinline__ R block_norm2(const R* v, int n, int lane) {
  R s = lane < n ? v[lane] * v[lane] : R(0);
  return t_sqrt(seg_sum<LPE>(s));
}
