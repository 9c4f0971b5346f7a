// tsim_policy_push.h — the TactilePush policy INSIDE the episode launch (closed loop in one launch each way).
//
// BASELINE config 3 as cfg/gd_tactile.yaml runs it (algorithms/gd.py:224-259) puts a policy between env-steps: observation
// (envs/tactile_push_env.py:72-131) -> DiagGaussianActor mean (utils/model.py:123-151: 393 -> 64 -> 64 -> 3, ELU) -> action mapping
// (:175-193) -> StepSimFunction.  With one launch per env-step every env-step waits for the slowest of the 4096 environments (4.4 M
// env-steps/s against 7.1 M for the open-loop episode launch).  Here the observation, the MLP and the action mapping of an
// environment run in ITS slot of the wavefront between two frames of k_forward, and their reverse between two frames of k_backward:
// the closed loop becomes one launch per episode each way, and an environment never waits for another wavefront.
//
// Lanes of a slot = hidden units (64 / LPE per lane); inputs are broadcast (the observation row straight from HBM / L2 — the tactile
// frame was written by this very slot a moment ago — the hidden vectors through the slot's idle pair-staging scratch in LDS); weights
// stream from L2 in the layouts that make a slot's read contiguous:  W1T [393][64], W2T [64][64], W3 [3][64] forward;
// W1p [64][W1S >= 393, padded to a multiple of 4], W2 [64][64] backward.  The per-layer (input, output-gradient) pairs go to HBM; the weight
// gradients are batched GEMMs over the whole episode afterwards (as in algorithms/batched_gd.py).
#pragma once
#include "tsim_device.h"

enum { PP_OBS = 393, PP_HID = 64, PP_ACT = 3, PP_GOAL = 3, PP_NTAC = 390 };

template <class R> struct PushPolicy {
  const R *W1T, *b1, *W2T, *b2, *W3, *b3;      // forward layouts
  const R *W1p, *W2; int w1s;                  // backward layouts (row stride of W1p)
  const R* goal;                               // [B][3] goal pose (x, y, yaw)
  const R* dist;                               // [T][B][2] external force on the box per env-step
  const R* tac0;                               // [B][390] tactile at the initial state (tsim_readout after the reset)
  R *u_out, *gl_out, *h1_out, *h2_out;         // forward records [T][B][3], [T][B][3], [T][B][64], [T][B][64]
  // backward only
  const R* du_direct;                          // [T][B][3] direct derivative of the loss w.r.t. the policy output (the reward's action term)
  R *g1_out, *g2_out, *g3_out;                 // [T][B][64], [T][B][64], [T][B][3] gradients w.r.t. the layers' pre-activations
  R* dobs_tac;                                 // [T][B][390] gradient w.r.t. the tactile part of frame f's observation = the seed of frame f - 1's tactile
                                               // read-out (one slice per frame: every address is written once and read once per launch — the
                                               // vector L1 is not coherent with a wavefront's own earlier stores to a line it has cached)
};

template <class R> __device__ __forceinline__ R pp_elu(R x) { return x > R(0) ? x : (R)expm1((double)x); }
template <class R> __device__ __forceinline__ R pp_elu_grad_from_output(R y) { return y > R(0) ? R(1) : y + R(1); }

// the 1/2/4 consecutive reals a lane owns, through one vector load
template <int OPL, class R> __device__ __forceinline__ void pp_ld(const R* p, R* out) {
#pragma unroll
  for (int o = 0; o < OPL; ++o) out[o] = p[o];
}

// Observation -> action for the environment of this slot.  tac_prev: the tactile frame the observation is built from (global; written by
// this slot, hence the fence + bypassing loads).  q (double, the state before the frame) gives the goal in the gripper frame.
// Writes c.u (the 6 actuator inputs of the frame) and the forward records.
template <int LPE, class R>
__device__ __forceinline__ void push_policy_forward(const Ctx<R>& c, int lane, bool valid, const PushPolicy<R>& P, size_t rec /* f * B + env */, int env, const R* tac_prev) {
  constexpr int OPL = PP_HID / LPE;                    // hidden units per lane
  R* scr = c.PT;                                       // >= 128 reals of idle pair-staging scratch per slot
  // goal pose in the gripper frame: rotation by -yaw, then the gripper's position is subtracted (tactile_push_env.py:84-92)
  R gl[3];
  {
    const R* g = P.goal + (size_t)env * 3;
    double sn, cs; t_sincos_d(c.q0D[0], sn, cs);
    const double gx = g[0], gy = g[1];
    gl[0] = (R)(cs * gx + sn * gy - c.q0D[1]); gl[1] = (R)(-sn * gx + cs * gy - c.q0D[2]); gl[2] = (R)((double)g[2] - c.q0D[0]);
  }
  R acc[OPL], w[OPL];
  pp_ld<OPL>(P.b1 + OPL * lane, acc);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    pp_ld<OPL>(P.W1T + (size_t)i * PP_HID + OPL * lane, w);
#pragma unroll
    for (int o = 0; o < OPL; ++o) acc[o] += w[o] * gl[i];
  }
#pragma unroll 8
  for (int i = 0; i < PP_NTAC; ++i) {
    const R x = __builtin_nontemporal_load(tac_prev + i);          // slot-uniform address: one transaction per slot
    pp_ld<OPL>(P.W1T + (size_t)(3 + i) * PP_HID + OPL * lane, w);
#pragma unroll
    for (int o = 0; o < OPL; ++o) acc[o] += w[o] * x;
  }
  R h1[OPL];
#pragma unroll
  for (int o = 0; o < OPL; ++o) { h1[o] = pp_elu(acc[o]); scr[OPL * lane + o] = h1[o]; }
  if (valid) {
#pragma unroll
    for (int o = 0; o < OPL; ++o) P.h1_out[rec * PP_HID + OPL * lane + o] = h1[o];
  }
  TS_SYNC();
  pp_ld<OPL>(P.b2 + OPL * lane, acc);
#pragma unroll 8
  for (int i = 0; i < PP_HID; ++i) {
    const R x = scr[i];
    pp_ld<OPL>(P.W2T + (size_t)i * PP_HID + OPL * lane, w);
#pragma unroll
    for (int o = 0; o < OPL; ++o) acc[o] += w[o] * x;
  }
  R h2[OPL];
#pragma unroll
  for (int o = 0; o < OPL; ++o) h2[o] = pp_elu(acc[o]);
  if (valid) {
#pragma unroll
    for (int o = 0; o < OPL; ++o) P.h2_out[rec * PP_HID + OPL * lane + o] = h2[o];
  }
  R up[3];
#pragma unroll
  for (int a_ = 0; a_ < 3; ++a_) {
    pp_ld<OPL>(P.W3 + (size_t)a_ * PP_HID + OPL * lane, w);
    R s = R(0);
#pragma unroll
    for (int o = 0; o < OPL; ++o) s += w[o] * h2[o];
    up[a_] = seg_sum<LPE>(s) + P.b3[a_];
  }
  TS_SYNC();                                           // scr is the pair staging of the next evaluation
  // action mapping (tactile_push_env.py:175-193): [tanh(u), force on the box, 0]
  if (lane < 6) {
    R v = R(0);
    if (lane < 3) v = (R)tanh((double)(lane == 0 ? up[0] : (lane == 1 ? up[1] : up[2])));
    else if (lane < 5) v = P.dist[rec * 2 + (lane - 3)];
    c.u[lane] = v;
  }
  if (valid && lane < 3) {
    P.u_out[rec * 3 + lane] = lane == 0 ? up[0] : (lane == 1 ? up[1] : up[2]);
    P.gl_out[rec * 3 + lane] = lane == 0 ? gl[0] : (lane == 1 ? gl[1] : gl[2]);
  }
}

// Reverse of the above for frame f: da = dL/d(action) of the frame (lane m < 6 holds entry m).  Writes the pre-activation gradients
// (g1, g2, g3), the gradient w.r.t. the tactile part of the observation into P.dobs_tac (the seed of the previous frame's tactile
// read-out) and returns, in lanes 0..2, what the goal part of the observation puts on q[0..2] of the state before the frame.
template <int LPE, class R>
__device__ __forceinline__ R push_policy_backward(const Ctx<R>& c, int lane, bool valid, const PushPolicy<R>& P, size_t rec, int env, R da, double yaw) {
  constexpr int OPL = PP_HID / LPE;
  R* scr = c.PT;
  R g3[3];
#pragma unroll
  for (int a_ = 0; a_ < 3; ++a_) {
    const double t = tanh((double)P.u_out[rec * 3 + a_]);
    g3[a_] = (R)((double)seg_bcast<LPE>(da, a_) * (1.0 - t * t)) + P.du_direct[rec * 3 + a_];
  }
  if (valid && lane < 3) P.g3_out[rec * 3 + lane] = lane == 0 ? g3[0] : (lane == 1 ? g3[1] : g3[2]);
  R w[OPL], h[OPL], g2[OPL];
  pp_ld<OPL>(P.h2_out + rec * PP_HID + OPL * lane, h);
#pragma unroll
  for (int o = 0; o < OPL; ++o) g2[o] = R(0);
#pragma unroll
  for (int a_ = 0; a_ < 3; ++a_) {
    pp_ld<OPL>(P.W3 + (size_t)a_ * PP_HID + OPL * lane, w);
#pragma unroll
    for (int o = 0; o < OPL; ++o) g2[o] += w[o] * g3[a_];
  }
#pragma unroll
  for (int o = 0; o < OPL; ++o) { g2[o] *= pp_elu_grad_from_output(h[o]); scr[OPL * lane + o] = g2[o]; }
  if (valid) {
#pragma unroll
    for (int o = 0; o < OPL; ++o) P.g2_out[rec * PP_HID + OPL * lane + o] = g2[o];
  }
  TS_SYNC();
  R g1[OPL];
#pragma unroll
  for (int o = 0; o < OPL; ++o) g1[o] = R(0);
#pragma unroll 8
  for (int j = 0; j < PP_HID; ++j) {                   // dh1[o] = sum_j W2[j][o] g2[j]
    const R x = scr[j];
    pp_ld<OPL>(P.W2 + (size_t)j * PP_HID + OPL * lane, w);
#pragma unroll
    for (int o = 0; o < OPL; ++o) g1[o] += w[o] * x;
  }
  pp_ld<OPL>(P.h1_out + rec * PP_HID + OPL * lane, h);
  TS_SYNC();
#pragma unroll
  for (int o = 0; o < OPL; ++o) { g1[o] *= pp_elu_grad_from_output(h[o]); scr[PP_HID + OPL * lane + o] = g1[o]; }
  if (valid) {
#pragma unroll
    for (int o = 0; o < OPL; ++o) P.g1_out[rec * PP_HID + OPL * lane + o] = g1[o];
  }
  TS_SYNC();
  // d obs[i] = sum_j W1[j][i] g1[j]: a lane owns the 4-element chunks i = 4 (lane + LPE m) .. + 3
  constexpr int NCH = (PP_OBS + 4 * LPE - 1) / (4 * LPE);
  R dob[NCH][4];
#pragma unroll
  for (int m = 0; m < NCH; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) dob[m][e] = R(0);
#pragma unroll 2
  for (int j = 0; j < PP_HID; ++j) {
    const R x = scr[PP_HID + j];
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
      const int i0 = 4 * (lane + LPE * m);
      if (i0 < P.w1s) {
        const R* wp = P.W1p + (size_t)j * P.w1s + i0;
#pragma unroll
        for (int e = 0; e < 4; ++e) dob[m][e] += wp[e] * x;
      }
    }
  }
  TS_SYNC();
  // tactile part -> the seed of the previous frame's tactile read-out; goal part -> q[0..2] of the state before the frame
#pragma unroll
  for (int m = 0; m < NCH; ++m) {
    const int i0 = 4 * (lane + LPE * m);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + e;
      if (valid && i >= 3 && i < PP_OBS) P.dobs_tac[rec * PP_NTAC + (i - 3)] = dob[m][e];
    }
  }
  // lane 0 holds d obs[0..2] in its first chunk
  const R d0 = seg_bcast<LPE>(dob[0][0], 0), d1 = seg_bcast<LPE>(dob[0][1], 0), d2 = seg_bcast<LPE>(dob[0][2], 0);
  double sn, cs; t_sincos_d(yaw, sn, cs);
  const R* g = P.goal + (size_t)env * 3;
  const double gx = g[0], gy = g[1];
  R out = R(0);
  if (lane == 0) out = (R)((double)d0 * (-sn * gx + cs * gy) + (double)d1 * (-cs * gx - sn * gy) - (double)d2);
  else if (lane == 1) out = -d0;
  else if (lane == 2) out = -d1;
  return out;
}
