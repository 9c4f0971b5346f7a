// tsim_policy_push.h — the TactilePush policy INSIDE the episode launch (closed loop in one launch each way).
//
// BASELINE config 3 as cfg/gd_tactile.yaml runs it (algorithms/gd.py:224-259) puts a policy between env-steps: observation
// (envs/tactile_push_env.py:72-131) -> DiagGaussianActor mean (utils/model.py:123-151: 393 -> 64 -> 64 -> 3, ELU) -> action mapping
// (:175-193) -> StepSimFunction.  With one launch per env-step every env-step waits for the slowest of the 4096 environments (4.4 M
// env-steps/s against 7.1 M for the open-loop episode launch).  Here the observation, the MLP and the action mapping of an
// environment run in ITS slot of the wavefront between two frames of k_forward, and their reverse between two frames of k_backward:
// the closed loop becomes one launch per episode each way, and an environment never waits for another wavefront.
//
// The dense layers run WAVE-wide: lane j of the wavefront = output unit j, for all the wavefront's environments at once.  A weight row
// is one coalesced 256-byte load per wavefront (one dword per lane), the inputs of the 1 / 2 / 4 environments are LDS broadcasts from the
// slots' idle tangent / pair-staging records, and a lane accumulates one output per environment.  What a lone wavefront pays for here is
// L2 latency — (rows / loads in flight) round trips — so the rows go 32 - 40 at a time, which a one-register row allows (the first version
// gave every slot its own 16-byte weight loads: 4 registers per row, 8 rows in flight, the same row fetched once per slot: 70 us per
// env-step forward, a fifth of the episode; profiles/r03_closed_loop_policy.md).  Layouts: W1T [393][64], W2T [64][64], W3 [3][64]
// forward; W1p [64][W1S >= 393, padded to a multiple of 4], W2 [64][64] backward.  The per-layer (input, output-gradient) pairs go to HBM; the weight
// gradients are batched GEMMs over the whole episode afterwards (as in algorithms/batched_gd.py).
#pragma once
#include "tsim_device.h"

enum { PP_OBS = 393, PP_HID = 64, PP_ACT = 3, PP_GOAL = 3, PP_NTAC = 390 };
enum { TSIM_PUSH_OBS_TACTILE = 0, TSIM_PUSH_OBS_NO_TACTILE = 1, TSIM_PUSH_OBS_PRIVILEGE = 2 };      // include/tsim_env.h tsim_push_policy.obs_mode

template <class R> struct PushPolicy {
  const R *W1T, *b1, *W2T, *b2, *W3, *b3;      // forward layouts
  const R *W1p, *W2; int w1s;                  // backward layouts (row stride of W1p)
  int mode, nin, nin_pad;                      // observation type (TSIM_PUSH_OBS_*), its length (393 / 3 / 6), padded to a multiple of PP_ROWS1
  // roll-out collection (PPO): the stochastic policy u = mean + exp(logstd) eps, observations normalised with FIXED statistics
  const R* eps;                                // [T][B][3] standard-normal draws, or null (deterministic: gd.py)
  const R* logstd;                             // [3]
  const R *obs_mean, *obs_istd; R obs_clip;    // [nin] each: x -> clamp((x - mean) * istd, +-clip), or null (forward-only launches)
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


enum { PP_OBS_PAD = 400, PP_ROWS1 = 40, PP_ROWS2 = 32, PP_JB = 8, PP_OCH = (PP_OBS + 63) / 64 };
// LDS scratch of the policy, per slot: the pair-staging records PP + PT (4 x 36 + 4 nd x 18 = 648 reals on the TactilePush model), idle
// between two frames and rewritten whole by every staging: xs = the first 448 reals, hs = the next 128.  (NOT the link tangent records:
// their entries for directions that do not move a link are zero from the start of the launch and never rewritten.)  x0 / h0 = slot 0's
// copy, slots are `stride` reals apart.
template <int LPE, class R> struct PPScr { R* xs; R* hs; R* xs0; R* hs0; int stride; };
template <int LPE, class R> __device__ __forceinline__ PPScr<LPE, R> pp_scr(const Ctx<R>& c) {
  PPScr<LPE, R> S;
  S.stride = ts_lds_env_reals(c.nl, c.nr, c.nu, (int)sizeof(R));
  const int slot = (int)threadIdx.x / LPE;
  S.xs = c.PP; S.hs = c.PP + 64 * PP_OCH; S.xs0 = S.xs - slot * S.stride; S.hs0 = S.hs - slot * S.stride;
  return S;
}
// acc[s] += sum_{i < nrows} W[min(i, wrows - 1)][j] * x_s[i]   (j = lane of the wavefront; x_s = x0 + s * stride in LDS; nrows a multiple
// of ROWS — rows past wrows meet zero inputs)
template <int NS, int ROWS, class R>
__device__ __forceinline__ void pp_dense64(const R* W, int nrows, int wrows, const R* x0, int stride, R (&acc)[NS]) {
  const R* Wj = W + threadIdx.x;
  for (int i0 = 0; i0 < nrows; i0 += ROWS) {
    R wb[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) wb[r] = Wj[(size_t)min(i0 + r, wrows - 1) * PP_HID];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s] += wb[r] * x0[s * stride + i0 + r];
  }
}

// MFMA (fp32, four environments per wavefront; every other shape and fp64 take the vector-ALU form pp_dense64).  The dense layers are the one contraction-shaped piece of this
// path with a non-trivial K: out[unit j][env s] = sum_k W[k][j] x_s[k], i.e. per wavefront a (64 x K) (K x 4) product, K = 393 / 64.
// v_mfma_f32_4x4x1_16b_f32 does one k of it per instruction: 16 blocks of (4 x 1)(1 x 4); block b takes A from lanes 4b .. 4b+3 (lane l: W[k][l],
// exactly what the coalesced row load leaves in the lanes) and B from the same lanes (lane l: x_{l % 4}[k], one LDS read with a per-lane
// address), and lane 4b + s accumulates D_b[0..3][s] = units 4b .. 4b+3 of environment s in four registers.  Per weight row: 1 global load,
// 1 LDS read, 1 MFMA (8 cycles) against 1 load, 4 broadcast reads and 4 FMAs (16 cycles) on the vector ALU.  Full fp32 FMAs, another
// summation order.  Result layout differs from pp_dense64 (lane = unit): lane 4b + s holds ITS environment's units 4b + i.
// Measured: profiles/r04_mfma_ab.md (closed GD epoch 55.8 -> 55.4 ms, +0.8 %: kept).
typedef float pp_v4f __attribute__((ext_vector_type(4)));
template <int ROWS>
__device__ __forceinline__ pp_v4f pp_dense64_mfma(const float* W, int nrows, int wrows, const float* x0, int stride) {
  const float* Wj = W + threadIdx.x;
  const float* xl = x0 + ((int)threadIdx.x & 3) * stride;
  pp_v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;      // four independent accumulation chains: no MFMA waits for the one before it
  for (int i0 = 0; i0 < nrows; i0 += ROWS) {
    float wb[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) wb[r] = Wj[(size_t)min(i0 + r, wrows - 1) * PP_HID];
#pragma unroll
    for (int r = 0; r < ROWS; r += 4) {
      d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[r], xl[i0 + r], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[r + 1], xl[i0 + r + 1], d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[r + 2], xl[i0 + r + 2], d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[r + 3], xl[i0 + r + 3], d3, 0, 0, 0);
    }
  }
  return (d0 + d1) + (d2 + d3);
}

// Observation -> action for the environment of this slot.  tac_prev: the tactile frame the observation is built from (global; written by
// this slot, hence the fence + bypassing loads).  q (double, the state before the frame) gives the goal in the gripper frame.
// Writes c.u (the 6 actuator inputs of the frame) and the forward records.
template <int LPE, class R>
__device__ __forceinline__ void push_policy_forward(const Ctx<R>& c, int lane, bool valid, const PushPolicy<R>& P, size_t rec /* f * B + env */, int env, const R* tac_prev) {
  constexpr int OPL = PP_HID / LPE;                    // hidden units per lane
  // goal pose in the gripper frame: rotation by -yaw, then the gripper's position is subtracted (tactile_push_env.py:84-92)
  R gl[3];
  {
    const R* g = P.goal + (size_t)env * 3;
    double sn, cs; t_sincos_d(c.q0D[0], sn, cs);
    const double gx = g[0], gy = g[1];
    gl[0] = (R)(cs * gx + sn * gy - c.q0D[1]); gl[1] = (R)(-sn * gx + cs * gy - c.q0D[2]); gl[2] = (R)((double)g[2] - c.q0D[0]);
  }
  constexpr int NS = TS_WAVE / LPE;
  const PPScr<LPE, R> S = pp_scr<LPE>(c);
  const int mode = ts_u(P.mode);
  if (mode == TSIM_PUSH_OBS_TACTILE) {
    // the observation of this slot into LDS: goal (3), the tactile frame (390; every lane fetches its share in one batch of loads), zeros
    constexpr int NX = (PP_OBS_PAD - 3 + LPE - 1) / LPE;
    R xv[NX];
#pragma unroll
    for (int m = 0; m < NX; ++m) { const int i = lane + LPE * m; xv[m] = i < PP_NTAC ? __builtin_nontemporal_load(tac_prev + i) : R(0); }
    if (lane < 3) S.xs[lane] = lane == 0 ? gl[0] : (lane == 1 ? gl[1] : gl[2]);
#pragma unroll
    for (int m = 0; m < NX; ++m) { const int i = lane + LPE * m; if (3 + i < PP_OBS_PAD) S.xs[3 + i] = xv[m]; }
  } else {
    // no_tactile: the goal (3); privilege: the box pose in the gripper frame (3), then the goal (tactile_push_env.py:104-131)
    for (int i = lane; i < P.nin_pad; i += LPE) S.xs[i] = R(0);
    const int go = mode == TSIM_PUSH_OBS_PRIVILEGE ? 3 : 0;
    if (lane < 3) S.xs[go + lane] = lane == 0 ? gl[0] : (lane == 1 ? gl[1] : gl[2]);
    if (mode == TSIM_PUSH_OBS_PRIVILEGE && lane < 3) {
      double sn, cs; t_sincos_d(c.q0D[0], sn, cs);
      const double ox = c.q0D[3], oy = c.q0D[4];
      S.xs[lane] = lane == 0 ? (R)(cs * ox + sn * oy - c.q0D[1]) : (lane == 1 ? (R)(-sn * ox + cs * oy - c.q0D[2]) : (R)(c.q0D[6] - c.q0D[0]));
    }
  }
  if (P.obs_mean) {                                    // VecNormalize with frozen statistics (a2c_ppo_acktr envs.py: norm_obs, clip_obs)
    TS_SYNC();
    const int nin = ts_u(P.nin);
    for (int i = lane; i < nin; i += LPE) S.xs[i] = t_min(t_max((S.xs[i] - P.obs_mean[i]) * P.obs_istd[i], -P.obs_clip), P.obs_clip);
  }
  TS_SYNC();
  R acc[NS], w[OPL];
  constexpr bool kMfma = sizeof(R) == 4 && NS == 4;
  if constexpr (kMfma) {
    const pp_v4f d = pp_dense64_mfma<PP_ROWS1>((const float*)P.W1T, ts_u(P.nin_pad), ts_u(P.nin), (const float*)S.xs0, S.stride);
    const int u0 = (int)threadIdx.x & ~3, es = (int)threadIdx.x & 3;          // this lane: units u0 .. u0 + 3 of environment es
#pragma unroll
    for (int i = 0; i < 4; ++i) S.hs0[es * S.stride + u0 + i] = pp_elu((R)d[i] + P.b1[u0 + i]);
  } else {
    {
      const R bj = P.b1[threadIdx.x];
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) acc[s_] = bj;
    }
    pp_dense64<NS, PP_ROWS1>(P.W1T, ts_u(P.nin_pad), ts_u(P.nin), S.xs0, S.stride, acc);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) S.hs0[s_ * S.stride + threadIdx.x] = pp_elu(acc[s_]);
  }
  TS_SYNC();
  R h1[OPL];
  pp_ld<OPL>(S.hs + OPL * lane, h1);
  if (valid) {
#pragma unroll
    for (int o = 0; o < OPL; ++o) P.h1_out[rec * PP_HID + OPL * lane + o] = h1[o];
  }
  if constexpr (kMfma) {
    const pp_v4f d = pp_dense64_mfma<PP_ROWS2>((const float*)P.W2T, PP_HID, PP_HID, (const float*)S.hs0, S.stride);
    const int u0 = (int)threadIdx.x & ~3, es = (int)threadIdx.x & 3;
    TS_SYNC();
#pragma unroll
    for (int i = 0; i < 4; ++i) S.hs0[es * S.stride + PP_HID + u0 + i] = pp_elu((R)d[i] + P.b2[u0 + i]);
  } else {
    {
      const R bj = P.b2[threadIdx.x];
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) acc[s_] = bj;
    }
    pp_dense64<NS, PP_ROWS2>(P.W2T, PP_HID, PP_HID, S.hs0, S.stride, acc);
    TS_SYNC();
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) S.hs0[s_ * S.stride + PP_HID + threadIdx.x] = pp_elu(acc[s_]);
  }
  TS_SYNC();
  R h2[OPL];
  pp_ld<OPL>(S.hs + PP_HID + OPL * lane, h2);
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
    if (P.eps) up[a_] += (R)exp((double)P.logstd[a_]) * P.eps[rec * 3 + a_];       // the sampled action; u_out records it
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
// read-out) and returns, in lanes 0..6, what the state part of the observation (goal; privilege: box pose) puts on q of the state before the frame.
template <int LPE, class R>
__device__ __forceinline__ R push_policy_backward(const Ctx<R>& c, int lane, bool valid, const PushPolicy<R>& P, size_t rec, int env, R da, const R* qb /* state before the frame */) {
  const double yaw = (double)qb[0];
  constexpr int OPL = PP_HID / LPE;
  constexpr int NS = TS_WAVE / LPE;
  const PPScr<LPE, R> S = pp_scr<LPE>(c);
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
  for (int o = 0; o < OPL; ++o) { g2[o] *= pp_elu_grad_from_output(h[o]); S.hs[OPL * lane + o] = g2[o]; }
  if (valid) {
#pragma unroll
    for (int o = 0; o < OPL; ++o) P.g2_out[rec * PP_HID + OPL * lane + o] = g2[o];
  }
  TS_SYNC();
  R g1[OPL];
  {                                                    // dh1_s[o] = sum_j W2[j][o] g2_s[j], wave-wide (o = lane of the wavefront)
    R a2[NS];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) a2[s_] = R(0);
    pp_dense64<NS, PP_ROWS2>(P.W2, PP_HID, PP_HID, S.hs0, S.stride, a2);
    TS_SYNC();
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) S.hs0[s_ * S.stride + PP_HID + threadIdx.x] = a2[s_];
    TS_SYNC();
    pp_ld<OPL>(S.hs + PP_HID + OPL * lane, g1);
  }
  pp_ld<OPL>(P.h1_out + rec * PP_HID + OPL * lane, h);
  TS_SYNC();
#pragma unroll
  for (int o = 0; o < OPL; ++o) { g1[o] *= pp_elu_grad_from_output(h[o]); S.hs[PP_HID + OPL * lane + o] = g1[o]; }
  if (valid) {
#pragma unroll
    for (int o = 0; o < OPL; ++o) P.g1_out[rec * PP_HID + OPL * lane + o] = g1[o];
  }
  TS_SYNC();
  // d obs_s[i] = sum_j W1[j][i] g1_s[j], wave-wide: lane l of the wavefront owns i = l + 64 m
  {
    R dob[PP_OCH][NS];
#pragma unroll
    for (int m = 0; m < PP_OCH; ++m)
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) dob[m][s_] = R(0);
    const R* g0 = S.hs0 + PP_HID;
    const int w1s = ts_u(P.w1s), och = (w1s + 63) / 64;          // 64-wide chunks of the observation (7 with the tactile frame, 1 without)
    for (int j0 = 0; j0 < PP_HID; j0 += PP_JB) {
      R wb[PP_JB][PP_OCH];
#pragma unroll
      for (int r = 0; r < PP_JB; ++r)
#pragma unroll
        for (int m = 0; m < PP_OCH; ++m) if (m < och) wb[r][m] = P.W1p[(size_t)(j0 + r) * w1s + min((int)threadIdx.x + 64 * m, w1s - 1)];
#pragma unroll
      for (int r = 0; r < PP_JB; ++r)
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
          const R x = g0[s_ * S.stride + j0 + r];
#pragma unroll
          for (int m = 0; m < PP_OCH; ++m) if (m < och) dob[m][s_] += wb[r][m] * x;
        }
    }
    TS_SYNC();
#pragma unroll
    for (int m = 0; m < PP_OCH; ++m)
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) if (m < och) S.xs0[s_ * S.stride + (int)threadIdx.x + 64 * m] = dob[m][s_];
    TS_SYNC();
  }
  const int mode = ts_u(P.mode);
  // tactile part -> the seed of the previous frame's tactile read-out (consecutive lanes, consecutive addresses); goal part -> q[0..2]
  if (valid && mode == TSIM_PUSH_OBS_TACTILE) {
    for (int i = lane; i < PP_NTAC; i += LPE) P.dobs_tac[rec * PP_NTAC + i] = S.xs[3 + i];
  }
  const int go = mode == TSIM_PUSH_OBS_PRIVILEGE ? 3 : 0;
  const R d0 = S.xs[go], d1 = S.xs[go + 1], d2 = S.xs[go + 2];
  const R e0 = S.xs[0], e1 = S.xs[1], e2 = S.xs[2];              // privilege: gradient w.r.t. the box pose in the gripper frame
  TS_SYNC();
  double sn, cs; t_sincos_d(yaw, sn, cs);
  const R* g = P.goal + (size_t)env * 3;
  const double gx = g[0], gy = g[1];
  R out = R(0);
  if (lane == 0) out = (R)((double)d0 * (-sn * gx + cs * gy) + (double)d1 * (-cs * gx - sn * gy) - (double)d2);
  else if (lane == 1) out = -d0;
  else if (lane == 2) out = -d1;
  if (mode == TSIM_PUSH_OBS_PRIVILEGE) {             // obj = (cs ox + sn oy - q1, -sn ox + cs oy - q2, q6 - q0)
    const double ox = (double)qb[3], oy = (double)qb[4];
    if (lane == 0) out += (R)((double)e0 * (-sn * ox + cs * oy) + (double)e1 * (-cs * ox - sn * oy) - (double)e2);
    else if (lane == 1) out -= e0;
    else if (lane == 2) out -= e1;
    else if (lane == 3) out = (R)((double)e0 * cs - (double)e1 * sn);
    else if (lane == 4) out = (R)((double)e0 * sn + (double)e1 * cs);
    else if (lane == 6) out = e2;
  }
  return out;
}
