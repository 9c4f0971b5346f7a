// TactilePush per-step formulas as fused element-wise kernels (include/tsim_env.h).  Included at the end of tsim_hip.hip.
// HBM-bound and tiny: one env-step moves ~3.2 KB per environment (13 MB at B = 4096, ~3 us at HBM rate), so these are
// launch-latency kernels; the point is ONE launch instead of ~45 each way.  Flat thread -> (environment, column) so that the
// observation rows are written / the tactile gradient rows are read fully coalesced; the 3 + 1 closed-form columns are
// computed by the first threads of each row.
#pragma once

template <class R>
__global__ void __launch_bounds__(256) k_push_action(int B, const R* __restrict__ u, const R* __restrict__ ext, R* __restrict__ a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * 6) return;
  const int e = i / 6, c = i - e * 6;
  a[i] = c < 3 ? (R)tanh((double)u[e * 3 + c]) : (c < 5 ? ext[e * 2 + c - 3] : (R)0);
}

template <class R>
__global__ void __launch_bounds__(256) k_push_action_bwd(int B, const R* __restrict__ u, const R* __restrict__ da, R* __restrict__ du) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * 3) return;
  const int e = i / 3, c = i - e * 3;
  const double t = tanh((double)u[i]);
  du[i] = (R)((double)da[e * 6 + c] * (1.0 - t * t));
}

template <class R>
__global__ void __launch_bounds__(256) k_push_observe(int B, int ntac, const R* __restrict__ q, const R* __restrict__ var,
                                                      const R* __restrict__ tac, const R* __restrict__ goal, const R* __restrict__ u,
                                                      R* __restrict__ obs, R* __restrict__ rew) {
  const int W = 3 + ntac;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * W) return;
  const int e = (int)(i / W), c = (int)(i - (long long)e * W);
  if (c >= 3) { obs[i] = tac[(long long)e * ntac + c - 3]; return; }
  const R* qe = q + e * 7; const R* g = goal + e * 3;
  const double th = qe[0], cs = cos(th), sn = sin(th), gx = g[0], gy = g[1];
  // goal pose in the gripper frame: rotation by -yaw, then the gripper's position is subtracted (tactile_push_env.py:84-92)
  const double v = c == 0 ? cs * gx + sn * gy - (double)qe[1] : (c == 1 ? -sn * gx + cs * gy - (double)qe[2] : (double)g[2] - th);
  obs[i] = (R)v;
  if (c == 0 && rew) {
    const double dx = ((double)qe[3] - gx) * 100.0, dy = ((double)qe[4] - gy) * 100.0, dr = ((double)qe[6] - (double)g[2]) * (36.0 / M_PI);
    double t2 = 0, a2 = 0;
    for (int k = 0; k < 3; ++k) { const double d = (double)var[e * 6 + k] - (double)var[e * 6 + 3 + k]; t2 += d * d; a2 += (double)u[e * 3 + k] * (double)u[e * 3 + k]; }
    rew[e] = (R)(-(dx * dx + dy * dy) * 0.01 - dr * dr * 0.1 - t2 * 2500.0 - a2 * 0.1);
  }
}

template <class R>
__global__ void __launch_bounds__(256) k_push_observe_bwd(int B, int ntac, const R* __restrict__ q, const R* __restrict__ var,
                                                          const R* __restrict__ goal, const R* __restrict__ u, const R* __restrict__ dobs,
                                                          const R* __restrict__ drew, long long drs, R* __restrict__ dq, R* __restrict__ dvar,
                                                          R* __restrict__ dtac, R* __restrict__ du) {
  const int W = 3 + ntac;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * W) return;
  const int e = (int)(i / W), c = (int)(i - (long long)e * W);
  if (c >= 3) { dtac[(long long)e * ntac + c - 3] = dobs[i]; return; }
  if (c != 0) return;
  const R* qe = q + e * 7; const R* g = goal + e * 3;
  const double th = qe[0], cs = cos(th), sn = sin(th), gx = g[0], gy = g[1];
  const double d0 = dobs[i], d1 = dobs[i + 1], d2 = dobs[i + 2];
  double o[7] = {d0 * (-sn * gx + cs * gy) + d1 * (-cs * gx - sn * gy) - d2, -d0, -d1, 0, 0, 0, 0};
  if (drew) {
    const double r = drew[(long long)e * drs];
    o[3] = -200.0 * ((double)qe[3] - gx) * r;                                   // -0.01 * 2 (q3 - gx) / 0.01^2
    o[4] = -200.0 * ((double)qe[4] - gy) * r;
    o[6] = -0.2 * ((double)qe[6] - (double)g[2]) * (36.0 / M_PI) * (36.0 / M_PI) * r;
    for (int k = 0; k < 3; ++k) {
      const double d = (double)var[e * 6 + k] - (double)var[e * 6 + 3 + k];
      dvar[e * 6 + k] = (R)(-5000.0 * d * r); dvar[e * 6 + 3 + k] = (R)(5000.0 * d * r);
      du[e * 3 + k] = (R)(-0.2 * (double)u[e * 3 + k] * r);
    }
  }
  for (int k = 0; k < 7; ++k) dq[e * 7 + k] = (R)o[k];
}

static inline unsigned push_blocks(long long n) { return (unsigned)((n + 255) / 256); }
#define TS_PUSH_ARGS_OK(B, dtype) do { if ((B) <= 0) return fail("push env: B must be > 0"); \
    if ((dtype) != TSIM_F32 && (dtype) != TSIM_F64) return fail("push env: dtype must be TSIM_F32 or TSIM_F64"); } while (0)

extern "C" int tsim_push_action(int B, int dtype, const void* u, const void* ext, void* action, void* stream) {
  TS_PUSH_ARGS_OK(B, dtype);
  if (!u || !ext || !action) return fail("tsim_push_action: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TSIM_F32) hipLaunchKernelGGL(k_push_action<float>, dim3(push_blocks((long long)B * 6)), dim3(256), 0, st, B, (const float*)u, (const float*)ext, (float*)action);
  else hipLaunchKernelGGL(k_push_action<double>, dim3(push_blocks((long long)B * 6)), dim3(256), 0, st, B, (const double*)u, (const double*)ext, (double*)action);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int tsim_push_action_backward(int B, int dtype, const void* u, const void* d_action, void* du, void* stream) {
  TS_PUSH_ARGS_OK(B, dtype);
  if (!u || !d_action || !du) return fail("tsim_push_action_backward: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TSIM_F32) hipLaunchKernelGGL(k_push_action_bwd<float>, dim3(push_blocks((long long)B * 3)), dim3(256), 0, st, B, (const float*)u, (const float*)d_action, (float*)du);
  else hipLaunchKernelGGL(k_push_action_bwd<double>, dim3(push_blocks((long long)B * 3)), dim3(256), 0, st, B, (const double*)u, (const double*)d_action, (double*)du);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int tsim_push_observe(int B, int ntac, int dtype, const void* q, const void* var, const void* tactile, const void* goal,
                                 const void* u, void* obs, void* rew, void* stream) {
  TS_PUSH_ARGS_OK(B, dtype);
  if (ntac < 0) return fail("tsim_push_observe: ntac < 0");
  if (!q || !goal || !obs || (ntac && !tactile)) return fail("tsim_push_observe: null pointer");
  if (rew && (!var || !u)) return fail("tsim_push_observe: the reward needs var and u");
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = push_blocks((long long)B * (3 + ntac));
  if (dtype == TSIM_F32) hipLaunchKernelGGL(k_push_observe<float>, dim3(nb), dim3(256), 0, st, B, ntac, (const float*)q, (const float*)var, (const float*)tactile, (const float*)goal, (const float*)u, (float*)obs, (float*)rew);
  else hipLaunchKernelGGL(k_push_observe<double>, dim3(nb), dim3(256), 0, st, B, ntac, (const double*)q, (const double*)var, (const double*)tactile, (const double*)goal, (const double*)u, (double*)obs, (double*)rew);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int tsim_push_observe_backward(int B, int ntac, int dtype, const void* q, const void* var, const void* goal, const void* u,
                                          const void* d_obs, const void* d_rew, long long d_rew_stride,
                                          void* dq, void* dvar, void* dtac, void* du, void* stream) {
  TS_PUSH_ARGS_OK(B, dtype);
  if (ntac < 0) return fail("tsim_push_observe_backward: ntac < 0");
  if (!q || !goal || !d_obs || !dq || (ntac && !dtac)) return fail("tsim_push_observe_backward: null pointer");
  if (d_rew && (!var || !u || !dvar || !du)) return fail("tsim_push_observe_backward: the reward gradient needs var, u, dvar and du");
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = push_blocks((long long)B * (3 + ntac));
  if (dtype == TSIM_F32) hipLaunchKernelGGL(k_push_observe_bwd<float>, dim3(nb), dim3(256), 0, st, B, ntac, (const float*)q, (const float*)var, (const float*)goal, (const float*)u, (const float*)d_obs, (const float*)d_rew, d_rew_stride, (float*)dq, (float*)dvar, (float*)dtac, (float*)du);
  else hipLaunchKernelGGL(k_push_observe_bwd<double>, dim3(nb), dim3(256), 0, st, B, ntac, (const double*)q, (const double*)var, (const double*)goal, (const double*)u, (const double*)d_obs, (const double*)d_rew, d_rew_stride, (double*)dq, (double*)dvar, (double*)dtac, (double*)du);
  HIPCHK(hipGetLastError());
  return 0;
}


// ================================================================================================ closed loop in one launch each way
// (tsim_policy_push.h: the policy between the frames of k_forward / k_backward)
static int push_obs_len(int mode) { return mode == TSIM_PUSH_OBS_TACTILE ? (int)PP_OBS : (mode == TSIM_PUSH_OBS_NO_TACTILE ? 3 : (mode == TSIM_PUSH_OBS_PRIVILEGE ? 6 : -1)); }
template <class R>
static PushPolicy<R> make_push_policy(const tsim_push_policy* p) {
  PushPolicy<R> P;
  memset(&P, 0, sizeof(P));
  P.W1T = (const R*)p->W1T; P.b1 = (const R*)p->b1; P.W2T = (const R*)p->W2T; P.b2 = (const R*)p->b2; P.W3 = (const R*)p->W3; P.b3 = (const R*)p->b3;
  P.W1p = (const R*)p->W1p; P.W2 = (const R*)p->W2; P.w1s = p->w1_stride;
  P.eps = (const R*)p->eps; P.logstd = (const R*)p->logstd; P.obs_mean = (const R*)p->obs_mean; P.obs_istd = (const R*)p->obs_istd; P.obs_clip = (R)p->obs_clip;
  P.mode = p->obs_mode; P.nin = push_obs_len(p->obs_mode); P.nin_pad = (P.nin + PP_ROWS1 - 1) / PP_ROWS1 * PP_ROWS1;
  return P;
}
static int push_closed_check(const tsim_batch* b, const tsim_push_policy* pol, int num_frames, int num_steps, const char* who) {
  if (!b || !pol) return fail(std::string(who) + ": null batch / policy");
  if (b->nr != 7 || b->nu != 6 || 3 * b->ntax != PP_NTAC || b->nvar != 2 || b->has_exp) return fail(std::string(who) + ": not the TactilePush model (ndof_r 7, ndof_u 6, 130 taxels, 2 end-effector points)");
  if (num_frames <= 0 || num_steps <= 0) return fail(std::string(who) + ": num_frames and num_steps must be positive");
  if (TS_PAIR_GROUP * ((int)PP_SIZE + b->nr * (int)PT_SIZE) < 64 * (int)PP_OCH + 2 * (int)PP_HID) return fail(std::string(who) + ": the pair-staging records are too small for the policy scratch");
  if (!pol->W1T || !pol->b1 || !pol->W2T || !pol->b2 || !pol->W3 || !pol->b3) return fail(std::string(who) + ": policy weights missing");
  if (push_obs_len(pol->obs_mode) < 0) return fail(std::string(who) + ": obs_mode must be 0 (tactile_flatten), 1 (no_tactile) or 2 (privilege)");
  if (pol->eps && !pol->logstd) return fail(std::string(who) + ": eps without logstd");
  if ((pol->obs_mean != nullptr) != (pol->obs_istd != nullptr)) return fail(std::string(who) + ": obs_mean and obs_istd come together");
  if (pol->obs_mean && b->record) return fail(std::string(who) + ": observation normalisation is for roll-out collection (reset with backward_flag = False); the adjoint launch does not undo it");
  return 0;
}
#define TS_LAUNCH_POLICY(KERNEL, R, b, st, a) do {                                                                                   \
    const LaunchShape L = launch_shape(b);                                                                                           \
    if constexpr (sizeof(R) == 4) {      /* the statically specialised TactilePush instantiation (tsim_static_pusher.hip) */           \
      if (kernel_mode(b) == TS_KM_STATIC && L.lpe == 16) { ts_static_pusher_launch_policy(a, L.grid, L.lds, st); break; }               \
      if (kernel_mode(b) == TS_KM_PARAM && L.lpe == 16) { ts_param_pusher_launch_policy(a, L.grid, L.lds, st); break; }                 \
    }                                                                                                                                 \
    if (L.lpe == 64) hipLaunchKernelGGL((KERNEL<R, 8, false, 64, true>), dim3(L.grid), dim3(TS_WAVE), L.lds, st, a);                  \
    else if (L.lpe == 32) hipLaunchKernelGGL((KERNEL<R, 8, false, 32, true>), dim3(L.grid), dim3(TS_WAVE), L.lds, st, a);             \
    else hipLaunchKernelGGL((KERNEL<R, 8, false, 16, true>), dim3(L.grid), dim3(TS_WAVE), L.lds, st, a);                              \
  } while (0)

template <class R>
static int push_closed_rollout_t(tsim_batch* b, const tsim_push_policy* pol, const void* goal, const void* dist, const void* tac0, int nframes, int nsub,
                                 void* q_out, void* qd_out, void* var_out, void* tac_out, void* u_out, void* gl_out, void* h1_out, void* h2_out, int32_t* status, hipStream_t st) {
  FwdArgs<R> a;
  memset(&a, 0, sizeof(a));
  a.I = b->dI; a.F = (const R*)b->dF; a.Fenv = (const R*)b->dFenv; a.fstride = b->nfrec; a.B = b->B; a.nsub = nsub; a.record = b->record; a.t0 = b->t_cur; a.nframes = nframes; a.tac_slot = nullptr;
  a.tape = (R*)b->tape; a.u = nullptr;
  a.q_out = (R*)q_out; a.qd_out = (R*)qd_out; a.var_out = (R*)var_out; a.tac_out = (R*)tac_out; a.status = status; a.evals = b->evals; a.order = nullptr;
  a.prev = (double*)b->prev; a.has_prev = b->has_prev; a.stage_cpt = b->stage_cpt;
  a.cross_kinks = b->cross_kinks; a.eval_budget = b->eval_budget; a.gnorm = b->gnorm; a.cull = b->pair_cull; a.vo_ls = b->value_trials;
  a.default_opts = default_options(b) ? 1 : 0;      // (helpers / value-first trials are off in closed-loop launches whatever the options say: k_forward)
  a.pol = make_push_policy<R>(pol);
  a.pol.goal = (const R*)goal; a.pol.dist = (const R*)dist; a.pol.tac0 = (const R*)tac0;
  a.pol.u_out = (R*)u_out; a.pol.gl_out = (R*)gl_out; a.pol.h1_out = (R*)h1_out; a.pol.h2_out = (R*)h2_out;
  TS_LAUNCH_POLICY(k_forward, R, b, st, a);
  HIPCHK(hipGetLastError());
  b->order_valid = 0; pose_invalidate(b, st);
  return 0;
}

extern "C" int tsim_push_closed_rollout(tsim_batch* b, const tsim_push_policy* pol, const void* goal, const void* dist, const void* tac0,
                                        int num_frames, int num_steps, void* q_out, void* qd_out, void* var_out, void* tac_out,
                                        void* u_out, void* gl_out, void* h1_out, void* h2_out, int32_t* status, void* stream) {
  if (int rc = push_closed_check(b, pol, num_frames, num_steps, "push_closed_rollout")) return rc;
  if (!goal || !dist || !u_out || !gl_out || !h1_out || !h2_out) return fail("push_closed_rollout: null pointer (goal, dist and the policy records are required)");
  if (pol->obs_mode == TSIM_PUSH_OBS_TACTILE && (!tac0 || !tac_out)) return fail("push_closed_rollout: tac0 and tac_out are required with the tactile observation (they feed it)");
  if (b->record && b->t_cur + (long long)num_frames * num_steps > b->cap) return fail("push_closed_rollout: tape capacity exceeded (" + std::to_string(b->cap) + " sub-steps)");
  TS_DEVICE(b);
  int rc = b->dtype == TSIM_F32 ? push_closed_rollout_t<float>(b, pol, goal, dist, tac0, num_frames, num_steps, q_out, qd_out, var_out, tac_out, u_out, gl_out, h1_out, h2_out, status, (hipStream_t)stream)
                                : push_closed_rollout_t<double>(b, pol, goal, dist, tac0, num_frames, num_steps, q_out, qd_out, var_out, tac_out, u_out, gl_out, h1_out, h2_out, status, (hipStream_t)stream);
  if (rc) return rc;
  if (b->record) b->t_cur += num_frames * num_steps;
  b->has_prev = 1;
  return 0;
}

template <class R>
static int push_closed_backward_t(tsim_batch* b, const tsim_push_policy* pol, const void* goal, int nframes, int nsub, const void* df_dq, const void* df_dvar,
                                  const void* du_direct, const void* u_out, const void* h1_out, const void* h2_out, void* g1_out, void* g2_out, void* g3_out,
                                  void* dobs_tac, void* df_du, hipStream_t st) {
  BwdArgs<R> a;
  memset(&a, 0, sizeof(a));
  a.I = b->dI; a.F = (const R*)b->dF; a.Fenv = (const R*)b->dFenv; a.fstride = b->nfrec; a.B = b->B; a.n = nframes * nsub; a.t_end = b->t_cur; a.seed_stride = nsub; a.frames = 1; a.tac_slot = nullptr;
  a.tape = (const R*)b->tape; a.df_dq = (const R*)df_dq; a.df_dvar = (const R*)df_dvar; a.df_dtac = nullptr;
  a.lamq = (R*)b->lamq; a.lamv = (R*)b->lamv; a.df_du = (R*)df_du; a.stage_cpt = b->stage_cpt; a.cyc = nullptr; a.cull = b->pair_cull;
  a.pol = make_push_policy<R>(pol);
  a.pol.goal = (const R*)goal; a.pol.du_direct = (const R*)du_direct;
  a.pol.u_out = (R*)u_out; a.pol.h1_out = (R*)h1_out; a.pol.h2_out = (R*)h2_out;
  a.pol.g1_out = (R*)g1_out; a.pol.g2_out = (R*)g2_out; a.pol.g3_out = (R*)g3_out; a.pol.dobs_tac = (R*)dobs_tac;
  TS_LAUNCH_POLICY(k_backward, R, b, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int tsim_push_closed_backward(tsim_batch* b, const tsim_push_policy* pol, const void* goal, int num_frames, int num_steps,
                                         const void* df_dq, const void* df_dvar, const void* du_direct, const void* u_out, const void* h1_out, const void* h2_out,
                                         void* g1_out, void* g2_out, void* g3_out, void* dobs_tac, void* df_du, void* stream) {
  if (int rc = push_closed_check(b, pol, num_frames, num_steps, "push_closed_backward")) return rc;
  if (!pol->W1p || !pol->W2 || pol->w1_stride < push_obs_len(pol->obs_mode) || pol->w1_stride % 4 || pol->w1_stride > 64 * (int)PP_OCH) return fail("push_closed_backward: W1p [64][w1_stride >= observation length, multiple of 4] and W2 are required");
  if (!b->record) return fail("push_closed_backward: reset(backward_flag=True) was not called");
  const long long n = (long long)num_frames * num_steps;
  if (n > b->t_cur) return fail("push_closed_backward: only " + std::to_string(b->t_cur) + " sub-steps on the tape");
  if (!goal || !du_direct || !u_out || !h1_out || !h2_out || !g1_out || !g2_out || !g3_out) return fail("push_closed_backward: null pointer");
  if (pol->obs_mode == TSIM_PUSH_OBS_TACTILE && !dobs_tac) return fail("push_closed_backward: dobs_tac is required with the tactile observation");
  TS_DEVICE(b);
  int rc = b->dtype == TSIM_F32 ? push_closed_backward_t<float>(b, pol, goal, num_frames, num_steps, df_dq, df_dvar, du_direct, u_out, h1_out, h2_out, g1_out, g2_out, g3_out, dobs_tac, df_du, (hipStream_t)stream)
                                : push_closed_backward_t<double>(b, pol, goal, num_frames, num_steps, df_dq, df_dvar, du_direct, u_out, h1_out, h2_out, g1_out, g2_out, g3_out, dobs_tac, df_du, (hipStream_t)stream);
  if (rc) return rc;
  b->t_cur -= (int)n;
  pose_invalidate(b, (hipStream_t)stream);
  return 0;
}
