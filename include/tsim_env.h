/* tsim_env.h — the TactilePush environment's per-step formulas, batched and fused (C ABI, libtsim_hip.so).
 *
 * Around every simulator step the reference's environment evaluates a handful of closed-form expressions per environment
 * in torch (envs/tactile_push_env.py: action mapping :175-193, observation :72-131, reward :202-211).  Batched, those
 * were ~90 element-wise launches per env-step including their autograd twins — more GPU time than the simulator's own
 * two launches (profiles/r02_closed_loop_kernels.md).  Here each direction is ONE launch.  Plain device pointers, [B][dim]
 * C order, dtype TSIM_F32 / TSIM_F64 (include/tsim.h); no batch handle: these are pure functions.  Return 0 or non-zero
 * with tsim_last_error().
 *
 * TactilePush layout (assets/tactile_push.xml through the model compiler): q[7] = gripper (yaw, x, y), box (x, y, z-rot
 * chain …, yaw at 6); var[6] = two 3-vectors whose distance is the "touch" term; tactile[ntac] flattened shear/normal
 * taxel values; goal[3] = (x, y, yaw); u[3] = policy output before tanh. */
#ifndef TSIM_ENV_H
#define TSIM_ENV_H
#ifdef __cplusplus
extern "C" {
#endif

/* robot_action = [tanh(u), external_force, 0]                      envs/tactile_push_env.py:175-193
 *   u [B][3], ext [B][2] -> action [B][6] */
int tsim_push_action(int B, int dtype, const void* u, const void* ext, void* action, void* stream);
/* du = d_action[:, 0:3] * (1 - tanh(u)^2) */
int tsim_push_action_backward(int B, int dtype, const void* u, const void* d_action, void* du, void* stream);

/* obs = [goal in the gripper frame (3), tactile (ntac)]            envs/tactile_push_env.py:84-114
 * rew = r_pos + r_rot + r_touch + r_act                            envs/tactile_push_env.py:202-211
 *   r_pos = -0.01 |(q[3:5] - goal[0:2]) / 0.01|^2,  r_rot = -0.1 ((q[6] - goal[2]) / (pi/36))^2,
 *   r_touch = -|var[0:3] - var[3:6]|^2 / 0.02^2,    r_act = -0.1 |u|^2
 * obs [B][3 + ntac]; rew [B] or NULL (then var and u may be NULL too: the observation after reset). */
int tsim_push_observe(int B, int ntac, int dtype, const void* q, const void* var, const void* tactile, const void* goal,
                      const void* u, void* obs, void* rew, void* stream);
/* Vector-Jacobian products of tsim_push_observe: d_obs [B][3 + ntac], d_rew [B] with element stride d_rew_stride (0 for a
 * broadcast scalar, as the gradient of a sum is) or NULL -> dq [B][7], dvar [B][6], dtac [B][ntac], du [B][3]
 * (dvar / du may be NULL when d_rew is). */
int tsim_push_observe_backward(int B, int ntac, int dtype, const void* q, const void* var, const void* goal, const void* u,
                               const void* d_obs, const void* d_rew, long long d_rew_stride,
                               void* dq, void* dvar, void* dtac, void* du, void* stream);

/* ---- The closed loop in ONE launch each way (BASELINE config 3 as cfg/gd_tactile.yaml runs it) -------------------------------------
 * algorithms/gd.py:224-259 alternates policy and env-step: obs -> DiagGaussianActor mean (utils/model.py:123-151: 393 -> 64 -> 64 -> 3, ELU)
 * -> action -> StepSimFunction.  One launch per env-step makes every env-step wait for the slowest environment of the batch; here the
 * observation, the MLP and the action mapping of an environment run inside ITS slot of the episode launch, between two frames
 * (csrc/tsim_policy_push.h), and their reverse inside the episode's adjoint launch.  Weight layouts (device, the batch's dtype):
 *   W1T [393][64], W2T [64][64]   layer weights transposed (input-major);   W3 [3][64], b1 [64], b2 [64], b3 [3] as torch stores them
 *   W1p [64][w1_stride]           layer-1 weights as torch stores them, rows padded to w1_stride >= 393, a multiple of 4 (backward only)
 *   W2  [64][64]                  layer-2 weights as torch stores them (backward only)                                              */
typedef struct tsim_push_policy {
  const void *W1T, *b1, *W2T, *b2, *W3, *b3, *W1p, *W2;
  int w1_stride;
  int obs_mode;      /* observation_type of envs/tactile_push_env.py:72-131: 0 tactile_flatten (393 inputs: goal in the gripper frame, tactile frame;
                        cfg/gd_tactile.yaml), 1 no_tactile (3: the goal; gd_no_tactile.yaml), 2 privilege (6: box pose in the gripper frame, goal;
                        gd_privilege.yaml).  W1T is [inputs][64], W1p [64][w1_stride >= inputs]. */
  /* Roll-out collection (cfg/ppo_tactile.yaml: the same actor, stochastic, on normalised observations).  All may be NULL. */
  const void* eps;      /* [T][B][3] standard-normal draws: u = mean + exp(logstd) * eps (the u record holds the sampled action) */
  const void* logstd;   /* [3] (utils/model.py:123-151 DiagGaussianActor.logstd) */
  const void *obs_mean, *obs_istd;  /* [inputs] each: observation -> clamp((x - obs_mean) * obs_istd, +-obs_clip) before the first layer (VecNormalize
                                       with the statistics frozen for the launch); forward-only launches (reset with backward_flag = 0) */
  double obs_clip;
} tsim_push_policy;

/* Forward: num_frames env-steps of num_steps sub-steps from the batch's current state; frame f acts with
 *   action_f = [tanh(policy(obs_f)), dist[f][env][0:2], 0],  obs_f = [goal in the gripper frame of the state before the frame, tactile frame before it]
 * (tac0 [B][390]: the tactile frame at the current state, tsim_readout).  Outputs per frame [T][B][.]: q, qd (may be NULL), var, tac (required with
 * the tactile observation, which it feeds; may be NULL otherwise, and the launch then skips the read-out), and the policy's records u (3, pre-tanh), gl (3, goal part of the observation), h1, h2 (64, ELU outputs).
 * Batches with per-environment parameter tables (tsim_set_env_tables) and edited models (tsim_update_model) are accepted (round 6): an fp32 batch
 * that keeps the TactilePush structure runs the structure-static closed-loop instantiation (tsim_kernel_variant "param:pusher"), any other the generic one. */
int tsim_push_closed_rollout(tsim_batch* b, const tsim_push_policy* pol, const void* goal, const void* dist, const void* tac0,
                             int num_frames, int num_steps, void* q_out, void* qd_out, void* var_out, void* tac_out,
                             void* u_out, void* gl_out, void* h1_out, void* h2_out, int32_t* status, void* stream);
/* Backward of that episode: df_dq / df_dvar [T][B][.] direct partials of the loss w.r.t. the frames' outputs (NULL = none), du_direct [T][B][3]
 * its direct partial w.r.t. the policy outputs (the reward's action term), the forward records, -> per frame the gradients w.r.t. the layers'
 * pre-activations g1, g2 [T][B][64], g3 [T][B][3] (weight gradients: dW_l = sum_t g_l[t]^T x_l[t], assembled by the caller as batched GEMMs),
 * dobs_tac [T][B][390] (workspace and result: gradient w.r.t. the tactile part of frame f's observation), df_du [T][B][6] or NULL.
 * The dependence of the FIRST observation on the initial state is not propagated (tsim_get_adjoint excludes it). */
int tsim_push_closed_backward(tsim_batch* b, const tsim_push_policy* pol, const void* goal, int num_frames, int num_steps,
                              const void* df_dq, const void* df_dvar, const void* du_direct, const void* u_out, const void* h1_out, const void* h2_out,
                              void* g1_out, void* g2_out, void* g3_out, void* dobs_tac, void* df_du, void* stream);

#ifdef __cplusplus
}
#endif
#endif
