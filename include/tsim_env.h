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

#ifdef __cplusplus
}
#endif
#endif
