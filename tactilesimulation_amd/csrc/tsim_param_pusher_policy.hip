// tsim_param_pusher_policy.hip — the closed-loop instantiations (the policy between the frames, tsim_policy_push.h) of the STRUCTURE-static
// TactilePush kernels (tsim_param_pusher.hip: structure compiled in, parameters from the batch's float records, shared or per environment).
// Round 6: what lets the fused closed loop (include/tsim_env.h tsim_push_closed_rollout / _backward) stay on compiled-in kernels after the env's
// update_* edits and with one parameter table per environment (domain randomisation: envs/tactile_push_env.py has none, but
// envs/tactile_insertion_env.py:238-281 and envs/dclaw_rotate_env.py:169-178 draw theirs at every reset — a batched TactilePush collector does the same).
// Four environments per wavefront, built at -Os like tsim_static_pusher_policy.hip (the policy's layers inlined: the kernels are as large as the
// instruction cache).
#include <hip/hip_runtime.h>
#include "tsim_kernels.h"
#include "tsim_static_pusher.h"

using TsParamPusher = TsParam<TsStaticPusher>;

void ts_param_pusher_launch_policy(const FwdArgs<float>& a, unsigned grid, size_t lds, hipStream_t st) {
  if (a.default_opts) hipLaunchKernelGGL((k_forward<float, 8, false, 16, true, TsDefaultOpts<TsParamPusher>>), dim3(grid), dim3(TS_WAVE), lds, st, a);      // every solver option at its default: as constants (tsim_static.h)
  else hipLaunchKernelGGL((k_forward<float, 8, false, 16, true, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
void ts_param_pusher_launch_policy(const BwdArgs<float>& a, unsigned grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((k_backward<float, 8, false, 16, true, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
