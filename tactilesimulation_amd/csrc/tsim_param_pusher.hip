// tsim_param_pusher.hip — the forward / adjoint kernels instantiated for the STRUCTURE of the TactilePush model (envs/assets/pusher/pusher.xml):
// tree, joint types, contact pairs, dof / motor layout and the structural floats of the compiled asset (identity joint frames, unit axes,
// absent limits: TsParam<TsStaticPusher>::Fk) are compile-time constants, every other parameter is read from the batch's float records in LDS
// (tsim_static.h ts_F).  The fused register-resident evaluation (tsim_static_eval.h) as in tsim_static_pusher.hip, but for ANY batch with this
// structure: after tsim_update_model (the env's update_* randomisers) and with tsim_set_env_tables (one table per environment).  Same flags as the
// fully static unit: the folds are those of the structural entries.
#include <hip/hip_runtime.h>
#include "tsim_kernels.h"
#include "tsim_static_pusher.h"

using TsParamPusher = TsParam<TsStaticPusher>;

void ts_param_pusher_launch(const FwdArgs<float>& a, int lpe, unsigned grid, size_t lds, hipStream_t st) {
  if (lpe == 16 && a.default_opts) hipLaunchKernelGGL((k_forward<float, 8, false, 16, false, TsDefaultOpts<TsParamPusher>>), dim3(grid), dim3(TS_WAVE), lds, st, a);      // every option at its default: as constants
  else if (lpe == 16) hipLaunchKernelGGL((k_forward<float, 8, false, 16, false, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else if (lpe == 32) hipLaunchKernelGGL((k_forward<float, 8, false, 32, false, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else hipLaunchKernelGGL((k_forward<float, 8, false, 64, false, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
void ts_param_pusher_launch(const BwdArgs<float>& a, int lpe, unsigned grid, size_t lds, hipStream_t st) {
  if (lpe == 16) hipLaunchKernelGGL((k_backward<float, 8, false, 16, false, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else if (lpe == 32) hipLaunchKernelGGL((k_backward<float, 8, false, 32, false, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else hipLaunchKernelGGL((k_backward<float, 8, false, 64, false, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
void ts_param_pusher_launch_debug(const DbgArgs<float>& a, unsigned grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((k_debug_eval<float, 16, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}

// ... and in fp64 (round 5): the reference's arithmetic type (envs/tactile_push_env.py:29).  Two or one environments per wavefront: four do not fit the
// block's LDS in fp64, and the host never asks for them (tsim_hip.hip TS_LAUNCH).  The Newton systems are solved with partial pivoting, as in every
// fp64 kernel (solve_newton).
void ts_param_pusher_launch(const FwdArgs<double>& a, int lpe, unsigned grid, size_t lds, hipStream_t st) {
  if (lpe == 32) hipLaunchKernelGGL((k_forward<double, 8, false, 32, false, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else hipLaunchKernelGGL((k_forward<double, 8, false, 64, false, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
void ts_param_pusher_launch(const BwdArgs<double>& a, int lpe, unsigned grid, size_t lds, hipStream_t st) {
  if (lpe == 32) hipLaunchKernelGGL((k_backward<double, 8, false, 32, false, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else hipLaunchKernelGGL((k_backward<double, 8, false, 64, false, TsParamPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
