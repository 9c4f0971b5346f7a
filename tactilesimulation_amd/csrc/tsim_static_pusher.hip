// tsim_static_pusher.hip — the forward / adjoint kernels instantiated for the TactilePush model (envs/assets/pusher/pusher.xml) with its
// compiled blob as compile-time constants (tsim_static_pusher.h, generated; tsim_static.h: what that buys).  Its own translation unit because
// it is built with -ffinite-math-only -fno-signed-zeros — x * 0 -> 0, x + 0 -> x fold away the model's identity joint frames and unit axes —
// and the generic kernels (tsim_hip.hip) are not.  fp32, every launch shape (the debug kernel: four environments per wavefront, the shape of
// BASELINE.json's headline batch); built at -O2 (host/buildhash.py).  The closed-loop instantiations live in tsim_static_pusher_policy.hip.
#include <hip/hip_runtime.h>
#include "tsim_kernels.h"
#include "tsim_static_pusher.h"

void ts_static_pusher_launch(const FwdArgs<float>& a, int lpe, unsigned grid, size_t lds, hipStream_t st) {
  if (lpe == 16 && a.default_opts) hipLaunchKernelGGL((k_forward<float, 8, false, 16, false, TsDefaultOpts<TsStaticPusher>>), dim3(grid), dim3(TS_WAVE), lds, st, a);      // every option at its default: as constants
  else if (lpe == 16) hipLaunchKernelGGL((k_forward<float, 8, false, 16, false, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else if (lpe == 32) hipLaunchKernelGGL((k_forward<float, 8, false, 32, false, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else hipLaunchKernelGGL((k_forward<float, 8, false, 64, false, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
void ts_static_pusher_launch(const BwdArgs<float>& a, int lpe, unsigned grid, size_t lds, hipStream_t st) {
  if (lpe == 16) hipLaunchKernelGGL((k_backward<float, 8, false, 16, false, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else if (lpe == 32) hipLaunchKernelGGL((k_backward<float, 8, false, 32, false, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else hipLaunchKernelGGL((k_backward<float, 8, false, 64, false, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
// one residual + Newton-matrix evaluation (tsim_debug_eval: parity tests, shader-clock stamps)
void ts_static_pusher_launch_debug(const DbgArgs<float>& a, unsigned grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((k_debug_eval<float, 16, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}

// ... and in fp64 (round 5): the reference's arithmetic type (envs/tactile_push_env.py:29).  Two or one environments per wavefront: four do not fit the
// block's LDS in fp64, and the host never asks for them (tsim_hip.hip TS_LAUNCH).  The Newton systems are solved with partial pivoting, as in every
// fp64 kernel (solve_newton).
void ts_static_pusher_launch(const FwdArgs<double>& a, int lpe, unsigned grid, size_t lds, hipStream_t st) {
  if (lpe == 32) hipLaunchKernelGGL((k_forward<double, 8, false, 32, false, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else hipLaunchKernelGGL((k_forward<double, 8, false, 64, false, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
void ts_static_pusher_launch(const BwdArgs<double>& a, int lpe, unsigned grid, size_t lds, hipStream_t st) {
  if (lpe == 32) hipLaunchKernelGGL((k_backward<double, 8, false, 32, false, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
  else hipLaunchKernelGGL((k_backward<double, 8, false, 64, false, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
