// tsim_static_pusher_policy.hip — the closed-loop instantiations of the static TactilePush kernels (the policy between the frames,
// tsim_policy_push.h): four environments per wavefront, the shape of BASELINE.json's headline batch.  Their own translation unit because they
// are built at -Os like the generic kernels: with the policy's layers inlined the forward kernel is 67 KB at -Os and 76 KB (512 registers,
// spills) at the -O2 the open-loop static kernels are built with (tsim_static_pusher.hip) — the instruction cache holds 64 KB.
#include <hip/hip_runtime.h>
#include "tsim_kernels.h"
#include "tsim_static_pusher.h"

void ts_static_pusher_launch_policy(const FwdArgs<float>& a, unsigned grid, size_t lds, hipStream_t st) {
  if (a.default_opts) hipLaunchKernelGGL((k_forward<float, 8, false, 16, true, TsDefaultOpts<TsStaticPusher>>), dim3(grid), dim3(TS_WAVE), lds, st, a);      // every solver option at its default: as constants (tsim_static.h)
  else hipLaunchKernelGGL((k_forward<float, 8, false, 16, true, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
void ts_static_pusher_launch_policy(const BwdArgs<float>& a, unsigned grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((k_backward<float, 8, false, 16, true, TsStaticPusher>), dim3(grid), dim3(TS_WAVE), lds, st, a);
}
