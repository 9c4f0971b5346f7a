cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "X_=1" "TSIM_NO_FREE_RUN=1" "TSIM_INKERNEL_READOUT=1" "TSIM_INKERNEL_READOUT=1 TSIM_NO_FREE_RUN=1"; do
  env $v python tools/sub_record_ab.py push_fwd dclaw insertion 2>/dev/null | grep '^{'
done | tee gpurun_out/r04z_sub_ab.log
