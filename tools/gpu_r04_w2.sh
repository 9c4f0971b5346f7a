cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for tag in base prev; do
  if [ "$tag" = base ]; then unset TSIM_HIP_LIB; else export TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_$tag.so; fi
  for E in 30 36 40 44; do
    python -c "
import sys, json; sys.path.insert(0, 'tools')
import trained_regime_grad_check as T
o = T.run($E, verbose=False)
f = o['f32']; print('$tag', $E, json.dumps({k: f[k] for k in ('q_err_max', 'branch_agree', 'grad_err_median', 'grad_err_max_agreeing', 'grad_err_max_all')}))" 2>/dev/null | grep "^$tag"
  done
done | tee gpurun_out/r04w2_trained.log
