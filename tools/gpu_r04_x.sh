cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04x
for tag in base f32solve; do
  if [ "$tag" = base ]; then unset TSIM_HIP_LIB; else export TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_$tag.so; fi
  echo "== $tag"
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -1
  for s in 20 20 100; do timeout 300 python bench.py --gpus 1 --steps $s --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps$s', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms'].items()})"; done
  python tools/round_stats.py 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('evals_mean','evals_max')})"
done 2>&1 | tee ${O}_ab.log
