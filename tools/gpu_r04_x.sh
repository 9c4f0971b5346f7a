cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for tag in base ${TAGS:-slp}; do
  if [ "$tag" = base ]; then unset TSIM_HIP_LIB; else export TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_$tag.so; fi
  echo "== $tag"
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
  for s in 20 20; do timeout 300 python bench.py --gpus 1 --steps $s --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps$s', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms'].items()})"; done
done 2>&1 | tee gpurun_out/r04x_ab.log
