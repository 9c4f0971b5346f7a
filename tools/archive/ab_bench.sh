# A/B of library builds on the GPU box: for each tactilesimulation_amd/csrc/ab/libtsim_<tag>.so named on the command line (and the
# in-tree build as "base"): a short parity subset, then the driver's timed region twice and the 100-step one once
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for tag in base "$@"; do
  if [ "$tag" = base ]; then unset TSIM_HIP_LIB; else export TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_$tag.so; fi
  echo "== $tag"
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
  for s in 20 20 100; do timeout 300 python bench.py --gpus 1 --steps $s --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps$s', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms'].items()})"; done
done
