# A/B bench of alternative builds of the HIP library: bash tools/gpu_ab.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  for dt in f32 f64; do
    TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/$lib timeout 300 python bench.py --steps 100 --warmup 10 --dtype $dt --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['dtype'], round(d['value']), d['roofline']['kernel_ms'], d['residual_evals_per_env_step']['mean_of_per_step_max'])"
  done
done
