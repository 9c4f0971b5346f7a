# bench.py under each lanes-per-environment setting (GPU box)
cd $GRAFT_REPO_ROOT
for dt in ${DTYPES:-f32 f64}; do for l in ${LPES:-64 32 16}; do
  TSIM_LPE=$l timeout 300 python bench.py --steps 100 --warmup 10 --dtype $dt --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('$dt LPE $l', round(r['value']), r['roofline']['kernel_ms_per_env_step'], 'step-mode', round(r['launch']['other_mode_value'] or 0), 'shape', r['launch_shape'], 'bad', r['nonconverged_envs_last_step'], r['nonconverged_warmup'])"
done; done
