# A/B of builds (csrc/*.so given as arguments) under each lanes-per-environment setting (GPU box)
cd $GRAFT_REPO_ROOT
for so in "$@"; do for dt in ${DTYPES:-f32}; do for l in ${LPES:-64 32 16}; do
  TSIM_HIP_LIB=$GRAFT_REPO_ROOT/tactilesimulation_amd/csrc/$so TSIM_LPE=$l timeout 300 python bench.py --steps 100 --warmup 10 --dtype $dt --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('$so $dt LPE $l', round(r['value']), {k: round(v,3) for k,v in r['roofline']['kernel_ms_per_env_step'].items()}, 'step-mode', round(r['launch']['other_mode_value'] or 0), 'bad', r['nonconverged_envs_last_step'], r['nonconverged_warmup'])"
done; done; done
