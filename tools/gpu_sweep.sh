cd $GRAFT_REPO_ROOT
for B in 1024 4096 16384 32768; do
  timeout 300 python bench.py --steps 100 --warmup 10 --batch $B --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd+bwd B=$B', d['dtype'], round(d['value']), d['roofline']['kernel_ms'], d['residual_evals_per_env_step'])"
done
timeout 300 python bench.py --steps 100 --warmup 10 --batch 1024 --forward-only --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd-only B=1024', d['dtype'], round(d['value']), d['roofline']['kernel_ms'])"
timeout 300 python bench.py --steps 100 --warmup 10 --batch 32768 --dtype f64 --episode 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd+bwd B=32768 f64', round(d['value']), d['roofline']['kernel_ms'])"
