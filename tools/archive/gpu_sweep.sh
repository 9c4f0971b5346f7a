# batch-size sweep of bench.py (GPU box); writes gpurun_out/batch_sweep_$1.json
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=gpurun_out/batch_sweep_$TAG.jsonl; : > $OUT
run() { # label args...
  l=$1; shift
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r={'label':'$l','value':round(d['value']),'dtype':d['dtype'],'kernel_ms_per_env_step':d['roofline']['kernel_ms_per_env_step'],'launch_shape':d['launch_shape'],'step_mode_value':round(d['launch']['other_mode_value'] or 0)}
print(json.dumps(r))" | tee -a $OUT
}
for B in 512 1024 2048 4096 8192 16384 32768; do run "fwd+bwd B=$B f32" --batch $B; done
for B in 1024 4096 16384; do run "fwd+bwd B=$B f64" --batch $B --dtype f64; done
for B in 1024 4096; do run "fwd-only B=$B f32" --batch $B --forward-only; done
