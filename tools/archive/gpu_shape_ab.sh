cd $GRAFT_REPO_ROOT
run() { # label env... -- args
  lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 100 --warmup 100 --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lab', round(d['value']), {k: round(v,3) for k,v in d['roofline']['kernel_ms_per_env_step'].items()}, d['launch_shape'], 'step', round(d['launch']['other_mode_value']))"
}
ARGS="--batch 8192" run "B8192 minw1" TSIM_MINW=1
ARGS="--batch 8192" run "B8192 minw2" TSIM_MINW=2
ARGS="--batch 16384" run "B16384 minw1" TSIM_MINW=1
ARGS="--batch 2048" run "B2048 lpe32" TSIM_LPE=32
ARGS="--batch 2048" run "B2048 lpe16" TSIM_LPE=16
ARGS="--batch 4096" run "B4096 default" A=1
ARGS="--batch 4096" run "B4096 default" A=1
ARGS="--batch 4096" run "B4096 default" A=1
ARGS="--batch 4096 --dtype f64" run "B4096 f64 lpe32 minw1" TSIM_LPE=32
ARGS="--batch 4096 --dtype f64" run "B4096 f64 lpe16 minw1" TSIM_LPE=16
