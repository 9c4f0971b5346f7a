cd $GRAFT_REPO_ROOT
for B in 256 1024 2048 3072 4096 5120 8192; do
TSIM_LPE=16 timeout 300 python bench.py --steps 100 --warmup 10 --batch $B --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('B=$B LPE16', round(r['value']), {k: round(v,3) for k,v in r['roofline']['kernel_ms_per_env_step'].items()}, r['launch_shape']['blocks'])"
done
