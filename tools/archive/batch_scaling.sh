# open-loop bench (forward + adjoint episode launches) against the batch size on one GPU: tools/batch_scaling.sh > gpurun_out/batch_scaling.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for s in 20 100; do for b in 1024 2048 4096 8192 16384 32768; do
  timeout 600 python bench.py --gpus 1 --batch $b --steps $s --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'batch': $b, 'steps': $s, 'env_steps_per_s': round(d['value']), 'ms_per_step': round(d['ms_per_step'],4), 'kernel_ms': {k: round(v,3) for k,v in d['kernel_ms'].items()}}))"
done; done
