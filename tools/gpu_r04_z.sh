cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/sub_record_ab.py push_fwd dclaw insertion 2>/dev/null | grep '^{' | sed "s/^/base /" | tee gpurun_out/r04z_sub_ab4.log
for i in 1 2; do python bench.py --steps 20 --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('push', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms'].items()})"; done
for lpe in 32 64; do TSIM_LPE=$lpe timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_rollout.py tests/test_gpu_edge_cases.py tests/test_gpu_literal.py -m gpu -q -x 2>&1 | tail -1; done
