cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04f
( timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 ) > ${O}_tests.log 2>&1
for v in 0 16; do for i in 1 2; do timeout 300 python examples/collect_dclaw_rollouts.py --batch 2048 --steps 100 --variants $v 2>&1 | tail -2 >> ${O}_dclaw_collect_v$v.log; done; done
( timeout 600 python bench.py --steps 20 --warmup 5 --no-pmc --no-closed-loop > ${O}_bench.json ) 2> ${O}_bench.err
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -5; cat ${O}_dclaw_collect_v0.log ${O}_dclaw_collect_v16.log
python -c "
import json
for l in open('${O}_bench.json'):
    if l.startswith('{'):
        b=json.loads(l); print(b['value'], b['ms_per_step']); 
        for k in ('push_forward_only_b1024','dclaw','insertion'): print(k, b[k].get('value'), b[k].get('nonconverged_envs'), b[k].get('error'), b[k].get('launch_shape'), b[k].get('residual_evals_per_substep_last_launch'))"
