cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04g
( timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 ) > ${O}_tests.log 2>&1
for i in 1 2; do
  timeout 300 python examples/collect_dclaw_rollouts.py --batch 2048 --steps 100 --variants 16 2>&1 | tail -1 >> ${O}_dclaw_v16_staged.log
  TSIM_NO_ENVTAB_CPT=1 timeout 300 python examples/collect_dclaw_rollouts.py --batch 2048 --steps 100 --variants 16 2>&1 | tail -1 >> ${O}_dclaw_v16_global.log
done
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -5; cat ${O}_dclaw_v16_staged.log ${O}_dclaw_v16_global.log
