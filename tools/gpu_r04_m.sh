cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04m
( timeout 900 python -m pytest tests/test_gpu_static_model.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_closed_loop.py tests/test_gpu_rollout.py -m gpu -q --tb=short 2>&1 | tail -8 ) > ${O}_tests_quick.log 2>&1
for i in 1 2 3; do timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_static.jsonl 2>/dev/null; done
tail -3 ${O}_tests_quick.log; cat ${O}_static.jsonl
