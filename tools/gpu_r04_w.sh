cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04w
rm -f ${O}_sites.jsonl
( TSIM_TEST_REPORT=$PWD/${O}_sites.jsonl timeout 600 python -m pytest tests/test_gpu_static_model.py tests/test_gpu_closed_loop.py -m gpu -q --tb=short 2>&1 | tail -40 ) > ${O}_tests.log 2>&1
grep -n "^FAILED\|passed\|failed\|^E " ${O}_tests.log | tail -8; cat ${O}_sites.jsonl | grep static
