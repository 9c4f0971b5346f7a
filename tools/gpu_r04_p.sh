cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04p
rm -f ${O}_sites.jsonl
( TSIM_TEST_REPORT=$PWD/${O}_sites.jsonl timeout 300 python -m pytest tests/test_gpu_static_model.py -m gpu -q --tb=short 2>&1 | head -40 ) > ${O}_static_test.log 2>&1
echo skip > ${O}_tests.log
head -30 ${O}_static_test.log | cut -c1-300; grep -n "^FAILED\|passed\|failed\|Error" ${O}_tests.log | tail -8; grep static ${O}_sites.jsonl
