cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04k
rm -f ${O}_sites.jsonl
( TSIM_TEST_REPORT=$PWD/${O}_sites.jsonl timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 ) > ${O}_tests.log 2>&1
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -8; grep static_vs ${O}_sites.jsonl
