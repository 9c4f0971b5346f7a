cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04b
rm -f ${O}_sites.jsonl
( TSIM_TEST_REPORT=$PWD/${O}_sites.jsonl timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150 ) > ${O}_tests.log 2>&1
TSIM_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-closed-loop --no-sub-records > ${O}_2ranks.json 2> ${O}_2ranks.err
tail -5 ${O}_2ranks.err; head -c 300 ${O}_2ranks.json
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail
du -sh gpurun_out
