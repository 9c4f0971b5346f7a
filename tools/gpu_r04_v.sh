cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04v
( timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 ) > ${O}_tests.log 2>&1
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -8
STEPS=20 bash tools/gpu_r04_t.sh
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --timed-only --no-pmc > /dev/null 2>&1 ); cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats.csv
head -5 ${O}_rocprof_kernel_stats.csv | cut -c1-160
