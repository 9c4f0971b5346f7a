cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04h; AB=$PWD/tactilesimulation_amd/csrc/ab
( timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 ) > ${O}_tests.log 2>&1
for i in 1 2 3; do
  timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_solve_dpp.jsonl 2>/dev/null
  TSIM_HIP_LIB=$AB/libtsim_pivot.so timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_solve_pivot.jsonl 2>/dev/null
done
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -8; cat ${O}_solve_dpp.jsonl ${O}_solve_pivot.jsonl
