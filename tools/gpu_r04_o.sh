cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04o; AB=$PWD/tactilesimulation_amd/csrc/ab
( timeout 900 python -m pytest tests/test_gpu_static_model.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_closed_loop.py -m gpu -q --tb=short 2>&1 | tail -12 ) > ${O}_tests_quick.log 2>&1
TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 > ${O}_fine_levels.json
for i in 1 2 3; do
  timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_levels.jsonl 2>/dev/null
  TSIM_HIP_LIB=$AB/libtsim_blocks.so timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_blocks.jsonl 2>/dev/null
done
tail -4 ${O}_tests_quick.log; cat ${O}_fine_levels.json ${O}_levels.jsonl ${O}_blocks.jsonl
