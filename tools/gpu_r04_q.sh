cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04q; AB=$PWD/tactilesimulation_amd/csrc/ab
( timeout 900 python -m pytest tests/test_gpu_static_model.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_closed_loop.py -m gpu -q --tb=line 2>&1 | tail -6 ) > ${O}_tests_quick.log 2>&1
TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 > ${O}_fine.json
for i in 1 2 3; do timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_timed.jsonl 2>/dev/null; done
tail -3 ${O}_tests_quick.log; cat ${O}_fine.json ${O}_timed.jsonl
