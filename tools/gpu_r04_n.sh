cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04n; AB=$PWD/tactilesimulation_amd/csrc/ab
TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 > ${O}_fine_static.json
TSIM_NO_STATIC=1 TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 > ${O}_fine_generic.json
( TSIM_LPE=16 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short 2>&1 | tail -4 ) > ${O}_parity_lpe16.log 2>&1
cat ${O}_fine_static.json ${O}_fine_generic.json; tail -2 ${O}_parity_lpe16.log
