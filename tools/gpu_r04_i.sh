cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04i; AB=$PWD/tactilesimulation_amd/csrc/ab
for i in 1 2; do
TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 >> ${O}_fine_dpp.json
TSIM_HIP_LIB=$AB/libtsim_fine_pivot.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 >> ${O}_fine_pivot.json
done
cat ${O}_fine_dpp.json ${O}_fine_pivot.json
