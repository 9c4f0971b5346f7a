cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_rounds.so python tools/round_stats.py 2>&1 | tail -1; TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_fine.so TSIM_LPE=16 python tools/fine_stamps.py 2>&1 | tail -1 ) | tee gpurun_out/r04u_rounds.log
