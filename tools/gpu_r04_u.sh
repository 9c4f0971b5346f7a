cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_rounds.so
( echo free; python tools/round_stats.py 2>&1 | tail -2; echo lockstep; TSIM_NO_FREE_RUN=1 TSIM_INKERNEL_READOUT=1 python tools/round_stats.py 2>&1 | tail -2; echo frame-sync; TSIM_INKERNEL_READOUT=1 python tools/round_stats.py 2>&1 | tail -2 ) | tee gpurun_out/r04u_rounds.log
