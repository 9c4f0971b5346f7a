cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04e; AB=$PWD/tactilesimulation_amd/csrc/ab
TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 > ${O}_fine_stamps.json
for lib in mfma valu; do
  rm -rf /tmp/prof_$lib
  ( cd /tmp && if [ $lib = valu ]; then export TSIM_HIP_LIB=$AB/libtsim_valu.so; fi; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$lib -o cl -- python $GRAFT_REPO_ROOT/tools/closed_loop_breakdown.py > $GRAFT_REPO_ROOT/${O}_breakdown_$lib.json 2>/dev/null )
  cp $(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1) ${O}_closed_loop_kernel_stats_$lib.csv
  python tools/kernel_stats_summary.py ${O}_closed_loop_kernel_stats_$lib.csv | head -5
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --timed-only --no-pmc > /dev/null 2>&1 ); cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats_f32_steps20.csv
python tools/kernel_stats_summary.py ${O}_rocprof_kernel_stats_f32_steps20.csv | head -6
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ro -o r -- python $GRAFT_REPO_ROOT/bench.py --readout-only --batch 4096 > /dev/null 2>&1 ); cp $(find /tmp/prof_ro -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats_readout_b4096.csv
python tools/kernel_stats_summary.py ${O}_rocprof_kernel_stats_readout_b4096.csv | head -5
cat ${O}_fine_stamps.json
