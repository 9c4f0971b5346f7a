# round 4, GPU session D: full suite on the final build (MFMA policy layers, non-temporal read-out stores), kernel durations of the closed loop
# under both builds of the policy layers, the driver-shaped bench line twice
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04d; AB=$PWD/tactilesimulation_amd/csrc/ab
( timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60 ) > ${O}_tests.log 2>&1
for lib in mfma valu; do
  rm -rf /tmp/prof_$lib
  ( cd /tmp && if [ $lib = valu ]; then export TSIM_HIP_LIB=$AB/libtsim_valu.so; fi; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$lib -o cl -- python $GRAFT_REPO_ROOT/tools/closed_loop_breakdown.py > $GRAFT_REPO_ROOT/${O}_breakdown_$lib.json 2>/dev/null )
  python tools/kernel_stats_summary.py $(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -8 > ${O}_closed_loop_kernels_$lib.txt
  cp $(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1) ${O}_closed_loop_kernel_stats_$lib.csv 2>/dev/null
done
for i in 1 2; do ( timeout 600 python bench.py --steps 20 --warmup 5 --pmc-dump gpurun_out/r04d_pmc_f32_$i.json > ${O}_bench_$i.json ) 2> ${O}_bench_$i.err; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --timed-only --no-pmc > /dev/null 2>&1 ); cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats_f32_steps20.csv
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail; cat ${O}_closed_loop_kernels_mfma.txt ${O}_closed_loop_kernels_valu.txt; cat ${O}_breakdown_mfma.json ${O}_breakdown_valu.json | grep "total\|M env"
python -c "
import json
for i in (1,2):
    for l in open('${O}_bench_%d.json' % i):
        if l.startswith('{'):
            b=json.loads(l); print(b['value'], b['ms_per_step'], b['roofline']['kernel_ms_per_env_step'], b['closed_loop']['value'], b['insertion']['value'], b['readout_hbm']['achieved'], [x['achieved'] for x in b['readout_hbm']['by_batch']])"
du -sh gpurun_out
