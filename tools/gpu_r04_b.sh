# round 4, GPU session B: full suite with the measured-error log, MFMA A/B of the in-kernel policy, non-temporal-store A/B of k_taxels,
# 2-rank shared-GPU run of the per-rank decomposition
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04b; AB=$PWD/tactilesimulation_amd/csrc/ab
rm -f ${O}_sites.jsonl
( TSIM_TEST_REPORT=$PWD/${O}_sites.jsonl timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > ${O}_tests.log 2>&1
( TSIM_HIP_LIB=$AB/libtsim_mfma.so timeout 600 python -m pytest tests/test_gpu_closed_loop.py -q 2>&1 | tail -5 ) > ${O}_mfma_tests.log 2>&1
for i in 1 2 3; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-sub-records 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(json.dumps({'lib':'valu','closed_loop':b['closed_loop']['value'],'s_per_epoch':b['closed_loop']['s_per_epoch']}))" >> ${O}_mfma_ab.jsonl
  TSIM_HIP_LIB=$AB/libtsim_mfma.so timeout 200 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-sub-records 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(json.dumps({'lib':'mfma','closed_loop':b['closed_loop']['value'],'s_per_epoch':b['closed_loop']['s_per_epoch']}))" >> ${O}_mfma_ab.jsonl
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04b_prof_valu -o cl -- python $GRAFT_REPO_ROOT/tools/closed_loop_breakdown.py > /dev/null 2>&1; TSIM_HIP_LIB=$AB/libtsim_mfma.so timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04b_prof_mfma -o cl -- python $GRAFT_REPO_ROOT/tools/closed_loop_breakdown.py > /dev/null 2>&1 )
for f in gpurun_out/r04b_prof_valu gpurun_out/r04b_prof_mfma; do python tools/kernel_stats_summary.py $(find $f -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -6 > ${f}_top.txt; done
for B in 1024 4096; do for i in 1 2; do
  timeout 200 python bench.py --readout-only --batch $B 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(json.dumps({'lib':'plain','B':$B,'ms':b['ms'],'achieved':b['achieved'],'ms_cold':b['ms_cold']}))" >> ${O}_taxnt_ab.jsonl
  TSIM_HIP_LIB=$AB/libtsim_taxnt.so timeout 200 python bench.py --readout-only --batch $B 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(json.dumps({'lib':'nt','B':$B,'ms':b['ms'],'achieved':b['achieved'],'ms_cold':b['ms_cold']}))" >> ${O}_taxnt_ab.jsonl
done; done
TSIM_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-closed-loop --no-sub-records > ${O}_2ranks.json 2> ${O}_2ranks.err
tail -8 ${O}_tests.log; cat ${O}_mfma_tests.log | tail -2; cat ${O}_mfma_ab.jsonl ${O}_taxnt_ab.jsonl; cat gpurun_out/r04b_prof_valu_top.txt gpurun_out/r04b_prof_mfma_top.txt; python -c "import json; b=json.load(open('${O}_2ranks.json')); print(b['value'], json.dumps(b['per_rank'])[:1200])"
