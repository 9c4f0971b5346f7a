# The round-4 GPU sessions, one case per session (merged from the 27 one-off scripts tools/gpu_r04_<x>.sh of that round; run on the GPU box through
# gpurun from the repo root): bash tools/gpu_r04.sh <session>.  What each produced is under profiles/r04_* (profiles/README.md).
S=$1
case $S in
a)
# round 4, GPU session A: full GPU suite, the driver-shaped bench line, LPT A/B for episode launches, the SURVEY-worded insertion leg
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04a
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > ${O}_tests.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 --pmc-dump gpurun_out/r04a_pmc_f32.json > ${O}_bench.json ) 2> ${O}_bench.err
for i in 1 2; do
  timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_lpt_on.jsonl 2>/dev/null
  TSIM_NO_EPISODE_LPT=1 timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_lpt_off.jsonl 2>/dev/null
done
timeout 300 python bench.py --workload insertion --steps 18 --warmup 9 --no-pmc --no-cpu-baseline --no-closed-loop > ${O}_insertion.json 2> ${O}_insertion.err
timeout 300 python tools/insertion_attempt_probe.py > ${O}_insertion_probe.json 2> ${O}_insertion_probe.err
tail -3 ${O}_tests.log; head -c 600 ${O}_bench.json; echo; cat ${O}_lpt_on.jsonl ${O}_lpt_off.jsonl; head -c 400 ${O}_insertion.json; echo; cat ${O}_insertion_probe.json | head -c 1500
;;
b)
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
;;
b2)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04b
rm -f ${O}_sites.jsonl
( TSIM_TEST_REPORT=$PWD/${O}_sites.jsonl timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150 ) > ${O}_tests.log 2>&1
TSIM_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-closed-loop --no-sub-records > ${O}_2ranks.json 2> ${O}_2ranks.err
tail -5 ${O}_2ranks.err; head -c 300 ${O}_2ranks.json
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail
du -sh gpurun_out
;;
c)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04c; AB=$PWD/tactilesimulation_amd/csrc/ab
( timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60 ) > ${O}_tests.log 2>&1
python -c "
import sys, json; sys.path.insert(0, 'tools')
import trained_regime_grad_check as T
print(json.dumps(T.run(40)))" > ${O}_trained_valu.json 2>/dev/null
TSIM_HIP_LIB=$AB/libtsim_mfma.so python -c "
import sys, json; sys.path.insert(0, 'tools')
import trained_regime_grad_check as T
print(json.dumps(T.run(40)))" > ${O}_trained_mfma.json 2>/dev/null
( timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench.json ) 2> ${O}_bench.err
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail; cat ${O}_trained_valu.json ${O}_trained_mfma.json; python -c "
import json
for l in open('${O}_bench.json'):
    if l.startswith('{'):
        b=json.loads(l); print(b['value'], b['ms_per_step'], b['insertion']['value'], b['readout_hbm']['achieved'], [x['achieved'] for x in b['readout_hbm']['by_batch']], b['readout_hbm'].get('pmc'))"
;;
d)
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
;;
e)
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
;;
f)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04f
( timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 ) > ${O}_tests.log 2>&1
for v in 0 16; do for i in 1 2; do timeout 300 python examples/collect_dclaw_rollouts.py --batch 2048 --steps 100 --variants $v 2>&1 | tail -2 >> ${O}_dclaw_collect_v$v.log; done; done
( timeout 600 python bench.py --steps 20 --warmup 5 --no-pmc --no-closed-loop > ${O}_bench.json ) 2> ${O}_bench.err
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -5; cat ${O}_dclaw_collect_v0.log ${O}_dclaw_collect_v16.log
python -c "
import json
for l in open('${O}_bench.json'):
    if l.startswith('{'):
        b=json.loads(l); print(b['value'], b['ms_per_step']); 
        for k in ('push_forward_only_b1024','dclaw','insertion'): print(k, b[k].get('value'), b[k].get('nonconverged_envs'), b[k].get('error'), b[k].get('launch_shape'), b[k].get('residual_evals_per_substep_last_launch'))"
;;
g)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04g
( timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 ) > ${O}_tests.log 2>&1
for i in 1 2; do
  timeout 300 python examples/collect_dclaw_rollouts.py --batch 2048 --steps 100 --variants 16 2>&1 | tail -1 >> ${O}_dclaw_v16_staged.log
  TSIM_NO_ENVTAB_CPT=1 timeout 300 python examples/collect_dclaw_rollouts.py --batch 2048 --steps 100 --variants 16 2>&1 | tail -1 >> ${O}_dclaw_v16_global.log
done
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -5; cat ${O}_dclaw_v16_staged.log ${O}_dclaw_v16_global.log
;;
h)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04h; AB=$PWD/tactilesimulation_amd/csrc/ab
( timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 ) > ${O}_tests.log 2>&1
for i in 1 2 3; do
  timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_solve_dpp.jsonl 2>/dev/null
  TSIM_HIP_LIB=$AB/libtsim_pivot.so timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_solve_pivot.jsonl 2>/dev/null
done
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -8; cat ${O}_solve_dpp.jsonl ${O}_solve_pivot.jsonl
;;
i)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04i; AB=$PWD/tactilesimulation_amd/csrc/ab
for i in 1 2; do
TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 >> ${O}_fine_dpp.json
TSIM_HIP_LIB=$AB/libtsim_fine_pivot.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 >> ${O}_fine_pivot.json
done
cat ${O}_fine_dpp.json ${O}_fine_pivot.json
;;
j)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04j
( timeout 600 python -m pytest tests/test_gpu_static_model.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q --tb=short -x 2>&1 | tail -30 ) > ${O}_tests_quick.log 2>&1
for i in 1 2 3; do
  timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_static.jsonl 2>/dev/null
  TSIM_NO_STATIC=1 timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_generic.jsonl 2>/dev/null
done
tail -5 ${O}_tests_quick.log; cat ${O}_static.jsonl ${O}_generic.jsonl
;;
k)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04k
rm -f ${O}_sites.jsonl
( TSIM_TEST_REPORT=$PWD/${O}_sites.jsonl timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 ) > ${O}_tests.log 2>&1
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -8; grep static_vs ${O}_sites.jsonl
;;
l)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04l
( timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 ) > ${O}_tests.log 2>&1
for i in 1 2; do
( timeout 600 python bench.py --steps 20 --warmup 5 --pmc-dump gpurun_out/r04l_pmc_f32_$i.json > ${O}_bench_$i.json ) 2> ${O}_bench_$i.err
done
TSIM_NO_STATIC=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-sub-records 2>/dev/null > ${O}_bench_generic.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --timed-only --no-pmc > /dev/null 2>&1 ); cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats_f32_steps20.csv
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -5
python -c "
import json
for f in ('${O}_bench_1.json','${O}_bench_2.json','${O}_bench_generic.json'):
    for l in open(f):
        if l.startswith('{'):
            b=json.loads(l); print(f[-14:], round(b['value']), round(b['ms_per_step'],4), b['roofline']['kernel_ms_per_env_step'], 'closed', round(b['closed_loop']['value']), 'other', round(b['launch']['other_mode_value']), [round(b[k]['value']) for k in ('f64','push_forward_only_b1024','dclaw','insertion') if k in b], b['roofline'].get('valu') and round(b['roofline']['valu']['wave_waiting_frac'],3))"
head -4 ${O}_rocprof_kernel_stats_f32_steps20.csv
;;
m)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04m
( timeout 900 python -m pytest tests/test_gpu_static_model.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_closed_loop.py tests/test_gpu_rollout.py -m gpu -q --tb=short 2>&1 | tail -8 ) > ${O}_tests_quick.log 2>&1
for i in 1 2 3; do timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_static.jsonl 2>/dev/null; done
tail -3 ${O}_tests_quick.log; cat ${O}_static.jsonl
;;
n)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04n; AB=$PWD/tactilesimulation_amd/csrc/ab
TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 > ${O}_fine_static.json
TSIM_NO_STATIC=1 TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 > ${O}_fine_generic.json
( TSIM_LPE=16 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short 2>&1 | tail -4 ) > ${O}_parity_lpe16.log 2>&1
cat ${O}_fine_static.json ${O}_fine_generic.json; tail -2 ${O}_parity_lpe16.log
;;
o)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04o; AB=$PWD/tactilesimulation_amd/csrc/ab
( timeout 900 python -m pytest tests/test_gpu_static_model.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_closed_loop.py -m gpu -q --tb=short 2>&1 | tail -12 ) > ${O}_tests_quick.log 2>&1
TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 > ${O}_fine_levels.json
for i in 1 2 3; do
  timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_levels.jsonl 2>/dev/null
  TSIM_HIP_LIB=$AB/libtsim_blocks.so timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_blocks.jsonl 2>/dev/null
done
tail -4 ${O}_tests_quick.log; cat ${O}_fine_levels.json ${O}_levels.jsonl ${O}_blocks.jsonl
;;
p)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04p
rm -f ${O}_sites.jsonl
( TSIM_TEST_REPORT=$PWD/${O}_sites.jsonl timeout 300 python -m pytest tests/test_gpu_static_model.py -m gpu -q --tb=short 2>&1 | head -40 ) > ${O}_static_test.log 2>&1
echo skip > ${O}_tests.log
head -30 ${O}_static_test.log | cut -c1-300; grep -n "^FAILED\|passed\|failed\|Error" ${O}_tests.log | tail -8; grep static ${O}_sites.jsonl
;;
q)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04q; AB=$PWD/tactilesimulation_amd/csrc/ab
( timeout 900 python -m pytest tests/test_gpu_static_model.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_closed_loop.py -m gpu -q --tb=line 2>&1 | tail -6 ) > ${O}_tests_quick.log 2>&1
TSIM_HIP_LIB=$AB/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1 > ${O}_fine.json
for i in 1 2 3; do timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_timed.jsonl 2>/dev/null; done
tail -3 ${O}_tests_quick.log; cat ${O}_fine.json ${O}_timed.jsonl
;;
r)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04r
timeout 200 python tools/bwd_stamps.py 2>/dev/null | tail -1 > ${O}_bwd_static.json
TSIM_NO_STATIC=1 timeout 200 python tools/bwd_stamps.py 2>/dev/null | tail -1 > ${O}_bwd_generic.json
cat ${O}_bwd_static.json ${O}_bwd_generic.json
;;
s)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04s
rm -f ${O}_sites.jsonl
( TSIM_TEST_REPORT=$PWD/${O}_sites.jsonl timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 ) > ${O}_tests.log 2>&1
for i in 1 2; do ( timeout 600 python bench.py --steps 20 --warmup 5 --pmc-dump gpurun_out/r04s_pmc_f32_$i.json > ${O}_bench_$i.json ) 2> ${O}_bench_$i.err; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --timed-only --no-pmc > /dev/null 2>&1 ); cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats_f32_steps20.csv
timeout 300 python bench.py --steps 100 --warmup 10 --no-pmc --no-cpu-baseline --no-sub-records --no-closed-loop 2>/dev/null > ${O}_bench_steps100.json
python -c "import __graft_entry__ as g; g.smoke()" > ${O}_smoke.log 2>&1
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -5; tail -1 ${O}_smoke.log
python -c "
import json
for f in ('${O}_bench_1.json','${O}_bench_2.json','${O}_bench_steps100.json'):
    for l in open(f):
        if l.startswith('{'):
            b=json.loads(l); print(f[-14:], round(b['value']), round(b['ms_per_step'],4), b['roofline']['kernel_ms_per_env_step'], 'closed', b.get('closed_loop',{}).get('value'), 'other', round(b['launch']['other_mode_value']), [round(b[k]['value']) for k in ('f64','push_forward_only_b1024','dclaw','insertion') if k in b], b['roofline'].get('valu') and (round(b['roofline']['valu']['wave_waiting_frac'],3), round(b['roofline']['valu']['valu_wave_insts_per_env_step'])))"
grep static ${O}_sites.jsonl | head -2; head -3 ${O}_rocprof_kernel_stats_f32_steps20.csv | cut -c1-200
;;
t)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04t
B="python bench.py --steps ${STEPS:-20} --warmup 5 --timed-only"
for i in 1 2; do
  for v in free "nofree:TSIM_NO_FREE_RUN=1" "inkernel:TSIM_INKERNEL_READOUT=1" "old:TSIM_INKERNEL_READOUT=1 TSIM_NO_FREE_RUN=1"; do
    n=${v%%:*}; e=${v#*:}; [ "$e" = "$v" ] && e="X_=1"
    env $e timeout 300 $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms'].items()})"
  done
done 2>&1 | tee ${O}_ab.log
;;
u)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_rounds.so python tools/round_stats.py 2>&1 | tail -1; TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_fine.so TSIM_LPE=16 python tools/fine_stamps.py 2>&1 | tail -1 ) | tee gpurun_out/r04u_rounds.log
;;
v)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04v
( timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 ) > ${O}_tests.log 2>&1
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -8
STEPS=20 bash tools/gpu_r04.sh t
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --timed-only --no-pmc > /dev/null 2>&1 ); cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats.csv
head -5 ${O}_rocprof_kernel_stats.csv | cut -c1-160
;;
w)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04w
( timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 ) > ${O}_tests.log 2>&1
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -8
python tools/sub_record_ab.py push_fwd 2>/dev/null | grep "^{" | cut -c1-130
for i in 1; do python bench.py --steps 20 --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('push', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms'].items()})"; done | tee ${O}_ab.log
;;
w2)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for tag in base prev; do
  if [ "$tag" = base ]; then unset TSIM_HIP_LIB; else export TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_$tag.so; fi
  for E in 30 36 40 44; do
    python -c "
import sys, json; sys.path.insert(0, 'tools')
import trained_regime_grad_check as T
o = T.run($E, verbose=False)
f = o['f32']; print('$tag', $E, json.dumps({k: f[k] for k in ('q_err_max', 'branch_agree', 'grad_err_median', 'grad_err_max_agreeing', 'grad_err_max_all')}))" 2>/dev/null | grep "^$tag"
  done
done | tee gpurun_out/r04w2_trained.log
;;
x)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for tag in base ${TAGS:-slp}; do
  if [ "$tag" = base ]; then unset TSIM_HIP_LIB; else export TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_$tag.so; fi
  echo "== $tag"
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
  for s in 20 20; do timeout 300 python bench.py --gpus 1 --steps $s --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps$s', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms'].items()})"; done
done 2>&1 | tee gpurun_out/r04x_ab.log
;;
y)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04y
rm -f ${O}_sites.jsonl
( TSIM_TEST_REPORT=$PWD/${O}_sites.jsonl timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 ) > ${O}_tests.log 2>&1
for i in 1 2; do ( timeout 600 python bench.py --steps 20 --warmup 5 --pmc-dump gpurun_out/r04y_pmc_f32_$i.json > ${O}_bench_$i.json ) 2> ${O}_bench_$i.err; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --timed-only --no-pmc > /dev/null 2>&1 ); cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats_f32_steps20.csv
timeout 300 python bench.py --steps 100 --warmup 10 --no-pmc --no-cpu-baseline --no-sub-records --no-closed-loop 2>/dev/null > ${O}_bench_steps100.json
python -c "import __graft_entry__ as g; g.smoke()" > ${O}_smoke.log 2>&1
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -5; tail -1 ${O}_smoke.log
python -c "
import json
for f in ('${O}_bench_1.json','${O}_bench_2.json','${O}_bench_steps100.json'):
    for l in open(f):
        if l.startswith('{'):
            b=json.loads(l); print(f[-14:], round(b['value']), round(b['ms_per_step'],4), b['roofline']['kernel_ms_per_env_step'], 'closed', b.get('closed_loop',{}).get('value'), 'other', round(b['launch']['other_mode_value']), [round(b[k]['value']) for k in ('f64','push_forward_only_b1024','dclaw','insertion') if k in b], b['roofline'].get('valu') and (round(b['roofline']['valu']['wave_waiting_frac'],3), round(b['roofline']['valu']['valu_wave_insts_per_env_step'])))"
grep static ${O}_sites.jsonl | head -2; head -3 ${O}_rocprof_kernel_stats_f32_steps20.csv | cut -c1-200
;;
z)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for tag in base o2; do
  if [ "$tag" = base ]; then unset TSIM_HIP_LIB; else export TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_$tag.so; fi
  python tools/sub_record_ab.py dclaw insertion 2>/dev/null | grep '^{' | sed "s/^/$tag /" | cut -c1-110
  python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
import bench
r = bench.sub_record("push", "f64", torch.device("cuda:0"), steps=20, warm=5)
print("f64", round(r["value"]), round(r["ms_per_step"], 4))
PY
done
;;
*) echo "usage: bash tools/gpu_r04.sh <a|b|b2|c|d|e|f|g|h|i|j|k|l|m|n|o|p|q|r|s|t|u|v|w|w2|x|y|z>"; exit 2 ;;
esac
