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
