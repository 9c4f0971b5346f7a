# The round-5 GPU sessions, one function per session (run on the GPU box through gpurun from the repo root): bash tools/gpu_r05.sh <session>
set -x
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
S=$1; O=gpurun_out/r05${S}
case $S in
a)  # pair cull + structure-static kernels: parity, then the legs they were built for
  timeout 900 python -m pytest tests/test_gpu_exact_options.py tests/test_gpu_param_model.py tests/test_gpu_static_model.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -15 > ${O}_tests.log
  timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "dclaw or insertion or ball_push or slider" 2>&1 | tail -8 >> ${O}_tests.log
  timeout 600 python tools/sub_record_ab.py dclaw insertion push_fwd > ${O}_legs_cull.jsonl 2> ${O}_legs_cull.err
  TSIM_NO_PAIR_CULL=1 timeout 600 python tools/sub_record_ab.py dclaw insertion push_fwd > ${O}_legs_nocull.jsonl 2> ${O}_legs_nocull.err
  timeout 300 python tools/param_tables_probe.py > ${O}_param_tables.json 2> ${O}_param_tables.err
  ;;
b)  # where an evaluation's cycles go on the generic kernels (D'Claw, TactileInsertion), the structure-static kernels on the bench's own inputs
  timeout 600 python -m pytest tests/test_gpu_param_model.py -x -q -m gpu 2>&1 | tail -5 > ${O}_tests.log
  for w in dclaw insertion; do for l in 32 16; do timeout 300 python tools/eval_stamps.py $w $l 20 >> ${O}_stamps.jsonl 2>> ${O}_stamps.err; done; done
  timeout 300 python tools/eval_stamps.py insertion 32 40 >> ${O}_stamps.jsonl 2>> ${O}_stamps.err
  timeout 300 python tools/param_tables_probe.py > ${O}_param_tables.json 2> ${O}_param_tables.err
  ;;
c)  # value-only line-search trials: exactness, then the straggler-bound legs with the option off / on
  timeout 900 python -m pytest tests/test_gpu_exact_options.py tests/test_gpu_param_model.py tests/test_gpu_static_model.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -15 > ${O}_tests.log
  timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_literal.py -x -q -m gpu 2>&1 | tail -8 >> ${O}_tests.log
  for v in 0 1 2 3; do TSIM_VALUE_TRIALS=$v timeout 600 python tools/sub_record_ab.py dclaw insertion push_fwd >> ${O}_legs.jsonl 2>> ${O}_legs.err; done
  for v in 0 2; do TSIM_VALUE_TRIALS=$v timeout 300 python tools/param_tables_probe.py >> ${O}_param_tables.jsonl 2>> ${O}_param_tables.err; done
  ;;
d)  # the shortcuts A/B in one process (launch-by-launch times): D'Claw, TactileInsertion, TactilePush forward-only and the headline's forward launch
  timeout 600 python -m pytest tests/test_gpu_exact_options.py -x -q -m gpu 2>&1 | tail -15 > ${O}_tests.log
  for w in dclaw insertion push_fwd push; do timeout 600 python tools/option_ab.py $w >> ${O}_option_ab.jsonl 2>> ${O}_option_ab.err; done
  timeout 600 python tools/option_ab.py dclaw 8192 3 >> ${O}_option_ab.jsonl 2>> ${O}_option_ab.err
  ;;
e)  # re-entry: where an evaluation's cycles go (generic kernels, D'Claw / TactileInsertion), the shortcuts A/B, evaluation totals per environment, static vs generic tolerance
  for w in dclaw insertion; do for l in 32 16; do timeout 300 python tools/eval_stamps.py $w $l 20 >> ${O}_stamps.jsonl 2>> ${O}_stamps.err; done; done
  timeout 300 python tools/static_vs_generic_probe.py > ${O}_static_vs_generic.json 2> ${O}_static_vs_generic.err
  for w in dclaw insertion; do timeout 300 python tools/evals_distribution.py $w >> ${O}_evals.jsonl 2>> ${O}_evals.err; done
  timeout 300 python tools/evals_distribution.py dclaw 8192 >> ${O}_evals.jsonl 2>> ${O}_evals.err
  for w in dclaw insertion; do timeout 600 python tools/option_ab.py $w >> ${O}_option_ab.jsonl 2>> ${O}_option_ab.err; done
  timeout 600 python tools/option_ab.py dclaw 8192 3 >> ${O}_option_ab.jsonl 2>> ${O}_option_ab.err
  ;;
f)  # helper slots: exactness, then the legs with the option off / on (one process per leg: launch-by-launch times)
  timeout 900 python -m pytest tests/test_gpu_exact_options.py -x -q -m gpu 2>&1 | tail -15 > ${O}_tests.log
  for w in dclaw insertion push_fwd push; do timeout 600 python tools/helpers_ab.py $w >> ${O}_helpers_ab.jsonl 2>> ${O}_helpers_ab.err; done
  timeout 600 python tools/helpers_ab.py dclaw 8192 3 >> ${O}_helpers_ab.jsonl 2>> ${O}_helpers_ab.err
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > ${O}_bench.json 2> ${O}_bench.err
  ;;
g)  # headline with the helper slots compiled in: option on / off, three runs each (timed region only)
  timeout 300 python -m pytest tests/test_gpu_exact_options.py -x -q -m gpu 2>&1 | tail -3 > ${O}_tests.log
  for i in 1 2 3; do
    timeout 300 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_on.jsonl 2>> ${O}_on.err
    TSIM_NO_TRIAL_HELPERS=1 timeout 300 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_off.jsonl 2>> ${O}_off.err
  done
  ;;
h)  # headline: this build against an A/B library (csrc/ab/libtsim_$2.so), interleaved, timed region only
  for i in 1 2 3; do
    timeout 300 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_new.jsonl 2>> ${O}_new.err
    TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_$2.so timeout 300 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_$2.jsonl 2>> ${O}_$2.err
  done
  ;;
i)  # value-first trials: exactness, the forward-only legs off / on, the headline against the round's first library
  timeout 900 python -m pytest tests/test_gpu_exact_options.py -x -q -m gpu 2>&1 | tail -15 > ${O}_tests.log
  for w in dclaw insertion push_fwd; do timeout 600 python tools/value_first_ab.py $w >> ${O}_value_first_ab.jsonl 2>> ${O}_value_first_ab.err; done
  ;;
z)  # the round's evidence run: full GPU suite, the driver's bench command (with its in-run PMC passes), kernel traces of the timed regions of the
    # headline and of the dclaw / insertion / fp64 legs, the 100-step line
  ( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > ${O}_tests.log 2>&1
  timeout 900 python bench.py --steps 20 --warmup 5 --pmc-dump ${O}_pmc_f32.json > ${O}_bench.json 2> ${O}_bench.err
  R=$PWD
  # (warm-up as long as an episode: every dispatch of a kernel in the stats is then ONE full launch — a 5-step warm-up launch would be averaged in)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_push -o kt -- python $R/bench.py --steps 20 --warmup 20 --timed-only --no-pmc > /dev/null 2>&1 ); cp $(find /tmp/kt_push -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats_f32_steps20.csv
  for w in dclaw insertion; do ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -o kt -- python $R/bench.py --workload $w --steps $([ $w = dclaw ] && echo 50 || echo 9) --warmup $([ $w = dclaw ] && echo 50 || echo 9) --timed-only --no-pmc --repeats 2 > /dev/null 2>&1 ); cp $(find /tmp/kt_$w -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats_$w.csv; done
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_f64 -o kt -- python $R/bench.py --dtype f64 --steps 20 --warmup 20 --timed-only --no-pmc > /dev/null 2>&1 ); cp $(find /tmp/kt_f64 -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats_f64_steps20.csv
  timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sub-records --no-closed-loop > ${O}_bench_steps100.json 2> ${O}_bench_steps100.err
  ;;
esac
