# The round-6 GPU sessions, one case per session (run on the GPU box through gpurun from the repo root): bash tools/gpu_r06.sh <session> [args]
set -x
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
S=$1; O=gpurun_out/r06${S}
R=$PWD
ktrace() {  # ktrace <tag> <bench args...>: rocprofv3 kernel trace + stats of a timed-only bench run -> ${O}_rocprof_kernel_stats_<tag>.csv
  local tag=$1; shift
  ( cd /tmp && rm -rf /tmp/kt_$tag && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o kt -- python $R/bench.py "$@" --timed-only > /dev/null 2>&1 )
  cp $(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1) ${O}_rocprof_kernel_stats_$tag.csv
}
case $S in
a)  # new bench line: the driver's command, then the kernel trace of the same timed region (per-kernel HIP-event times must agree with it)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-dump ${O}_pmc_f32.json > ${O}_bench.json 2> ${O}_bench.err; cp bench_detail.json ${O}_bench_detail.json
  wc -c ${O}_bench.json
  ktrace f32_steps20 --steps 20 --warmup 20
  ( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > ${O}_tests.log 2>&1
  ;;
t)  # full GPU suite only
  ( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > ${O}_tests.log 2>&1
  ;;
b)  # D'Claw / TactileInsertion: lanes per environment (= helper slots per wavefront) 64 / 32 / 16 on the straggler-bound collection legs
  for l in 64 32 16; do TSIM_LPE=$l timeout 600 python tools/sub_record_ab.py dclaw insertion >> ${O}_lpe.jsonl 2>> ${O}_lpe.err; done
  ;;
c)  # changed tests, then: the <= 256-register build (two wavefronts per SIMD; csrc/ab/libtsim_w2.so = -DTS_WAVES_PER_EU=2) against the shipped one,
    # timed region only, interleaved, at the batch sizes where a SIMD gets 1 / 2 / 4 wavefronts of four environments (B = 4096 / 8192 / 16 384)
    # and with two environments per wavefront (TSIM_LPE=32: twice the wavefronts)
  ( timeout 1500 python -m pytest tests/test_gpu_param_model.py tests/test_gpu_static_model.py tests/test_gpu_exact_options.py tests/test_gpu_models.py tests/test_gpu_shim.py tests/test_gpu_dclaw.py tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -12 ) > ${O}_tests.log 2>&1
  for B in 4096 8192 16384; do for L in 16 32; do for i in 1 2; do
    TSIM_LPE=$L timeout 300 python bench.py --steps 20 --warmup 5 --batch $B --timed-only >> ${O}_shipped_B${B}_L${L}.jsonl 2>> ${O}_ab.err
    TSIM_LPE=$L TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_w2.so timeout 300 python bench.py --steps 20 --warmup 5 --batch $B --timed-only >> ${O}_w2_B${B}_L${L}.jsonl 2>> ${O}_ab.err
  done; done; done
  ;;
d)  # fused closed loop with per-environment tables (structure-static closed-loop kernels): parity, then the epoch with / without tables;
    # the cost of a tape WITHOUT the Newton matrix (csrc/ab/libtsim_reeval.so = -DTS_BWD_REEVAL: k_backward evaluates it again), interleaved with the shipped library
  ( timeout 1500 python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_exact_options.py tests/test_gpu_batched_env.py -m gpu -q -x 2>&1 | tail -12 ) > ${O}_tests.log 2>&1
  for i in 1 2 3; do
    timeout 300 python bench.py --steps 20 --warmup 5 --timed-only >> ${O}_shipped.jsonl 2>> ${O}_ab.err
    TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_reeval.so timeout 300 python bench.py --steps 20 --warmup 5 --timed-only >> ${O}_reeval.jsonl 2>> ${O}_ab.err
  done
  timeout 600 python tools/closed_loop_tables_bench.py > ${O}_closed_loop_tables.json 2> ${O}_closed_loop_tables.err
  ;;
e)  # CNN policy on the tactile_map observation: graphed == eager, then the bench with the closed-loop legs only
  ( timeout 900 python -m pytest tests/test_gpu_batched_env.py -m gpu -q -x 2>&1 | tail -6 ) > ${O}_tests.log 2>&1
  timeout 600 python bench.py --steps 20 --warmup 5 --legs closed_loop > ${O}_bench.json 2> ${O}_bench.err; cp bench_detail.json ${O}_bench_detail.json
  ;;
z)  # the round's evidence run: full GPU suite, the driver's bench command (with its in-run PMC passes), kernel traces of the timed regions of the
    # headline and of the dclaw / insertion / fp64 legs, the 100-step line, the suite once more with the exact shortcuts OFF
  ( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > ${O}_tests.log 2>&1
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-dump ${O}_pmc_f32.json > ${O}_bench.json 2> ${O}_bench.err; cp bench_detail.json ${O}_bench_detail.json
  ktrace f32_steps20 --steps 20 --warmup 20
  ktrace dclaw --workload dclaw --steps 50 --warmup 50 --repeats 2
  ktrace insertion --workload insertion --steps 9 --warmup 9 --repeats 2
  ktrace f64_steps20 --dtype f64 --steps 20 --warmup 20
  timeout 600 python bench.py --steps 100 --warmup 10 --legs pmc > ${O}_bench_steps100.json 2> ${O}_bench_steps100.err
  ( TSIM_NO_TRIAL_HELPERS=1 TSIM_NO_VALUE_FIRST=1 TSIM_VALUE_TRIALS=0 TSIM_NO_PAIR_CULL=1 timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_exact_options.py 2>&1 | tail -6 ) > ${O}_tests_options_off.log 2>&1
  ;;
f)  # A/B of a library variant (csrc/ab/libtsim_$2.so) against the shipped one on the headline, interleaved, timed region only
  for i in 1 2 3; do
    timeout 300 python bench.py --steps 20 --warmup 5 --timed-only >> ${O}_$2_shipped.jsonl 2>> ${O}_ab.err
    TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_$2.so timeout 300 python bench.py --steps 20 --warmup 5 --timed-only >> ${O}_$2.jsonl 2>> ${O}_ab.err
  done
  ;;
g)  # default-options instantiation: exactness + parity tests, then the headline with it / without it (TSIM_NO_DEFAULT_OPTS=1), interleaved
  ( timeout 1500 python -m pytest tests/test_gpu_exact_options.py tests/test_gpu_static_model.py tests/test_gpu_param_model.py tests/test_gpu_configs.py tests/test_gpu_literal.py tests/test_gpu_edge_cases.py tests/test_gpu_bdf2_adjoint.py -m gpu -q -x 2>&1 | tail -8 ) > ${O}_tests.log 2>&1
  for i in 1 2 3; do
    timeout 300 python bench.py --steps 20 --warmup 5 --timed-only >> ${O}_default_opts.jsonl 2>> ${O}_ab.err
    TSIM_NO_DEFAULT_OPTS=1 timeout 300 python bench.py --steps 20 --warmup 5 --timed-only >> ${O}_runtime_opts.jsonl 2>> ${O}_ab.err
  done
  ;;
h)  # default-options closed-loop instantiation: parity, then the closed-loop leg with / without it
  ( timeout 1500 python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_exact_options.py -m gpu -q -x 2>&1 | tail -6 ) > ${O}_tests.log 2>&1
  for i in 1 2; do
    timeout 300 python tools/closed_loop_tables_bench.py >> ${O}_default_opts.json 2>> ${O}_ab.err
    TSIM_NO_DEFAULT_OPTS=1 timeout 300 python tools/closed_loop_tables_bench.py >> ${O}_runtime_opts.json 2>> ${O}_ab.err
  done
  ;;
esac
