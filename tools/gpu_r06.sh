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
esac
