# usage (on the GPU box, via gpurun): bash tools/gpu_bench.sh [tag] [what...]
# every command is wrapped in its own timeout: a wedged profiler must not eat the GPU budget.
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}; shift
WHAT=${@:-"tests acc bench prof"}
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in $WHAT; do
case $w in
tests) timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "^E  *\+|^$" | tail -25 ;;
smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
acc) timeout 600 python tools/accuracy_report.py $TAG 2>&1 | tail -80 ;;
bench) timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_f32_$TAG.json 2> gpurun_out/bench_f32_$TAG.err; cat gpurun_out/bench_f32_$TAG.json; tail -3 gpurun_out/bench_f32_$TAG.err
       timeout 600 python bench.py --steps 100 --warmup 10 --dtype f64 --no-cpu-baseline > gpurun_out/bench_f64_$TAG.json 2> gpurun_out/bench_f64_$TAG.err; cat gpurun_out/bench_f64_$TAG.json; tail -3 gpurun_out/bench_f64_$TAG.err ;;
prof) timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o prof -- python bench.py --steps 20 --warmup 20 --no-cpu-baseline --timed-only > gpurun_out/prof_$TAG.log 2>&1
      f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); head -8 $f ;;
esac
done
