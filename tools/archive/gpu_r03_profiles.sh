# round-3 evidence run: the driver's bench command with the in-run PMC passes, a kernel-trace of the same timed region, the 100-step
# variant, the f64 headline, the read-out kernels.  Everything lands in gpurun_out/ (copied into profiles/ by hand afterwards).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --pmc-dump gpurun_out/r03_pmc_f32.json > gpurun_out/r03_bench_f32_steps20.json 2> gpurun_out/r03_bench_f32_steps20.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_kt -o kt -- python bench.py --steps 20 --warmup 20 --no-cpu-baseline --timed-only --no-pmc > gpurun_out/r03_kt.log 2>&1
cp $(find gpurun_out/r03_kt -name "*kernel_stats.csv" | head -1) gpurun_out/r03_rocprof_kernel_stats_f32_steps20.csv
timeout 300 python bench.py --steps 100 --warmup 10 --no-pmc --no-cpu-baseline --no-closed-loop --no-sub-records > gpurun_out/r03_bench_f32_steps100.json 2>/dev/null
timeout 300 python bench.py --dtype f64 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-closed-loop --no-sub-records > gpurun_out/r03_bench_f64_steps20.json 2>/dev/null
timeout 300 python bench.py --workload dclaw --steps 40 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/r03_bench_dclaw.json 2>/dev/null
timeout 300 python bench.py --workload insertion --steps 18 --warmup 9 --no-pmc --no-cpu-baseline > gpurun_out/r03_bench_insertion.json 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_ro -o ro -- python tools/readout_ab.py > gpurun_out/r03_readout.log 2>&1
cp $(find gpurun_out/r03_ro -name "*kernel_stats.csv" | head -1) gpurun_out/r03_rocprof_kernel_stats_readout.csv
TSIM_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --batch 2048 --no-cpu-baseline --no-pmc > gpurun_out/r03_bench_2ranks_shared_gpu.json 2>/dev/null
head -c 300 gpurun_out/r03_bench_f32_steps20.json; echo; head -4 gpurun_out/r03_rocprof_kernel_stats_f32_steps20.csv; for f in steps100 ; do head -c 250 gpurun_out/r03_bench_f32_$f.json; echo; done; head -c 250 gpurun_out/r03_bench_f64_steps20.json; echo; head -c 250 gpurun_out/r03_bench_dclaw.json; echo; head -c 250 gpurun_out/r03_bench_insertion.json; echo; head -c 300 gpurun_out/r03_bench_2ranks_shared_gpu.json; echo
