# hardware counters of the bench kernels (separate rocprofv3 --pmc passes; no trace domains combined with --pmc)
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$TAG
timeout 120 rocprofv3 -L > gpurun_out/pmc_$TAG/counters_list.txt 2>&1
FRAMES=${2:-20}
run() { # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$TAG/$n -o pmc -- python bench.py --steps $FRAMES --warmup $FRAMES --no-cpu-baseline --timed-only > gpurun_out/pmc_$TAG/$n.log 2>&1
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU
run p3 FETCH_SIZE
run p4 WRITE_SIZE
python tools/pmc_to_json.py gpurun_out/pmc_$TAG $FRAMES gpurun_out/pmc_$TAG.json
