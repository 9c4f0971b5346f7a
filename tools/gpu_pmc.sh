# hardware counters of the bench kernels (separate rocprofv3 --pmc passes; no trace domains combined with --pmc)
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$TAG
timeout 120 rocprofv3 -L > gpurun_out/pmc_$TAG/counters_list.txt 2>&1
grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|GRBM_[A-Z_]+)\b" gpurun_out/pmc_$TAG/counters_list.txt | sort -u | tr '\n' ' ' | head -c 6000; echo
run() { # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$TAG/$n -o pmc -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$TAG/$n.log 2>&1
  f=$(find gpurun_out/pmc_$TAG/$n -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r.get('Kernel_Name','')[:30]
    if 'k_forward' not in k and 'k_backward' not in k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k in agg:
    print(k, {c: round(v/ max(cnt[(k,c)],1)) for c,v in agg[k].items()}, 'dispatches', max(cnt[(k,c)] for c in agg[k]))
PY
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU
run p3 FETCH_SIZE
run p4 WRITE_SIZE
