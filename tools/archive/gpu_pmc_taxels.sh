cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_tax_$i -o pmc -- python tools/readout_ab.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_tax_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_taxels" in k or "k_readout" in k:
            per[(k[:40], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c, d), v in per.items():
        acc[k][c].append(v)
for k in acc:
    print(k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()})
PY
