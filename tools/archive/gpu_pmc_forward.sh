# wait / issue decomposition of k_forward and k_backward on the bench's timed region (separate rocprofv3 --pmc passes, counters only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAIT_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_fwd_$i -o pmc -- python bench.py --steps 20 --warmup 20 --no-cpu-baseline --timed-only --no-pmc > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_fwd_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_forward" in k or "k_backward" in k:
            per[(k[:30], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c, d), v in per.items():
        acc[k][c].append(v)
for k in acc:
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    wc = m.get("SQ_WAVE_CYCLES", 1.0)
    print(k)
    for c in sorted(m): print("   %-24s %14.0f  %6.3f of wave cycles" % (c, m[c], m[c] / wc))
PY
