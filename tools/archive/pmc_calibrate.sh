# units of SQ_THREAD_CYCLES_VALU vs SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU: a full-lane elementwise torch kernel as the known case
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_cal
timeout 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_cal -o cal -- python -c "
import torch
a = torch.randn(1 << 26, device='cuda'); b = torch.randn(1 << 26, device='cuda')
for _ in range(3): c = a * b + a
torch.cuda.synchronize()" > gpurun_out/pmc_cal/log.txt 2>&1
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_cal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "elementwise" in r["Kernel_Name"] and int(r["Grid_Size"]) >= (1 << 24):
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
out = []
for d, c in list(agg.items())[-3:]:
    out.append({"counters": dict(c), "thread_cycles_per_valu_inst": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"],
                "thread_cycles_per_active_quadcycle": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]})
print(json.dumps(out, indent=1)); json.dump(out, open("gpurun_out/pmc_calibration.json", "w"), indent=1)
PY
