"""Collect the rocprofv3 --pmc passes of tools/gpu_pmc.sh into profiles-style JSON (mean per dispatch)."""
import csv, sys, json, glob, collections, os
d, frames, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_forward" not in k and "k_backward" not in k:
            continue
        k = k.split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
res = {"note": "rocprofv3 --pmc (tools/gpu_pmc.sh), separate passes (SQ x2, FETCH_SIZE, WRITE_SIZE), mean per dispatch of "
               "`python bench.py --steps %d --warmup %d --timed-only` fp32 B=4096 (one k_forward / k_backward launch = %d env-steps); FETCH_SIZE/"
               "WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE under-reports wide reads by 2x, MI355X_MICROARCH.md §HBM); "
               "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles" % (frames, frames, frames),
       "frames_per_launch": frames,
       "per_kernel": {k: {c: v / cnt[(k, c)] for c, v in agg[k].items()} for k in agg},
       "dispatches": {k: max(cnt[(k, c)] for c in agg[k]) for k in agg}}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["per_kernel"], indent=1)[:3000])
