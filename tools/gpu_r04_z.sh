cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for tag in base o2; do
  if [ "$tag" = base ]; then unset TSIM_HIP_LIB; else export TSIM_HIP_LIB=$PWD/tactilesimulation_amd/csrc/ab/libtsim_$tag.so; fi
  python tools/sub_record_ab.py dclaw insertion 2>/dev/null | grep '^{' | sed "s/^/$tag /" | cut -c1-110
  python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
import bench
r = bench.sub_record("push", "f64", torch.device("cuda:0"), steps=20, warm=5)
print("f64", round(r["value"]), round(r["ms_per_step"], 4))
PY
done
