cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dclaw.py tests/test_gpu_models.py -m gpu -q -x 2>&1 | tail -2
for v in "X_=1" "TSIM_NO_STATIC=1"; do env $v python tools/sub_record_ab.py dclaw 2>/dev/null | grep '^{' | sed "s/^/$v /" | cut -c1-200; done | tee gpurun_out/r04z_dclaw_static.log
