cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04r
timeout 200 python tools/bwd_stamps.py 2>/dev/null | tail -1 > ${O}_bwd_static.json
TSIM_NO_STATIC=1 timeout 200 python tools/bwd_stamps.py 2>/dev/null | tail -1 > ${O}_bwd_generic.json
cat ${O}_bwd_static.json ${O}_bwd_generic.json
