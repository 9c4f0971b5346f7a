cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04t
B="python bench.py --steps ${STEPS:-20} --warmup 5 --timed-only"
for i in 1 2; do
  for v in free "nofree:TSIM_NO_FREE_RUN=1" "inkernel:TSIM_INKERNEL_READOUT=1" "old:TSIM_INKERNEL_READOUT=1 TSIM_NO_FREE_RUN=1"; do
    n=${v%%:*}; e=${v#*:}; [ "$e" = "$v" ] && e="X_=1"
    env $e timeout 300 $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms'].items()})"
  done
done 2>&1 | tee ${O}_ab.log
