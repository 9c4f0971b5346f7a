cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04w
( timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30 ) > ${O}_tests.log 2>&1
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail -8
python tools/sub_record_ab.py push_fwd 2>/dev/null | grep "^{" | cut -c1-130
for i in 1; do python bench.py --steps 20 --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('push', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms'].items()})"; done | tee ${O}_ab.log
