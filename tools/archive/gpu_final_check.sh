cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_static_model.py -m gpu -q 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-sub-records --no-closed-loop 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'])"
