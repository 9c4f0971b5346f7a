cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04c; AB=$PWD/tactilesimulation_amd/csrc/ab
( timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60 ) > ${O}_tests.log 2>&1
python -c "
import sys, json; sys.path.insert(0, 'tools')
import trained_regime_grad_check as T
print(json.dumps(T.run(40)))" > ${O}_trained_valu.json 2>/dev/null
TSIM_HIP_LIB=$AB/libtsim_mfma.so python -c "
import sys, json; sys.path.insert(0, 'tools')
import trained_regime_grad_check as T
print(json.dumps(T.run(40)))" > ${O}_trained_mfma.json 2>/dev/null
( timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench.json ) 2> ${O}_bench.err
grep -n "^FAILED\|passed\|failed" ${O}_tests.log | tail; cat ${O}_trained_valu.json ${O}_trained_mfma.json; python -c "
import json
for l in open('${O}_bench.json'):
    if l.startswith('{'):
        b=json.loads(l); print(b['value'], b['ms_per_step'], b['insertion']['value'], b['readout_hbm']['achieved'], [x['achieved'] for x in b['readout_hbm']['by_batch']], b['readout_hbm'].get('pmc'))"
