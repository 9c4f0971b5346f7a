cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 600 python tools/accuracy_report.py ${1:-q} > /dev/null 2>&1; python -c "
import json; a=json.load(open('gpurun_out/accuracy_${1:-q}.json'))
for r in a['accuracy']: print(r['dtype'], 'grad_max', '%.2e'%r['grad_max_rel_err'], 'tac_max', '%.2e'%r['tactile_max_rel_err'], 'q_max', '%.2e'%r['q_max_abs_err'])
for r in a['phase_cycles']: print(r['dtype'], r['B'], [round(x) for x in r['stamp_deltas']], round(r['total']))
"
bash tools/gpu_ab.sh libtsim_hip.so
