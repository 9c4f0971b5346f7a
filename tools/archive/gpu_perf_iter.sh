# one perf iteration on the GPU box: correctness subset, fine stamps (A/B build), the driver's bench command (timed region only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-x}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_models.py -m gpu -x -q 2>&1 | tail -4
if [ -f tactilesimulation_amd/csrc/ab/libtsim_fine.so ]; then TSIM_HIP_LIB=tactilesimulation_amd/csrc/ab/libtsim_fine.so TSIM_LPE=16 timeout 200 python tools/fine_stamps.py 2>&1 | tail -1; fi
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-closed-loop --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps20', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms_per_env_step'], d['launch']['other_mode_value'])"; done
timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --no-pmc --no-closed-loop --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps100', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms_per_env_step'])"
