set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc; lscpu | grep "Model name"; rocm-smi --showproductname 2>/dev/null | head -5
python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | grep -vE "^E  *\+|^$" | tail -60
