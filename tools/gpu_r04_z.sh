cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_static_model.py tests/test_gpu_configs.py tests/test_gpu_edge_cases.py -m gpu -q 2>&1 | tail -3
python tools/sub_record_ab.py push_fwd 2>/dev/null | grep '^{' | cut -c1-200
