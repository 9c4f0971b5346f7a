cd $GRAFT_REPO_ROOT
timeout 300 python tools/rolling_ball_speed.py 2>/dev/null | tail -1
bash tools/gpu_ab.sh libtsim_hip.so
