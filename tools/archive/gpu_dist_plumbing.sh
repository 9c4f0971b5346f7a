# exercises bench.py's N > 1 code path on a 1-GPU box: 2 ranks share cuda:0, collectives over gloo (NOT a scaling number)
cd $GRAFT_REPO_ROOT
TSIM_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --batch 1024 --backend gloo --no-cpu-baseline 2>&1 | tail -3 | cut -c1-400
