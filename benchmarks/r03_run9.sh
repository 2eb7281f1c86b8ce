cd $GRAFT_REPO_ROOT
export MVDB_BENCH_SHARE_GPU=1 MVDB_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 23555 bench.py --gpus 2 --steps 12 --warmup 2 --rows 1000000 --dim 512 --k 10 --dump /tmp/dump.npz > /tmp/b.out 2> /tmp/b.err; echo "rc=$?"
tail -c 600 /tmp/b.out; echo; tail -c 2500 /tmp/b.err
