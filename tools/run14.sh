R=$GRAFT_REPO_ROOT; cd $R
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["n_gpus"], d["value"], d["ms_per_step"], d["roofline"]["achieved"])'
BENCH_ONE_DEVICE=1 BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --frames 64 --no-cpu 2> gpurun_out/two_rank.err | python -c "$P" two_ranks_one_gpu_gloo; tail -3 gpurun_out/two_rank.err
python -m pytest tests -q -m gpu 2>&1 | tail -5
