R=$GRAFT_REPO_ROOT; cd $R
BENCH_ONE_DEVICE=1 BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --frames 64 --no-cpu > gpurun_out/two_rank.out 2> gpurun_out/two_rank.err; echo rc=$?; echo STDOUT; cat gpurun_out/two_rank.out; echo STDERR; tail -5 gpurun_out/two_rank.err
python -m pytest tests/test_python_api.py -q -m gpu 2>&1 | tail -15
