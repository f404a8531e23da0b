R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/gpu_tests3.log; cat gpurun_out/gpu_tests3.log
./tools/membench > gpurun_out/membench.log 2>&1; cat gpurun_out/membench.log
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["kernel_ms_avg"])'
for v in full xyz planes planes+dst; do python bench.py --steps 10 --warmup 2 --no-cpu --outputs $v 2>/dev/null | python -c "$P" $v; done > gpurun_out/ablate1.log 2>&1
OUSTER_HIP_TILE=32 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "$P" tile32 >> gpurun_out/ablate1.log
OUSTER_HIP_XCD=0 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "$P" noxcd >> gpurun_out/ablate1.log
python bench.py --steps 10 --warmup 2 --no-cpu --frames 1024 2>/dev/null | python -c "$P" frames1024 >> gpurun_out/ablate1.log
python bench.py --steps 10 --warmup 2 --no-cpu --frames 64 2>/dev/null | python -c "$P" frames64 >> gpurun_out/ablate1.log
cat gpurun_out/ablate1.log
