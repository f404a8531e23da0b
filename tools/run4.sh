R=$GRAFT_REPO_ROOT; cd $R
python tools/debug1.py > gpurun_out/debug1.log 2>&1; cat gpurun_out/debug1.log
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/gpu_tests4.log; cat gpurun_out/gpu_tests4.log
bash tools/gpu_ab.sh tools/ab/libouster_hip_v1.so ouster_sdk_amd/lib/libouster_hip.so 3 > gpurun_out/ab4.log 2>&1; cat gpurun_out/ab4.log
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["kernel_ms_avg"])'
for v in xyz planes planes+dst; do python bench.py --steps 10 --warmup 2 --no-cpu --outputs $v 2>/dev/null | python -c "$P" $v; done > gpurun_out/ablate4.log 2>&1; cat gpurun_out/ablate4.log
