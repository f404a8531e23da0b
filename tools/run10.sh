R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
bash tools/gpu_ab.sh tools/ab/libouster_hip_v2.so ouster_sdk_amd/lib/libouster_hip.so 3 > gpurun_out/ab10.log 2>&1; cat gpurun_out/ab10.log
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d["roofline"]["box_d2d_copy_GBps"])'
for w in single fused4 dual; do python bench.py --steps 20 --warmup 3 --no-cpu --workload $w 2>/dev/null | python -c "$P" $w; done > gpurun_out/workloads10.log 2>&1; cat gpurun_out/workloads10.log
