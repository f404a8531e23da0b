R=$GRAFT_REPO_ROOT; cd $R
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["kernel_ms_avg"])'
for i in 1 2 3; do
python bench.py --steps 40 --warmup 5 --no-cpu 2>/dev/null | python -c "$P" A_256
OUSTER_HIP_TILE=64 OUSTER_HIP_SO=$R/tools/ab/libouster_hip_nt512.so python bench.py --steps 40 --warmup 5 --no-cpu 2>/dev/null | python -c "$P" B_512_tile64_1wg
OUSTER_HIP_SO=$R/tools/ab/libouster_hip_nt512.so python bench.py --steps 40 --warmup 5 --no-cpu 2>/dev/null | python -c "$P" C_512_tile32_2wg
done
OUSTER_HIP_TILE=64 OUSTER_HIP_SO=$R/tools/ab/libouster_hip_nt512.so python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
