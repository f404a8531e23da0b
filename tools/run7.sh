R=$GRAFT_REPO_ROOT; cd $R
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["kernel_ms_avg"])'
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "$P" null_w3
OUSTER_HIP_OWN_STREAM=1 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "$P" own_w3
python bench.py --steps 200 --warmup 50 --no-cpu 2>/dev/null | python -c "$P" null_w50_s200
OUSTER_HIP_OWN_STREAM=1 python bench.py --steps 200 --warmup 50 --no-cpu 2>/dev/null | python -c "$P" own_w50_s200
done
rocm-smi --showclocks 2>/dev/null | head -20
