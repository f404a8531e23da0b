R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "odd_geometries" 2>&1 | tail -12
python bench.py --steps 10 --warmup 2 --no-cpu --pcie 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"], d["pcie_inclusive"])'
python bench.py --steps 10 --warmup 2 --no-cpu --frames 1 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("single-frame", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"])'
python bench.py --steps 10 --warmup 2 --no-cpu --frames 8 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("8-frame", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"])'
