R=$GRAFT_REPO_ROOT; cd $R
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"])'
for n in 1 2 4 8 16; do
 for t in 64 32 16; do OUSTER_HIP_TILE=$t python bench.py --steps 30 --warmup 5 --no-cpu --frames $n 2>/dev/null | python -c "$P" frames${n}_tile$t; done
 python bench.py --steps 30 --warmup 5 --no-cpu --frames $n 2>/dev/null | python -c "$P" frames${n}_auto
done
