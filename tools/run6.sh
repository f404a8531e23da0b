R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/gpu_tests6.log; cat gpurun_out/gpu_tests6.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 > gpurun_out/bench6.json 2> gpurun_out/bench6.err; tail -3 gpurun_out/bench6.err; cat gpurun_out/bench6.json
