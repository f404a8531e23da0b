R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -q -m gpu -x 2>&1 | tail -4
bash tools/gpu_ab.sh tools/ab/libouster_hip_head.so ouster_sdk_amd/lib/libouster_hip.so 4 2>&1 | tee gpurun_out/ab19.log
LD_LIBRARY_PATH=$R/ouster_sdk_amd/lib:/opt/rocm/lib ./tests/cpp/_build/bench_host_api 20 | tee gpurun_out/host_api_latency.json
