R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/gpu_tests9.log; cat gpurun_out/gpu_tests9.log
bash tools/gpu_ab.sh tools/ab/libouster_hip_v2.so ouster_sdk_amd/lib/libouster_hip.so 3 > gpurun_out/ab9.log 2>&1; cat gpurun_out/ab9.log
