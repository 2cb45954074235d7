cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_audit.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r06_plan_tests.txt; cat gpurun_out/r06_plan_tests.txt
python tools/probes/b1_host_probe.py 2>&1 | grep -v amdgpu.ids | head -3
python tools/loop_bench.py 2>&1 | grep -v amdgpu.ids | head -4
A3D_LAUNCH_PLANS=0 python tools/loop_bench.py 2>&1 | grep -v amdgpu.ids | head -4
