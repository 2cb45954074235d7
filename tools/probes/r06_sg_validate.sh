cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_precision.py tests/test_gpu_e2e.py tests/test_gpu_presplit.py tests/test_gpu_audit.py tests/test_known_answers.py -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_sg_tests.txt; cat gpurun_out/r06_sg_tests.txt
