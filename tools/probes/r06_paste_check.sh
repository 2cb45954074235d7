cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_oracle_golden.py tests/test_frontend.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r06_paste_tests.txt; cat gpurun_out/r06_paste_tests.txt
python tools/loop_bench.py 2>&1 | grep -v amdgpu.ids | head -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes --no-operating-points --no-train-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[0]); print(d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r06_pp; rocprofv3 --kernel-trace --stats -d gpurun_out/r06_pp --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes --no-operating-points --no-train-leg > /dev/null 2>&1
grep -h "paste_lsq" gpurun_out/r06_pp/*/*kernel_stats.csv | cut -c1-160; rm -rf gpurun_out/r06_pp
