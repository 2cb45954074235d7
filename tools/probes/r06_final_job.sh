cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_presplit.py tests/test_gpu_training.py tests/test_known_answers.py -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_tests5.txt; cat gpurun_out/r06_tests5.txt
bash tools/profile_train.sh r06 > gpurun_out/r06_profile_train.log 2>&1; tail -8 gpurun_out/r06_profile_train.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r06_xchg_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_xchg_prof --output-format csv -- python3 tools/probes/exchange_probe_run.py 2 > gpurun_out/r06_xchg_prof.json 2> gpurun_out/r06_xchg_prof.err
python3 tools/trace_streams.py gpurun_out/r06_xchg_prof > gpurun_out/r06_train_exchange_streams.txt 2>&1; head -60 gpurun_out/r06_train_exchange_streams.txt
python tools/kernel_table.py --out gpurun_out/r06_kernel_table.md 2>&1 | tail -1
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1; tail -3 gpurun_out/r06_profile_round.log | cut -c1-200
