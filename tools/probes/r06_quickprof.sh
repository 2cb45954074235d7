cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r06_qprof
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes --no-operating-points --no-train-leg"
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_qprof --output-format csv -- python3 bench.py $ARGS > gpurun_out/r06_qprof.json 2>/dev/null
python3 tools/summarize_rocprof.py gpurun_out/r06_qprof gpurun_out/r06_qprof.json 10 3 gpurun_out/r06_qprof_summary.md > /dev/null
sed -n 4,50p gpurun_out/r06_qprof_summary.md | cut -c1-150
