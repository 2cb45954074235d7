cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r06_b1prof
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_b1prof --output-format csv -- python3 bench.py --batch 1 --steps 40 --warmup 10 --no-cpu-baseline --no-alt-modes --no-operating-points --no-train-leg > gpurun_out/r06_b1.json 2>/dev/null
python3 tools/summarize_rocprof.py gpurun_out/r06_b1prof gpurun_out/r06_b1.json 40 10 gpurun_out/r06_b1_kernel_summary.md > /dev/null
head -45 gpurun_out/r06_b1_kernel_summary.md | cut -c1-150
