# rocprofv3 kernel summary of bench.py --precision bf16x3 (profiles/r01_bf16x3_kernel_summary.md): run through gpurun, then tools/summarize_rocprof.py.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/prof_x3; rm -rf $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16x3 > gpurun_out/bench_x3_prof.json 2> gpurun_out/prof_x3.err
tail -c 300 gpurun_out/bench_x3_prof.json
