cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/b1prof
rocprofv3 --kernel-trace -d gpurun_out/b1prof --output-format csv -- python3 bench.py --batch 1 --steps 40 --warmup 10 --no-cpu-baseline --no-alt-modes --no-operating-points > gpurun_out/b1.json 2>/dev/null
tail -1 gpurun_out/b1.json | cut -c1-200
python3 bench.py --batch 1 --steps 40 --warmup 10 --no-cpu-baseline --no-alt-modes --no-operating-points 2>/dev/null | tail -1 | cut -c60-200
