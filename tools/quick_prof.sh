cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/sp
rocprofv3 --kernel-trace --stats -d /tmp/sp --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes --no-operating-points $QP_ARGS > /tmp/sp.json 2>/dev/null
tail -1 /tmp/sp.json | cut -c60-140
python3 tools/summarize_rocprof.py /tmp/sp /tmp/sp.json 10 3 /tmp/sp.md > /dev/null; sed -n 8,32p /tmp/sp.md | cut -c1-150
