cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for lib in liba3d_hip.so liba3d_hip_oldsel.so; do
  A3D_ALT_LIB=$lib python tools/probes/run_with_alt_lib.py bench.py --batch 1 --no-cpu-baseline --no-train-leg --no-alt-modes --no-operating-points --steps 60 --warmup 10 2>/dev/null > /tmp/b.json; python -c "import json;d=json.loads(open('/tmp/b.json').read().splitlines()[0]);print('$lib bench1', d['value'], d['ms_per_step'])"
  A3D_ALT_LIB=$lib python tools/probes/run_with_alt_lib.py tools/loop_bench.py 2>&1 | grep "reference-style"
  A3D_ALT_LIB=$lib python tools/probes/run_with_alt_lib.py tools/train_bench.py --precision bf16 --batch 2 --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().splitlines()[-1]);print('$lib train2', d['value'], d['ms_per_step'])"
  A3D_ALT_LIB=$lib python tools/probes/run_with_alt_lib.py bench.py --no-cpu-baseline --no-train-leg --no-alt-modes --no-operating-points --steps 20 2>/dev/null > /tmp/b.json; python -c "import json;d=json.loads(open('/tmp/b.json').read().splitlines()[0]);print('$lib bench64', d['value'], d['ms_per_step'])"
done; done
