#!/bin/bash
# Dev tool (GPU box): the bench step under a list of environment settings, interleaved and repeated (clock drift hits all alike).
#   bash tools/step_ab.sh "A3D_WINO_TUNE=0" "A3D_WINO_TUNE=23" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPS=${REPS:-2}
for r in $(seq $REPS); do
  for cfg in "$@"; do
    env $cfg python3 bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-alt-modes --no-operating-points ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['all_conv_kernels']
fam = lambda p: sum(v['ms_per_step'] for n, v in k.items() if n.startswith(p))
print('$cfg', '|', d['value'], 'frames/s', d['ms_per_step'], 'ms | wino_gemm %.2f wino_input %.2f conv_h2<2> %.2f conv_h2<1> %.2f h2w %.2f ph4p %.2f xs %.2f' % (
    fam('wino_gemm'), fam('wino_input'), fam('conv_h2_kernel<2>'), fam('conv_h2_kernel<1>'), fam('conv_h2w'), fam('conv_ph4p') + fam('conv_c3p'), fam('conv_h2xs')))
"
  done
done
