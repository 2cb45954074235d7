#!/bin/bash
# Operating point B (SCORE_THRESH_TEST = 0.0: 100 detections per frame, 6400 ROIs through the mask / plane / axis heads): bench line,
# rocprofv3 kernel trace + stats, HBM traffic and matrix-pipe occupancy in their own passes.
#   bash tools/profile_pointB.sh r05
set -u
TAG=${1:-r05}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-alt-modes --no-operating-points --no-train-leg --score-thresh 0.0"
CMD="python3 bench.py $ARGS"
rm -rf $OUT/${TAG}_prof_pointB
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_pointB --output-format csv -- python3 bench.py $ARGS > $OUT/${TAG}_bench_prof_pointB.json 2> $OUT/${TAG}_prof_pointB.err
python3 tools/summarize_rocprof.py $OUT/${TAG}_prof_pointB $OUT/${TAG}_bench_prof_pointB.json 5 2 $OUT/${TAG}_kernel_summary_pointB.md "$CMD" > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_pmcB_$C
  rocprofv3 --pmc $C -d $OUT/${TAG}_pmcB_$C --output-format csv -- python3 bench.py $ARGS > $OUT/${TAG}_bench_pmcB_$C.json 2> $OUT/${TAG}_pmcB_$C.err
done
python3 tools/summarize_pmc_traffic.py $OUT/${TAG}_prof_pointB $OUT/${TAG}_pmcB_FETCH_SIZE $OUT/${TAG}_pmcB_WRITE_SIZE $OUT/${TAG}_bench_prof_pointB.json 5 $OUT/${TAG}_traffic_pointB.json > $OUT/${TAG}_traffic_pointB.txt 2>&1
rm -rf $OUT/${TAG}_pmcB_MFMA
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/${TAG}_pmcB_MFMA --output-format csv -- python3 bench.py $ARGS > $OUT/${TAG}_bench_pmcB_MFMA.json 2> $OUT/${TAG}_pmcB_MFMA.err
python3 tools/summarize_pmc_mfma.py $OUT/${TAG}_pmcB_MFMA $OUT/${TAG}_prof_pointB 5 $OUT/${TAG}_pmc_mfma_kernels_pointB.json > $OUT/${TAG}_mfma_pointB.txt 2>&1
rm -rf $OUT/${TAG}_pmcB_FETCH_SIZE $OUT/${TAG}_pmcB_WRITE_SIZE $OUT/${TAG}_pmcB_MFMA
find $OUT/${TAG}_prof_pointB -name "*agent_info.csv" -delete 2>/dev/null
find $OUT/${TAG}_prof_pointB -name "*kernel_trace.csv" -delete 2>/dev/null
head -40 $OUT/${TAG}_kernel_summary_pointB.md
head -20 $OUT/${TAG}_traffic_pointB.txt
head -20 $OUT/${TAG}_mfma_pointB.txt
