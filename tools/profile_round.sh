#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel trace + stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in their own
# passes), matrix-pipe occupancy.  Summaries land in gpurun_out/ (copy the ones to be judged into profiles/).
#   bash tools/profile_round.sh r03
set -u
TAG=${1:-r05}
OUT=gpurun_out
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p gpurun_out
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes --no-operating-points --no-train-leg"
python3 bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
for MODE in fp16x2 bf16x3 fp32; do
  SUF=""; [ $MODE = fp32 ] && SUF="_fp32"; [ $MODE = bf16x3 ] && SUF="_bf16x3"
  rm -rf $OUT/${TAG}_prof$SUF
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof$SUF --output-format csv -- python3 bench.py $ARGS --precision $MODE > $OUT/${TAG}_bench_prof$SUF.json 2> $OUT/${TAG}_prof$SUF.err
  python3 tools/summarize_rocprof.py $OUT/${TAG}_prof$SUF $OUT/${TAG}_bench_prof$SUF.json 10 3 $OUT/${TAG}_kernel_summary$SUF.md > /dev/null
done
# the default arithmetic once more with the depth decoder and the ROI heads NOT on their own streams: every kernel has the chip alone
rm -rf $OUT/${TAG}_prof_alone
A3D_DEPTH_OVERLAP=0 A3D_HEADS_CONCURRENT_ROWS=0 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_alone --output-format csv -- python3 bench.py $ARGS > $OUT/${TAG}_bench_prof_alone.json 2> $OUT/${TAG}_prof_alone.err
python3 tools/summarize_rocprof.py $OUT/${TAG}_prof_alone $OUT/${TAG}_bench_prof_alone.json 10 3 $OUT/${TAG}_kernel_summary_alone.md > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_pmc_$C
  rocprofv3 --pmc $C -d $OUT/${TAG}_pmc_$C --output-format csv -- python3 bench.py $ARGS > $OUT/${TAG}_bench_pmc_$C.json 2> $OUT/${TAG}_pmc_$C.err
done
python3 tools/summarize_pmc_traffic.py $OUT/${TAG}_prof $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_bench_prof.json 10 $OUT/${TAG}_traffic.json > $OUT/${TAG}_traffic.txt 2>&1
rm -rf $OUT/${TAG}_pmc_MFMA
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/${TAG}_pmc_MFMA --output-format csv -- python3 bench.py $ARGS > $OUT/${TAG}_bench_pmc_MFMA.json 2> $OUT/${TAG}_pmc_MFMA.err
python3 tools/summarize_pmc_mfma.py $OUT/${TAG}_pmc_MFMA $OUT/${TAG}_prof 10 $OUT/${TAG}_pmc_mfma_kernels.json > $OUT/${TAG}_mfma.txt 2>&1
# the raw counter CSVs are tens of MB: keep the summaries only
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_MFMA
find $OUT/${TAG}_prof $OUT/${TAG}_prof_fp32 $OUT/${TAG}_prof_bf16x3 $OUT/${TAG}_prof_alone -name "*agent_info.csv" -delete 2>/dev/null
tail -3 $OUT/${TAG}_bench.json | cut -c1-600
cat $OUT/${TAG}_traffic.txt | head -30
cat $OUT/${TAG}_mfma.txt | head -20
