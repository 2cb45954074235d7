#!/bin/bash
# Training-step profile on the GPU box: bench lines with roofline + CPU leg, rocprofv3 kernel summaries and HBM traffic
# for the bf16 step at 2 and 16 images per GPU.  Summaries land in gpurun_out/ (copy the ones to be judged into profiles/).
#   bash tools/profile_train.sh r04
set -u
TAG=${1:-r04}
OUT=gpurun_out
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p gpurun_out
python3 tools/train_bench.py --precision bf16 --batch 2 --cpu-baseline > $OUT/${TAG}_train_bench_b2.json 2> $OUT/${TAG}_train_bench_b2.err
python3 tools/train_bench.py --precision bf16 --batch 16 > $OUT/${TAG}_train_bench_b16.json 2> $OUT/${TAG}_train_bench_b16.err
python3 tools/train_bench.py --precision bf16x3 --batch 2 > $OUT/${TAG}_train_bench_x3_b2.json 2>/dev/null
python3 tools/train_bench.py --precision bf16x3 --batch 16 > $OUT/${TAG}_train_bench_x3_b16.json 2>/dev/null
for B in 2 16; do
  ARGS="tools/train_bench.py --precision bf16 --batch $B --steps 10 --warmup 3"
  rm -rf $OUT/${TAG}_train_prof_b$B
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_train_prof_b$B --output-format csv -- python3 $ARGS > $OUT/${TAG}_train_bench_prof_b$B.json 2> $OUT/${TAG}_train_prof_b$B.err
  # (the instrumented step after the timed loop is an 11th step: 11 preprocess marks end the trace)
  python3 tools/summarize_rocprof.py $OUT/${TAG}_train_prof_b$B $OUT/${TAG}_train_bench_prof_b$B.json 11 3 $OUT/${TAG}_train_kernel_summary_b$B.md "python3 $ARGS" > /dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/${TAG}_train_pmc_${C}_b$B
    rocprofv3 --pmc $C -d $OUT/${TAG}_train_pmc_${C}_b$B --output-format csv -- python3 $ARGS > /dev/null 2> $OUT/${TAG}_train_pmc_${C}_b$B.err
  done
  python3 tools/summarize_pmc_traffic.py $OUT/${TAG}_train_prof_b$B $OUT/${TAG}_train_pmc_FETCH_SIZE_b$B $OUT/${TAG}_train_pmc_WRITE_SIZE_b$B $OUT/${TAG}_train_bench_prof_b$B.json 11 $OUT/${TAG}_train_traffic_b$B.json "python3 $ARGS" > $OUT/${TAG}_train_traffic_b$B.txt 2>&1
  rm -rf $OUT/${TAG}_train_pmc_FETCH_SIZE_b$B $OUT/${TAG}_train_pmc_WRITE_SIZE_b$B
  find $OUT/${TAG}_train_prof_b$B -name "*agent_info.csv" -delete 2>/dev/null
done
for f in b2 b16 x3_b2 x3_b16; do python3 - <<PY
import json
d = json.loads(open("$OUT/${TAG}_train_bench_$f.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$f", d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["whole_step"], d.get("cpu_baseline"))
PY
done
head -30 $OUT/${TAG}_train_kernel_summary_b16.md
head -20 $OUT/${TAG}_train_traffic_b16.txt
