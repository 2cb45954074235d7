#!/bin/bash
# Regenerates the round's measurement artifacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/bench_final.json           python bench.py (with the CPU baseline)
#   gpurun_out/prof_final/                rocprofv3 --kernel-trace --stats of bench.py
#   gpurun_out/pmc_fetch_f, pmc_write_f   the two PMC passes (one TCC counter per pass, each under its own timeout)
# Afterwards: tools/summarize_rocprof.py / tools/summarize_pmc_traffic.py turn them into profiles/rNN_*.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes > $O/bench_final_prof.json 2> $O/prof_final.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-modes > $O/bench_pmc_fetch_f.json 2> $O/pmc_fetch_f.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-modes > $O/bench_pmc_write_f.json 2> $O/pmc_write_f.err
tail -c 400 $O/bench_final.json; echo; tail -c 300 $O/bench_final_prof.json; echo
ls $O/prof_final/* $O/pmc_fetch_f/* $O/pmc_write_f/* | head
