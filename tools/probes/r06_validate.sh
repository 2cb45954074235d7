cd $GRAFT_REPO_ROOT
python -m pytest tests/ -q -m gpu 2>&1 | tail -12 > gpurun_out/r06_gpu_suite_final.txt; tail -6 gpurun_out/r06_gpu_suite_final.txt
python __graft_entry__.py --smoke 2>&1 | tail -2
python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err; wc -l gpurun_out/r06_bench_final.json; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_bench_final.json").read().splitlines()[0])
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["like_for_like"]["bf16x3"]["value"], r["like_for_like"]["fp32"]["value"], r["loop_b1"]["frames_per_s"], r["train_step"])
PY
