cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_known_answers.py tests/test_gpu_presplit.py -q -m gpu -k "roi or pool or stage or known or presplit" 2>&1 | tail -3
python tools/roi_bench.py 2>&1 | tail -3
python bench.py --batch 1 --no-cpu-baseline --no-train-leg --no-alt-modes --no-operating-points --steps 60 --warmup 10 2>/dev/null > /tmp/b.json; python -c "import json;d=json.loads(open('/tmp/b.json').read().splitlines()[0]);print('bench1', d['value'], d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-alt-modes --steps 20 2>/dev/null > /tmp/b.json; python -c "
import json;d=json.loads(open('/tmp/b.json').read().splitlines()[0]);r=d['roofline'];print('bench64', d['value'], d['ms_per_step'], r['operating_points'], r['loop_b1']['frames_per_s']); print({k:(v['value'],v['ms_per_step'],v['timing']) for k,v in d['train_step'].items()})"
