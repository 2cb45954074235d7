cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_training.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r06_bsg_tests.txt; cat gpurun_out/r06_bsg_tests.txt
for i in 1 2; do python tools/train_bench.py --precision bf16 --batch 2 --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('new', d['value'], d['ms_per_step'], d.get('timing'))"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r06_tb2; rocprofv3 --kernel-trace --stats -d gpurun_out/r06_tb2 --output-format csv -- python3 tools/train_bench.py --precision bf16 --batch 2 --steps 10 --warmup 3 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r06_tb2/*/*_kernel_trace.csv')[0]; rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
print('launches per step', [idx[i+1]-idx[i] for i in range(len(idx)-1)][-5:])
import collections
c=collections.Counter(); t=collections.Counter()
for r in rows[idx[-6]:idx[-1]]:
    n=r['Kernel_Name'].split('(')[0][-60:]; c[n]+=1; t[n]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for n,v in t.most_common(8): print(f"{n:62s} x{c[n]/5:6.1f} {v/5e3:8.1f} us")
PY
rm -rf gpurun_out/r06_tb2
