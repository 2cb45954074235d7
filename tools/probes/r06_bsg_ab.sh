cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in 1 0; do A3D_BF16_SG=$v python tools/train_bench.py --precision bf16 --batch 2 --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().splitlines()[-1]); print('A3D_BF16_SG=$v', d['value'], d['ms_per_step'], d.get('timing','')[-40:])"; done; done
for v in 1 0; do A3D_BF16_SG=$v python tools/train_bench.py --precision bf16 --batch 16 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().splitlines()[-1]); print('b16 A3D_BF16_SG=$v', d['value'], d['ms_per_step'])"; done
