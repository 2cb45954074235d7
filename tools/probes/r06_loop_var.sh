cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python tools/loop_bench.py 2>&1 | grep -v amdgpu.ids | head -1; done
for i in 1 2; do A3D_LAUNCH_PLANS=0 python tools/loop_bench.py 2>&1 | grep -v amdgpu.ids | head -1; done
python - <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch
from bench import build_detector
from loop_bench import loop_b1
model, cfg = build_detector(0.5, torch.device("cuda:0"))
for _ in range(4): print(loop_b1(model, cfg))
PY
