cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch
from bench import build_detector
from loop_bench import loop_b1
model, cfg = build_detector(0.5, torch.device("cuda:0"))
print("first call:", loop_b1(model, cfg))
print("second call:", loop_b1(model, cfg))
PY
