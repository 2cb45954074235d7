import os, sys, torch
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from train_bench import train_leg
dev = torch.device("cuda:0")
order = [int(b) for b in sys.argv[1].split(",")]
for b in order:
    r = train_leg(dev, b, 10, 5, precision="bf16")
    print("leg", b, r["value"], r["ms_per_step"], flush=True)
