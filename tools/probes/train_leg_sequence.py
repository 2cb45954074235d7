"""Training legs of different batch sizes one after the other in ONE process (what bench.py does): `train_leg_sequence.py 2,16`."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from train_bench import train_leg
dev = torch.device("cuda:0")
order = [int(b) for b in sys.argv[1].split(",")]
for b in order:
    r = train_leg(dev, b, 10, 5, precision="bf16")
    print("leg", b, r["value"], r["ms_per_step"], flush=True)
