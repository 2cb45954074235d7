"""Does the number of HIP streams a process has created before the trainer's two side streams change the step time?  (ROCm maps streams onto
GPU_MAX_HW_QUEUES hardware queues, 4 by default: streams that share a queue do not overlap.)   usage: train_stream_queues.py <dummy streams> [batch]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from train_bench import train_leg
dev = torch.device("cuda:0")
n, b = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 16
keep = []
for _ in range(n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        keep.append(torch.zeros(16, device=dev) + 1)
    keep.append(s)
torch.cuda.synchronize()
r = train_leg(dev, b, 10, 5, precision="bf16")
print(f"dummy streams {n}, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}: batch {b}: {r['value']} images/s, {r['ms_per_step']} ms", flush=True)
