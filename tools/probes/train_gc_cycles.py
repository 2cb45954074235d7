"""Which objects of a training step are cyclic garbage (freed by the collector only, not by reference counts)?  GPU tensors among them keep
their blocks away from the caching allocator until a collection runs."""
import collections, gc, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import build_detector
from train_bench import synthetic_targets
from articulation3d_amd.training import DetectorTrainer
from articulation3d_amd.utils.synthetic import synthetic_frames
dev = torch.device("cuda:0")
model, _ = build_detector(0.5, dev)
B = 2
tr = DetectorTrainer(model, seed=2020, precision="bf16")
frames = torch.from_numpy(synthetic_frames(B, seed=2020)).to(dev)
tg = synthetic_targets(B, 2020)
gtb, gtc = [t[0] for t in tg], [t[1] for t in tg]
for _ in range(4): tr.step(frames, gtb, gtc)
torch.cuda.synchronize()
gc.collect()
gc.disable()
gc.set_debug(gc.DEBUG_SAVEALL)
tr.step(frames, gtb, gtc)
torch.cuda.synchronize()
n = gc.collect()
print("unreachable objects after one step:", n)
cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
print(cnt.most_common(20))
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
print("tensors:", len(tens), sum(t.numel() * t.element_size() for t in tens) / 1e6, "MB")
for o in gc.garbage:
    if type(o).__name__ in ("function", "cell", "frame", "dict", "list", "tuple") and len(repr(o)) < 300:
        print(type(o).__name__, repr(o)[:200])
