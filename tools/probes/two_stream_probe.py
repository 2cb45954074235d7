"""Timing probe: one 64-frame batch vs two 32-frame halves on two streams driven by two host threads (potential of interleaving)."""
import os, sys, time, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_detector
from articulation3d_amd.utils.synthetic import synthetic_frames

model, cfg = build_detector(0.5, "cuda:0")
model2, _ = build_detector(0.5, "cuda:0")
x = torch.from_numpy(synthetic_frames(64)).cuda()
def full():
    model.inference_batched(x)
for _ in range(3): full()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): full()
torch.cuda.synchronize()
print(f"one batch of 64: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
s = [torch.cuda.Stream(), torch.cuda.Stream()]
ms = [model, model2]
def half(i, n):
    with torch.cuda.stream(s[i]):
        for _ in range(n):
            ms[i].inference_batched(x[32 * i:32 * i + 32])
def pair(n):
    th = [threading.Thread(target=half, args=(i, n)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
pair(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
pair(10)
torch.cuda.synchronize()
print(f"two halves of 32 on two streams / threads: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per 64 frames")
for i in range(2):
    with torch.cuda.stream(s[i]):
        for _ in range(3): ms[i].inference_batched(x[:32])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): model.inference_batched(x[:32])
torch.cuda.synchronize()
print(f"one batch of 32: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
