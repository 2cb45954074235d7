"""Do an HBM-bound kernel and a matrix-pipe-bound kernel overlap when they run on two streams?  Two independent 3x3 Winograd layers at the p2
size (64 x 120 x 160 x 256 -> 256: input transform 1.4 ms, HBM-bound; GEMM 2.0 ms, L2 -> LDS / MFMA-bound), each transform -> GEMM on its own
stream, against the same four launches on one stream."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from articulation3d_amd import ops
from articulation3d_amd.streams import side
B = int(os.environ.get("B", "64"))
torch.manual_seed(0)
xs = [ops.keep_amax(torch.randn(B, 120, 160, 256, device="cuda"), None) if hasattr(ops, "keep_amax") else torch.randn(B, 120, 160, 256, device="cuda") for _ in range(2)]
pk = [ops.pack_conv(torch.randn(256, 256, 3, 3) / 48, torch.randn(256) * 0.1, None, 1, 1, ops.ACT_RELU) for _ in range(2)]
def layer(i):
    return ops.conv2d(xs[i], pk[i])
for i in range(2):
    layer(i)
print("variant:", ops.last_conv_variant())
torch.cuda.synchronize()
def timed(fn, n=6):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n
def serial():
    layer(0); layer(1)
def forked():
    main = torch.cuda.current_stream(); s = side(1)
    ev = torch.cuda.Event(); ev.record(main)
    with torch.cuda.stream(s):
        s.wait_event(ev); y = layer(1); done = torch.cuda.Event(); done.record(s)
    layer(0); main.wait_event(done); y.record_stream(main)
print(f"one layer {timed(lambda: layer(0)):.3f} ms | two layers, one stream {timed(serial):.3f} ms | two streams {timed(forked):.3f} ms")
