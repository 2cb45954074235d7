import sys, json, io, contextlib
sys.path.insert(0, "/root/repo")
import bench
from articulation3d_amd import streams
from articulation3d_amd.modeling import backbone, rpn
from articulation3d_amd.modeling.roi_heads import roi_heads
for sb in (0, 16):
    for mod in (streams, backbone, rpn, roi_heads):
        mod.SMALL_BATCH = sb
    for b in (1, 2, 4, 8, 16):
        sys.argv = ["bench.py", "--batch", str(b), "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-alt-modes"]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        d = json.loads(buf.getvalue().strip().splitlines()[-1])
        print(sb, b, d["value"], d["ms_per_step"], flush=True)
