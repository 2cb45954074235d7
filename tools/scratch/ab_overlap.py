import sys, json, io, contextlib
sys.path.insert(0, "/root/repo")
import bench
from articulation3d_amd.modeling.meta_arch import PlaneRCNN
for ov in (0, 64):
    PlaneRCNN.small_batch_overlap = ov
    for b in (48, 64, 64):
        sys.argv = ["bench.py", "--batch", str(b), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        d = json.loads(buf.getvalue().strip().splitlines()[-1])
        print(ov, b, d["value"], d["ms_per_step"], flush=True)
