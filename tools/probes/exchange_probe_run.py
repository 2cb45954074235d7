"""The gradient-exchange probe of tools/train_bench.py on its own (one JSON line): python tools/probes/exchange_probe_run.py [batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import articulation3d_amd  # noqa: E402,F401
from articulation3d_amd.streams import side  # noqa: E402
from bench import build_detector  # noqa: E402
from train_bench import exchange_probe  # noqa: E402

torch.cuda.set_device(0)
side(0)
model, _ = build_detector(0.5, "cuda:0")
r = exchange_probe("cuda:0", model, int(sys.argv[1]) if len(sys.argv) > 1 else 2, 10, 5)
sys.stdout.write(json.dumps({"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), **{k: r[k] for k in ("exposed_ms", "ms_per_step_with_exchange", "ms_per_step_without", "host_ms_per_step")}}) + "\n")
