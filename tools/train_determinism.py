"""Debug aid: two identical forward/backward passes of the trainer must agree (up to the atomics of ROIAlign backward)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_detector
from articulation3d_amd.training import DetectorTrainer, SolverCfg
from articulation3d_amd.utils.synthetic import synthetic_frames
from tools.train_bench import synthetic_targets

model, _ = build_detector(0.5, "cuda:0")
frames = torch.from_numpy(synthetic_frames(2)).cuda()
tg = synthetic_targets(2, 2020)
tr = DetectorTrainer(model, SolverCfg(base_lr=0.01, warmup_iters=0), seed=3)
runs = []
for r in range(4):
    tr.iter = 0
    losses, aux = tr.forward_backward(frames, [t[0] for t in tg], [t[1] for t in tg])
    torch.cuda.synchronize()
    runs.append(dict(losses={k: v.item() for k, v in losses.items()}, grads={k: v.cpu() for k, v in tr.export_grads().items()},
                     labels=aux["anchor_labels"].cpu(), ridx=aux["roi_index"].cpu(), rc=aux["roi_count"].cpu(), props=aux["proposals"][0].cpu(),
                     pred=aux["pred"].cpu(), feats={k: v.cpu() for k, v in aux["feats"].items()}))
    print(r, runs[-1]["losses"])
a = runs[0]
for r in range(1, 4):
    b = runs[r]
    print("run", r, "labels eq", torch.equal(a["labels"], b["labels"]), "ridx eq", torch.equal(a["ridx"], b["ridx"]), "props eq", torch.equal(a["props"], b["props"]),
          "pred eq", torch.equal(a["pred"], b["pred"]), "feats eq", all(torch.equal(a["feats"][k], b["feats"][k]) for k in a["feats"]))
    worst = max((((a["grads"][k] - b["grads"][k]).norm() / (a["grads"][k].norm() + 1e-30)).item(), k) for k in a["grads"])
    print("   worst grad rel diff", worst)
