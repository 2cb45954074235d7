"""cProfile of the host side of the reference-style per-frame loop (tools/loop_bench.py)."""
import cProfile, os, pstats, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from articulation3d_amd.config import get_cfg, get_planercnn_cfg_defaults
from articulation3d_amd.utils.arti_vis import PlaneRCNN_Branch, create_instances
from articulation3d_amd.utils.synthetic import synthetic_frames, calibrate_batchnorm
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = get_cfg(); get_planercnn_cfg_defaults(cfg)
cfg.merge_from_file(os.path.join(ROOT, "configs", "planercnn_inference.yaml"))
cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.5
torch.manual_seed(2020)
branch = PlaneRCNN_Branch(cfg, load_weights=False)
model = branch.predictor.model
frames = synthetic_frames(40)
calibrate_batchnorm(model, torch.from_numpy(synthetic_frames(2, 2021)).cuda())
def loop(fr):
    out = []
    for im in fr:
        pred = branch.inference(im)
        d = branch.process(pred)
        out.append(create_instances(d["instances"], im.shape[:2], pred_planes=d["pred_plane"].numpy(), pred_rot_axis=d["pred_rot_axis"],
                                    pred_tran_axis=d["pred_tran_axis"], conf_threshold=0.5))
    return out
loop(frames[:4]); torch.cuda.synchronize()
t = time.perf_counter(); loop(frames[4:36]); torch.cuda.synchronize(); print("ms/frame", 1e3 * (time.perf_counter() - t) / 32)
pr = cProfile.Profile(); pr.enable(); loop(frames[4:36]); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40); st.sort_stats("cumulative").print_stats(45)
