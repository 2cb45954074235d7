"""GPU suite (-m gpu): the scripts a user / the driver runs, end to end in fresh processes.

  * BASELINE configs[0]: tools/inference.py on a 4-frame synthetic clip (plumbing), and the clear refusal of MODEL.DEVICE=cpu;
  * bench.py's multi-rank forms: `--gpus 2` spawns two ranks itself (gloo here: a 1-GPU box cannot give RCCL two devices)
    and reports n_gpus = 2; the single-rank RCCL form (A3D_BENCH_FORCE_DIST=1) runs the real nccl all-gather;
  * a missing checkpoint is an error, never a silent random initialisation.
Child processes are started with subprocess (fork + exec of a fresh interpreter): nothing is exec'd INSIDE a process that
holds the GPU.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)


def _run(cmd, timeout=900, env=ENV):
    return subprocess.run([sys.executable, *cmd], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_inference_script_on_synthetic_clip(tmp_path):
    """configs[0] plumbing on the GPU: build from the YAML, detect 4 frames, track, optimise, write both JSON files."""
    out = tmp_path / "out"
    r = _run(["tools/inference.py", "--config", "configs/planercnn_inference.yaml", "--input", "synthetic:4", "--output", str(out),
              "--random-init", "--calibrate-bn", "--conf-threshold", "0.3", "--batch", "2",
              "MODEL.ROI_HEADS.SCORE_THRESH_TEST", "0.3"])
    assert r.returncode == 0, r.stderr[-2000:]
    preds = json.load(open(out / "predictions.json"))
    tracks = json.load(open(out / "tracks.json"))
    assert len(preds) == 4 and set(tracks) == {"rot", "trans"}
    assert sum(len(p["instances"]) for p in preds) > 0
    inst = next(i for p in preds for i in p["instances"])
    assert set(inst) == {"bbox", "score", "category_id", "pred_plane", "pred_rot_axis", "pred_tran_axis", "segmentation"}
    assert inst["score"] > 0.3 and len(inst["pred_plane"]) == 3 and inst["segmentation"]["size"] == [480, 640]


def test_inference_script_refuses_cpu_device_and_missing_weights(tmp_path):
    r = _run(["tools/inference.py", "--config", "configs/planercnn_inference.yaml", "--input", "synthetic:2", "--output", str(tmp_path / "o"),
              "--random-init", "MODEL.DEVICE", "cpu"])
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    # the YAML's default MODEL.WEIGHTS (exps/model_final.pth) does not exist offline: without --random-init that is an error
    r = _run(["tools/inference.py", "--config", "configs/planercnn_inference.yaml", "--input", "synthetic:2", "--output", str(tmp_path / "o2")])
    assert r.returncode != 0 and "does not exist" in (r.stderr + r.stdout)
    assert not (tmp_path / "o2" / "predictions.json").exists()


def test_inference_script_sharded_over_two_ranks(tmp_path):
    """BASELINE configs[3] in miniature: torchrun with two ranks (gloo: both on this box's one GPU), 7 frames -> uneven blocks
    of 4 + 3, ONE gather of the detection records, rank 0 alone tracks / optimises / writes; the result equals the single-process
    run frame for frame."""
    import socket

    common = ["--config", "configs/planercnn_inference.yaml", "--input", "synthetic:7", "--random-init", "--calibrate-bn",
              "--conf-threshold", "0.3", "--batch", "2"]
    opts = ["MODEL.ROI_HEADS.SCORE_THRESH_TEST", "0.3"]
    one, two = tmp_path / "one", tmp_path / "two"
    r = _run(["tools/inference.py", *common, "--output", str(one), *opts])
    assert r.returncode == 0, r.stderr[-2000:]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
              "tools/inference.py", *common, "--output", str(two), "--dist-backend", "gloo", *opts])
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = json.load(open(one / "predictions.json")), json.load(open(two / "predictions.json"))
    assert len(a) == len(b) == 7
    for fa, fb in zip(a, b):
        assert len(fa["instances"]) == len(fb["instances"])
        for ia, ib in zip(fa["instances"], fb["instances"]):
            assert ia["bbox"] == ib["bbox"] and ia["category_id"] == ib["category_id"] and ia["segmentation"] == ib["segmentation"]
            assert ia["pred_plane"] == ib["pred_plane"] and ia["pred_rot_axis"] == ib["pred_rot_axis"]


def test_bench_spawns_its_ranks(tmp_path):
    """`python bench.py --gpus 2` outside torchrun must launch two ranks and report n_gpus = 2 (the flag was dead)."""
    r = _run(["bench.py", "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "4",
              "--cpu-frames", "2", "--no-alt-modes", "--no-operating-points"], timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_frames_per_step"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    # the collective record (round 5): backend, world, every rank's device identity gathered over the group, what the gathers moved
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world"] == 2 and [x["rank"] for x in c["devices"]] == [0, 1]
    assert c["distinct_devices"] == 1 and c["gathers_per_step"] >= 1 and c["gather_bytes_per_step"] > 0  # (two ranks share this box's one GPU)
    # ... and the training-step leg runs data-parallel over the same group (one all-reduce per step)
    t = d["train_step"]["images_per_gpu_2"]
    assert t["config"]["global_batch"] == 4 and t["value"] > 0 and 0 < t["roofline"]["whole_step"]["frac"] < 1
    # an N > 1 line carries the CPU leg and the matched-detections check too (rank 0 runs them after the timed region)
    assert d["cpu_baseline"]["cores"] >= 1 and d["matched_detections"]["frames"] == 5 and d["matched_detections"]["matched"] is True
    # a launcher / flag disagreement is refused instead of mislabelled
    r = _run(["bench.py", "--gpus", "1", "--steps", "1"], env=dict(ENV, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_bench_with_eight_ranks_reports_eight(tmp_path):
    """BASELINE configs[3]'s launch shape on the one GPU of this box: `bench.py --gpus 8` spawns eight ranks (gloo, all on device 0: the
    driver's 8-GPU run uses RCCL with one device per rank -- LOCAL_RANK is the device ordinal there), each detects its own frames, every
    step's records are all-gathered, rank 0 runs the CPU leg and the matched-detections check while the others wait at the barrier,
    and the line says n_gpus = 8 with whole-job frames per step."""
    r = _run(["bench.py", "--gpus", "8", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "2",
              "--cpu-frames", "2", "--no-alt-modes", "--no-operating-points", "--no-train-leg"], timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["config"]["global_frames_per_step"] == 16 and d["scaling"] == "weak" and d["value"] > 0
    assert d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0 and d["matched_detections"]["matched"] is True
    # (eight ranks time-share ONE GPU here: the dominant kernel's rate can round to 0.0000 of the roof -- what must hold is that it was measured)
    assert "RCCL all-gather" in d["config"]["sharding"] and d["roofline"]["launches"] > 0 and d["roofline"]["avg_launch_ms"] > 0
    assert d["collective"]["world"] == 8 and len(d["collective"]["devices"]) == 8 and d["collective"]["gather_bytes_per_step"] > 0


def _two_gpus():
    import torch

    return torch.cuda.device_count() >= 2  # (does not initialise HIP)


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs: RCCL gives every rank its own device")
def test_two_real_rccl_ranks_bench_and_inference(tmp_path):
    """On a box with >= 2 GPUs: `bench.py --gpus 2` over RCCL (two ranks, one GPU each, the two-phase gather of the records, the CPU
    leg and the matched-detections check on rank 0 while rank 1 waits) and `torchrun -n 2 tools/inference.py` over nccl against the
    single-process run.  Skips itself on the 1-GPU boxes of the development pool."""
    import socket

    r = _run(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--cpu-frames", "2", "--no-alt-modes", "--no-operating-points"],
             timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_frames_per_step"] == 16 and "cpu_baseline" in d and d["matched_detections"]["matched"] is True
    # RCCL saw two DISTINCT GPUs: checkable from the line alone
    c = d["collective"]
    assert c["backend"] == "nccl" and c["world"] == 2 and c["distinct_devices"] == 2 and len({x["pci"] for x in c["devices"]}) == 2
    assert c["gather_bytes_per_step"] > 0 and d["train_step"]["images_per_gpu_2"]["config"]["global_batch"] == 4
    common = ["--config", "configs/planercnn_inference.yaml", "--input", "synthetic:7", "--random-init", "--calibrate-bn",
              "--conf-threshold", "0.3", "--batch", "2"]
    opts = ["MODEL.ROI_HEADS.SCORE_THRESH_TEST", "0.3"]
    one, two = tmp_path / "one", tmp_path / "two"
    assert _run(["tools/inference.py", *common, "--output", str(one), *opts]).returncode == 0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
              "tools/inference.py", *common, "--output", str(two), "--dist-backend", "nccl", *opts])
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.load(open(one / "predictions.json")) == json.load(open(two / "predictions.json"))


def test_bench_single_rank_rccl_gather(tmp_path):
    """The nccl (= RCCL) process group and the asynchronous all-gather of the detection records, with one rank on this box."""
    r = _run(["bench.py", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-alt-modes", "--no-operating-points", "--no-train-leg"],
             env=dict(ENV, A3D_BENCH_FORCE_DIST="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["collective"]["backend"] == "nccl" and d["collective"]["world"] == 1 and d["collective"]["devices"][0]["pci"]
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] <= 1.0
    assert d["roofline"]["kernel"].startswith(("wino_", "conv_")) and d["roofline"]["pipe"] in ("f16x3", "bf16x6", "f32")


def test_bench_default_line_has_the_contract_fields():
    r = _run(["bench.py", "--steps", "2", "--warmup", "1", "--batch", "16", "--cpu-frames", "2", "--no-alt-modes"], timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "matched_detections", "operating_points", "value_with_transfers"):
        assert k in d, k
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["pipe"] == "f16x3" and rf["peak"] == 2500.0 and d["dtype"].startswith("f32")
    assert rf["algorithmic_speedup"] >= 1.0
    assert set(d["operating_points"]) == {"A_thresh0.7", "B_thresh0.0", "C_given4"}
    assert d["operating_points"]["B_thresh0.0"]["detections_per_frame"] == 100.0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["batch8_frames_per_s"] > 0 and cb["os_cpu_count"] >= cb["cores"]
    md = d["matched_detections"]
    assert md["frames"] == 5 and md["matched"] is True, md
    assert all(k in md for k in ("max_raw_plane_err", "max_raw_rot_err", "max_raw_tran_err", "max_plane_cond", "max_tran_axis_cond")), md
    assert d["roi_out_of_window"] == 0 and "22-bit significand" in d["dtype"] and "block exponent" in d["dtype"]
    # round 5: the transfers ride a copy stream beside the step (the transfer-inclusive figure sits within a few percent of `value` even on
    # this short run), the collective record is there at N = 1 too, and the training step (BASELINE configs[4]) reports where the driver looks
    assert d["value_with_transfers"]["vs_resident"] > 0.9 and d["value_transfer_inclusive"] == d["value_with_transfers"]["value"]
    assert d["collective"]["world"] == 1 and d["collective"]["backend"] is None and len(d["collective"]["devices"]) == 1
    for k in ("images_per_gpu_2", "images_per_gpu_16"):
        t = d["train_step"][k]
        assert t["unit"] == "images/s" and t["value"] > 0 and t["roofline"]["peak"] == 2500.0 and 0 < t["roofline"]["whole_step"]["frac"] < 1
        assert "ONE stream" in t["roofline"]["source"]  # (per-kernel times come from the one-stream instrumented step)
    t2 = d["train_step"]["images_per_gpu_2"]  # (the CPU port's autograd step: the 2-image leg only -- minutes of host time at 16)
    assert t2["cpu_baseline"]["kind"] == "port" and t2["cpu_baseline"]["value"] > 0 and "cpu_baseline" not in d["train_step"]["images_per_gpu_16"]
    # round 6: the N > 1 step's gradient exchange measured on this box (four segments on a one-rank RCCL group), and the numeric leaves the
    # driver's record keeps in full -- like-for-like arithmetics, operating points, the reference's per-frame loop, the training legs
    c = t2["collective"]
    assert c["segments"] == 4 and c["bytes"] > 80e6 and c["payload"] == "bf16" and -0.5 < c["exposed_ms"] < 1.5, c
    if "alt_modes" in d:
        assert set(rf["like_for_like"]) >= {"bf16x3", "fp32"} and rf["like_for_like"]["bf16x3"]["value"] == d["alt_modes"]["bf16x3"]["value"]
    assert set(rf["operating_points"]) == {"A", "B", "C"} and rf["operating_points"]["B"]["value"] == d["operating_points"]["B_thresh0.0"]["value"]
    assert rf["loop_b1"]["frames"] == 64 and rf["loop_b1"]["frames_per_s"] > 50 and rf["loop_b1"]["detections_per_frame"] > 0
    assert set(rf["train_step"]) == {"images_per_gpu_2", "images_per_gpu_16"} and rf["train_step"]["images_per_gpu_2"]["exposed_exchange_ms"] == c["exposed_ms"]


def test_first_small_batch_of_a_process_equals_the_steady_state():
    """Regression: freshly packed / split weights are produced on the current stream and were then read by the concurrent branches of
    a 1-2 frame batch on OTHER streams without an ordering -- the first batch of a process could come out wrong (seen as a 1-in-8
    failure of the sharded-inference test, reproduced 8 / 24 with two processes contending for the GPU).  `_publish()` in
    modeling/layers.py and the lazy splits in ops.conv2d now drain the producing stream once.  Two contending processes, twice."""
    import subprocess

    for _ in range(2):
        procs = [subprocess.Popen([sys.executable, "tools/determinism_probe.py", str(b), "4", "0.3"], cwd=ROOT, env=ENV, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for b in (2, 1)]
        for pr in procs:
            out, _ = pr.communicate(timeout=600)
            assert pr.returncode == 0, out[-2000:]
            assert "0 runs differ from the first" in out, out[-2000:]


def test_single_frame_pass_without_its_host_read_and_as_a_hip_graph():
    """Round 4: `roi_heads.fixed_rows` sizes the per-ROI head tensors for every detection slot instead of reading
    the live total back, and `PlaneRCNN.inference_graphed` captures that pass -- side-stream branches included -- as ONE HIP graph.
    Both must give the bits of the eager, sized pass (records, depth, pasted masks), on the frame they were captured with and on another
    one.  (Timing: tools/loop_bench.py -- the graph is not the default.)"""
    import torch

    sys.path.insert(0, ROOT)
    from bench import build_detector
    from articulation3d_amd.utils.synthetic import synthetic_frames

    model, _ = build_detector(0.5, "cuda:0")
    fr = torch.from_numpy(synthetic_frames(3, 2020)).cuda()
    ref = [model.inference_batched(fr[i:i + 1], want_masks=True) for i in range(3)]
    ref = [(o.records.clone(), o.rec_count.clone(), o.depth.clone(), o.masks.clone()) for o in ref]
    assert sum(int(r[1].sum()) for r in ref) > 0
    model.roi_heads.fixed_rows = True
    try:
        for i in range(3):
            o = model.inference_batched(fr[i:i + 1], want_masks=True)
            assert torch.equal(o.records, ref[i][0]) and torch.equal(o.rec_count, ref[i][1]) and torch.equal(o.depth, ref[i][2]) and torch.equal(o.masks, ref[i][3])
    finally:
        model.roi_heads.fixed_rows = False
    for i in (0, 1, 2, 0):
        o = model.inference_graphed(fr[i:i + 1], want_masks=True)
        torch.cuda.synchronize()
        assert torch.equal(o.records, ref[i][0]) and torch.equal(o.rec_count, ref[i][1]) and torch.equal(o.depth, ref[i][2]) and torch.equal(o.masks, ref[i][3]), i
    assert len(model._graphs) == 1
    # an eager pass after the captures is untouched by them (its maxima slots are its own)
    o = model.inference_batched(fr[1:2], want_masks=True)
    assert torch.equal(o.records, ref[1][0]) and torch.equal(o.depth, ref[1][2])


def test_the_package_rule_for_stream_placement_gives_the_fast_order():
    """VERDICT r5 item 7.  Stream placement on this runtime follows the ORDER of first use (four hardware queues by default).  The package's
    rule -- its stream pool first, foreign streams and RCCL afterwards (bench.py, tools/train_bench.py, tools/inference.py) -- must be the
    fast order of the 16-image training step, with a one-rank RCCL group and three foreign streams present; the opposite order is run too
    and reported (18.1 against 15.1 ms).  More hardware queues would remove the dependence (6 or 8: both orders 15.0 ms) but make the
    gradient exchange's cross-stream waits cost 5-6 ms per step -- measured in round 6 and not taken (articulation3d_amd/__init__.py)."""
    r = _run(["tools/probes/stream_order_check.py"], timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    pf, ff = d["package_first"], d["foreign_first"]
    assert pf["queues"]["GPU_MAX_HW_QUEUES"] is None and pf["queues"]["pool_first"] is True, d
    assert pf["ms_per_step"] <= 1.06 * min(pf["ms_per_step"], ff["ms_per_step"]), d  # (the rule's order is never the slow one)
    print("stream order:", d["ratio_foreign_over_package"], pf["ms_per_step"], ff["ms_per_step"])
