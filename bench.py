#!/usr/bin/env python3
"""Headline benchmark: frames/sec through the PlaneRCNN detector at 480x640 on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 and no torchrun environment: this process spawns the N ranks itself (python -m torch.distributed.run,
one rank per GPU, RCCL) BEFORE it touches the GPU and relays rank 0's JSON line; under the driver's own
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one of the ranks (WORLD_SIZE must equal N).

One "step" = one 64-frame clip (BASELINE configs[2]) of synthetic 480x640 uint8 BGR frames (resident in HBM) through the whole
detector: normalise -> ResNet50-FPN -> RPN (top-k, NMS) -> ROIAlign -> box head -> detection NMS -> mask /
plane / axis heads -> depth head -> fused post-process + mask paste + plane-offset LSQ -> packed detection
records (+ an RCCL all-gather of the records when N > 1: frames are sharded across ranks, weak scaling).
Weights: random init of the reference architecture with calibrated batch-norm statistics (no checkpoint is
available offline); fp32 end to end.  Prints ONE JSON line on rank 0; besides the contract fields it carries
  roofline            dominant kernel: EXECUTED matrix FLOPs / fp32-MFMA peak (frac), algorithmic_speedup separately
  operating_points    SURVEY 8d: A (thresh 0.7, D=0), B (0.0, D=100), C (4 injected boxes per frame through given_boxes)
  value_with_transfers the same step with the uint8 clip copied H2D and the packed records D2H inside the timed region
  cpu_baseline        the CPU oracle on this box's host cores (B=1 as the reference loops, and a B=8 figure)
  matched_detections  HIP detections of the CPU leg's frames compared with the oracle's (oracle/matching.py)
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense f32-input MFMA peak (MI355X_MICROARCH.md)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense; never the 2:1-sparsity figure)
# What a wave that issues nothing but v_mfma_f32_32x32x16_f16 on registers sustains on this part (tools/probes/mfma_peak.hip, whole chip,
# 2 waves per SIMD, 0.3-2.6 ms launches): 2.04-2.09 PFLOP/s = the clock it holds under matrix load.  Informational: `peak` stays the
# guide's figure.
F16_MFMA_SUSTAINED_MEASURED_TFLOPS = 2050.0
PIPE_PEAK = {"f32": FP32_MFMA_PEAK_TFLOPS, "bf16": BF16_MFMA_PEAK_TFLOPS, "bf16x6": BF16_MFMA_PEAK_TFLOPS, "f16x3": BF16_MFMA_PEAK_TFLOPS}
PIPE_FLOPS_PER_FMA = {"f32": 1.0, "bf16": 1.0, "bf16x6": 6.0, "f16x3": 3.0, "none": 0.0}  # matrix-pipe products issued per fp32 multiply-add


DTYPES = {
    "bf16x3": "f32 (fp32 tensors and accumulation; every product formed from exact 3-way bf16 operand splits, six bf16 MFMAs per k step: "
              "error vs float64 <= the fp32-input MFMA's, the whole parity suite runs in this arithmetic)",
    "fp16x2": "f32 storage and accumulation; operands enter the matrix pipe with a 22-bit significand under ONE power-of-two block exponent "
              "per image (per ROI behind the poolers, per layer for filters): x*s = h + l in fp16, three fp16 MFMAs per k step (h.h + h.l + l.h). "
              "Elementwise law |y - y64| <= c (2^-22 sum|x||w| + 2^-39 (max|x| sum|w| + max|w| sum|x|)): fp32-grade while an output's "
              "receptive field lies within 2^18 of its image's maximum (tests/test_gpu_precision.py); ROIs fainter than 2^-16 of their pyramid "
              "level are counted (roi_out_of_window)",
    "fp32": "f32 (fp32-input MFMA)",
    "bf16": "bf16 arithmetic (fp32 accumulate, fp32 tensors) -- opt-in autocast mode, not comparable with the f32 figures",
}


def rocprof_name(variant: str) -> str:
    """The dispatcher's variant label (a3d_last_conv_variant) -> the kernel instantiation as a rocprofv3 trace spells it.  The fp16x2
    kernels are the F16 = true instantiations of the split-operand templates ("conv_h2_kernel<2>" = conv_x3_kernel<2, false, true, true>)."""
    import re

    v = variant.split(" sk")[0]
    m = re.match(r"conv_(x3|h2)_kernel<(\d)>( stem)?$", v)
    if m:
        h2 = "true" if m.group(1) == "h2" else "false"  # (F16, and WDMA: module-cached filters arrive pre-split by LDS-DMA)
        return f"conv_x3_kernel<{m.group(2)}, {'true' if m.group(3) else 'false'}, {h2}, {h2}>"
    m = re.match(r"conv_(x3|h2)w_kernel( ph4| xd)?$", v)
    if m:  # (<F16, fused four-phase form, activations pre-split (dual DMA)>)
        return (f"conv_x3w_kernel<{'true' if m.group(1) == 'h2' else 'false'}, {'true' if m.group(2) == ' ph4' else 'false'}, "
                f"{'true' if m.group(2) == ' xd' else 'false'}>")
    m = re.match(r"conv_ph4p_kernel<(\d+)>( dot)?$", v)  # (<four-phase form, 32-column blocks per wave, tile width, tap-product epilogue>)
    if m:
        return f"conv_ph4p_kernel<true, 4, {m.group(1)}, {'true' if m.group(2) else 'false'}>"
    m = re.match(r"conv_c3p_kernel<(\d), (\d+)>$", v)
    if m:
        return f"conv_ph4p_kernel<false, {m.group(1)}, {m.group(2)}, false>"
    if v == "wino_input_kernel":  # (the fp16x2 arithmetic's transform writes V pre-split: its own kernel)
        return "wino_input_h2_kernel"
    m = re.match(r"conv_h2xs_kernel<(\d+)>$", v)  # (activation-stationary pointwise kernel: <Cin / 16, N groups, chunks per ring stage>)
    if m:
        return {"64": "conv_xs_kernel<4, 4, 1>", "128": "conv_xs_kernel<8, 2, 2>", "256": "conv_xs_kernel<16, 2, 2>"}[m.group(1)]
    m = re.match(r"conv_h2xs_b2b_kernel<(\d+),(\d+)>$", v)  # (the back-to-back pointwise pair: <Cin / 16, N groups, chunks per stage, Cout2 / 32, prefetch>)
    if m:
        return {"64": "conv_xs_b2b_kernel<4, 2, 2, 2, true>", "128": "conv_xs_b2b_kernel<8, 2, 2, 4, false>"}[m.group(1)]
    m = re.match(r"conv_h2sg_kernel<(\d)>$", v)  # (small-grid pointwise kernel: <chunks in flight per wave>)
    if m:
        return f"conv_sg_kernel<{m.group(1)}>"
    if v == "wino_gemm_h2w_kernel<2> planes + wino_fold_kernel":  # (the plane-split form of small problems: its GEMM launch)
        return "wino_gemm_x3w_kernel<2, true, true, 0>"
    m = re.match(r"wino_gemm_(x3|h2)w_kernel<(\d)>$", v)
    if m:  # (<WM, F16, plane-split, ping-pong loop>: round 5's fp16x2 launches are all <4, true, false, 1>, one map or several)
        h2 = m.group(1) == "h2"
        return f"wino_gemm_x3w_kernel<{m.group(2)}, {'true' if h2 else 'false'}, false, {1 if h2 and m.group(2) == '4' else 0}>"
    return v


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per step per GPU (BASELINE configs[2]: a 64-frame clip)")
    ap.add_argument("--score-thresh", type=float, default=0.5,
                    help="MODEL.ROI_HEADS.SCORE_THRESH_TEST of the headline; 0.5 gives a realistic handful of detections per frame "
                         "on random-init weights (0.7, the reference default, gives none; 0.0 gives 100: both are in operating_points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-modes", action="store_true", help="skip the secondary timed loop in the other arithmetic (fp32-input MFMA)")
    ap.add_argument("--no-operating-points", action="store_true", help="skip the A / B / C operating points and the transfer-inclusive loop")
    ap.add_argument("--precision", default="fp16x2", choices=["fp32", "bf16", "bf16x3", "fp16x2"],
                    help="fp16x2 (default: fp32-grade products from a two-way fp16 split of each operand, scaled per image, three fp16 MFMAs "
                         "per k step; fp32 tensors and accumulation; the arithmetic the package and its parity suite run in), bf16x3 (exact "
                         "3-way bf16 splits, six MFMAs), fp32 (fp32-input MFMA, the round-1 default) or bf16 (opt-in autocast arithmetic "
                         "-- reported with its own dtype, not comparable)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI; the default) or gloo (test rigs with fewer GPUs than ranks)")
    ap.add_argument("--cpu-frames", type=int, default=12, help="timed frames of the bounded CPU-baseline sample (~10-20 s of host work)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the short training-step leg (BASELINE configs[4]: `train_step` in the line)")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes (the reference's analogue is
    detectron2's launch(num_gpus_per_machine=N), tools/train_net.py:110-117).  This parent never initialises HIP
    (device_count() does not), so nothing is exec'd or forked from a GPU-holding process."""
    import torch

    have = torch.cuda.device_count()
    if args.dist_backend == "nccl" and have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s); RCCL needs one GPU per rank "
                         "(--dist-backend gloo oversubscribes a smaller test rig)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def build_detector(score_thresh: float, device: str, seed: int = 2020):
    import torch

    from articulation3d_amd.config import get_cfg, get_planercnn_cfg_defaults
    from articulation3d_amd.modeling import build_model
    from articulation3d_amd.utils.synthetic import calibrate_batchnorm, synthetic_frames

    cfg = get_cfg()
    get_planercnn_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "planercnn_inference.yaml"))
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = score_thresh
    cfg.MODEL.DEVICE = device
    torch.manual_seed(seed)
    model = build_model(cfg).eval()
    calib = torch.from_numpy(synthetic_frames(2, seed + 1)).to(device)
    calibrate_batchnorm(model, calib)
    return model, cfg


def cpu_baseline(model, frames_u8, score_thresh: float, nframes: int, gpu_results):
    """The CPU oracle (oracle/planercnn_oracle.py, a pure-PyTorch fp32 restatement of the same graph) timed on this box's
    host cores: B=1 per call exactly as the reference's loop (tools/inference.py:215-219), 3 warm-up frames, median of
    `nframes`; plus ONE batched B=8 call so the ratio is not inflated by the reference's batch-1 habit.
    Its detections double as the checker of the HIP run on the same frames (`matched_detections`)."""
    import numpy as np
    import torch

    from oracle import matching as M
    from oracle import planercnn_oracle as O

    # threads = CPUs this process may run on (affinity mask, not the machine total), capped at 64: oneDNN convs at
    # batch 1 do not scale past that and oversubscription makes the oracle pathologically slow
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, 64))
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    P = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    ocfg = O.OracleCfg(score_thresh=score_thresh)
    warm = 3
    imgs = O.frames_to_chw(frames_u8[: warm + nframes])
    ts, outs = [], []
    for i in range(warm + nframes):
        t0 = time.perf_counter()
        out = O.detect(imgs[i:i + 1], P, ocfg)
        if i >= warm:
            ts.append(time.perf_counter() - t0)
        outs.append(out[0])
    med = float(np.median(ts))
    t0 = time.perf_counter()
    O.detect(imgs[:8], P, ocfg)
    t_b8 = time.perf_counter() - t0
    ms = [M.compare_frame(g, o) for g, o in zip(gpu_results, outs)]
    summ = M.summarize(ms)
    matched = dict(matched=bool(all(m["matched"] for m in ms)),
                   definition="per frame: equal detection count; every oracle detection pairs with a HIP detection of the same class, "
                              "box within 5e-3 px, score within 1e-4, rank exchanged only between scores tied to 2e-4 "
                              "(oracle/matching.py); continuous head outputs are reported -- incl. the RAW head vectors before F.normalize "
                              "(max_raw_*_err, no 1/|r| amplification) and the normalised ones weighted by their conditioning (max_*_cond), "
                              "here against the oracle's fp32 run; their bound is tests/test_gpu_e2e.py's float64 yardstick",
                   frames=len(ms), frames_matched=sum(1 for m in ms if m["matched"]),
                   **{k: v for k, v in summ.items() if k.startswith("max_") or k in ("detections", "detections_gpu", "mask_hamming_px")})
    base = {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{nframes} synthetic 480x640 frames, batch 1 per call as the reference loops, after {warm} warm-up frames, median; "
                      f"detections/frame={[len(o['scores']) for o in outs[warm:]]}",
            "batch8_frames_per_s": round(8.0 / t_b8, 4), "os_cpu_count": os.cpu_count(), "affinity_cpus": avail,
            "torch_threads": torch.get_num_threads(),
            "host_tflops": round(205e9 / med / 1e12, 3),
            "caveat": "a stated baseline, never the target: ~205 GFLOP per frame at this rate is a fraction of a TFLOP/s on the threads used "
                      "(oneDNN convolutions at batch 1, restated ROIAlign / NMS in vectorised torch; batch 8 is no faster), so the GPU : CPU "
                      "ratio says how slow this port is on the host, not how good the kernels are -- the roofline fraction does"}
    return base, matched


def kernel_sums(events):
    """CONV_TIMING records -> {label: [algorithmic flops, seconds, launches, executed fp32 multiply-add flops, pipe]}."""
    per = {}
    for name, flops, e0, e1, _shape, executed, pipe, _st in events:
        d = per.setdefault(name, [0.0, 0.0, 0, 0.0, pipe])
        d[0] += flops
        d[1] += e0.elapsed_time(e1) * 1e-3
        d[2] += 1
        d[3] += executed
    return per


def dominant_roofline(per, step_sec=None):
    """The kernel with the largest summed duration among `per` (kernel_sums) and its matrix-pipe roofline entry."""
    # (among the kernels that multiply: at a couple of frames per rank a zero-FLOP transform can hold the largest summed duration)
    cand = {k: v for k, v in per.items() if v[3] > 0} or per
    dname, (dflops, dsec, dn, dexec, dpipe) = max(cand.items(), key=lambda kv: kv[1][1])
    peak = PIPE_PEAK.get(dpipe, FP32_MFMA_PEAK_TFLOPS)
    achieved = dexec * PIPE_FLOPS_PER_FMA[dpipe] / dsec / 1e12  # FLOPs the matrix pipe actually executes
    r = {"kernel": dname, "rocprof_name": rocprof_name(dname), "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
         "frac": round(achieved / peak, 4), "traffic": None, "pipe": dpipe, "launches": dn, "avg_launch_ms": round(1e3 * dsec / dn, 4),
         "fp32_equivalent_tflops": round(dexec / dsec / 1e12, 2), "algorithmic_tflops": round(dflops / dsec / 1e12, 2)}
    if step_sec:
        r["share_of_step_time"] = round(dsec / step_sec, 3)
    return r


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        raise SystemExit(launch_ranks(args))
    force_dist = os.environ.get("A3D_BENCH_FORCE_DIST") == "1"  # (lets a 1-GPU box exercise the RCCL path with one rank)
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree "
                         "(the JSON line reports n_gpus = ranks that actually ran)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP) device: there is no CPU fallback for the product path")
    local_dev = local_rank % torch.cuda.device_count()  # == local_rank on a real N-GPU node
    torch.cuda.set_device(local_dev)
    from articulation3d_amd.streams import side

    side(0)  # the package's side streams take their hardware queues now, in front of the collective library's (articulation3d_amd/streams.py)
    dev = f"cuda:{local_dev}"
    dist = None
    use_dist = world > 1 or force_dist
    if world > 1:
        # N ranks share one host: torch's default of one intra-op worker per host core would put N x cores spinning OpenMP threads
        # next to the N HIP runtimes (utils/arti_vis.py measured what that does to a launch-bound loop); the step needs none of them
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, min(8, avail // world)))
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")  # (only used by the single-process A3D_BENCH_FORCE_DIST form)
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # (RCCL prints a version banner to STDOUT when its communicator comes up; the contract is ONE JSON line there: descriptor 1 points
        # at stderr while the group is created and its first collective runs, and the C library's buffer is flushed before it returns)
        import ctypes

        sys.stdout.flush()
        keep_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.dist_backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device(dev))  # nccl == RCCL on ROCm
                dist.all_reduce(torch.zeros(1, device=dev))
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend=args.dist_backend)
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(keep_fd, 1)
            os.close(keep_fd)

    from articulation3d_amd import ops
    from articulation3d_amd.parallel import gather_records_async
    from articulation3d_amd.utils.synthetic import synthetic_frames

    model, cfg = build_detector(args.score_thresh, dev)
    ops.DEFAULT_PRECISION = {"fp32": 0, "bf16": 1, "bf16x3": 2, "fp16x2": 3}[args.precision]
    B = args.batch
    # contiguous block of the synthetic clip per rank (temporal order is restored by rank order)
    frames_np = synthetic_frames(B, seed=2020 + rank)
    frames = torch.from_numpy(frames_np).to(dev)  # resident in HBM before the timed region

    pending = []  # the all-gather of batch i travels while batch i+1 is computed; it is waited for one step later
    gloo_cpu = use_dist and args.dist_backend != "nccl"  # (gloo test rigs: parallel.gather_records_async stages through the host)

    def step(given=None, src=frames):
        out = model.inference_batched(src, given_boxes=given)
        if use_dist:
            pending.append(gather_records_async(out.records, out.rec_count, rows=out.records.shape[0]))
            if len(pending) > 1:
                pending.pop(0).wait_compact()  # (counts first, then the live records only: parallel.gather_records_async)
        return out

    def drain():
        while pending:
            pending.pop(0).wait_compact()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds: float) -> float:
        if not use_dist:
            return seconds
        t = torch.tensor([seconds], device="cpu" if gloo_cpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(nsteps, nwarm, fn):
        for _ in range(nwarm):
            fn()
        drain()
        barrier()
        t0 = time.perf_counter()
        o = None
        for _ in range(nsteps):
            o = fn()
        drain()  # every batch's records have arrived on every rank before the clock stops
        barrier()
        return max_over_ranks(time.perf_counter() - t0), o

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    # One fully instrumented step OUTSIDE the timed region (HIP events around every conv-GEMM launch, on the launch stream): the
    # per-kernel table (`all_conv_kernels`) and the ranking that names the dominant kernel.  Inside the timed region only the
    # dominant kernel's launches carry events -- ~700 event pairs per step cost 0.8 ms of its 44, and the headline does not pay
    # for the table.
    ops.CONV_TIMING = []
    step()
    drain()
    barrier()
    survey, ops.CONV_TIMING = ops.CONV_TIMING, None
    tot = {}
    for t in survey:
        tot[t[0]] = tot.get(t[0], 0.0) + t[2].elapsed_time(t[3])
    dominant = max(tot.items(), key=lambda kv: kv[1])[0]
    ops.CONV_TIMING_ONLY = frozenset({dominant})
    ops.CONV_TIMING = []
    from articulation3d_amd import parallel as _par

    _par.STATS.update(calls=0, bytes=0, host_s=0.0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    parallel_stats = dict(_par.STATS)
    timing, ops.CONV_TIMING, ops.CONV_TIMING_ONLY = ops.CONV_TIMING, None, None
    timing = [t for t in timing if t[0] == dominant]  # (a launch seen for the first time is instrumented whatever its label)
    elapsed = max_over_ranks(elapsed)
    total_frames = B * world * args.steps
    fps = total_frames / elapsed

    # ---- the same step with the transfers SURVEY 8d puts in the region: uint8 clip H2D (pinned) and packed records D2H
    extra = {}
    if not args.no_operating_points:
        host_frames = torch.from_numpy(frames_np).pin_memory()
        rec_host = torch.empty(out.records.shape, dtype=out.records.dtype).pin_memory()
        cnt_host = torch.empty(out.rec_count.shape, dtype=out.rec_count.dtype).pin_memory()

        # Round 5: the transfers ride a COPY stream beside the compute stream -- clip i + 1 goes H2D into the other of two device buffers
        # while step i runs, the records of step i go D2H while step i + 1 runs (two pinned host buffers each way).  Events order them:
        # a step waits for ITS clip's copy, a copy into a buffer waits for the step that last read it, a record copy for the step that
        # wrote it.  What a caller with host frames would run (pipeline.detect_clip takes the same form).
        from articulation3d_amd.streams import side

        copy_stream = side(2, dev)  # (the package's one pool of side streams: articulation3d_amd/streams.py)
        dev_buf = [torch.empty_like(frames), torch.empty_like(frames)]
        rec_hosts = [rec_host, torch.empty_like(rec_host).pin_memory()]
        cnt_hosts = [cnt_host, torch.empty_like(cnt_host).pin_memory()]
        ev_in = [torch.cuda.Event(), torch.cuda.Event()]    # clip landed in dev_buf[k]
        ev_step = [torch.cuda.Event(), torch.cuda.Event()]  # the step that read dev_buf[k] / wrote its records is done
        state = {"i": 0, "primed": False}

        def prefetch(k):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ev_step[k])  # (a never-recorded event is complete)
                dev_buf[k].copy_(host_frames, non_blocking=True)
                ev_in[k].record(copy_stream)

        def step_xfer():
            k = state["i"] & 1
            if not state["primed"]:
                prefetch(k)
                state["primed"] = True
            prefetch(k ^ 1)  # the NEXT clip, behind this step on the copy stream's own time
            torch.cuda.current_stream().wait_event(ev_in[k])
            o = step(src=dev_buf[k])
            ev_step[k].record()
            o.records.record_stream(copy_stream)
            o.rec_count.record_stream(copy_stream)
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ev_step[k])
                rec_hosts[k].copy_(o.records, non_blocking=True)
                cnt_hosts[k].copy_(o.rec_count, non_blocking=True)
            state["i"] += 1
            return o

        def drain_xfer():
            copy_stream.synchronize()

        n_x = max(2, min(args.steps, 5))
        for _ in range(2):
            step_xfer()
        drain()
        drain_xfer()
        barrier()
        t0x = time.perf_counter()
        for _ in range(n_x):
            step_xfer()
        drain()
        drain_xfer()  # every clip's records are in host memory before the clock stops
        barrier()
        el = max_over_ranks(time.perf_counter() - t0x)
        extra["value_with_transfers"] = {
            "value": round(B * world * n_x / el, 2), "unit": "frames/s", "ms_per_step": round(1e3 * el / n_x, 3), "steps": n_x,
            "vs_resident": round((B * world * n_x / el) / fps, 4),
            "includes": f"H2D of the uint8 clip ({frames_np.nbytes / 1e6:.1f} MB per step, pinned) and D2H of the packed records "
                        f"({rec_host.numel() * 4 / 1e6:.1f} MB per step) inside the timed region, on a copy stream beside the compute stream "
                        "(double-buffered both ways: clip i + 1 travels under step i, the records of step i under step i + 1); `value` has "
                        "the clip resident, as the bench contract defines it"}
        # ---- operating points of SURVEY 8d on the same clip (headline = --score-thresh)
        pts = {}
        pred = model.roi_heads.box_predictor
        saved = pred.test_score_thresh
        n_p = max(2, min(args.steps, 3))
        try:
            for name, th in (("A_thresh0.7", 0.7), ("B_thresh0.0", 0.0)):
                pred.test_score_thresh = th
                el, o = timed(n_p, 1, step)
                pts[name] = {"value": round(B * world * n_p / el, 2), "ms_per_step": round(1e3 * el / n_p, 3), "steps": n_p,
                             "detections_per_frame": round(o.det.count.float().mean().item(), 2)}
        finally:
            pred.test_score_thresh = saved
        R = pred.test_topk_per_image
        gb = torch.zeros((B, R, 4), device=dev)
        gb[:, :4] = torch.tensor([[40.0, 60.0, 300.0, 400.0], [200.0, 100.0, 600.0, 460.0], [10.0, 10.0, 120.0, 90.0], [320.0, 40.0, 420.0, 140.0]], device=dev)
        gc = torch.full((B,), 4, device=dev, dtype=torch.int32)
        el, o = timed(n_p, 1, lambda: step(given=(gb, gc)))
        pts["C_given4"] = {"value": round(B * world * n_p / el, 2), "ms_per_step": round(1e3 * el / n_p, 3), "steps": n_p, "detections_per_frame": 4.0,
                           "note": "4 fixed boxes per frame injected through forward_with_given_boxes (roi_heads.py:147): backbone + depth + "
                                   "mask / plane / axis heads, no RPN / box head"}
        extra["operating_points"] = pts

    # Secondary figure, same clip, same timing discipline: the other fp32-grade arithmetic (fp32-input MFMA when the headline runs
    # bf16x3, and vice versa).  Reported beside the headline, never as it.
    alt = None
    if args.precision in ("fp32", "bf16x3", "fp16x2") and not args.no_alt_modes:
        alt = {}
        for other in [m for m in ("bf16x3", "fp32", "fp16x2") if m != args.precision][:2]:
            saved = ops.DEFAULT_PRECISION
            ops.DEFAULT_PRECISION = {"fp32": 0, "bf16x3": 2, "fp16x2": 3}[other]
            try:
                n_alt = max(2, args.steps // 2)
                el, _ = timed(n_alt, max(1, min(args.warmup, 2)), step)
                # this mode's own roofline object: one fully instrumented step (HIP events around every conv-GEMM launch, on the launch
                # stream) after its timed loop -- the dominant kernel of THIS arithmetic, executed FLOPs against its own pipe's peak
                ops.CONV_TIMING = []
                step()
                drain()
                barrier()
                alt_events, ops.CONV_TIMING = ops.CONV_TIMING, None
            finally:
                ops.DEFAULT_PRECISION = saved
                ops.CONV_TIMING = None
            alt_per = kernel_sums(alt_events)
            alt_roof = dominant_roofline(alt_per, el / n_alt)
            alt_roof["source"] = "one fully instrumented step of this mode (every conv-GEMM launch bracketed by HIP events), after its timed loop"
            alt_roof["conv_kernels_ms_per_step"] = round(1e3 * sum(v[1] for v in alt_per.values()), 3)
            alt_roof["whole_step_fp32_equivalent_tflops"] = round(sum(v[3] for v in alt_per.values()) / (el / n_alt) / 1e12, 2)
            alt[other] = {"value": round(B * world * n_alt / el, 2), "unit": "frames/s", "ms_per_step": round(1e3 * el / n_alt, 3), "steps": n_alt,
                          "dtype": DTYPES[other], "roofline": alt_roof, "note": "python bench.py --precision " + other}

    dets = out.rec_count.float().mean().item()
    raw = out.det.count.float().mean().item()

    # roofline of the dominant kernel: per kernel sums of FLOPs and HIP-event durations over the timed steps.  Names are the
    # dispatcher's own record of what it launched (a3d_last_conv_variant), not a host-side mirror of its rules.
    per, per_main, stream_sec = {}, {}, {}
    for name, flops, e0, e1, _shape, executed, pipe, st in timing:
        sec = e0.elapsed_time(e1) * 1e-3
        d = per.setdefault(name, [0.0, 0.0, 0, 0.0, pipe])
        d[0] += flops
        d[1] += sec
        d[2] += 1
        d[3] += executed
    # The dominant kernel = the largest sum of launch durations over ALL its launches in the timed region (what a rocprofv3 --stats of this
    # command ranks first, and its average duration is the figure that summary shows).  The depth decoder and the small-batch ROI heads
    # run on side streams beside the trunk: an event pair there brackets a kernel that SHARES the chip, so such launches read long
    # (the ROI heads' 16 Winograd GEMMs: 0.12 ms each alone, ~0.4 ms here).  `main_stream` repeats the figures over the launches of the
    # trunk's stream only -- the same kernel with the chip (mostly) to itself.
    for t in survey:  # (the trunk's stream = the one that carries most of the conv time of a whole step)
        stream_sec[t[7]] = stream_sec.get(t[7], 0.0) + t[2].elapsed_time(t[3])
    main_stream = max(stream_sec.items(), key=lambda kv: kv[1])[0]
    for name, flops, e0, e1, _shape, executed, pipe, st in timing:
        if st == main_stream:
            d = per_main.setdefault(name, [0.0, 0.0, 0, 0.0, pipe])
            d[0] += flops
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += 1
            d[3] += executed
    dom = max(per.items(), key=lambda kv: kv[1][1])
    dname, (dflops, dsec, dn, dexec, dpipe) = dom
    conv_sec = sum(v[1] for v in per.values())
    peak = PIPE_PEAK.get(dpipe, FP32_MFMA_PEAK_TFLOPS)
    issued = dexec * PIPE_FLOPS_PER_FMA[dpipe]  # FLOPs the matrix pipe actually executes
    achieved = issued / dsec / 1e12
    roofline = {
        "kernel": dname, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": None, "pipe": dpipe,
        "flops_counted": "EXECUTED on the matrix pipe: fp16x2 issues THREE fp16 MFMA products per fp32 multiply-add, bf16x3 SIX bf16 ones (peak = "
                         "the dense 16-bit MFMA rate, the same for fp16 and bf16); Winograd F(2x2,3x3) performs 16 multiply-adds per 2x2 output "
                         "tile and channel pair where the direct form performs 36",
        "fp32_equivalent_tflops": round(dexec / dsec / 1e12, 2),
        "fp32_equivalent_vs_fp32_mfma_peak": round(dexec / dsec / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),  # (what the fp32-input MFMA could do at best)
        "algorithmic_tflops": round(dflops / dsec / 1e12, 2), "algorithmic_speedup": round(dflops / dexec, 4) if dexec else None,
        "sustained_mfma_rate_measured": F16_MFMA_SUSTAINED_MEASURED_TFLOPS if dpipe in ("f16x3", "bf16x6", "bf16") else None,
        "launches": dn, "avg_launch_ms": round(1e3 * dsec / dn, 4), "avg_launch_gflop_executed_fp32_equivalent": round(dexec / dn / 1e9, 3),
        "avg_launch_gflop_algorithmic": round(dflops / dn / 1e9, 3), "share_of_step_time": round(dsec / elapsed, 3),
    }
    if dname in per_main and per_main[dname][1] > 0:
        mf, ms_, mn, mx, _ = per_main[dname]
        roofline["main_stream"] = {"launches": mn, "avg_launch_ms": round(1e3 * ms_ / mn, 4), "achieved": round(mx * PIPE_FLOPS_PER_FMA[dpipe] / ms_ / 1e12, 2),
                                   "frac": round(mx * PIPE_FLOPS_PER_FMA[dpipe] / ms_ / 1e12 / peak, 4),
                                   "note": "the same kernel over its launches on the trunk's stream only (side-stream launches share the chip and read long)"}
    tpath = next((q for q in (os.path.join(ROOT, "profiles", f"r0{n}_traffic.json") for n in (6, 5, 4, 3, 2)) if os.path.exists(q)), "")
    if tpath:  # HBM bytes per launch from separate rocprofv3 --pmc passes of this command (tools/summarize_pmc_traffic.py)
        tr = json.load(open(tpath))
        rname = rocprof_name(dname)
        roofline["rocprof_name"] = rname
        k = next((v for n, v in tr.get("kernels", {}).items() if n.replace(" ", "") == rname.replace(" ", "")), None)
        if k:
            roofline["traffic"] = k["hbm_bytes_per_launch"]
            roofline["traffic_source"] = "committed " + os.path.relpath(tpath, ROOT) + ": separate rocprofv3 --pmc passes of this command, NOT measured in this run"
    if getattr(model, "small_batch_overlap", 0) >= B:
        roofline["overlap_note"] = ("the depth decoder runs on a second HIP stream beside the ROI branch (A3D_DEPTH_OVERLAP; +1.3 % frames/s): launch durations "
                                    "of kernels that run while it does are those of kernels SHARING the chip, so per-kernel rates here are lower bounds; "
                                    "kernels alone: A3D_DEPTH_OVERLAP=0, profiles/r03_kernel_summary_alone.md")
    per_all = {}
    for name, flops, e0, e1, _shape, executed, pipe, st in survey:
        d = per_all.setdefault(name, [0.0, 0.0, 0, 0.0, pipe])
        d[0] += flops
        d[1] += e0.elapsed_time(e1) * 1e-3
        d[2] += 1
        d[3] += executed
    roofline["all_conv_kernels"] = {k: {"pipe": v[4], "fp32_equivalent_tflops": round(v[3] / v[1] / 1e12, 2) if v[3] else 0.0,
                                        "frac_of_pipe_peak": round(v[3] * PIPE_FLOPS_PER_FMA[v[4]] / v[1] / 1e12 / PIPE_PEAK[v[4]], 4) if v[3] else None,
                                        "algorithmic_tflops": round(v[0] / v[1] / 1e12, 2), "ms_per_step": round(1e3 * v[1], 3),
                                        "launches_per_step": v[2]} for k, v in sorted(per_all.items())}
    if tpath:  # HBM side of every kernel the committed counter passes saw: bytes per launch (FETCH + WRITE) over this run's average duration
        for k, v in roofline["all_conv_kernels"].items():
            rn = rocprof_name(k).replace(" ", "")
            tk = next((w for n, w in tr.get("kernels", {}).items() if n.replace(" ", "") == rn), None)
            if tk is None and k == "wino_input_kernel":
                tk = tr.get("kernels", {}).get("wino_input_kernel")
            # (only where the label and the kernel correspond one to one: a kernel that several labels share -- the wide direct kernel
            # under "conv_h2w_kernel" and "... sk32" -- has ONE average in the counter summary, over launches of very different sizes)
            if tk and v["launches_per_step"] and abs(tk.get("launches_per_step", 0) - v["launches_per_step"]) < 0.5:
                v["hbm_bytes_per_launch"] = tk["hbm_bytes_per_launch"]
                v["hbm_tb_per_s"] = round(tk["hbm_bytes_per_launch"] * v["launches_per_step"] / (v["ms_per_step"] * 1e-3) / 1e12, 2)
                v["hbm_frac_of_8tb_s"] = round(v["hbm_tb_per_s"] / 8.0, 3)
    roofline["all_conv_kernels_source"] = "ONE fully instrumented step run before the timed region (the timed region instruments the dominant kernel only)"
    conv_sec = sum(v[1] for v in per_all.values())
    step_sec = elapsed / args.steps
    roofline["conv_kernels_share_of_step_time"] = round(conv_sec / step_sec, 3)
    roofline["whole_step_fp32_equivalent_tflops"] = round(sum(v[3] for v in per_all.values()) / step_sec / 1e12, 2)

    result = {
        "metric": "frames/sec through PlaneRCNN detector at 480x640",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPES[args.precision], "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: full PlaneRCNN detector (ResNet50-FPN + RPN + ROIAlign + box/mask/plane/axis heads "
                               "+ depth head + NMS + mask paste + plane-offset LSQ + record pack), random-init weights with "
                               "calibrated BN, synthetic 480x640 uint8 frames resident in HBM (arithmetic: see `dtype`)",
                   "frames_per_step_per_gpu": B, "global_frames_per_step": B * world, "score_thresh_test": args.score_thresh,
                   "raw_detections_per_frame": round(raw, 2), "kept_detections_per_frame": round(dets, 2),
                   "proposals_per_frame": round(out.proposals[4].float().mean().item(), 1),
                   "sharding": "contiguous frame blocks per rank" + (", RCCL all-gather of detection records per step" if world > 1 else "")},
        "roofline": roofline,
        # SURVEY 8d's definition of the metric puts the transfers inside the clock: the same step with the uint8 clip copied H2D and
        # the packed records copied D2H in the timed region (details under `value_with_transfers`); `value` has the clip resident
        **({"value_transfer_inclusive": extra["value_with_transfers"]["value"],
            "ms_per_step_transfer_inclusive": extra["value_with_transfers"]["ms_per_step"]} if "value_with_transfers" in extra else {}),
        **extra,
        **({"alt_modes": alt} if alt else {}),
    }
    result["roi_out_of_window"] = ops.roi_window_count(dev) if args.precision == "fp16x2" else None  # (the window monitor: 0 expected)
    # What a SCALE run needs to show that the collective backend saw N distinct GPUs: every rank's device identity, gathered over the group
    # (the reference's analogue: detectron2's launch + comm.gather, tools/train_net.py:110-117, evaluation/arti_evaluation.py:195-199).
    props = torch.cuda.get_device_properties(local_dev)
    ident = {"rank": rank, "local_rank": local_rank, "device_index": local_dev, "name": props.name,
             "uuid": str(getattr(props, "uuid", "")), "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))}
    idents = [ident]
    if use_dist:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    gs = parallel_stats
    result["collective"] = {
        "backend": (dist.get_backend() if use_dist else None), "library": ("RCCL (torch.distributed 'nccl' on ROCm)" if use_dist and dist.get_backend() == "nccl" else None),
        "world": world, "devices": idents, "distinct_devices": len({(d["uuid"], d["pci"]) for d in idents}),
        "gathers_per_step": round(gs["calls"] / max(1, args.steps), 2) if use_dist else 0,
        "gather_bytes_per_step": int(gs["bytes"] / max(1, args.steps)) if use_dist else 0,
        "gather_ms_per_step": round(1e3 * gs["host_s"] / max(1, args.steps), 4) if use_dist else 0.0,
        "note": "two-phase all-gather of the detection records per step (counts, then the live records only: articulation3d_amd/parallel.py); bytes = "
                "what one rank receives over both phases, ms = host time inside GatherHandle.wait_compact (the wait for the counts; the payload "
                "collective is stream-ordered under the next step's kernels)"}
    if not args.no_cpu_baseline:
        # The CPU leg runs on rank 0 AFTER the timed region, at every N (the other ranks wait at the barrier below, their GPUs idle):
        # the oracle on the host cores, and the HIP detections of the same frames compared with its detections.
        if rank == 0:
            from oracle import matching as M

            n_cmp = 3 + args.cpu_frames  # rank 0's clip starts with exactly the frames the CPU leg runs (same seed, same generator)
            cmp_np = synthetic_frames(n_cmp, 2020)
            heads = model.roi_heads
            heads.plane_head.keep_raw = heads.axis_head.keep_raw = True  # checker hook: the head vectors before F.normalize
            try:
                cmp_out = model.inference_batched(torch.from_numpy(cmp_np).to(dev), want_masks=True)
                torch.cuda.synchronize()
            finally:
                heads.plane_head.keep_raw = heads.axis_head.keep_raw = False
            base, matched = cpu_baseline(model, cmp_np, args.score_thresh, args.cpu_frames, M.gpu_frame_results(cmp_out))
            result["cpu_baseline"] = base
            result["matched_detections"] = matched
            result["gpu_over_cpu"] = round(fps / base["value"], 1)
            result["gpu_over_cpu_batch8"] = round(fps / base["batch8_frames_per_s"], 1)
        if use_dist:
            barrier()
    # ---- numeric leaves the driver's record keeps in full (it stores `roofline`, `cpu_baseline` and `config` whole and only NAMES the other
    # top-level keys): the like-for-like arithmetics, the operating points of SURVEY 8d and the reference's unchanged per-frame loop
    if alt:
        result["roofline"]["like_for_like"] = {
            k: {"value": v["value"], "ms_per_step": v["ms_per_step"], "frac": v["roofline"]["frac"], "peak": v["roofline"]["peak"],
                "kernel": v["roofline"]["kernel"]} for k, v in alt.items()}
        result["roofline"]["like_for_like"]["note"] = ("the same 64-frame step in the two arithmetics that multiply full 24-bit operands, as BASELINE "
                                                       "configs[2]'s 'fp32' reads: bf16x3 = exact 3-way bf16 operand splits, fp32 = fp32-input MFMA")
    if "operating_points" in extra:
        result["roofline"]["operating_points"] = {k[0]: {"value": v["value"], "ms_per_step": v["ms_per_step"], "detections_per_frame": v["detections_per_frame"]}
                                                  for k, v in extra["operating_points"].items()}
    if "value_with_transfers" in extra:
        result["roofline"]["value_transfer_inclusive"] = extra["value_with_transfers"]["value"]
    if world == 1 and not args.no_operating_points:
        # the reference's loop shape (tools/inference.py:215-228: one frame per model call, host records per frame), unchanged, on this model
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from loop_bench import loop_b1

        result["roofline"]["loop_b1"] = loop_b1(model, cfg, n=64, conf_threshold=args.score_thresh)
        result["roofline"]["loop_b1"]["note"] = ("tools/loop_bench.py: PlaneRCNN_Branch.inference -> process -> create_instances per frame, 64 frames after 64 "
                                                 "OTHER untimed ones (a video's steady state: every per-frame detection count seen once); the batched figure is `value`")
    if not args.no_train_leg:
        # BASELINE configs[4] where the driver sees it: a short leg of the step1_bbox training step (bf16 autocast arithmetic, as the config
        # asks) at the reference's 2 images per GPU and at 16, each with its own roofline object (dominant kernel + whole step) and, on
        # rank 0 at N = 1, the CPU port timed on the host cores (2-image leg only: the oracle's autograd step takes minutes at 16).  After
        # everything else: it cannot disturb the detection figures.  The trainer takes the detector that is already resident (its frozen
        # stem / res2 are used as they are, everything trainable is copied into the flat buffers) instead of building a second one.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from train_bench import exchange_probe, train_leg

        out = None
        torch.cuda.empty_cache()
        legs = {}
        for tb, tsteps in ((2, 10), (16, 10)):
            r = train_leg(dev, tb, tsteps, 5, precision="bf16", cpu_baseline=(world == 1 and tb == 2 and not args.no_cpu_baseline), rank=rank, world=world,
                          dist=dist if use_dist else None, model=model)
            roof = r["roofline"]
            legs[f"images_per_gpu_{tb}"] = {
                "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"], "timing": r["timing"], "dtype": r["dtype"],
                "config": r["config"], "launches_per_step": r.get("launches_per_step"), "collective": r.get("collective"),
                "roofline": {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "pipe", "launches", "avg_launch_ms", "whole_step", "source")},
                "top_kernels": dict(list(roof["all_gemm_kernels"].items())[:6]),
                **({"cpu_baseline": r["cpu_baseline"], "gpu_over_cpu": r["gpu_over_cpu"]} if "cpu_baseline" in r else {})}
        if world == 1 and args.dist_backend == "nccl":
            # the N > 1 step's gradient exchange on this box: every segment collective of the 2-image step on a ONE-rank RCCL group, against
            # the same step without the exchange -> what the exchange leaves exposed (articulation3d_amd/parallel.py GradientExchange)
            legs["images_per_gpu_2"]["collective"] = exchange_probe(dev, model, 2, 10, 5, dist_module=dist if use_dist else None)
        result["train_step"] = legs
        result["roofline"]["train_step"] = {k: {"value": v["value"], "ms_per_step": v["ms_per_step"], "whole_step_frac": v["roofline"]["whole_step"]["frac"],
                                                "dominant_frac": v["roofline"]["frac"], "launches_per_step": v.get("launches_per_step"),
                                                **({"exposed_exchange_ms": v["collective"]["exposed_ms"]} if v.get("collective") and "exposed_ms" in v["collective"] else {})}
                                            for k, v in legs.items()}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
