"""Is the training step's time independent of the ORDER in which a process first uses its streams?  (the `-m gpu` test
tests/test_gpu_bench.py::test_stream_placement_does_not_depend_on_first_use_order runs this file.)

    python tools/probes/stream_order_check.py            # parent: two fresh children, one JSON line
    python tools/probes/stream_order_check.py --child package_first | foreign_first

package_first: streams.side() right after the device is set, then three foreign streams and a one-rank RCCL group (its communicator
stream) -- the package's rule.  foreign_first: three foreign streams and RCCL FIRST, the package's pool last -- the order that costs 18.1
instead of 15.1 ms per 16-image step with the runtime's default of four hardware queues.  `--queues N` exports GPU_MAX_HW_QUEUES=N to the
children (round 6 measured 5, 6 and 8: articulation3d_amd/__init__.py says why the default stays)."""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(order, batch):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch

    import articulation3d_amd  # noqa: F401
    from articulation3d_amd import streams
    from train_bench import train_leg

    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    keep = []

    def foreign():
        import torch.distributed as dist

        for _ in range(3):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                keep.append(torch.zeros(16, device=dev) + 1)
            keep.append(s)
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        os.dup2(2, 1)  # (RCCL's banner: stdout of this child is a pipe the parent parses)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        dist.all_reduce(torch.zeros(1, device=dev))
        torch.cuda.synchronize()

    if order == "package_first":
        streams.side(0)
        foreign()
    else:
        foreign()
    r = train_leg(dev, batch, 10, 5, precision="bf16")
    sys.stderr.write("RESULT " + json.dumps({"order": order, "ms_per_step": r["ms_per_step"], "images_per_s": r["value"],
                                             "queues": streams.queue_setting()}) + "\n")
    sys.stderr.flush()
    import torch.distributed as dist

    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--queues", default=None)
    a = ap.parse_args()
    if a.child:
        child(a.child, a.batch)
        return
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GPU_MAX_HW_QUEUES", None)
    if a.queues:
        env["GPU_MAX_HW_QUEUES"] = a.queues
    out = {}
    for order in ("package_first", "foreign_first"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", order, "--batch", str(a.batch)], env=env, capture_output=True, text=True,
                           timeout=900)
        line = [l for l in r.stderr.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and line, r.stderr[-2000:]
        out[order] = json.loads(line[-1][7:])
    a_, b_ = out["package_first"]["ms_per_step"], out["foreign_first"]["ms_per_step"]
    out["ratio_foreign_over_package"] = round(b_ / a_, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
