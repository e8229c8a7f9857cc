#!/usr/bin/env python3
"""The inner MAML step (bench.inner_step_rate: 16 frames per batch at LR 176x320) in the orders a driver can bring the
process up, each in its own process:
    plain       dynavsr_amd.configure_runtime() first (six hardware queues), no process group
    rccl_first  a one-rank RCCL group is initialised FIRST (train_dynavsr.py:23-30), configure_runtime() comes too late
    one_stream  DVSR_BWD_STREAMS=0 (weight gradients on the launch stream)
Prints one JSON object per mode: ms per frame-step (min / median of 5), the runtime report, the probe's answer.
usage: python tools/queue_probe_bench.py [mode]   (no mode: runs all three as children)"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(mode):
    import torch
    sys.path.insert(0, ROOT)
    dev = torch.device("cuda", 0)
    if mode == "rccl_first":
        import torch.distributed as tdist
        torch.cuda.set_device(0)
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        tdist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        t = torch.ones(1 << 20, device=dev)
        tdist.all_reduce(t)
        torch.cuda.synchronize()
    import warnings
    import dynavsr_amd
    from dynavsr_amd import _lib as L
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        rt = dynavsr_amd.configure_runtime()
    import bench
    times = [bench.inner_step_rate(dev)["ms_per_step"] for _ in range(5)]
    rep = L.runtime_report(dev)
    print(json.dumps({"mode": mode, "ms_min": min(times), "ms_median": sorted(times)[2], "ms_all": times, "runtime": rep,
                      "configure_effective": rt["effective"]}))


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    out = {}
    for mode in ("plain", "rccl_first", "one_stream", "plain", "rccl_first"):
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        if mode == "one_stream":
            env["DVSR_BWD_STREAMS"] = "0"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print(line[-1] if line else "%s FAILED: %s" % (mode, r.stderr[-800:]), flush=True)


if __name__ == "__main__":
    main()
