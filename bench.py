#!/usr/bin/env python3
"""Benchmark of the EDVR hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One step = one pass of the hot path over one synthetic clip per GPU:
EDVR-M x4 forward, 1x5x3x180x320 -> 3x720x1280, fp32 (BASELINE.json configs[1]).  Clips are
independent work items, so ranks shard clips with no data-path collective:
"scaling": "weak".  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0           # same guide, HBM3E peak


def cpu_baseline(cfg, h, w, seed, y_gpu=None):
    """The CPU oracle (torch CPU convs + C restatement of the DCN) timed on this box's host cores
    on ONE clip of the same workload.  A reported baseline, not the target.  The same clip was run by the
    timed loop on rank 0, so the oracle's output doubles as the full-size parity check (`y_gpu`)."""
    from dynavsr_amd import synth
    from oracle import dcn as odcn, edvr as oedvr
    odcn.lib()
    P = synth.edvr_state_dict(seed)
    x = synth.clip(1, 1, cfg["nframes"], h, w, smooth=False)
    threads = torch.get_num_threads()
    with torch.no_grad():
        t0 = time.time()
        ref = oedvr.edvr_forward(P, x)
        dt = time.time() - t0
    out = {"value": 1.0 / dt, "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": "1 clip forward (EDVR-M x4, 1x5x3x%dx%d), fp32, oracle/edvr.py on %d torch/OpenMP "
                     "threads of %d host cpus, no warm-up" % (h, w, threads, os.cpu_count())}
    if y_gpu is not None:
        d = (y_gpu.detach().cpu().double() - ref.double())
        out["parity_vs_this_run"] = {"rel_l2": float(d.norm() / ref.double().norm()), "max_abs": float(d.abs().max()),
                                     "psnr_db": float(10 * torch.log10(1.0 / (d ** 2).mean()))}
    return out


def inner_step_rate(dev, steps=8):
    """Secondary figure (north_star target >= 50 clips/s): one inner MAML step through the wrapper
    API at LR 176x320 -> SLR 44x80: MFDN forward with grad, EDVR forward+backward on the SLR clip,
    Charbonnier + 10*L1 losses, Adam step over G u E parameters (test_dynavsr.py:235-277)."""
    from copy import deepcopy
    import torch.nn.functional as F
    from dynavsr_amd import synth
    from dynavsr_amd.adapt import make_inner_optimizer
    from dynavsr_amd.models import create_model
    from dynavsr_amd.options import options as option
    opt = option.dict_to_nonedict(option.parse(os.path.join(
        ROOT, "dynavsr_amd", "options", "test", "EDVR", "EDVR_M_S4.yml"), is_train=False))
    opt["dist"] = False
    for k in ("pretrain_model_G", "pretrain_model_E"):
        opt["path"][k] = None
    model, est = create_model(opt)
    _, est_fixed = create_model(opt)
    model.netG.load_state_dict(synth.edvr_state_dict(0))
    est.netE.load_state_dict(synth.mfdn_state_dict(0))
    est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
    netG, netE = deepcopy(model.netG), deepcopy(est.netE)
    model.netG, est.netE = netG, netE
    inner = make_inner_optimizer(opt, netG, netE)
    lqs = synth.clip(3, 1, 5, 176, 320, smooth=False).to(dev)
    data = {"LQs": lqs}
    est_fixed.feed_data(data); est_fixed.test()
    slr_fixed = est_fixed.fake_L

    def step():
        est.feed_data(data); est.forward_without_optim()
        inner.zero_grad()
        model.feed_data({"LQs": est.fake_L, "GT": lqs[:, 2]})
        loss = model.calculate_loss() + 10 * F.l1_loss(est.fake_L, slr_fixed)
        loss.backward()
        inner.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"value": 1e3 / ms, "unit": "clips/s", "ms_per_step": ms, "steps": steps,
            "workload": "1 inner MAML step, EDVR-M x4 + MFDN, LR 1x5x3x176x320 -> SLR 44x80, fp32, Adam; "
                        "MFDN runs on the native estimator tape (dvsr_estimator_*)"}


def split_mode_rate(cfg, h, w, x, y_fp32, steps, warmup):
    """The same forward with network_G.bf16_mfma = 2: every fp32 operand of the 3x3 convs split
    into three bf16 pieces, six products on the bf16 MFMA, fp32 accumulation.  Reported BESIDE the headline,
    never as it: `value` above is the exact-fp32 MFMA path."""
    from dynavsr_amd import synth
    from dynavsr_amd.models.archs.EDVR_arch import EDVR
    net = EDVR(bf16_mfma=2, **cfg)
    net.load_state_dict(synth.edvr_state_dict(0, **cfg), strict=True)
    net = net.to(x.device)
    with torch.no_grad():
        for _ in range(warmup):
            y = net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = net(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    d = (y - y_fp32).double()
    return {"value": steps / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / steps, "dtype": "f32 results from "
            "3-way bf16 operand split on v_mfma_f32_32x32x16_bf16 (fp32 accumulate)",
            "max_abs_vs_fp32_mfma_path": float(d.abs().max()),
            "rel_l2_vs_fp32_mfma_path": float(d.norm() / y_fp32.double().norm()),
            "note": "opt-in (bf16_mfma = 2), held to the fp32 parity bars by tests/test_gpu_edvr.py"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--height", type=int, default=180)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inner-step", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="skip the experimental bf16-split timing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run (also with 1 rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from dynavsr_amd import engine, synth
    from dynavsr_amd.models.archs.EDVR_arch import EDVR

    cfg = dict(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=4)
    h, w = args.height, args.width
    net = EDVR(**cfg)
    net.load_state_dict(synth.edvr_state_dict(0, **cfg), strict=True)   # random init, seed-fixed
    net = net.to(dev)
    x = synth.clip(1 + rank, 1, cfg["nframes"], h, w, smooth=False).to(dev)   # U[0,1) clip, resident in HBM

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            y = net(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = net(x)
        barrier()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    assert torch.isfinite(y).all()

    line = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        line = {
            "metric": "inner-loop frames/sec/GPU (EDVR-M x4, 5x3x180x320)",
            "value": world * args.steps / elapsed, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "EDVR-M x4 forward (PCD deformable alignment + TSA fusion + "
                                   "reconstruction), 1 clip 1x5x3x%dx%d -> 3x%dx%d per GPU per step "
                                   "(BASELINE.json configs[1])" % (h, w, 4 * h, 4 * w),
                       "clips_per_step": world, "sharding": "independent clips per rank, no collective"},
        }
        # ---- roofline of the dominant kernel: separate pass, hipEvents around every launch
        plan = engine.get_plan(net._cfg(), 1, h, w)
        params = [p.detach().contiguous() for p in net.ordered_parameters()]
        ws = torch.empty(plan.workspace_bytes(False), dtype=torch.uint8, device=dev)
        out = torch.empty_like(y)
        info = plan.op_info()
        acc = {}
        reps = max(3, min(10, args.steps))
        for _ in range(reps):
            for (kind, _name, fl, by), t_ms in zip(info, plan.forward_timed(params, x, out, ws)):
                a = acc.setdefault(kind, [0.0, 0.0, 0.0, 0])
                a[0] += t_ms; a[1] += fl; a[2] += by; a[3] += 1
        dom = max(acc, key=lambda k: acc[k][0])
        t_ms, fl, by, cnt = acc[dom]
        total_ms = sum(a[0] for a in acc.values())
        if fl / max(by, 1) > FP32_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
            ach = fl / (t_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / FP32_MFMA_PEAK_TFLOPS}
        else:
            ach = by / (t_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS}
        traffic, traffic_src = None, None
        try:   # HBM bytes per launch from the committed PMC passes (cannot be collected inside this process)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f).get(dom)
            if t and (h, w) == (180, 320):
                traffic, traffic_src = t["bytes_per_launch"], t["source"]
        except (OSError, ValueError):
            pass
        roof.update({"traffic": traffic, "traffic_source": traffic_src, "kernel": dom, "launches_per_step": cnt // reps,
                     "avg_launch_ms": t_ms / cnt, "share_of_step": t_ms / total_ms,
                     "method": "hipEvent pair around every launch on the launch stream, %d instrumented "
                               "passes after the timed region" % reps})
        line["roofline"] = roof
        line["kernel_breakdown_ms_per_step"] = {k: round(a[0] / reps, 4) for k, a in
                                                sorted(acc.items(), key=lambda kv: -kv[1][0])}
        line["end_to_end_tflops"] = sum(a[1] for a in acc.values()) / reps / (ms * 1e-3) / 1e12
        if world == 1 and not args.no_split:
            line["experimental_bf16_split"] = split_mode_rate(cfg, h, w, x, y, args.steps, args.warmup)
        if world == 1 and not args.no_inner_step:
            line["inner_step"] = inner_step_rate(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, h, w, 0, y)   # same clip (seed 1 + rank 0), same weights
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
