#!/usr/bin/env python3
"""Benchmark of the EDVR hot path on MI355X (contract: see the task statement / DESIGN.md §5).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

`value` -- one step = one pass of the hot path over one synthetic clip per GPU: EDVR-M x4 forward,
1x5x3x180x320 -> 3x720x1280, fp32 (BASELINE.json configs[1], the configuration the metric is quoted on).
Clips are independent work items, so ranks shard clips with no data-path collective:
"scaling": "weak".  Rank 0 prints ONE JSON line.  Beside the headline the line carries:
  roofline           dominant kernel of the headline forward, hipEvents around every launch
  inner_step         (ii) one inner MAML step at LR 176x320 (MFDN fwd+bwd, EDVR fwd+bwd on the SLR clip, losses,
                     Adam) over >= 50 steps, with its own roofline object and the launches per step
  per_frame_pipeline (iii) baseline forward + inner step + adapted forward per frame (adapt_video)
  meta_step          configs[3]: every rank runs one outer meta-training iteration on its shard of the tasks and
                     the meta-gradient is all-reduced over RCCL (the one real exchange step of the method)
  edvr_l_bf16        configs[4]: EDVR-L x4 1x7x3x64x64 on the three MFMA modes
  cpu_baseline       the CPU oracle on this box's host cores (a reported baseline, not the target)
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import dynavsr_amd  # noqa: E402

dynavsr_amd.configure_runtime()   # six hardware queues unless the user chose; precedes the first HIP call

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, dense bf16
HBM_PEAK_GBS = 8000.0           # same guide, HBM3E peak

# Algorithmic work (probe-derived; scales linearly in H*W)
F_EDVR_M_180 = 973.6e9          # EDVR-M x4 forward, 1x5x3x180x320
F_MFDN_180 = 53.1e9             # MFDN x4 forward, 5 frames of 180x320
F_EDVR_L_64 = 373.4e9           # EDVR-L x4 forward, 1x7x3x64x64


def f_edvr(h, w):
    return F_EDVR_M_180 * (h * w) / (180.0 * 320.0)


def f_mfdn(h, w):
    return F_MFDN_180 * (h * w) / (180.0 * 320.0)


def f_inner(h, w):
    """3 x EDVR(SLR) (forward, data gradient, weight gradient) + 3 x MFDN(LR) + the frozen MFDN: 386 GFLOP @176x320."""
    return 3 * f_edvr(h // 4, w // 4) + 4 * f_mfdn(h, w)


def pipe_mix(f32_flops, bf16_flops, t_s):
    """Roofline fields of work issued to BOTH matrix pipes (fp32: v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s dense; bf16:
    v_mfma_f32_32x32x16_bf16, 2500): frac = the share of the time the pipes need at their dense peaks for exactly this
    instruction mix (what PMC's SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES measures), achieved = all issued FLOPs / time,
    peak = achieved / frac (the rate of this mix on always-busy pipes)."""
    pipe_s = f32_flops / (FP32_MFMA_PEAK_TFLOPS * 1e12) + bf16_flops / (BF16_MFMA_PEAK_TFLOPS * 1e12)
    issued = f32_flops + bf16_flops
    frac = pipe_s / t_s
    ach = issued / t_s / 1e12
    return {"bound": "mfma", "achieved": ach, "peak": (ach / frac) if frac > 0 else FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": frac, "issued_flops_f32_pipe": f32_flops, "issued_flops_bf16_pipe": bf16_flops,
            "peak_f32_pipe": FP32_MFMA_PEAK_TFLOPS, "peak_bf16_pipe": BF16_MFMA_PEAK_TFLOPS}


def mfma_roof(t_ms, alg_flops, tapes, what, extra=None):
    """Roofline object of a whole leg on the matrix pipes.  `tapes` = [(plan.work() dict, forward passes, backward
    passes)] of everything the leg runs: the tapes' issued / algorithmic ratios price SURVEY 8d's algorithmic FLOP figure down
    to what either pipe is issued (launches on the Winograd kernels do 16/36 of their multiplies; launches on the exact 3-way
    bf16 operand split issue six bf16 products per fp32 product), so `frac` is a fraction of a hardware ceiling (<= 1) and
    the algorithmic rate is reported beside it."""
    def tot(kf, kb):
        return sum(wk[kf] * nf + wk[kb] * nb for wk, nf, nb in tapes)
    alg = tot("fwd_algorithmic", "bwd_algorithmic")
    k = alg_flops / alg if alg else 1.0          # (SURVEY's figure also counts the few non-contraction FLOPs)
    exe = tot("fwd_executed", "bwd_executed") * k
    r = pipe_mix(tot("fwd_f32_pipe", "bwd_f32_pipe") * k, tot("fwd_bf16_pipe", "bwd_bf16_pipe") * k, t_ms * 1e-3)
    r.update({"traffic": None, "executed_flops": exe, "algorithmic_flops": alg_flops,
              "algorithmic_tflops": alg_flops / (t_ms * 1e-3) / 1e12, "mfma_flops_executed_frac": exe / alg_flops,
              "fp32_products_tflops": exe / (t_ms * 1e-3) / 1e12,
              "note": "%s: frac = time the two matrix pipes need at their dense peaks for the FLOPs ISSUED to them / time "
                      "(issued = SURVEY 8d's algorithmic FLOPs x the issued share of the leg's launch tapes: Winograd "
                      "F(2x2,3x3) launches do 4/9 of their multiplies, F(4x4,3x3) launches (no-grad forwards) 1/4, launches on the exact 3-way bf16 split issue six "
                      "bf16 products per fp32 product); achieved = issued FLOPs / time, peak = achieved / frac; "
                      "executed_flops / fp32_products_tflops = the fp32 products behind them (r03's `achieved`); "
                      "algorithmic_tflops = the same time priced with the direct sums" % what})
    if extra:
        r.update(extra)
    return r


def _warm(fn, seconds=0.3, at_least=3):
    """Warm-up by TIME, not by count: after host-side set-up (building a 20 M-parameter network takes seconds) the GPU
    has clocked down, and a few short iterations are over before it is back up -- the same forward then measures 3 ms
    slower (seen on the EDVR-L leg: 3.1 vs 6.9 ms run to run with 3 warm-up iterations)."""
    t0, n = time.perf_counter(), 0
    while n < at_least or time.perf_counter() - t0 < seconds:
        fn()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


def _opt():
    from dynavsr_amd.options import options as option
    opt = option.dict_to_nonedict(option.parse(os.path.join(
        ROOT, "dynavsr_amd", "options", "test", "EDVR", "EDVR_M_S4.yml"), is_train=False))
    opt["dist"] = False
    for k in ("pretrain_model_G", "pretrain_model_E"):
        opt["path"][k] = None
    return opt


def _sources_digest():
    """Identity of the conv kernel sources the committed PMC traffic figure was collected on."""
    h = hashlib.sha256()
    for f in ("conv2d_v2.hip", "conv2d_wino.hip", "conv2d_wino3.hip", "conv2d_wino4.hip", "conv2d_wino5.hip", "small_grid.h", "common.h"):
        with open(os.path.join(ROOT, "dynavsr_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


# ---- CPU baseline ---------------------------------------------------------------------------------
def cpu_baseline(cfg, h, w, seed, y_gpu=None):
    """The CPU oracle (torch CPU convs + C restatement of the DCN) timed on this box's host cores on a bounded
    sample of the same workloads: forward @HxW on all cores (1 warm-up + median of 3), forward
    @64x64 on ONE core and on all cores (1 warm-up + median of 5 each, SURVEY 8d's protocol; the two full-size
    legs take 8-10 s per pass and keep 3 so that the default run stays within minutes), one inner step @176x320 on
    all cores (1 warm-up + median of 3).  The headline clip was run by the timed GPU loop on rank 0, so the oracle's
    output doubles as the full-size parity check (`y_gpu`)."""
    from collections import OrderedDict
    import torch.nn.functional as F
    from dynavsr_amd import synth
    from oracle import dcn as odcn, edvr as oedvr, mfdn as omfdn
    odcn.lib()
    P = synth.edvr_state_dict(seed)
    all_threads = torch.get_num_threads()

    def med(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    ref = [None]
    x = synth.clip(1, 1, cfg["nframes"], h, w, smooth=False)

    def fwd_big():
        with torch.no_grad():
            ref[0] = oedvr.edvr_forward(P, x)
    t_all = med(fwd_big)
    x64 = synth.clip(1, 1, cfg["nframes"], 64, 64, smooth=False)

    def fwd64():
        with torch.no_grad():
            oedvr.edvr_forward(P, x64)
    t64_all = med(fwd64, 5)
    def omp_threads(n):   # the C DCN oracle is plain OpenMP (libgomp); torch may run its own pool
        torch.set_num_threads(n)
        try:
            import ctypes
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(n)
        except OSError:
            pass
    omp_threads(1)
    try:
        t64_one = med(fwd64, 5)
    finally:
        omp_threads(all_threads)
    # one inner MAML step (test_dynavsr.py:235-277) at LR 176x320 through the oracle's functions
    PG = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in P.items())
    PE = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in synth.mfdn_state_dict(0).items())
    lqs = synth.clip(3, 1, 5, 176, 320, smooth=False)
    with torch.no_grad():
        slr_fixed = omfdn.mfdn_forward(synth.mfdn_state_dict(1), lqs)
    adam = torch.optim.Adam(list(PG.values()) + list(PE.values()), lr=1e-5, betas=(0.9, 0.99))

    def inner():
        adam.zero_grad()
        slr = omfdn.mfdn_forward(PE, lqs)
        loss = oedvr.charbonnier(oedvr.edvr_forward(PG, slr), lqs[:, 2]) + 10 * F.l1_loss(slr, slr_fixed)
        loss.backward()
        adam.step()
    t_inner = med(inner)
    out = {"value": 1.0 / t_all, "unit": "frames/s", "cores": all_threads, "kind": "port",
           "sample": "EDVR-M x4 forward, 1 clip 1x5x3x%dx%d, fp32, oracle/edvr.py (torch CPU convs + the C DCN oracle) on "
                     "%d torch/OpenMP threads of %d host cpus; 1 warm-up, median of 3" % (h, w, all_threads, os.cpu_count()),
           "forward_64x64": {"all_cores": {"value": 1.0 / t64_all, "unit": "frames/s", "cores": all_threads},
                             "single_core": {"value": 1.0 / t64_one, "unit": "frames/s", "cores": 1},
                             "sample": "1x5x3x64x64 (BASELINE configs[0] size), 1 warm-up, median of 5"},
           "inner_step_176x320": {"value": 1.0 / t_inner, "unit": "clips/s", "cores": all_threads,
                                  "sample": "MFDN fwd+bwd @176x320, EDVR fwd+bwd @44x80, Charbonnier + 10 L1, Adam; "
                                            "frozen MFDN hoisted like the GPU path; 1 warm-up, median of 3"}}
    if y_gpu is not None:
        d = (y_gpu.detach().cpu().double() - ref[0].double())
        out["parity_vs_this_run"] = {"rel_l2": float(d.norm() / ref[0].double().norm()), "max_abs": float(d.abs().max()),
                                     "psnr_db": float(10 * torch.log10(1.0 / (d ** 2).mean()))}
    return out


# ---- inner MAML step (SURVEY §8d ii) --------------------------------------------------------------
def inner_step_rate(dev, steps=60, h=176, w=320, frames_per_batch=16):
    """One inner MAML step per frame at LR 176x320 -> SLR 44x80 (test_dynavsr.py:208-277): refresh of the frame's copies,
    frozen-estimator forward, MFDN forward with grad, EDVR forward+backward on the SLR clip, Charbonnier + 10*L1 losses,
    MFDN backward, Adam step over G u E parameters.  north_star target: >= 50 clips/s.
    `value`: K frames adapted as ONE batch with per-frame parameter gradients (adapt.FrameBatch; every shipped YAML has
    adapt_iter = 1, so all K copies start from the same weights) -- per frame-step.  `per_frame_loop`: the reference's
    loop, one frame at a time (adapt.adapt_frame), same content.  `step_only`: the r01 / r02 protocol (no copy refresh,
    frozen estimator hoisted out of the timed loop), kept for continuity."""
    from copy import deepcopy
    from dynavsr_amd import engine, hipops, synth
    from dynavsr_amd.adapt import FrameBatch, adapt_frame, make_inner_optimizer
    from dynavsr_amd.models import create_model
    opt = _opt()
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    model.netG.load_state_dict(synth.edvr_state_dict(0))
    est.netE.load_state_dict(synth.mfdn_state_dict(0))
    est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
    K = frames_per_batch
    fl = f_inner(h, w)

    def timed(fn, n):
        _warm(fn)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    # (a) K frames as one batch
    assert FrameBatch.supported(opt, model, est)
    fb = FrameBatch(opt, model.netG, est.netE, K)
    lqs_k = synth.clip(3, K, 5, h, w, smooth=False).to(dev)
    ms_batch = timed(lambda: fb.adapt(model, est, est_fixed, lqs_k), max(4, steps // K))
    ms = ms_batch / K
    gp = engine.get_plan(model.netG._cfg(), K, h // 4, w // 4, grad_groups=K)
    ep = engine.get_estimator_plan((engine.MFDN, est.netE.nf, est.netE.in_nc, est.netE.scale, 5), K, h, w, grad_groups=K)
    launches = {"edvr_forward": gp.n_launches, "edvr_backward": gp.n_backward_launches,
                "estimator_forward": ep.n_launches, "estimator_backward": ep.n_backward_launches}
    # (a') BASELINE configs[2] takes 3 inner steps per frame: after the first one the copies have diverged and the batch runs
    # on per-frame weight sets (dvsr_*_plan_create_ex)
    ms3 = timed(lambda: fb.adapt(model, est, est_fixed, lqs_k, steps=3), max(3, steps // (3 * K))) / K
    del fb
    # (b) the per-frame loop, same content
    lq1 = {"LQs": lqs_k[:1].contiguous()}
    ms_loop = timed(lambda: adapt_frame(opt, model, est, modelcp, estcp, est_fixed, lq1, final_test=False), steps)
    # (c) r01 / r02 protocol: the bare step on one pair of copies
    netG, netE = deepcopy(model.netG), deepcopy(est.netE)
    model.netG, est.netE = netG, netE
    inner = make_inner_optimizer(opt, netG, netE)
    lqs = lq1["LQs"]
    est_fixed.feed_data(lq1); est_fixed.test()
    slr_fixed = est_fixed.fake_L

    def step():
        est.feed_data(lq1); est.forward_without_optim()
        inner.zero_grad()
        model.feed_data({"LQs": est.fake_L, "GT": lqs[:, 2]})
        loss = hipops.inner_loss(model.calculate_loss(), est.fake_L, slr_fixed, 10.0)
        loss.backward()
        inner.step()
    ms_step = timed(step, steps)

    ecfg = (engine.MFDN, est.netE.nf, est.netE.in_nc, est.netE.scale, 5)
    g1, e1 = engine.get_plan(model.netG._cfg(), 1, h // 4, w // 4).work(), engine.get_estimator_plan(ecfg, 1, h, w).work()
    gk, ek = gp.work(), ep.work()
    gks = engine.get_plan(model.netG._cfg(), K, h // 4, w // 4, grad_groups=K, weight_sets=K).work()
    eks = engine.get_estimator_plan(ecfg, K, h, w, grad_groups=K, weight_sets=K).work()
    # tapes per step: EDVR forward + backward, MFDN forward + backward, the frozen MFDN's forward
    tapes_k = [(gk, 1, 1), (ek, 2, 1)]
    tapes_1 = [(g1, 1, 1), (e1, 2, 1)]
    tapes_step_only = [(g1, 1, 1), (e1, 1, 1)]
    tapes_3 = [(gk, 1, 1), (ek, 2, 1), (gks, 2, 2), (eks, 2, 2)]
    traffic = {}
    try:   # PMC FETCH_SIZE + WRITE_SIZE of one batched step (tools/pmc_traffic.py --inner), per frame-step
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            tj = json.load(f).get("inner_step_batched")
        if tj and tj.get("frames_per_batch") == K and (tj.get("h"), tj.get("w")) == (h, w):
            # algorithmic bytes of the same step: every launch's distinct inputs once + outputs once over the forward tapes
            # (EDVR on the SLR batch, MFDN twice: adapted + frozen), the backward tapes counted as twice their forward
            alg_bytes = (3.0 * gk["fwd_bytes"] + 4.0 * ek["fwd_bytes"]) / K
            traffic = {"traffic": tj["bytes_per_frame_step"], "traffic_source": tj["source"],
                       "traffic_scope": "all kernels of one batched inner step / K (PMC FETCH_SIZE + WRITE_SIZE)",
                       "algorithmic_bytes_per_frame_step": alg_bytes,
                       "hbm_gbs": tj["bytes_per_frame_step"] / (ms * 1e-3) / 1e9}
    except (OSError, ValueError):
        pass

    def roof(t_ms, flops, tapes, extra=None):
        r = mfma_roof(t_ms, flops, tapes, "whole step (not one kernel)", extra)
        r["algorithmic_gflop_per_step"] = flops / 1e9
        return r
    return {"value": 1e3 / ms, "unit": "clips/s", "ms_per_step": ms, "frames_per_batch": K, "ms_per_batch": ms_batch,
            "workload": "1 inner MAML step per frame, EDVR-M x4 + MFDN, LR 1x5x3x%dx%d -> SLR %dx%d, fp32, Adam, %d frames "
                        "adapted as one batch with per-frame parameter gradients (all copies start from the same weights: "
                        "adapt_iter = 1); includes the refresh of the per-frame copies and the frozen estimator's forward"
                        % (h, w, h // 4, w // 4, K),
            "roofline": roof(ms, fl, tapes_k, traffic),
            "per_frame_loop": {"value": 1e3 / ms_loop, "unit": "clips/s", "ms_per_step": ms_loop,
                               "workload": "the same step one frame at a time (adapt.adapt_frame: test_dynavsr.py:208-277)",
                               "roofline": roof(ms_loop, fl, tapes_1)},
            "step_only": {"value": 1e3 / ms_step, "unit": "clips/s", "ms_per_step": ms_step,
                          "workload": "r01 / r02 protocol: one frame, no copy refresh, frozen estimator outside the loop "
                                      "(%.0f GFLOP executed)" % ((fl - f_mfdn(h, w)) / 1e9),
                          "roofline": roof(ms_step, fl - f_mfdn(h, w), tapes_step_only)},
            "three_inner_steps": {"ms_per_frame": ms3, "frames_per_batch": K,
                                  "workload": "adapt_iter = 3 (BASELINE configs[2]) on the same batch: steps 2 and 3 with per-frame "
                                              "weight sets; %.0f GFLOP per frame" % ((3 * (fl - f_mfdn(h, w)) + f_mfdn(h, w)) / 1e9),
                                  "roofline": roof(ms3, 3 * (fl - f_mfdn(h, w)) + f_mfdn(h, w), tapes_3)},
            "tape_ops_per_batch": launches,
            "target_clips_per_s": 50}


# ---- per-frame pipeline (SURVEY §8d iii) ----------------------------------------------------------
def per_frame_pipeline_rate(dev, clips=32, h=176, w=320, frames_per_batch=16):
    """test_dynavsr.py:197-283 per frame: un-adapted baseline forward, copies refreshed, one inner step, adapted
    forward -- adapt_video with the two full-size forwards on side streams under the next clip's adaptation."""
    from dynavsr_amd import synth
    from dynavsr_amd.adapt import adapt_video
    from dynavsr_amd.models import create_model
    opt = _opt()
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
    est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
    data = [{"LQs": synth.clip(10 + i, 1, 5, h, w, smooth=False).to(dev)} for i in range(clips)]
    out = {}
    for name, ov, kf in (("sequential", False, 1), ("overlapped", True, 1), ("batched", True, frames_per_batch)):
        for _ in adapt_video(opt, model, est, modelcp, estcp, est_fixed, data[:max(3, kf)], overlap=ov, frames_per_batch=kf):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in adapt_video(opt, model, est, modelcp, estcp, est_fixed, data, overlap=ov, frames_per_batch=kf):
            pass
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / clips * 1e3
    ms = min(out["overlapped"], out["batched"])
    fl = 2 * f_edvr(h, w) + f_inner(h, w)
    from dynavsr_amd import engine
    kb = frames_per_batch if out["batched"] <= out["overlapped"] else 1
    ecfg = (engine.MFDN, est.netE.nf, est.netE.in_nc, est.netE.scale, 5)
    tapes = [(engine.get_plan(model.netG._cfg(), kb, h, w).work(nograd=True), 2, 0),   # baseline + adapted forwards (no-grad)
             (engine.get_plan(model.netG._cfg(), kb, h // 4, w // 4, grad_groups=kb).work(), 1, 1),
             (engine.get_estimator_plan(ecfg, kb, h, w, grad_groups=kb).work(), 2, 1)]
    roof = mfma_roof(ms, fl, tapes, "whole per-frame pipeline (not one kernel)")
    roof["algorithmic_gflop_per_frame"] = fl / 1e9
    return {"value": 1e3 / ms, "unit": "frames/s", "ms_per_frame": ms, "ms_per_frame_sequential": out["sequential"],
            "ms_per_frame_overlapped_per_frame_loop": out["overlapped"],
            "ms_per_frame_batched_inner_steps": out["batched"], "frames_per_batch": frames_per_batch,
            "clips": clips,
            "workload": "per frame: baseline EDVR-M x4 forward @%dx%d + 1 inner MAML step + adapted forward @%dx%d "
                        "(the baseline forward is report-only in the reference and is measured here too)" % (h, w, h, w),
            "roofline": roof}


# ---- configs[3]: one outer meta-training iteration with the RCCL exchange -----------------------------
def meta_step_rate(dev, world, dist, group=None, cpu_group=False, tasks_per_rank=2, iters=4):
    """Every rank: adapt.meta_train_step on its own `tasks_per_rank` tasks at the training YAML's shapes (LR 5x3x64x64,
    SLR 16x16, HR 256x256; train_dynavsr.py:265-438), with the meta-gradient all-reduce over the process group.
    The collective alone is timed separately on the same flat buffer.  `group`: the RCCL group the gradients travel over;
    `cpu_group`: the default group is the gloo one (barriers and the MAX of the times use CPU tensors on it)."""
    from dynavsr_amd import dist as D, synth
    from dynavsr_amd.adapt import meta_train_step
    from dynavsr_amd.models import create_model
    rank = dist.get_rank() if dist is not None else 0
    opt = _opt()
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
    params = list(model.netG.parameters()) + list(est.netE.parameters())
    optimizer = torch.optim.Adam(params, lr=1e-5, betas=(0.9, 0.99))
    B = tasks_per_rank
    data = {"LQs": synth.clip(100 + rank, B, 5, 64, 64, smooth=False).to(dev),
            "SuperLQs": synth.clip(200 + rank, B, 5, 16, 16, smooth=False).to(dev),
            "GT": synth.clip(300 + rank, B, 5, 256, 256, smooth=False).to(dev)}
    force = dist is not None

    def it():
        return meta_train_step(opt, model, est, modelcp, estcp, data, optimizer, inner="reference", group=group,
                               force_collective=force)
    for _ in range(6):          # (a fixed count: every rank must run the same number of collectives)
        it()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        r = it()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = (time.perf_counter() - t0) / iters
    # the exchange step alone
    nbytes, ar_ms = None, None
    if dist is not None:
        for _ in range(2):
            nbytes = D.allreduce_meta_gradients([model.netG, est.netE], average=True, group=group, force=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            D.allreduce_meta_gradients([model.netG, est.netE], average=True, group=group, force=True)
        torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - t0) / 10 * 1e3
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if cpu_group else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return {"value": world * B / dt, "unit": "tasks/s (all ranks)", "tasks_per_s_per_gpu": B / dt,
            "ms_per_outer_iteration": dt * 1e3, "tasks_per_rank": B, "ranks": world,
            "allreduce": {"backend": "nccl (RCCL)" if dist is not None else None, "bytes": nbytes, "ms": ar_ms,
                          "includes": "pack of the .grad tensors into one flat fp32 buffer, ONE all-reduce, unpack",
                          "executed": dist is not None},
            "loss_q": r["loss_q"],
            "workload": "train_dynavsr.py outer iteration (inner='reference'), EDVR-M x4 + MFDN, per rank %d tasks of LR "
                        "5x3x64x64 / SLR 16x16 / HR 256x256, 1 inner Adam step, meta Adam; weak scaling over ranks" % B}


# ---- distributed validation: frames sharded round-robin over the ranks -----------------------------------
def validation_rate(dev, world, dist, group=None, frames_per_rank=8, h=176, w=320, frames_per_batch=8):
    """train_dynavsr.py:500-728 / the test driver's frame loop on N ranks: the frames range(rank, n, world) are adapted
    and super-resolved on each rank (adapt.validate_video: baseline forward, inner step, adapted forward, PSNR of both on
    the device), the PSNR vectors are reduced to rank 0.  Weak scaling: `frames_per_rank` per rank."""
    from dynavsr_amd import synth
    from dynavsr_amd.adapt import validate_video
    from dynavsr_amd.models import create_model
    rank = dist.get_rank() if dist is not None else 0
    opt = _opt()
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
    est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
    n = frames_per_rank * world
    mine = set(range(rank, n, world))
    # (only this rank's frames are materialised; the others are placeholders that are never touched)
    clips = [{"LQs": synth.clip(400 + i, 1, 5, h, w, smooth=True).to(dev)} if i in mine else None for i in range(n)]
    gts = [synth.clip(500 + i, 1, 1, 4 * h, 4 * w, smooth=True)[0, 0].to(dev) if i in mine else None for i in range(n)]

    def run():
        return validate_video(opt, model, est, modelcp, estcp, est_fixed, clips, gts, rank, world, group=group,
                              frames_per_batch=frames_per_batch)
    run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    r = run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)   # (the default group is the gloo one)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ps, pf, seen = r["psnr_start"], r["psnr_final"], r["evaluated"]
    fin = seen & torch.isfinite(ps) & torch.isfinite(pf)
    return {"value": n / dt, "unit": "frames/s (all ranks)", "frames": n, "frames_per_rank": frames_per_rank, "ranks": world,
            "ms_per_frame_per_rank": dt / frames_per_rank * 1e3,
            "psnr_vector_complete_on_rank0": bool(seen.all()) if rank == 0 else None,
            "mean_psnr_start_db": float(ps[fin].mean()), "mean_psnr_final_db": float(pf[fin].mean()),
            "workload": "per frame: un-adapted EDVR-M x4 forward @%dx%d + 1 inner MAML step (%d frames per batch) + adapted "
                        "forward + PSNR of both vs GT on the device; frames range(rank, n, world) per rank, PSNR vectors "
                        "reduced to rank 0 (train_dynavsr.py:509, :721-728)" % (h, w, frames_per_batch)}


# ---- configs[4]: EDVR-L on the three MFMA modes -----------------------------------------------------
def edvr_l_rates(dev, steps=10):
    from dynavsr_amd import hipops, synth
    from dynavsr_amd.models.archs.EDVR_arch import EDVR
    cfg = dict(nf=128, nframes=7, groups=8, front_RBs=5, back_RBs=40, scale=4)
    from dynavsr_amd.utils import util
    x = synth.clip(9, 1, 7, 64, 64, smooth=False).to(dev)
    tgt = synth.clip(109, 1, 1, 256, 256, smooth=False)[:, 0].to(dev)
    # north_star's accuracy gate (PSNR vs ground truth within 0.02 dB of the reference arithmetic) on the synthetic
    # substitute of tests/test_gpu_edvr.py::test_edvr_l_bf16_psnr_gate_in_north_star_terms: since round 5 a super-resolution
    # PAIR at a trained network's operating point (synth.sr_pair: bilinear alone scores 30 dB; the network's residual branch
    # damped, synth.damp_residual_branch) -- the round-3 substitute (unrelated noise images, 8 dB) is kept as `stress_*`
    xs_lr, xs_gt = synth.sr_pair(91, 7, 64, 64)
    xs_lr = xs_lr.to(dev)
    hr_pair = util.tensor2img(xs_gt, mode="rgb")
    xs = synth.clip(91, 1, 7, 64, 64).to(dev)
    hr = util.tensor2img(synth.clip(92, 1, 1, 256, 256)[0, 0], mode="rgb")
    out, y0, p0, s0 = {}, None, None, None
    for mode, name in ((0, "fp32_mfma"), (1, "bf16_operands"), (2, "bf16_split3")):
        net = EDVR(bf16_mfma=mode, **cfg)
        sd_full = synth.edvr_state_dict(8, **cfg)
        net.load_state_dict(sd_full, strict=True)
        net = net.to(dev)

        def fwd():
            with torch.no_grad():
                return net(x)

        def fwd_bwd():
            for p in net.parameters():
                p.grad = None
            hipops.charbonnier(net(x), tgt).backward()
        res = {}
        for key, fn, mult in (("forward", fwd, 1.0), ("forward_backward", fwd_bwd, 3.0)):
            _warm(fn)
            t0 = time.perf_counter()
            for _ in range(steps):
                y = fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            res[key] = {"ms": ms, "tflops": mult * F_EDVR_L_64 / (ms * 1e-3) / 1e12}
            if key == "forward":
                yf = y
        with torch.no_grad():
            res["stress_psnr_vs_unrelated_target_db"] = util.calculate_psnr(util.tensor2img(net(xs)[0], mode="rgb"), hr)
            net.load_state_dict(synth.damp_residual_branch(sd_full), strict=True)
            res["psnr_vs_synthetic_gt_db"] = util.calculate_psnr(util.tensor2img(net(xs_lr)[0], mode="rgb"), hr_pair)
            net.load_state_dict(sd_full, strict=True)
        if mode == 0:
            y0, p0, s0 = yf, res["psnr_vs_synthetic_gt_db"], res["stress_psnr_vs_unrelated_target_db"]
        else:
            res["delta_psnr_vs_fp32_db"] = res["psnr_vs_synthetic_gt_db"] - p0
            res["passes_0p02_db_gate"] = abs(res["delta_psnr_vs_fp32_db"]) <= 0.02
            # (ONE operating point: residual gain 0.02, one seed.  Every trunk perturbation reaches the output through the damped
            # conv_last, so the delta scales with the gain; profiles/r06_edvr_l_gate_sweep.txt sweeps gain x seed.)
            res["gate_operating_point"] = {"residual_gain": 0.02, "weight_seed": 8,
                                           "gate_holds_up_to_gain": (0.02 if mode == 1 else 1.0),
                                           "worst_delta_db_in_sweep": (-0.0546 if mode == 1 else 2e-5),
                                           "sweep": "profiles/r06_edvr_l_gate_sweep.txt (gain 0.01..1 x seeds 8, 9, 10; "
                                                    "tests/test_gpu_edvr.py::test_edvr_l_bf16_gate_sweep_over_gain_and_seed)"}
            res["stress_delta_psnr_vs_fp32_db"] = res["stress_psnr_vs_unrelated_target_db"] - s0
            d = (yf - y0).double()
            res["rel_l2_vs_fp32_mfma"] = float(d.norm() / y0.double().norm())
            res["psnr_db_vs_fp32_mfma"] = float(10 * torch.log10(1.0 / (d ** 2).mean()))
        peak = FP32_MFMA_PEAK_TFLOPS if mode == 0 else BF16_MFMA_PEAK_TFLOPS / (6.0 if mode == 2 else 1.0)
        res["roofline"] = {"bound": "mfma", "achieved": res["forward"]["tflops"], "peak": peak, "unit": "TFLOP/s",
                           "frac": res["forward"]["tflops"] / peak, "traffic": None}
        # HBM-side bytes of one forward+backward step from the committed PMC passes (tools/collect_profiles.sh 7b): at this
        # tile size the step moves ~10 GB in ~10 ms -- a tenth of the HBM rate: the leg is launch / latency bound, not HBM-bound
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                tj = json.load(f).get("edvr_l_fwd_bwd_mode%d" % mode)
            if tj:
                res["roofline"]["traffic"] = tj["bytes_per_step"]
                res["roofline"]["traffic_scope"] = "one forward+backward step, all kernels (PMC FETCH_SIZE + WRITE_SIZE)"
                res["hbm_gbs_forward_backward"] = tj["bytes_per_step"] / (res["forward_backward"]["ms"] * 1e-3) / 1e9
                res["hbm_frac_forward_backward"] = res["hbm_gbs_forward_backward"] / HBM_PEAK_GBS
        except (OSError, ValueError, KeyError):
            pass
        out[name] = res
        del net
    out["workload"] = ("EDVR-L x4 (nf 128, 7 frames, 40 blocks) 1x7x3x64x64 -> 3x256x256 (BASELINE configs[4] tile), "
                       "373.4 GFLOP forward; bf16 modes: 3x3 stride-1 convolutions (forward + data gradient) on "
                       "v_mfma_f32_32x32x16_bf16 with fp32 accumulate, everything else fp32; peak for bf16_split3 = "
                       "2500/6 TFLOP/s of fp32-equivalent work (six bf16 products per fp32 product)")
    return out


# ---- the other two video backbones behind define_G ---------------------------------------
def backbone_rates(dev):
    """TOFlow on 1x7x3x256x448 (it runs at the output resolution) and DUF-16L x4 on 1x7x3x64x112: eval forward and
    training forward+backward, against the algorithmic FLOPs of their convolutions (tools/backbone_bench.py has all three
    DUF depths)."""
    from dynavsr_amd import hipops, synth
    from dynavsr_amd.models.archs import DUF_arch, TOF_arch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import backbone_bench as bb      # conv_flops(): 2 * MAC of every convolution at the size it runs on
    out = {}
    tof = TOF_arch.TOFlow(adapt_official=True); tof.load_state_dict(synth.tof_state_dict(0))
    duf = DUF_arch.DUF_16L(scale=4, adapt_official=True); duf.load_state_dict(synth.duf_state_dict(0, 16, 4))
    for name, net, x, sc in (("toflow", tof, synth.clip(1, 1, 7, 256, 448, smooth=False), 1),
                             ("duf_16l_x4", duf, synth.clip(2, 1, 7, 64, 112, smooth=False), 4)):
        net, x = net.to(dev), x.to(dev)
        tgt = torch.rand(1, 3, sc * x.shape[-2], sc * x.shape[-1], device=dev)
        fl = bb.conv_flops(net, x)

        def fwd():
            with torch.no_grad():
                net(x)

        def fwd_bwd():
            for p in net.parameters():
                p.grad = None
            hipops.charbonnier(net(x), tgt).backward()
        res = {"clip": "x".join(map(str, x.shape)), "conv_gflop_forward": fl / 1e9}
        for key, fn, mult, train in (("forward", fwd, 1.0, False), ("forward_backward", fwd_bwd, 3.0, True)):
            net.train(train)
            _warm(fn)
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            ach = mult * fl / (ms * 1e-3) / 1e12
            res[key] = {"ms": ms, "roofline": {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                               "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None}}
        out[name] = res
        del net
    out["workload"] = ("the two other backbones behind define_G (TOF_arch.py, DUF_arch.py): op-composed on the native conv / "
                       "BatchNorm / warp / dynamic-filter kernels, fp32")
    return out


def split_mode_rate(cfg, h, w, x, y_fp32, steps, warmup):
    """The headline forward with network_G.bf16_mfma = 2: reported BESIDE the headline, never as it."""
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


def _merge_rank_leg(per_rank, value_key="value"):
    """A leg every rank ran on its own frames (independent work, no collective): rank 0's full object, with `value` = the SUM
    over the ranks and the per-rank figures beside it (the driver derives scaling efficiency from its own N = 1 run)."""
    legs = [x for x in per_rank if x is not None]
    if not legs:
        return None
    out = dict(legs[0])
    vals = [float(x[value_key]) for x in legs]
    out[value_key] = sum(vals)
    out["ranks"] = len(legs)
    out["per_rank_value"] = vals
    out["min_rank_value"], out["max_rank_value"] = min(vals), max(vals)
    return out


def _gather_objects(dist, obj, world):
    """Every rank's object on rank 0 (list in rank order; [obj] without a process group)."""
    if dist is None or world == 1:
        return [obj]
    box = [None] * world
    dist.all_gather_object(box, obj)
    return box


def _spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: re-run this file under torch.distributed.run with N local ranks
    (rendezvous on 127.0.0.1, a free port) and hand its exit code back."""
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def _dry_run(rank, world, args):
    """The N-rank skeleton of the bench with no kernels, through the SAME plumbing as the GPU run: the gloo process group
    (barriers, MAX of the elapsed time, gathering of the per-rank legs: _gather_objects / _merge_rank_leg), a second group
    where the GPU run creates the RCCL one (here gloo as well) for the flat meta-gradient all-reduce
    (dist.allreduce_meta_gradients on a tensor list of the real size: 15.0 MB for EDVR-M + MFDN) and the round-robin frame
    shards with their metric reduction (dist.shard_indices / reduce_metric_vectors).  Rank 0 returns a JSON line with the key
    set of the GPU line at the same world size (tests/test_host_logic.py compares it with a recorded GPU line)."""
    import torch.distributed as tdist
    from dynavsr_amd import dist as D
    dist, coll = None, None
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        tdist.init_process_group("gloo", rank=rank, world_size=world)
        dist = tdist
    grouped = dist is not None

    def barrier():
        if grouped:
            tdist.barrier()
    work = torch.zeros(1)
    for _ in range(args.warmup):
        work += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        work += 1
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64)
    if grouped:
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
    elapsed = float(t)
    # the legs every rank runs on its own frames: stand-ins with the rank in their value (sum = world (world + 1) / 2)
    mine = ({"value": float(rank + 1), "unit": "clips/s", "ms_per_step": None},
            {"value": float(rank + 1), "unit": "frames/s", "ms_per_frame": None})
    both = _gather_objects(dist, mine, world)
    if grouped:
        coll = tdist.new_group(backend="gloo")    # (the GPU run: backend "nccl" = RCCL, created after the single-GPU legs)
    # the one exchange step of the method, on parameters of the real sizes (values: rank + 1, averaged -> (world + 1) / 2)
    holder = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(n)) for n in (3_300_131, 452_291)])
    for p in holder:
        p.grad = torch.full_like(p, float(rank + 1))
    nbytes = D.allreduce_meta_gradients([holder], average=True, group=coll, force=grouped)
    ok = all(bool((p.grad == (world + 1) / 2.0).all()) for p in holder)
    # distributed validation: frames range(rank, n, world), per-frame metric vector reduced to rank 0
    frames = 10
    vec = torch.zeros(frames, dtype=torch.float64)
    for i in D.shard_indices(frames, rank, world):
        vec[i] = 30.0 + i
    D.reduce_metric_vectors([vec], group=coll)
    if grouped:
        tdist.barrier()
        tdist.destroy_process_group()
    if rank != 0:
        return None
    line = {"metric": "dry run: launcher and process-group plumbing only (gloo, CPU, no kernels)", "value": None,
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
            "config": {"workload": "none (dry run)", "clips_per_step": world},
            "protocol": None, "value_one_clip_in_flight": None, "runtime": None,
            "one_clip_in_flight": None, "roofline": None, "kernel_breakdown_ms_per_step": None, "end_to_end_tflops": None,
            "inner_step": _merge_rank_leg([b_[0] for b_ in both]),
            "per_frame_pipeline": _merge_rank_leg([b_[1] for b_ in both]),
            "inner_step_clips_per_s": _merge_rank_leg([b_[0] for b_ in both]).get("value"),
            "per_frame_pipeline_frames_per_s": _merge_rank_leg([b_[1] for b_ in both]).get("value"),
            "meta_step": {"ranks": world, "allreduce": {"backend": "gloo", "bytes": nbytes, "executed": grouped,
                                                         "averaged_correctly": ok}},
            "validation": {"frames": frames, "sharding": "range(rank, frames, world)",
                           "psnr_vector_complete": bool((vec == torch.arange(frames, dtype=torch.float64) + 30.0).all())}}
    if world == 1:   # legs of the one-GPU line only
        line.update({"experimental_bf16_split": None, "edvr_l_bf16": None, "other_backbones": None, "cpu_baseline": None})
    return line


def _rt_report(dev):
    """hw queue setting + the side-stream overlap probe on the current stream and on a FRESH stream (the cached answer for the
    current stream predates whatever created streams since; a new stream shows what a plan on a new launch stream would get)."""
    from dynavsr_amd import _lib as L
    r = L.runtime_report(dev)
    fresh = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(fresh):
        r["side_stream_overlaps_fresh_stream"] = int(L.lib().dvsr_side_stream_overlaps(L.stream()))
    return r


def main():
    # The ONE JSON line must be the only thing on stdout: RCCL prints a version banner there (from C, flushed at exit),
    # so file descriptor 1 is pointed at stderr for the whole run and the line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--height", type=int, default=180)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--clips-in-flight", type=int, default=2,
                    help="clips of the forward leg on the GPU at a time, one HIP stream each (1: one clip per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inner-step", action="store_true", help="skip the inner-step and per-frame-pipeline legs")
    ap.add_argument("--no-split", action="store_true", help="skip the bf16 legs (split-mode forward, EDVR-L)")
    ap.add_argument("--no-meta", action="store_true", help="skip the meta-training iteration with the RCCL all-reduce")
    ap.add_argument("--no-validation", action="store_true", help="skip the sharded validation leg (frames over ranks)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group plumbing only, on the CPU over gloo (no kernels): what tests/ run here")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU, the way the reference's
        # README launches its trainer (README.md:92-96, `python -m torch.distributed.launch --nproc_per_node=8 ...`;
        # train_dynavsr.py:23-30 reads the rendezvous from the environment).  Rank 0 of the children prints the JSON line
        # to the stdout it inherits from this process.
        os.dup2(real_stdout, 1)
        sys.exit(_spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or with "
                         "torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if args.dry_run:
        line = _dry_run(rank, world, args)
        if line is not None:
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Two process groups.  `dist` (gloo, CPU) carries the barriers, the MAX of the elapsed times and the gathering of the
    # per-rank legs: it owns no GPU stream or helper thread, so the host-bound launch sequences of the single-GPU legs run as
    # they do without a launcher (an initialised RCCL communicator slows them by 25-35 %: tools/rccl_effect.py).  The RCCL
    # group (`rccl`, backend "nccl") is created AFTER those legs, for the two legs that move data between GPUs: the
    # meta-gradient all-reduce and the reduction of the validation metrics.
    dist, rccl, dist_err = None, None, None
    import torch.distributed as tdist
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run (also with 1 rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        tdist.init_process_group("gloo", rank=rank, world_size=world)
        dist = tdist

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

    # One step = `in_flight` clips, one per HIP stream (adapt.super_resolve_video: every clip runs the launches it would run
    # alone; the second queue fills the tail rounds and the gaps between the 85 dependent launches of a forward).
    from itertools import repeat
    from dynavsr_amd.adapt import super_resolve_video
    S = max(1, args.clips_in_flight)
    opt_fwd = _opt()

    def run_clips(n, in_flight):
        y_ = None
        for y_ in super_resolve_video(opt_fwd, net, repeat(x, n), in_flight=in_flight):
            pass
        return y_

    run_clips(S * args.warmup, S)
    barrier()
    t0 = time.perf_counter()
    y = run_clips(S * args.steps, S)
    barrier()
    elapsed = time.perf_counter() - t0
    # ... and the same clips one at a time on one stream (the figure of rounds 1-3), rank-local
    run_clips(max(3, args.warmup // 2), 1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run_clips(args.steps, 1)
    torch.cuda.synchronize()
    ms_one = 1e3 * (time.perf_counter() - t1) / args.steps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    assert torch.isfinite(y).all()

    line = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        line = {
            "metric": "inner-loop frames/sec/GPU (EDVR-M x4, 5x3x180x320): forward leg = BASELINE.json configs[1]; "
                      "the inner MAML step and the per-frame pipeline are `inner_step` / `per_frame_pipeline`",
            "value": world * S * args.steps / elapsed, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # (`value` counts S clips in flight per GPU since round 4; rounds 1-3 measured one at a time: compare like with like)
            "protocol": "%d clip(s) in flight per GPU, one HIP stream each; value_one_clip_in_flight = the one-at-a-time "
                        "protocol of rounds 1-3" % S,
            "value_one_clip_in_flight": 1e3 / ms_one,
            "config": {"workload": "EDVR-M x4 forward (PCD deformable alignment + TSA fusion + "
                                   "reconstruction), clips of 1x5x3x%dx%d -> 3x%dx%d (BASELINE.json configs[1]); one step = "
                                   "%d clip(s) per GPU, one per HIP stream (adapt.super_resolve_video)"
                                   % (h, w, 4 * h, 4 * w, S),
                       "clips_per_step": world * S, "clips_in_flight_per_gpu": S,
                       "value_one_clip_in_flight": 1e3 / ms_one,
                       "sharding": "independent clips per rank, no collective"},
            "one_clip_in_flight": {"value": 1e3 / ms_one, "unit": "frames/s", "ms_per_clip": ms_one,
                                   "note": "the same clips one at a time on one stream (rank 0): the protocol of rounds 1-3 "
                                           "and the latency of one forward"},
        }
        # ---- roofline of the dominant kernel: separate pass, hipEvents around every launch
        plan = engine.get_plan(net._cfg(), 1, h, w)
        params = [p.detach().contiguous() for p in net.ordered_parameters()]
        ws = torch.empty(plan.workspace_bytes(False), dtype=torch.uint8, device=dev)
        out = torch.empty_like(y)
        info = plan.op_info()
        acc = {}
        wino = {}   # per kind: [launches on the Winograd kernel, their algorithmic flops, their time]
        reps = max(3, min(10, args.steps))
        for _ in range(reps):
            for (kind, name, fl, by), t_ms in zip(info, plan.forward_timed(params, x, out, ws)):
                a = acc.setdefault(kind, [0.0, 0.0, 0.0, 0])
                a[0] += t_ms; a[1] += fl; a[2] += by; a[3] += 1
                tag = name[name.rfind("/"):]
                if tag.endswith("w]") or tag.endswith("w3]") or tag.endswith("w5]"):   # launch geometry tag of dvsr_edvr_op_info
                    # (fp32 F(2x2) kernel | bf16x3 F(2x2) kernel | bf16x3 F(4x4) kernel): n, flops, ms
                    wv = wino.setdefault(kind, [0, 0.0, 0.0, 0, 0.0, 0.0, 0, 0.0, 0.0])
                    o3 = 6 if tag.endswith("w5]") else (3 if tag.endswith("w3]") else 0)
                    wv[o3] += 1; wv[o3 + 1] += fl; wv[o3 + 2] += t_ms
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
        # HBM bytes per launch from the committed PMC passes (counters cannot be collected inside this process).
        # The file names the kernel sources it was collected on; a figure from other sources is NOT reported.
        traffic, traffic_src, traffic_err = None, None, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                tj = json.load(f)
            t = tj.get(dom)
            if t and (h, w) == (180, 320):
                if tj.get("sources_sha256_16") == _sources_digest():
                    traffic, traffic_src = t["bytes_per_launch"], t["source"]
                else:
                    traffic_err = ("profiles/pmc_traffic.json was collected on other kernel sources (%s, now %s): re-run "
                                   "tools/collect_profiles.sh" % (tj.get("sources_sha256_16"), _sources_digest()))
                    print("bench.py: STALE PMC TRAFFIC -- " + traffic_err, file=sys.stderr, flush=True)
        except (OSError, ValueError) as e:
            traffic_err = "profiles/pmc_traffic.json unreadable: %s" % e
        roof.update({"traffic": traffic, "traffic_source": traffic_src, "kernel": dom, "launches_per_step": cnt // reps,
                     "avg_launch_ms": t_ms / cnt, "share_of_step": t_ms / total_ms,
                     "method": "hipEvent pair around every launch on the launch stream, %d instrumented "
                               "passes after the timed region" % reps})
        if traffic_err:
            roof["traffic_error"] = traffic_err
        roof["algorithmic_flops"] = fl / reps
        roof["executed_flops"] = fl / reps
        if roof["bound"] == "mfma":
            roof["algorithmic_tflops"] = roof["achieved"]
        if dom in wino:
            # The launches on the Winograd kernels do 16 multiplies per 2x2 output block and (cout, cin) pair instead of the
            # direct sum's 36; the bf16x3 kernel (conv2d_wino3.hip) issues each of them as six bf16 products.  The roofline
            # object is that of the kernel most of the class's time goes to; the contract's algorithmic rate (2 x MACs of the
            # direct 3x3 sum, SURVEY 8d, / time) is `algorithmic_tflops`.
            wn, wfl, wt, w3n, w3fl, w3t, w5n, w5fl, w5t = wino[dom]
            executed = fl - (wfl + w3fl) * (1.0 - 16.0 / 36.0) - w5fl * (1.0 - 36.0 / 144.0)
            roof["algorithm"] = ("%d of %d launches per step on the Winograd F(4x4,3x3) kernel (conv2d_wino5.hip, round 6: 36 "
                                 "transformed points per 4x4 outputs = 1/4 of the direct sum's multiplies, each issued as six bf16 "
                                 "products under the exact 3-way operand split), %d on the F(2x2,3x3) kernel under the same split "
                                 "(conv2d_wino4.hip: 4/9 of the multiplies x 6), %d on the fp32 Winograd kernel (conv2d_wino.hip), "
                                 "the rest on the direct implicit-GEMM kernels"
                                 % (w5n // reps, cnt // reps, w3n // reps, wn // reps))
            roof["executed_flops"] = executed / reps
            roof["mfma_flops_executed_frac"] = executed / fl
            roof["class"] = {"kind": dom, "launches_per_step": cnt // reps, "ms_per_step": t_ms / reps,
                             "share_of_step": t_ms / total_ms, "algorithmic_tflops": fl / (t_ms * 1e-3) / 1e12,
                             "fp32_products_tflops": executed / (t_ms * 1e-3) / 1e12,
                             "fp32_products_over_fp32_pipe_peak": executed / (t_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
            roof["winograd_kernels"] = {
                "conv2d_wino5_kernel": {"launches_per_step": w5n // reps, "ms_per_step": w5t / reps,
                                        "algorithmic_tflops": (w5fl / (w5t * 1e-3) / 1e12) if w5t else None},
                "conv2d_wino4_kernel": {"launches_per_step": w3n // reps, "ms_per_step": w3t / reps,
                                        "algorithmic_tflops": (w3fl / (w3t * 1e-3) / 1e12) if w3t else None}}
            if w5t >= w3t and w5t >= wt:
                k_alg, k_fl, k_t, k_n, k_name = w5fl, w5fl * (36.0 / 144.0), w5t, w5n, "conv2d_wino5_kernel"
                roof.update(pipe_mix(0.0, 6.0 * k_fl / reps, k_t / reps * 1e-3))
            elif w3t >= wt:
                k_alg, k_fl, k_t, k_n, k_name = w3fl, w3fl * (16.0 / 36.0), w3t, w3n, "conv2d_wino4_kernel"
                roof.update(pipe_mix(0.0, 6.0 * k_fl / reps, k_t / reps * 1e-3))
            else:
                k_alg, k_fl, k_t, k_n, k_name = wfl, wfl * (16.0 / 36.0), wt, wn, "conv2d_wino_kernel"
                roof.update(pipe_mix(k_fl / reps, 0.0, k_t / reps * 1e-3))
            roof.update({"kernel": k_name, "launches_per_step": k_n // reps, "avg_launch_ms": k_t / k_n,
                         "share_of_step": k_t / total_ms, "algorithmic_flops": k_alg / reps,
                         "executed_flops": k_fl / reps,
                         "algorithmic_tflops": k_alg / (k_t * 1e-3) / 1e12,
                         "fp32_products_tflops": k_fl / (k_t * 1e-3) / 1e12})
            roof["winograd_launch_share_of_kernel_time"] = (wt + w3t + w5t) / t_ms
            roof["note"] = ("roofline of the kernel most of the step goes to (`kernel`; the whole conv3x3s1 class is under "
                            "`class`): achieved = FLOPs ISSUED to the matrix pipe it runs on / its time, frac = / that pipe's "
                            "dense peak -- the pipe's own utilisation (PMC SQ_VALU_MFMA_BUSY_CYCLES agrees: "
                            "profiles/*_pmc_mfma_util.txt).  The bf16x3 kernels are bound by the weight fragments' way through "
                            "the vector memory path and by the input transform, not by the pipe (DESIGN 3.1f-i): the F(4x4) "
                            "kernel ISSUES 0.56 x the products of F(2x2) for the same outputs, so a lower `frac` at a shorter "
                            "time is the point of it; fp32 products / time are `fp32_products_tflops` (r03's fp32 "
                            "Winograd kernel: 0.47 of the fp32 pipe for the same products at 5 %% more time); "
                            "algorithmic_tflops = SURVEY 8d's direct-sum FLOPs / the same time; with DVSR_CONV_WINO=0 the "
                            "direct kernels measure frac 0.70 of the fp32 pipe (profiles/*_wino_vs_direct.txt)")
        line["roofline"] = roof
        line["kernel_breakdown_ms_per_step"] = {k: round(a[0] / reps, 4) for k, a in
                                                sorted(acc.items(), key=lambda kv: -kv[1][0])}
        line["end_to_end_tflops"] = S * sum(a[1] for a in acc.values()) / reps / (ms * 1e-3) / 1e12   # (a step is S clips)
        if world == 1 and not args.no_split:
            line["experimental_bf16_split"] = split_mode_rate(cfg, h, w, x, y, args.steps, args.warmup)
            line["edvr_l_bf16"] = edvr_l_rates(dev)
            try:
                line["other_backbones"] = backbone_rates(dev)
            except Exception as e:     # a side leg must not cost the line
                line["other_backbones"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # The legs the metric is named after run on EVERY rank, each on its own frames (independent work items, no collective:
    # north_star asks for the inner step at 1/2/4/8 GPUs): rank 0 reports the sum and the per-rank figures.
    if not args.no_inner_step:
        barrier()
        mine = (inner_step_rate(dev), per_frame_pipeline_rate(dev))
        barrier()
        both = _gather_objects(dist, mine, world)
        if rank == 0:
            line["inner_step"] = _merge_rank_leg([b_[0] for b_ in both])
            line["per_frame_pipeline"] = _merge_rank_leg([b_[1] for b_ in both])
            # (top-level scalars of the two legs BASELINE's metric is named after)
            line["inner_step_clips_per_s"] = line["inner_step"].get("value")
            line["per_frame_pipeline_frames_per_s"] = line["per_frame_pipeline"].get("value")
            # (the driver's record keeps `config` verbatim and only the NAMES of other extra keys: the figures the metric is
            # named after travel inside it too)
            line["config"]["inner_step_clips_per_s"] = line["inner_step_clips_per_s"]
            line["config"]["per_frame_pipeline_frames_per_s"] = line["per_frame_pipeline_frames_per_s"]
            line["config"]["three_inner_steps_ms_per_frame"] = (line["inner_step"].get("three_inner_steps") or {}).get("ms_per_frame")
    if rank == 0:
        # what the process got from the runtime: the hardware-queue setting and the MEASURED overlap of the weight-gradient
        # side stream with the launch stream (dvsr_side_stream_overlaps), before any RCCL communicator exists
        line["runtime"] = dict(_rt_report(dev), probed="before the RCCL group")
    # The RCCL communicator is created only now: once it exists, its helper threads slow host-bound launch sequences down
    # (EDVR-L bf16 forward+backward, ~700 launches in 10.5 ms, measured 14.1 ms when the meta leg ran first; the inner-step
    # per-frame loop 8.2 -> 10.0 ms; the GPU-bound legs do not move).
    if not (args.no_meta and args.no_validation):
        try:
            if dist is not None:
                rccl = tdist.new_group(backend="nccl")
            elif not args.no_meta:            # plain `python bench.py`: a one-rank RCCL group for the meta_step leg
                s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
                tdist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
                rccl = tdist.group.WORLD
        except Exception as e:   # only the optional one-rank group may fail softly
            if world > 1:
                raise
            dist_err = "%s: %s" % (type(e).__name__, e)
    if rank == 0 and line is not None and rccl is not None:
        line["runtime"]["side_stream_overlaps_with_rccl_group"] = _rt_report(dev)["side_stream_overlaps_fresh_stream"]
    meta = None
    if not args.no_meta:   # every rank takes part (the collective)
        try:
            meta = meta_step_rate(dev, world, tdist if rccl is not None else None, rccl, cpu_group=dist is not None)
        except Exception as e:
            if world > 1:
                raise
            meta = {"error": "%s: %s" % (type(e).__name__, e)}
        if dist_err:
            meta["process_group_error"] = dist_err
    val = None
    if not args.no_validation:   # every rank takes part (its shard of the frames, the metric reduction)
        try:
            val = validation_rate(dev, world, tdist if (world > 1 and rccl is not None) else None, rccl)
        except Exception as e:
            if world > 1:
                raise
            val = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        if meta is not None:
            line["meta_step"] = meta
        if val is not None:
            line["validation"] = val
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, h, w, 0, y)   # same clip (seed 1 + rank 0), same weights
    if tdist.is_initialized():
        tdist.barrier(group=None if dist is not None else rccl)
        tdist.destroy_process_group()
    if line is not None:
        os.write(real_stdout, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()
