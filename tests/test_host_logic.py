"""CPU: host-side logic around the hot path -- options YAML surface, model factory / wrapper
contract, checkpoint compatibility, MFDN restatement vs golden, and the data-parallel helpers
(world_size-2 gloo)."""
import os
import socket
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, relerr
from dynavsr_amd import synth
from dynavsr_amd.options import options as option

YAML = os.path.join(ROOT, "dynavsr_amd", "options", "test", "EDVR", "EDVR_M_S4.yml")


def cpu_opt(is_train=False):
    opt = option.dict_to_nonedict(option.parse(YAML, is_train=is_train))
    opt["gpu_ids"] = None
    opt["dist"] = False
    for k in ("pretrain_model_G", "pretrain_model_E"):
        opt["path"][k] = None
    return opt


def test_options_surface():
    opt = option.parse(YAML, is_train=False)
    assert isinstance(opt, OrderedDict) and opt["is_train"] is False
    assert opt["datasets"]["val"]["phase"] == "val" and opt["datasets"]["val"]["scale"] == 4
    assert opt["datasets"]["val"]["data_type"] == "img"
    assert opt["network_G"]["scale"] == 4 and "results_root" in opt["path"]
    n = option.dict_to_nonedict(opt)
    assert n["missing"] is None and n["train"]["maml"]["missing"] is None
    assert n["train"]["maml"]["adapt_iter"] == 1 and n["train"]["pixel_criterion"] == "cb"
    t = option.parse(YAML, is_train=True, exp_name="unit_debug")
    assert t["name"] == "unit_debug" and t["train"]["val_freq"] == 8 and t["path"]["models"].endswith("models")
    assert "network_G" in option.dict2str(opt)


def test_create_model_contract(tmp_path):
    from dynavsr_amd.models import create_model
    opt = cpu_opt()
    model, est = create_model(opt)
    assert type(model).__name__ == "VideoBaseModel" and type(est).__name__ == "LRimgestimator_Model"
    for attr in ("netG", "device", "log_dict", "optimizers", "schedulers", "feed_data", "calculate_loss",
                 "optimize_parameters", "optimize_by_loss", "test", "get_current_visuals", "get_current_log",
                 "load_network", "save", "save_training_state", "resume_training", "update_learning_rate"):
        assert hasattr(model, attr), attr
    for attr in ("netE", "feed_data", "forward_without_optim", "test", "MyLoss", "save", "load_network"):
        assert hasattr(est, attr), attr
    # parameters are ordinary leaves that external optimizers / deepcopy / .grad writes can use
    from copy import deepcopy
    cp = deepcopy(model.netG)
    p = next(cp.parameters())
    assert p.is_leaf and p.requires_grad
    p.grad = torch.zeros_like(p)
    p.grad += 1
    torch.optim.SGD(cp.parameters(), lr=1.0).step()
    assert not torch.equal(p, next(model.netG.parameters()))
    # checkpoints: reference-style files, optional 'module.' prefix
    sd = synth.edvr_state_dict(0)
    f = tmp_path / "G.pth"
    torch.save(OrderedDict(("module." + k, v) for k, v in sd.items()), f)
    model.load_network(str(f), model.netG, strict=True)
    assert torch.equal(model.netG.state_dict()["conv_last.bias"], sd["conv_last.bias"])
    opt["path"]["models"] = str(tmp_path)
    model.save("7")
    back = torch.load(tmp_path / "7_G.pth")
    assert list(back.keys()) == list(sd.keys())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.feed_data({"LQs": torch.zeros(1, 5, 3, 16, 16)}, need_GT=False)
        model.test()
    with pytest.raises(NotImplementedError):
        bad = cpu_opt()
        bad["model"] = "sr"
        create_model(bad)


def test_train_mode_builds_optimizers():
    from dynavsr_amd.models import create_model
    opt = cpu_opt(is_train=True)
    model, est = create_model(opt)
    assert len(model.optimizers) == 1 and len(model.schedulers) == 1 and len(est.optimizers) == 1
    assert model.get_current_learning_rate() == [1e-5]


def test_estimator_modules_state_dict_and_no_cpu_path():
    """MFDN / SFDN keep the reference's parameter names and shapes (reference *_E.pth load strict) and,
    like the EDVR backbone, refuse to run without the MI355X: there is no CPU fallback to hide behind."""
    from dynavsr_amd.models.archs.LRimg_estimator import DirectKernelEstimator_CMS, DirectKernelEstimatorVideo
    for scale in (2, 4):
        net = DirectKernelEstimatorVideo(nf=64, in_nc=3, scale=scale)
        net.load_state_dict(synth.mfdn_state_dict(0, scale=scale), strict=True)
        assert [tuple(p.shape) for p in net.ordered_parameters()] == \
            [tuple(v.shape) for v in synth.mfdn_state_dict(0, scale=scale).values()]
    assert sum(p.numel() for p in DirectKernelEstimatorVideo(64, 3, 4).parameters()) == 452291
    sf = DirectKernelEstimator_CMS(nf=16)
    sf.load_state_dict(synth.sfdn_state_dict(0, nf=16), strict=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(synth.clip(1, 1, 5, 16, 16).transpose(1, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sf(synth.clip(1, 1, 1, 16, 16)[:, 0])


def test_util_metrics_golden():
    from dynavsr_amd.utils import util
    g = load_golden("psnr")
    gt = synth.clip(int(g["gtseed"]), 1, 1, 256, 256)[0, 0]
    assert abs(util.calculate_psnr(g["img"], util.tensor2img(gt, mode="rgb")) - float(g["psnr"])) < 1e-12


def test_dcn_dropin_rejects_cpu():
    from dynavsr_amd.models.archs.dcn import ModulatedDeformConvPack
    m = ModulatedDeformConvPack(8, 8, 3, stride=1, padding=1, dilation=1, deformable_groups=2,
                                extra_offset_mask=True)
    assert sorted(k for k, _ in m.named_parameters()) == ["bias", "conv_offset_mask.bias",
                                                          "conv_offset_mask.weight", "weight"]
    assert float(m.conv_offset_mask.weight.abs().sum()) == 0.0
    with pytest.raises((NotImplementedError, RuntimeError)):
        m([torch.zeros(1, 8, 4, 4), torch.zeros(1, 8, 4, 4)])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    from dynavsr_amd import dist as ddist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddist.init_dist(backend="gloo")
    torch.manual_seed(0)
    netG, netE = torch.nn.Linear(6, 3), torch.nn.Linear(3, 2)       # same init on every rank
    # every rank owns a disjoint shard of 5 "clips"; grads accumulate over the local shard
    clips = [torch.full((1, 6), float(i + 1)) for i in range(5)]
    for i in ddist.shard_indices(len(clips), rank, world):
        netE(netG(clips[i])).sum().backward()
    netE.bias.grad = None                                           # a param without grad on this rank
    nbytes = ddist.allreduce_meta_gradients([netG, netE], average=False)
    vec = [torch.tensor([float(rank + 1), 1.0])]
    ddist.reduce_metric_vectors(vec, dst=0)
    if rank == 0:
        out.put((nbytes, netG.weight.grad.clone(), netE.bias.grad.clone(), vec[0].clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_meta_gradient_allreduce_gloo():
    """world_size 2 on CPU: sharded clips + one flat all-reduce == single-process gradient."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    nbytes, gw, gb, vec = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    netG, netE = torch.nn.Linear(6, 3), torch.nn.Linear(3, 2)
    for i in range(5):
        netE(netG(torch.full((1, 6), float(i + 1)))).sum().backward()
    assert nbytes == 4 * (6 * 3 + 3 + 3 * 2 + 2)
    assert torch.allclose(gw, netG.weight.grad, atol=1e-5)
    assert torch.allclose(gb, torch.zeros(2))      # both ranks dropped it -> zeros, not garbage
    assert vec.tolist() == [3.0, 2.0]


def test_shard_indices_cover_everything():
    from dynavsr_amd.dist import shard_indices
    for n in (0, 1, 7, 34):
        for world in (1, 2, 8):
            got = sorted(i for r in range(world) for i in shard_indices(n, r, world))
            assert got == list(range(n))


class _StubG:
    """The slice of VideoBaseModel that adapt.meta_train_step touches, over a CPU toy network."""
    def __init__(self):
        torch.manual_seed(1)
        self.netG = torch.nn.Sequential(torch.nn.Flatten(2), torch.nn.Linear(16, 16))

    def feed_data(self, data, need_GT=True):
        self.var_L, self.real_H = data["LQs"], data.get("GT")

    def calculate_loss(self):
        self.fake_H = self.netG(self.var_L).mean(1)
        return ((self.fake_H - self.real_H.flatten(1)[:, :16]) ** 2 + 1e-6).sqrt().mean()


class _StubE:
    """... and of LRimgestimator_Model."""
    def __init__(self):
        torch.manual_seed(2)
        self.netE = torch.nn.Linear(16, 16)
        self.MyLoss = torch.nn.L1Loss()

    def feed_data(self, data):
        self.real_H, self.real_L = data["LQs"], data.get("SuperLQs")

    def forward_without_optim(self):
        b, n = self.real_H.shape[:2]
        self.fake_L = self.netE(self.real_H.flatten(2)).reshape(b, n, 1, 4, 4)


def _meta_opt():
    from dynavsr_amd.options.options import dict_to_nonedict
    return dict_to_nonedict({"train": {"use_real": False, "maml": {"adapt_iter": 2, "optimizer": "SGD", "lr_alpha": 1e-2,
                                                                      "use_patch": False}}})


def _meta_data():
    g = torch.Generator().manual_seed(7)
    return {"LQs": torch.rand(2, 3, 1, 4, 4, generator=g), "SuperLQs": torch.rand(2, 3, 1, 4, 4, generator=g),
            "GT": torch.rand(2, 3, 1, 4, 4, generator=g)}


def _meta_worker(rank, world, port, out):
    import torch.distributed as dist
    from dynavsr_amd import dist as ddist
    from dynavsr_amd.adapt import meta_train_step
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddist.init_dist(backend="gloo")
    model, est, modelcp, estcp = _StubG(), _StubE(), _StubG(), _StubE()
    params = list(model.netG.parameters()) + list(est.netE.parameters())
    data = _meta_data()
    shard = {k: v[rank:rank + 1] for k, v in data.items()}          # one task per rank (range(rank, B, world))
    meta_train_step(_meta_opt(), model, est, modelcp, estcp, shard, torch.optim.SGD(params, lr=0.5))
    if rank == 0:
        out.put([p.detach().numpy().copy() for p in params])    # plain arrays: nothing to share after exit
    dist.barrier()
    dist.destroy_process_group()


def test_meta_train_step_data_parallel_gloo():
    """world_size 2 on CPU (stub wrappers with the surface meta_train_step touches): every rank runs its shard of the
    tasks, ONE all-reduce averages the accumulated meta-gradients, the meta step then moves every rank's
    parameters by the mean of the per-rank gradients."""
    from dynavsr_amd.adapt import meta_train_step
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_meta_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    grads = []
    for r in range(world):                                           # the same shards, one process, no collective
        model, est, modelcp, estcp = _StubG(), _StubE(), _StubG(), _StubE()
        params = list(model.netG.parameters()) + list(est.netE.parameters())
        shard = {k: v[r:r + 1] for k, v in _meta_data().items()}
        meta_train_step(_meta_opt(), model, est, modelcp, estcp, shard, torch.optim.SGD(params, lr=0.0))
        grads.append([p.grad.clone() for p in params])
        start = [p.detach().clone() for p in params]
    for p_new, p0, g0, g1 in zip(got, start, grads[0], grads[1]):
        assert torch.allclose(torch.from_numpy(p_new), p0 - 0.5 * 0.5 * (g0 + g1), atol=1e-6)


def test_use_patch_positions_follow_the_reference_draw():
    """train.maml.use_patch: the patch positions come out of python's `random` in common_crop's order (py, then px,
    per patch; preprocessing.py:76-77) -- against the positions the reference's common_crop drew for the golden, and the
    oracle's crop against the stored crops."""
    import random
    from dynavsr_amd.adapt import draw_patch_positions
    from oracle import inner as oin
    g = load_golden("crop_patches")
    t, c, h, w, s, n, psz = (int(g[k]) for k in ("t", "c", "h", "w", "scale", "n", "patch_size"))
    random.seed(int(g["seed"]))
    py, px = draw_patch_positions(h, w, psz // 2, n)
    assert py == list(g["py"]) and px == list(g["px"])
    r = np.random.RandomState(int(g["dseed"]))
    seq = torch.from_numpy(r.rand(1, t, c, h, w).astype(np.float32))
    hr = torch.from_numpy(r.rand(1, c, s * h, s * w).astype(np.float32))
    random.seed(int(g["seed"]))
    lr_p, hr_p = oin.crop(seq, hr, n, psz)
    assert torch.equal(lr_p, torch.from_numpy(g["lr"])) and tuple(hr_p.shape) == (n, c, s * psz // 2, s * psz // 2)
    assert np.allclose(hr_p.double().sum(dim=(1, 2, 3)).numpy(), g["hr_sum"], rtol=0, atol=1e-9)


def test_log_dict_reads_as_floats_without_an_eager_sync():
    """log_dict['l_pix'] is what the reference's `l_pix.item()` gives, but the conversion happens at read time."""
    from dynavsr_amd.models.base_model import LogDict
    d = LogDict()
    t = torch.tensor(0.25)
    d['l_pix'] = t
    assert isinstance(OrderedDict.__getitem__(d, 'l_pix'), torch.Tensor)       # stored as it is: no .item() on write
    assert d['l_pix'] == 0.25 and isinstance(d['l_pix'], float)
    d['l_pix'] = torch.tensor(0.5)
    d['lr'] = 1e-4
    assert d.items() == [('l_pix', 0.5), ('lr', 1e-4)] and d.values() == [0.5, 1e-4] and d.get('none', 3) == 3
    assert all(isinstance(v, float) for _, v in d.items())
    from dynavsr_amd.models import create_model
    model, est = create_model(cpu_opt())
    assert isinstance(model.log_dict, LogDict) and model.get_current_log() is model.log_dict
    # the drivers add the SLR term to the returned loss IN PLACE (`loss_train += 10 * F.l1_loss(...)`, test_dynavsr.py:264-274):
    # the logged value must stay the pixel loss the reference's eager .item() captured
    model._pixel_loss = lambda: torch.tensor(2.0, requires_grad=True) * 1.0
    loss = model.calculate_loss()
    with torch.no_grad():
        loss += 5.0
    assert float(loss) == 7.0 and model.log_dict['l_pix'] == 2.0


def test_hw_queue_default_is_explicit_and_respects_the_user(monkeypatch):
    """Importing the package leaves GPU_MAX_HW_QUEUES alone; dynavsr_amd.configure_runtime() asks ROCm for six hardware
    queues unless the user chose a value, once per process."""
    import importlib
    import dynavsr_amd
    import dynavsr_amd._lib as L
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    importlib.reload(L)
    assert "GPU_MAX_HW_QUEUES" not in os.environ            # import time: no process-wide side effect
    r = dynavsr_amd.configure_runtime()
    assert os.environ["GPU_MAX_HW_QUEUES"] == "6" and r == {"hw_queues": "6", "effective": True}
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    assert dynavsr_amd.configure_runtime()["hw_queues"] == "6"   # idempotent: the first call decided
    importlib.reload(L)
    assert L.configure_runtime() == {"hw_queues": "4", "effective": True} and os.environ["GPU_MAX_HW_QUEUES"] == "4"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    importlib.reload(L)
    from dynavsr_amd.models import create_model
    create_model(cpu_opt())                                  # the wrappers configure the runtime themselves
    assert os.environ["GPU_MAX_HW_QUEUES"] == "6"


def test_create_model_warns_when_the_queue_request_comes_too_late(monkeypatch):
    """create_model in a process whose HIP runtime is already up (the reference's trainer initialises its process group first,
    train_dynavsr.py:23-30): GPU_MAX_HW_QUEUES cannot be raised any more -- configure_runtime() reports effective = False and
    create_model says so loudly (a RuntimeWarning naming the remedy), once per call, instead of silently running on."""
    import importlib
    import warnings
    import torch
    from dynavsr_amd import _lib as L
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    importlib.reload(L)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    from dynavsr_amd.models import create_model
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        create_model(cpu_opt())
    msgs = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]
    assert any("GPU_MAX_HW_QUEUES" in m and "configure_runtime" in m for m in msgs), msgs
    assert L.configure_runtime() == {"hw_queues": None, "effective": False} and "GPU_MAX_HW_QUEUES" not in os.environ
    monkeypatch.undo()
    importlib.reload(L)


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_dry_run(launcher):
    """`python bench.py --gpus 2` must start its own ranks (no launcher: README.md:92-96 starts the reference's trainer
    as one process per GPU; train_dynavsr.py:23-30), and the torch.distributed.run form the driver uses for N > 1 must
    keep working.  --dry-run runs the N-rank skeleton over gloo on the CPU: barriers, MAX of the elapsed time, the flat
    15 MB meta-gradient all-reduce, the frame shards + metric reduction; rank 0 prints ONE JSON line."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    args = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"]
    if launcher == "self":
        cmd = [sys.executable, bench] + args
    else:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), bench] + args
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run"] and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["meta_step"]["ranks"] == 2 and line["meta_step"]["allreduce"]["executed"]
    assert line["meta_step"]["allreduce"]["averaged_correctly"] and line["meta_step"]["allreduce"]["bytes"] == 4 * (3300131 + 452291)
    assert line["validation"]["psnr_vector_complete"]
    # the legs every rank runs on its own frames arrive on rank 0 as sum + per-rank figures (stand-in values: rank + 1)
    for leg in ("inner_step", "per_frame_pipeline"):
        assert line[leg]["ranks"] == 2 and line[leg]["per_rank_value"] == [1.0, 2.0] and line[leg]["value"] == 3.0
        assert line[leg]["min_rank_value"] == 1.0 and line[leg]["max_rank_value"] == 2.0
    # ... and the line has the key set of a GPU line (recorded on one MI355X: profiles/r05_bench_line.json) minus the legs
    # that only the one-GPU line carries
    rec_path = os.path.join(root, "profiles", "r05_bench_line.json")
    if os.path.exists(rec_path):
        with open(rec_path) as f:
            rec = json.loads(f.read().strip().splitlines()[-1])
        one_gpu_only = {"experimental_bf16_split", "edvr_l_bf16", "other_backbones", "cpu_baseline"}
        assert set(line) - {"dry_run"} == set(rec) - one_gpu_only, (sorted(set(line) ^ set(rec)))
        for leg in ("inner_step", "per_frame_pipeline"):
            assert {"value", "unit", "ranks", "per_rank_value", "min_rank_value", "max_rank_value"} <= set(rec[leg]) & set(line[leg])


def _val_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as tdist
    from dynavsr_amd import dist as D
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    n = 7
    seen = []

    def run(indices):
        for i in indices:
            seen.append(i)
            yield (30.0 + i, torch.tensor(31.0 + i))
    a, b = D.validate_sharded(n, run, rank, world)
    q.put((rank, seen, a.tolist(), b.tolist()))
    tdist.barrier()
    tdist.destroy_process_group()


def test_validate_sharded_round_robin_and_reduce_gloo():
    """train_dynavsr.py:500-728: frames range(rank, n, world) per rank, zero-initialised vectors, reduce(sum) to rank 0."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_val_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
    (r0, seen0, a0, b0), (r1, seen1, a1, b1) = out
    assert seen0 == [0, 2, 4, 6] and seen1 == [1, 3, 5]
    assert a0 == [30.0 + i for i in range(7)] and b0 == [31.0 + i for i in range(7)]          # complete on rank 0
    assert [a1[i] for i in (1, 3, 5)] == [31.0, 33.0, 35.0]                                   # own entries elsewhere


def test_process_env_switches_are_snapshot_and_a_change_warns(monkeypatch):
    """The once-per-process native switches (engine._PROCESS_ENV): the first plan build snapshots them, a later build under another
    value warns (the native side keeps what it read first) -- and the list covers every getenv("DVSR_...") site of csrc/ that is
    not a per-plan switch (engine._GEOMETRY_ENV) or a debug-build aid."""
    import glob
    import os
    import re
    import warnings
    from dynavsr_amd import engine
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynavsr_amd", "csrc")
    sites = set()
    for f in glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")):
        sites |= set(re.findall(r'getenv\("(DVSR_[A-Z0-9_]+)"\)', open(f).read()))
    debug_only = {"DVSR_CONV_ABLATE", "DVSR_WGRAD_NOFLUSH"}
    missing = sites - set(engine._PROCESS_ENV) - set(engine._GEOMETRY_ENV) - debug_only
    assert not missing, missing
    monkeypatch.setattr(engine, "_process_env_seen", None)
    monkeypatch.delenv("DVSR_DCN_FWD", raising=False)
    engine._check_process_env()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        engine._check_process_env()
        assert not w
        monkeypatch.setenv("DVSR_DCN_FWD", "dma")
        engine._check_process_env()
        assert len(w) == 1 and "DVSR_DCN_FWD" in str(w[0].message)


def test_tape_functions_read_the_callers_grad_mode():
    """engine._TapeFunction: inside autograd.Function.forward grad mode is always off and ctx.needs_input_grad reports
    requires_grad of the inputs even under torch.no_grad() -- so the wrapper captures the CALLER's grad mode in apply().
    (Round 6: without it every no-grad forward of a trainable network sized the gradient workspace and missed the engine's
    no-grad launch geometries; the GPU suite holds the workspace size, this holds the decision.)"""
    import torch
    from dynavsr_amd import engine
    seen = []

    class Probe(engine._TapeFunction):
        @staticmethod
        def forward(ctx, x, w):
            seen.append((engine._need_grad(ctx), tuple(ctx.needs_input_grad), torch.is_grad_enabled()))
            return x * w

        @staticmethod
        def backward(ctx, g):
            return g, g

    w = torch.nn.Parameter(torch.ones(3))
    x = torch.ones(3)
    y = Probe.apply(x, w)
    assert seen[-1] == (True, (False, True), False) and y.requires_grad
    with torch.no_grad():
        y = Probe.apply(x, w)
    assert seen[-1] == (False, (False, True), False) and not y.requires_grad     # needs_input_grad alone would have said True
    y = Probe.apply(x, w.detach())
    assert seen[-1][0] is False                                                   # nothing requires grad: no tape either
    with torch.no_grad():
        with torch.enable_grad():
            y = Probe.apply(x, w)
    assert seen[-1][0] is True
    y.sum().backward()
    assert torch.equal(w.grad, torch.ones(3))
