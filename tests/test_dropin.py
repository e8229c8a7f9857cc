"""CPU: the Level-1 drop-in recipe of INTEGRATION.md, checked against the reference drivers themselves.

With INTEGRATION's exact PYTHONPATH (dynavsr_amd first, then the repo, then codes/) a fresh interpreter
AST-walks codes/test_dynavsr.py and codes/train_dynavsr.py and resolves
  * every project-local import (``data.*``, ``utils.*``, ``options.*``, ``models.*``) to a module file,
  * every ``util.X`` / ``option.X`` attribute the drivers touch, on THIS build's modules,
  * every method the drivers call on the model wrappers, on this build's wrapper classes.
Nothing of the reference is copied or shipped: the drivers are only parsed, and the whole file is skipped where
/root/reference does not exist (the GPU box).  The host helpers of utils/util.py are also checked here
against the oracle's independent restatements (tests may import oracle/)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

REF = "/root/reference/codes"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")

_WALKER = r"""
import ast, importlib, importlib.machinery, importlib.util, json, sys
drivers, report = sys.argv[1:], {"imports": {}, "missing_modules": [], "missing_attrs": [], "missing_methods": []}
LOCAL = {"data", "utils", "options", "models"}
OURS = ("/dynavsr_amd/",)
import options.options as option
from utils import util
import models
from models.Video_base_model import VideoBaseModel
from models.LRestimator_model import LRimgestimator_Model
alias = {"util": util, "option": option}
import inspect
def provides(cls, name):     # a method, or an attribute some method of the class (or a base) assigns on self
    if callable(getattr(cls, name, None)):
        return True
    for k in cls.__mro__[:-1]:
        for n in ast.walk(ast.parse(inspect.getsource(k))):
            if isinstance(n, ast.Attribute) and isinstance(n.ctx, ast.Store) and isinstance(n.value, ast.Name) \
                    and n.value.id == "self" and n.attr == name:
                return True
    return False
wrappers = {"model": VideoBaseModel, "modelcp": VideoBaseModel, "est_model": LRimgestimator_Model,
            "est_modelcp": LRimgestimator_Model, "est_model_fixed": LRimgestimator_Model}
for path in drivers:
    tree = ast.parse(open(path).read(), path)
    for node in ast.walk(tree):
        mods = []
        if isinstance(node, ast.Import):
            mods = [a.name for a in node.names]
        elif isinstance(node, ast.ImportFrom) and node.level == 0:
            mods = [node.module]
            for a in node.names:    # `from data.meta_learner import loader` may name a submodule
                if node.module.split(".")[0] in LOCAL and importlib.util.find_spec(node.module) and \
                        getattr(importlib.util.find_spec(node.module), "submodule_search_locations", None) and \
                        importlib.machinery.PathFinder.find_spec(
                            node.module + "." + a.name, importlib.util.find_spec(node.module).submodule_search_locations):
                    mods.append(node.module + "." + a.name)
        for m in mods:
            if m.split(".")[0] not in LOCAL:
                continue
            try:
                spec = importlib.util.find_spec(m)
            except Exception as e:
                spec = None
            if spec is None or not spec.origin:
                report["missing_modules"].append([path, m])
            else:
                report["imports"][m] = spec.origin
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in alias:
            if not hasattr(alias[node.value.id], node.attr):
                report["missing_attrs"].append([path, node.value.id + "." + node.attr, node.lineno])
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in wrappers:
            if not provides(wrappers[node.value.id], node.attr):     # methods called and attributes read / written
                report["missing_methods"].append([path, node.value.id + "." + node.attr, node.lineno])
report["util_file"], report["option_file"], report["models_file"] = util.__file__, option.__file__, models.__file__
print(json.dumps(report))
"""


@needs_ref
def test_level1_recipe_resolves_everything_the_drivers_use(tmp_path):
    env = dict(os.environ)
    # INTEGRATION.md, Level 1: PYTHONPATH=<repo>/dynavsr_amd:<repo>:<codes>
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "dynavsr_amd"), ROOT, REF])
    env["PYTHONDONTWRITEBYTECODE"] = "1"                 # the reference tree is read-only by contract
    drivers = [os.path.join(REF, "test_dynavsr.py"), os.path.join(REF, "train_dynavsr.py")]
    r = subprocess.run([sys.executable, "-c", _WALKER] + drivers, env=env, cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["missing_modules"] == [], rep["missing_modules"]
    assert rep["missing_attrs"] == [], rep["missing_attrs"]
    assert rep["missing_methods"] == [], rep["missing_methods"]
    # who provides what: models / options / utils.util are this build's, the dataset code is the reference's
    for k in ("util_file", "option_file", "models_file"):
        assert os.path.join(ROOT, "dynavsr_amd") in rep[k], rep[k]
    imp = rep["imports"]
    assert imp["data.data_sampler"].startswith(REF) and imp["data.meta_learner"].startswith(REF)
    assert imp["options.options"].startswith(ROOT) and imp["models"].startswith(ROOT) and imp["utils"].startswith(ROOT)


def test_packages_do_not_shadow_a_codes_tree(tmp_path):
    """Same mechanism without the reference: a stand-in codes/ tree with data/other.py and utils/other.py next to
    this build on the path -- both must stay importable, and `data.random_kernel_generator` of that tree wins
    (the DataLoader workers call it with CPU tensors; the device version is an explicit opt-in)."""
    codes = tmp_path / "codes"
    for pkg in ("data", "utils"):
        (codes / pkg).mkdir(parents=True)
        (codes / pkg / "__init__.py").write_text("")
        (codes / pkg / "other.py").write_text("WHO = 'codes'\n")
    (codes / "data" / "random_kernel_generator.py").write_text("WHO = 'codes'\n")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "dynavsr_amd"), ROOT, str(codes)])
    code = ("import data.other, utils.other, data.random_kernel_generator as r\n"
            "from utils import util\n"
            "assert data.other.WHO == utils.other.WHO == r.WHO == 'codes'\n"
            "assert 'dynavsr_amd' in util.__file__ and hasattr(util, 'ProgressBar')\n"
            "import dynavsr_amd.data.random_kernel_generator as ours\n"
            "assert hasattr(ours, 'Degradation')\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


# ---- utils/util.py host helpers -------------------------------------------------------------------
def test_util_ssim_psnr_match_the_oracle():
    from dynavsr_amd.utils import util
    from oracle import metrics as om
    rng = np.random.RandomState(3)
    a = rng.randint(0, 256, (37, 45, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.randint(-9, 10, a.shape), 0, 255).astype(np.uint8)
    assert abs(util.calculate_ssim(a, b) - om.calculate_ssim(a, b)) < 1e-12
    assert abs(util.calculate_ssim(a[:, :, 0], b[:, :, 0]) - om.calculate_ssim(a[:, :, 0], b[:, :, 0])) < 1e-12
    assert abs(util.calculate_ssim(a[:, :, :1], b[:, :, :1]) - util.calculate_ssim(a[:, :, 0], b[:, :, 0])) < 1e-15
    assert util.calculate_psnr(a, b) == om.calculate_psnr(a, b) and util.calculate_psnr(a, a) == float("inf")
    assert util.calculate_ssim(a, a) == pytest.approx(1.0, abs=1e-12)
    with pytest.raises(ValueError):
        util.calculate_ssim(a, b[:-1])
    with pytest.raises(ValueError):
        util.calculate_ssim(a[None], b[None])


def test_util_tensor2img_variants():
    from dynavsr_amd.utils import util
    from oracle import metrics as om
    t = torch.rand(3, 9, 11) * 1.4 - 0.2
    rgb, bgr = util.tensor2img(t, mode='rgb'), util.tensor2img(t)
    assert rgb.dtype == np.uint8 and rgb.shape == (9, 11, 3) and np.array_equal(rgb[:, :, ::-1], bgr)
    assert np.array_equal(rgb, om.tensor2img_rgb(t.numpy()))
    assert np.array_equal(util.tensor2rgbimg(t), rgb)
    assert util.tensor2img(t[0]).shape == (9, 11)
    assert util.tensor2img(t, out_type=np.float32, mode='rgb').max() <= 1.0
    # 4-D: make_grid tiling, nrow = floor(sqrt(B)), 2 px of zero padding around every tile
    g = util.tensor2img(torch.ones(5, 3, 4, 6), mode='rgb')
    assert g.shape == (3 * 6 + 2, 2 * 8 + 2, 3)
    assert g[:2].max() == 0 and g[2:6, 2:8].min() == 255 and g[2:6, 8:10].max() == 0 and g[14:18, 10:16].max() == 0
    assert util.tensor2img(torch.ones(1, 3, 4, 6)).shape == (4, 6, 3)       # squeeze() drops the batch of one
    with pytest.raises(TypeError):
        util.tensor2img(torch.ones(2, 2, 3, 4, 6))
    assert [v.shape for v in util.crop_border([rgb, bgr], 2)] == [(5, 7, 3)] * 2 and util.crop_border([rgb], 0)[0] is rgb


def test_util_files_logger_progress(tmp_path, capsys):
    import logging
    from dynavsr_amd.utils import util
    d = tmp_path / "exp"
    util.mkdirs([str(d), str(tmp_path / "a" / "b")])
    util.mkdir(str(d))
    (d / "x.txt").write_text("1")
    util.mkdir_and_rename(str(d))
    assert d.is_dir() and not list(d.iterdir())
    assert any(p.name.startswith("exp_archived_") and (p / "x.txt").exists() for p in tmp_path.iterdir())
    util.setup_logger("dropin_test", str(tmp_path), "val", level=logging.INFO, screen=False, tofile=True)
    logging.getLogger("dropin_test").info("hello")
    logs = [p for p in tmp_path.iterdir() if p.name.startswith("val_") and p.suffix == ".log"]
    assert len(logs) == 1 and "INFO: hello" in logs[0].read_text()
    util.set_random_seed(7)
    a = (np.random.rand(), torch.rand(1).item())
    util.set_random_seed(7)
    assert a == (np.random.rand(), torch.rand(1).item())
    bar = util.ProgressBar(3)
    for i in range(3):
        bar.update("step %d" % i)
    out = capsys.readouterr().out
    assert "3/3" in out and "step 2" in out and bar.completed == 3
    util.ProgressBar(0).update()
    assert "completed: 1" in capsys.readouterr().out
    # PNG writer: decodable by an independent inflate + un-filter
    import struct, zlib
    img = np.random.RandomState(0).randint(0, 256, (5, 7, 3)).astype(np.uint8)
    util.save_img(img, str(tmp_path / "f.png"))                               # BGR in, like cv2.imwrite
    raw = (tmp_path / "f.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", raw[16:24]) == (7, 5)
    n = struct.unpack(">I", raw[33:37])[0]
    rows = np.frombuffer(zlib.decompress(raw[41:41 + n]), np.uint8).reshape(5, 1 + 7 * 3)
    assert rows[:, 0].max() == 0 and np.array_equal(rows[:, 1:].reshape(5, 7, 3), img[:, :, ::-1])


def test_util_duf_downsample_and_flip_forward():
    from dynavsr_amd.utils import util
    x = torch.rand(1, 2, 3, 32, 40)
    y = util.DUF_downsample(x, scale=4)
    assert y.shape == (1, 2, 3, 8, 10)
    c = torch.full((1, 1, 3, 32, 32), 0.25)
    assert torch.allclose(util.DUF_downsample(c, scale=2), torch.full((1, 1, 3, 16, 16), 0.25), atol=1e-6)
    net = torch.nn.Conv2d(3, 3, 3, padding=1)
    inp = torch.rand(1, 3, 8, 8)
    ref = util.single_forward(net, inp)
    assert torch.equal(ref, net(inp).detach())
    sym = util.flipx4_forward(lambda t: (t * 2, None), inp)                   # tuple outputs: first element
    assert torch.allclose(sym, inp * 2, atol=1e-7)


@needs_ref
def test_backbone_state_dicts_match_the_reference_modules():
    """networks.define_G's three video backbones expose exactly the reference modules' state-dict keys and shapes, so
    reference checkpoints load with strict=True.  The reference modules are only constructed (CPU, plain torch; EDVR's
    CUDA extension import is stubbed like oracle/gen_golden.py does), nothing is copied."""
    code = r'''
import importlib.util, json, sys, types
sys.dont_write_bytecode = True
repo, ref = sys.argv[1], sys.argv[2]
sys.path.insert(0, repo)
from dynavsr_amd.models.archs import DUF_arch, EDVR_arch, TOF_arch
sys.path.insert(0, ref)
sys.modules["models.archs.dcn.deform_conv_cuda"] = types.ModuleType("deform_conv_cuda")
import models.archs.DUF_arch as RD, models.archs.EDVR_arch as RE, models.archs.TOF_arch as RT
bad = []
pairs = [("TOF", TOF_arch.TOFlow(adapt_official=True), RT.TOFlow(adapt_official=True))]
for n in ("DUF_16L", "DUF_28L", "DUF_52L"):
    for s in (2, 3, 4):
        pairs.append((n + "x%d" % s, getattr(DUF_arch, n)(scale=s, adapt_official=True), getattr(RD, n)(scale=s, adapt_official=True)))
for s in (2, 4):
    cfg = dict(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=s)
    pairs.append(("EDVR-Mx%d" % s, EDVR_arch.EDVR(**cfg), RE.EDVR(**cfg)))
for name, a, b in pairs:
    sa, sb = a.state_dict(), b.state_dict()
    if list(sa) != list(sb) or any(tuple(sa[k].shape) != tuple(sb[k].shape) for k in sa):
        bad.append(name)
print(json.dumps({"checked": len(pairs), "bad": bad}))
'''
    env = dict(os.environ)
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, "-c", code, ROOT, REF], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["checked"] == 12 and rep["bad"] == [], rep


def test_async_png_writer(tmp_path):
    """Background PNG writer: same bytes as the synchronous path, the caller's buffer may be reused
    right after the call, errors surface at flush_png()."""
    from dynavsr_amd.utils import util
    rs = np.random.RandomState(1)
    imgs = [rs.randint(0, 256, (24, 40, 3)).astype(np.uint8) for _ in range(6)]
    for i, im in enumerate(imgs):
        util.write_png(str(tmp_path / ("s%d.png" % i)), im)
    util.async_png(True)
    try:
        buf = np.empty_like(imgs[0])
        for i, im in enumerate(imgs):
            buf[...] = im
            util.write_png(str(tmp_path / ("a%d.png" % i)), buf)      # the writer copies at the call
            buf[...] = 0
        util.save_img(imgs[0], str(tmp_path / "a_bgr.png"))
        util.flush_png()
        for i in range(6):
            assert (tmp_path / ("a%d.png" % i)).read_bytes() == (tmp_path / ("s%d.png" % i)).read_bytes()
        util.write_png(str(tmp_path / "no_such_dir" / "x.png"), imgs[0])
        with pytest.raises(OSError):
            util.flush_png()
        util.flush_png()                                               # the error is reported once
    finally:
        util.async_png(False)
    util.save_img(imgs[0], str(tmp_path / "s_bgr.png"))
    assert (tmp_path / "a_bgr.png").read_bytes() == (tmp_path / "s_bgr.png").read_bytes()
