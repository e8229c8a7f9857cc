"""GPU parity of Degradation.apply (dvsr_degrade_apply) against the goldens produced by the
reference's class and against the CPU oracle.  fp32 kernel, row-major tap order: tolerance 2e-6 absolute on
[0,1] data (the reference's own conv2d vs a float64 accumulation differ by 7e-7)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from dynavsr_amd.data.random_kernel_generator import Degradation
from oracle import degradation as od

pytestmark = pytest.mark.gpu
CASES = ["s4_aniso", "s2_aniso", "s4_iso", "s4_delta", "s2_k11", "s2_k10_even"]


def frames(seed_shape):
    seed, shape = int(seed_shape[0]), tuple(int(v) for v in seed_shape[1:])
    return np.random.RandomState(seed).rand(*shape).astype(np.float32)


@pytest.mark.parametrize("tag", CASES)
def test_apply_golden(tag):
    g = load_golden("degradation")
    ks, scale, theta, sx, sy = g[tag + "__params"]
    d = Degradation(int(ks), int(scale), theta=theta, sigma=[sx, sy])
    lr = d.apply(torch.from_numpy(frames(g[tag + "__seed_shape"])).cuda())
    assert lr.is_cuda and tuple(lr.shape) == g[tag + "__lr"].shape
    assert np.abs(lr.cpu().numpy() - g[tag + "__lr"]).max() < 2e-6


def test_single_image_per_frame_kernels_and_chain():
    g = load_golden("degradation")
    ks, scale, theta, sx, sy = g["single__params"]
    d = Degradation(int(ks), int(scale), theta=theta, sigma=[sx, sy])
    lr = d.apply(torch.from_numpy(frames(g["single__seed_shape"])).cuda())          # C H W in, C h w out
    assert lr.dim() == 3 and np.abs(lr.cpu().numpy() - g["single__lr"]).max() < 2e-6
    d = Degradation(21, 4)
    d.set_kernel_directly(g["perframe__kernels"])
    for n in (5, 7):
        lr = d.apply(torch.from_numpy(frames(g["perframe%d__seed_shape" % n])).cuda())
        assert np.abs(lr.cpu().numpy() - g["perframe%d__lr" % n]).max() < 2e-6
    ks, scale, theta, sx, sy = g["chain__params"]                                  # vsrbase.py:184-186
    d = Degradation(int(ks), int(scale), theta=theta, sigma=[sx, sy])
    lr = d.apply(torch.from_numpy(frames(g["chain__seed_shape"])).cuda(), quantise=True)
    dl = np.abs(lr.cpu().numpy() - g["chain__lr"])
    assert dl.max() <= 1.0 / 255 + 1e-7 and (dl > 1e-7).mean() < 1e-3              # a level may flip on a last-bit tie
    lv = lr.cpu().numpy() * 255
    assert np.abs(lv - np.round(lv)).max() < 1e-4                                  # really on the 8-bit grid
    slr = d.apply(torch.from_numpy(g["chain__lr"]).cuda())
    assert np.abs(slr.cpu().numpy() - g["chain__slr"]).max() < 2e-6


@pytest.mark.parametrize("shape,ks,scale", [((1, 3, 17, 19), 21, 4), ((2, 1, 33, 16), 11, 2), ((1, 3, 256, 256), 21, 4),
                                            ((5, 3, 720, 1280), 21, 4), ((1, 3, 45, 70), 21, 3)])
def test_apply_vs_oracle_ragged_and_full_size(shape, ks, scale):
    """Sizes that are not multiples of the 16x16 output tile or of the stride, a full HR frame, pads close to
    the image size; and linearity in the image (a size-independent property)."""
    r = np.random.RandomState(3)
    img = r.rand(*shape).astype(np.float32)
    d = Degradation(ks, scale, theta=0.9, sigma=[2.5, 1.1])
    y = d.apply(torch.from_numpy(img).cuda()).cpu().numpy()
    if np.prod(shape) <= 3 * 256 * 256:
        assert np.abs(y - od.apply(img, d.kernel, scale)).max() < 2e-6
    else:   # full size: a crop of the oracle (rows/cols of the output depend on a bounded input window)
        p = d.kernel_shift(d.kernel).shape[0] // 2
        sub = img[:1, :, :200 + 2 * p, :264 + 2 * p]
        yo = od.apply(sub, d.kernel, scale)
        n = (200 // scale) - p // scale - 2
        assert np.abs(y[:1, :, :n, :n] - yo[:, :, :n, :n]).max() < 2e-6
    img2 = r.rand(*shape).astype(np.float32)
    y2 = d.apply(torch.from_numpy(img2).cuda()).cpu().numpy()
    y12 = d.apply(torch.from_numpy(img + img2).cuda()).cpu().numpy()
    assert np.abs(y12 - (y + y2)).max() < 5e-6
    assert abs(float(y.mean()) - float(img.mean())) < 2e-2                        # the kernel sums to 1


def test_apply_errors():
    d = Degradation(21, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d.apply(torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match="reflection pad"):
        d.apply(torch.zeros(1, 3, 8, 64).cuda())                                   # pad 13 >= H, like ReflectionPad2d
