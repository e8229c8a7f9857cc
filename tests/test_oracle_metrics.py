"""The metrics oracle (oracle/metrics.py) against what can pin it without cv2 (see its header): the Gaussian
kernel formula, closed-form SSIM values, an independent scipy evaluation, and the product's host helpers."""
import numpy as np
import torch

from dynavsr_amd.utils import util
from oracle import metrics as om


def _frames(seed, h, w, c=3, noise=0.05):
    r = np.random.RandomState(seed)
    a = r.rand(c, h, w).astype(np.float32) * 1.2 - 0.1          # leaves [0,1]: exercises the clamp
    b = np.clip(a + noise * r.standard_normal(a.shape).astype(np.float32), -0.2, 1.3)
    return a, b


def test_gaussian_kernel_is_cv2_formula():
    k = om.gaussian_kernel(11, 1.5)
    assert abs(k.sum() - 1) < 1e-15 and np.allclose(k, k[::-1])
    # ratio of neighbours is the Gaussian's: k[i+1]/k[i] = exp(-((i-4)^2 - (i-5)^2) / 4.5)
    i = np.arange(10)
    assert np.allclose(k[1:] / k[:-1], np.exp(-(((i - 4) ** 2) - ((i - 5) ** 2)) / 4.5))


def test_ssim_closed_forms():
    a, _ = _frames(0, 24, 31)
    ia = om.tensor2img_rgb(a)
    assert abs(om.calculate_ssim(ia, ia) - 1.0) < 1e-12
    c1 = np.full((20, 20, 3), 60, np.uint8)
    c2 = np.full((20, 20, 3), 200, np.uint8)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    want = (2 * 60 * 200 + C1) * C2 / ((60 ** 2 + 200 ** 2 + C1) * C2)
    assert abs(om.calculate_ssim(c1, c2) - want) < 1e-9
    assert abs(om.calculate_ssim(c1, c2) - om.calculate_ssim(c2, c1)) < 1e-15


def test_ssim_against_scipy_correlate():
    from scipy import ndimage
    a, b = _frames(1, 40, 52)
    ia, ib = om.tensor2img_rgb(a), om.tensor2img_rgb(b)
    k = om.gaussian_kernel()
    win = np.outer(k, k)[:, :, None]                    # one 2-D window per channel, like cv2.filter2D

    def f(x):
        return ndimage.correlate(x.astype(np.float64), win, mode="mirror")[5:-5, 5:-5]
    x, y = ia.astype(np.float64), ib.astype(np.float64)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mu1, mu2 = f(x), f(y)
    m = ((2 * mu1 * mu2 + C1) * (2 * (f(x * y) - mu1 * mu2) + C2)) / \
        ((mu1 ** 2 + mu2 ** 2 + C1) * (f(x * x) - mu1 ** 2 + f(y * y) - mu2 ** 2 + C2))
    assert abs(m.mean() - om.calculate_ssim(ia, ib)) < 1e-12


def test_host_helpers_match_oracle():
    a, b = _frames(2, 16, 20)
    ia = util.tensor2img(torch.from_numpy(a), mode="rgb")
    assert np.array_equal(ia, om.tensor2img_rgb(a))
    ib = om.tensor2img_rgb(b)
    assert util.calculate_psnr(ia, ib) == om.calculate_psnr(ia, ib)
    assert util.calculate_psnr(ia, ia) == float("inf")
