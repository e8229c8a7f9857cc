"""GPU parity of dvsr_frame_metrics against the CPU oracle of util.tensor2img /
calculate_psnr / calculate_ssim: uint8 image bit-exact, squared-error sum exact (PSNR to the last ulp of the
host's log10), SSIM (float64 on both sides, separable vs 2-D window) within 1e-10."""
import math

import numpy as np
import pytest
import torch

from dynavsr_amd.utils import util
from oracle import metrics as om

pytestmark = pytest.mark.gpu


def _frames(seed, c, h, w, noise=0.05):
    r = np.random.RandomState(seed)
    a = r.rand(c, h, w).astype(np.float32) * 1.2 - 0.1          # leaves [0,1]: exercises the clamp
    b = np.clip(a + noise * r.standard_normal(a.shape).astype(np.float32), -0.2, 1.3).astype(np.float32)
    return a, b


@pytest.mark.parametrize("c,h,w", [(3, 11, 11), (3, 37, 53), (1, 64, 64), (3, 33, 11), (3, 180, 320), (3, 720, 1280)])
def test_frame_metrics_vs_oracle(c, h, w):
    a, b = _frames(h * 1000 + w, c, h, w)
    ia, ib = om.tensor2img_rgb(a if c > 1 else a[0]), om.tensor2img_rgb(b if c > 1 else b[0])
    psnr, ssim, img = util.frame_metrics(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), need_img=True)
    assert img.dtype == np.uint8 and np.array_equal(img, ia)
    want_psnr = om.calculate_psnr(ia, ib)
    assert abs(psnr - want_psnr) <= 1e-12 * want_psnr
    want_ssim = om.calculate_ssim(ia, ib)
    assert abs(ssim - want_ssim) < 1e-10, (ssim, want_ssim)


def test_frame_metrics_edge_cases():
    a, b = _frames(5, 3, 24, 40)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    psnr, ssim = util.frame_metrics(ta, ta)
    assert psnr == float("inf") and abs(ssim - 1.0) < 1e-12                   # identical frames
    p1, s1 = util.frame_metrics(ta, tb)
    p2, s2 = util.frame_metrics(ta[None], tb[None])                            # the [1,3,H,W] form of fake_H
    assert (p1, s1) == (p2, s2)                                                # deterministic, squeeze like tensor2img
    assert util.frame_metrics(ta, tb) == (p1, s1)
    # ties of the rounding: k + 0.5 quantises to the even neighbour (numpy's round), exactly representable inputs
    t = torch.tensor([0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255], dtype=torch.float32).repeat(3, 12, 3).cuda()
    _, _, img = util.frame_metrics(t, t, need_img=True)
    assert np.array_equal(img, om.tensor2img_rgb(t.cpu().numpy()))
    # no "valid" region: the reference's mean of an empty SSIM map is NaN; PSNR is still defined
    p, s = util.frame_metrics(ta[:, :10], tb[:, :10])
    assert math.isnan(s) and abs(p - om.calculate_psnr(om.tensor2img_rgb(a[:, :10]), om.tensor2img_rgb(b[:, :10]))) < 1e-9
    # another value range (min_max of tensor2img)
    p, s = util.frame_metrics(ta * 2 - 1, tb * 2 - 1, min_max=(-1, 1))
    ia, ib = om.tensor2img_rgb(a * 2 - 1, (-1, 1)), om.tensor2img_rgb(b * 2 - 1, (-1, 1))
    assert abs(p - om.calculate_psnr(ia, ib)) < 1e-9 and abs(s - om.calculate_ssim(ia, ib)) < 1e-10


def test_frame_metrics_errors():
    a, b = _frames(6, 3, 16, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        util.frame_metrics(torch.from_numpy(a), torch.from_numpy(b))
    with pytest.raises(ValueError, match="same dimensions"):
        util.frame_metrics(torch.from_numpy(a).cuda(), torch.from_numpy(b[:, :8]).cuda())
