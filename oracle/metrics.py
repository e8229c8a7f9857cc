"""ORACLE (test infrastructure only -- never imported by dynavsr_amd/).

CPU restatement of the per-frame metrics of codes/utils/util.py:
  tensor2img      :112-142   clamp, rescale, x255, round half to even, uint8 HWC (3-D tensors, mode='rgb')
  calculate_psnr  :262-269
  ssim            :271-293   C1 = (0.01*255)^2, C2 = (0.03*255)^2, window = outer(k, k) with
                             k = cv2.getGaussianKernel(11, 1.5), five cv2.filter2D passes in float64 cropped to
                             the "valid" region [5:-5, 5:-5], mean of the SSIM map
  calculate_ssim  :295-313   3-channel images: the mean of three IDENTICAL ssim(img1, img2) calls on the whole
                             HWC array (the loop never indexes the channel), i.e. ssim over all channels

PARITY UNPINNED for ssim: cv2 is not installed in this image, so the reference's function cannot be executed
here and no golden vector can be generated from it.  What pins this file instead (tests/test_oracle_metrics.py):
cv2.getGaussianKernel's documented formula for ksize = 11, sigma = 1.5 (exp(-(i-5)^2 / (2 sigma^2)), normalised),
closed-form values (identical frames -> 1, two constant frames -> (2ab + C1)/(a^2 + b^2 + C1)), and an
independent evaluation through scipy.ndimage.correlate.  tensor2img / calculate_psnr are pinned by the golden
vectors of the PSNR gate (tests/golden/, produced by importing the reference)."""
import math

import numpy as np


def tensor2img_rgb(t, min_max=(0, 1)):
    """util.py:112-142 for a [3,H,W] / [H,W] float tensor or array, mode='rgb'."""
    a = np.asarray(t, dtype=np.float32)
    a = np.clip(a, np.float32(min_max[0]), np.float32(min_max[1]))
    a = (a - np.float32(min_max[0])) / np.float32(min_max[1] - min_max[0])
    if a.ndim == 3:
        a = a.transpose(1, 2, 0)
    return (a * 255.0).round().astype(np.uint8)


def calculate_psnr(img1, img2):
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    return float('inf') if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))


def gaussian_kernel(ksize=11, sigma=1.5):
    """cv2.getGaussianKernel for ksize > 7 (no fixed small-kernel table): normalised samples of the Gaussian."""
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return k / k.sum()


def _filter_valid(img, window):
    """cv2.filter2D(img, -1, window)[5:-5, 5:-5]: correlation with the 11x11 window, per channel; the crop
    removes every pixel the border mode could have touched."""
    kh, kw = window.shape
    h, w = img.shape[:2]
    out = np.zeros((h - kh + 1, w - kw + 1) + img.shape[2:], dtype=np.float64)
    for dy in range(kh):
        for dx in range(kw):
            out += window[dy, dx] * img[dy:dy + h - kh + 1, dx:dx + w - kw + 1]
    return out


def ssim(img1, img2):
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    img1, img2 = img1.astype(np.float64), img2.astype(np.float64)
    k = gaussian_kernel(11, 1.5)
    window = np.outer(k, k)
    mu1, mu2 = _filter_valid(img1, window), _filter_valid(img2, window)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = _filter_valid(img1 ** 2, window) - mu1_sq
    s2 = _filter_valid(img2 ** 2, window) - mu2_sq
    s12 = _filter_valid(img1 * img2, window) - mu1_mu2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean()


def calculate_ssim(img1, img2):
    if img1.shape != img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    if img1.ndim == 2:
        return ssim(img1, img2)
    if img1.ndim == 3:
        if img1.shape[2] == 3:
            return float(np.array([ssim(img1, img2) for _ in range(3)]).mean())
        if img1.shape[2] == 1:
            return ssim(np.squeeze(img1), np.squeeze(img2))
    raise ValueError('Wrong input image dimensions.')
