"""The two metric/IO helpers the PSNR gate depends on, restated from codes/utils/util.py:
``tensor2img`` (:112-142, 2-D/3-D tensors; the 4-D grid variant needs torchvision and is not used
by test_dynavsr.py) and ``calculate_psnr`` (:262-269, uint8 full frame, no border crop)."""
import math

import numpy as np


def tensor2img(tensor, out_type=np.uint8, min_max=(0, 1), mode='bgr'):
    t = tensor.squeeze().float().cpu().clamp_(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() == 3:
        a = t.numpy()
        a = a.transpose(1, 2, 0) if mode == 'rgb' else a[[2, 1, 0]].transpose(1, 2, 0)
    elif t.dim() == 2:
        a = t.numpy()
    else:
        raise TypeError('Only support 3D and 2D tensor. But received with dimension: {:d}'.format(t.dim()))
    if out_type == np.uint8:
        a = (a * 255.0).round()
    return a.astype(out_type)


def calculate_psnr(img1, img2):
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    return float('inf') if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))


def frame_metrics(sr, gt, min_max=(0, 1), need_img=False):
    """PSNR / SSIM of a super-resolved frame against its ground truth, computed on the GPU.

    The per-frame tail of test_dynavsr.py:285-292: ``img = tensor2img(sr, mode='rgb')``,
    ``calculate_psnr(img, hr_image)``, ``calculate_ssim(img, hr_image)`` with ``hr_image = tensor2img(GT)``
    (codes/utils/util.py:112-142, :262-269, :271-313) -- same quantisation, same float64 SSIM -- without
    moving the fp32 frame to the host.  sr, gt: [3,H,W] or [1,3,H,W] fp32 tensors on the GPU (there is no
    CPU path: calculate_psnr above is the host-side reference helper).  Returns (psnr, ssim) or, with
    need_img, (psnr, ssim, uint8 HWC RGB numpy image of sr for the PNG writer)."""
    import torch
    from .. import _lib as L
    sr, gt = sr.squeeze(), gt.squeeze()
    if sr.dim() == 2:
        sr, gt = sr[None], gt[None]
    if sr.dim() != 3 or sr.shape != gt.shape:
        raise ValueError('Input images must have the same dimensions.')   # calculate_ssim's message
    if not sr.is_cuda:
        raise RuntimeError("frame_metrics runs on the GPU (libdynavsr_hip); there is no CPU fallback")
    sr, gt = sr.float().contiguous(), gt.float().contiguous()
    c, h, w = sr.shape
    lib = L.lib()
    ws = torch.empty(lib.dvsr_frame_metrics_workspace_bytes(c, h, w), dtype=torch.uint8, device=sr.device)
    out = torch.empty(2, dtype=torch.float64, device=sr.device)
    img = torch.empty((h, w, c), dtype=torch.uint8, device=sr.device) if need_img else None
    L.check(lib.dvsr_frame_metrics(L.ptr(sr), L.ptr(gt), c, h, w, float(min_max[0]), float(min_max[1]),
                                   img.data_ptr() if need_img else None, out.data_ptr(), ws.data_ptr(),
                                   ws.numel(), L.stream()), "frame_metrics")
    mse, ssim = out.tolist()
    psnr = float('inf') if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))
    if need_img:
        a = img.cpu().numpy()
        return psnr, ssim, (a[:, :, 0] if c == 1 else a)
    return psnr, ssim
