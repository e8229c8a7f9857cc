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
