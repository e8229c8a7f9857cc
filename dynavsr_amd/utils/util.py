"""Host-side helpers of the drivers, same names and behaviour as codes/utils/util.py.

test_dynavsr.py / train_dynavsr.py touch: get_timestamp, mkdir(s), mkdir_and_rename, set_random_seed,
setup_logger (:43-95), crop_border, tensor2img (:97-142), save_img (:172), calculate_psnr / ssim /
calculate_ssim (:262-313) and ProgressBar (:316-363).  The reference needs cv2 (Gaussian window, filter2D,
imwrite) and torchvision (make_grid); neither is a dependency here: the window is built in numpy, the SSIM
filter is the separable "valid" form of the reference's outer-product window (same region as its [5:-5] crop),
make_grid's tiling is restated, PNGs are written with zlib.  ``frame_metrics`` is the device-side
replacement of the per-frame tail; everything else here is plain host code.
"""
import logging
import math
import os
import random
import shutil
import struct
import sys
import time
import zlib
from collections import OrderedDict
from datetime import datetime

import numpy as np
import yaml


def OrderedYaml():
    """(Loader, Dumper) that keep mapping order (util.py:23-36)."""
    try:
        from yaml import CDumper as Dumper, CLoader as Loader
    except ImportError:
        from yaml import Dumper, Loader
    tag = yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG
    Dumper.add_representer(OrderedDict, lambda dumper, data: dumper.represent_dict(data.items()))
    Loader.add_constructor(tag, lambda loader, node: OrderedDict(loader.construct_pairs(node)))
    return Loader, Dumper


# ---- miscellaneous (util.py:43-95) ---------------------------------------------------------------
def get_timestamp():
    return datetime.now().strftime('%y%m%d-%H%M%S')


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def mkdirs(paths):
    for p in ([paths] if isinstance(paths, str) else paths):
        mkdir(p)


def mkdir_and_rename(path):
    """An existing directory is archived under a time-stamped name, then a fresh one is made."""
    if os.path.exists(path):
        archived = path + '_archived_' + get_timestamp()
        msg = 'Path already exists. Rename it to [{:s}]'.format(archived)
        print(msg)
        logging.getLogger('base').info(msg)
        os.rename(path, archived)
    os.makedirs(path)


def set_random_seed(seed):
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def setup_logger(logger_name, root, phase, level=logging.INFO, screen=False, tofile=False):
    lg = logging.getLogger(logger_name)
    fmt = logging.Formatter('%(asctime)s.%(msecs)03d - %(levelname)s: %(message)s', datefmt='%y-%m-%d %H:%M:%S')
    lg.setLevel(level)
    handlers = []
    if tofile:
        handlers.append(logging.FileHandler(os.path.join(root, '{}_{}.log'.format(phase, get_timestamp())), mode='w'))
    if screen:
        handlers.append(logging.StreamHandler())
    for h in handlers:
        h.setFormatter(fmt)
        lg.addHandler(h)


# ---- image conversion (util.py:97-173) -----------------------------------------------------------
def crop_border(img_list, crop_border):
    if crop_border == 0:
        return img_list
    return [v[crop_border:-crop_border, crop_border:-crop_border] for v in img_list]


def _make_grid(t, nrow, padding=2):
    """torchvision.utils.make_grid(t, nrow, normalize=False) for a [B,C,H,W] tensor: single-channel images are
    repeated to 3 channels, tiles are laid out row-major with `padding` zero pixels around each."""
    import torch
    if t.size(1) == 1:
        t = t.expand(-1, 3, -1, -1)
    b, c, h, w = t.shape
    if b == 1:
        return t[0]
    xmaps = min(nrow, b)
    ymaps = int(math.ceil(b / xmaps))
    th, tw = h + padding, w + padding
    grid = torch.zeros((c, th * ymaps + padding, tw * xmaps + padding), dtype=t.dtype)
    for k in range(b):
        y, x = divmod(k, xmaps)
        grid[:, y * th + padding:y * th + padding + h, x * tw + padding:x * tw + padding + w] = t[k]
    return grid


def _tensor2array(tensor, out_type, min_max, to_bgr):
    t = tensor.squeeze().float().cpu().clamp_(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() == 4:
        t = _make_grid(t, nrow=int(math.sqrt(len(t))))
    if t.dim() == 3:
        a = t.numpy()
        a = (a[[2, 1, 0]] if to_bgr else a).transpose(1, 2, 0)
    elif t.dim() == 2:
        a = t.numpy()
    else:
        raise TypeError('Only support 4D, 3D and 2D tensor. But received with dimension: {:d}'.format(t.dim()))
    if out_type == np.uint8:
        a = (a * 255.0).round()      # round half to even, like the reference's numpy call
    return a.astype(out_type)


def tensor2img(tensor, out_type=np.uint8, min_max=(0, 1), mode='bgr'):
    """[B,C,H,W] (tiled), [C,H,W] or [H,W] float tensor, RGB -> uint8 HWC image, BGR unless mode='rgb'."""
    return _tensor2array(tensor, out_type, min_max, to_bgr=(mode != 'rgb'))


def tensor2rgbimg(tensor, out_type=np.uint8, min_max=(0, 1), mode='bgr'):
    return _tensor2array(tensor, out_type, min_max, to_bgr=False)


def _png_bytes(rgb):
    """Minimal PNG encoder (8-bit grey / RGB), zlib level 3: the frames are evaluation output, not archives."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * c)], axis=1).tobytes()   # filter 0 per row

    def chunk(tag, data):
        body = tag + data
        return struct.pack('>I', len(data)) + body + struct.pack('>I', zlib.crc32(body) & 0xFFFFFFFF)

    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, ctype, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(raw, 3)) + chunk(b'IEND', b''))


# ---- PNG output off the critical path (test_dynavsr.py:285-288 writes three PNGs per frame inside the
# adaptation loop) -------------------------------------------------------------------------------------------------
# async_png(True), or DVSR_ASYNC_PNG=1 in the environment, hands the encode + write of save_img / write_png to ONE
# background thread (zlib and file I/O release the GIL, so they overlap the next frame's kernels); the array is copied
# at the call, so the caller may reuse its buffer.  flush_png() waits for the queue and re-raises the first error;
# it also runs at interpreter exit.  Off by default: the reference's calls are synchronous.
_png_state = {'on': os.environ.get('DVSR_ASYNC_PNG', '0') not in ('', '0'), 'q': None, 'thread': None, 'err': None}


def _png_worker(q):
    while True:
        item = q.get()
        try:
            if item is None:
                return
            path, arr = item
            with open(path, 'wb') as f:
                f.write(_png_bytes(arr))
        except Exception as e:  # surfaced by flush_png()
            if _png_state['err'] is None:
                _png_state['err'] = e
        finally:
            q.task_done()


def async_png(on=True):
    """Switch the background PNG writer on / off (off flushes first)."""
    if not on:
        flush_png()
    _png_state['on'] = bool(on)


def flush_png():
    """Wait until every queued PNG is on disk; raise the first error a write hit."""
    q = _png_state['q']
    if q is not None:
        q.join()
    err, _png_state['err'] = _png_state['err'], None
    if err is not None:
        raise err


def _emit_png(path, arr):
    if not _png_state['on']:
        with open(path, 'wb') as f:
            f.write(_png_bytes(arr))
        return
    if _png_state['q'] is None:
        import atexit
        import queue
        import threading
        _png_state['q'] = queue.Queue(maxsize=64)   # bounded: at most 64 frames (~170 MB at 720x1280) wait in memory
        _png_state['thread'] = threading.Thread(target=_png_worker, args=(_png_state['q'],), daemon=True)
        _png_state['thread'].start()
        atexit.register(flush_png)
    _png_state['q'].put((path, np.array(arr, dtype=np.uint8, copy=True)))


def save_img(img, img_path, mode='RGB'):
    """cv2.imwrite(img_path, img) of the reference: the array is taken as BGR (cv2's convention; `mode` is
    ignored there too).  PNG only."""
    a = np.asarray(img)
    if a.ndim == 3 and a.shape[2] >= 3:
        a = a[:, :, [2, 1, 0] + list(range(3, a.shape[2]))]
    _emit_png(img_path, a)


def write_png(path, rgb):
    """imageio.imwrite(path, rgb) for the uint8 RGB frames of test_dynavsr.py:288."""
    _emit_png(path, rgb)


def DUF_downsample(x, scale=4, sigma=None):
    """x [B,T,C,H,W] -> Gaussian-blurred (13 taps, sigma 0.4*scale) and sub-sampled frames (util.py:176-210)."""
    import torch
    import torch.nn.functional as F
    from scipy.ndimage import gaussian_filter
    assert scale in [2, 3, 4], 'Scale [{}] is not supported'.format(scale)
    B, T, C, H, W = x.size()
    x = x.reshape(-1, 1, H, W)
    pad = 6 + scale * 2
    r_h, r_w = (3 - (H % 3), 3 - (W % 3)) if scale == 3 else (0, 0)
    x = F.pad(x, [pad, pad + r_w, pad, pad + r_h], 'reflect')
    delta = np.zeros((13, 13))
    delta[6, 6] = 1
    k = torch.from_numpy(gaussian_filter(delta, 0.4 * scale if sigma is None else sigma)).type_as(x)[None, None]
    x = F.conv2d(x, k, stride=scale)[:, :, 2:-2, 2:-2]
    return x.view(B, T, C, x.size(2), x.size(3))


def single_forward(model, inp):
    import torch
    with torch.no_grad():
        out = model(inp)
    if isinstance(out, (list, tuple)):
        out = out[0]
    return out.data.float().cpu()


def flipx4_forward(model, inp):
    """Self-ensemble over the four flips (util.py:232-256)."""
    import torch
    acc = single_forward(model, inp)
    for dims in ((-1,), (-2,), (-2, -1)):
        acc = acc + torch.flip(single_forward(model, torch.flip(inp, dims)), dims)
    return acc / 4


# ---- metrics (util.py:262-313) -------------------------------------------------------------------
def calculate_psnr(img1, img2):
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    return float('inf') if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))


def _gauss11():
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))       # cv2.getGaussianKernel(11, 1.5)
    return g / g.sum()


def _valid_blur(a, g):
    """'valid' correlation with outer(g, g), separably: the reference filters the whole image and crops
    [5:-5, 5:-5], which removes exactly the border-dependent part."""
    n = len(g)
    rows = sum(g[i] * a[i:a.shape[0] - n + 1 + i] for i in range(n))
    return sum(g[i] * rows[:, i:rows.shape[1] - n + 1 + i] for i in range(n))


def ssim(img1, img2):
    """Single-plane SSIM, float64, 11x11 Gaussian window sigma 1.5 (util.py:272-292).  A 3-channel array is
    filtered per channel like cv2.filter2D does, and the mean runs over everything."""
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    g = _gauss11()
    mu1, mu2 = _valid_blur(a, g), _valid_blur(b, g)
    s11 = _valid_blur(a * a, g) - mu1 * mu1
    s22 = _valid_blur(b * b, g) - mu2 * mu2
    s12 = _valid_blur(a * b, g) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean()


def calculate_ssim(img1, img2):
    """[0,255] images, HW or HWC.  For 3 channels the reference averages `ssim(img1, img2)` of the FULL
    3-channel arrays three times (util.py:305-309), i.e. the mean over all channels at once."""
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    if img1.ndim == 2:
        return ssim(img1, img2)
    if img1.ndim == 3:
        if img1.shape[2] == 3:
            return np.array([ssim(img1, img2)] * 3).mean()
        if img1.shape[2] == 1:
            return ssim(np.squeeze(img1), np.squeeze(img2))
        return None                                        # the reference falls through here, too
    raise ValueError('Wrong input image dimensions.')


class ProgressBar(object):
    """Two-line terminal progress bar (util.py:316-363): bar + message, redrawn in place."""

    def __init__(self, task_num=0, bar_width=50, start=True):
        self.task_num = task_num
        self.bar_width = min(bar_width, self._get_max_bar_width())
        self.completed = 0
        if start:
            self.start()

    def _get_max_bar_width(self):
        cols = shutil.get_terminal_size()[0]
        width = min(int(cols * 0.6), cols - 50)
        if width < 10:
            print('terminal width is too small ({}), please consider widen the terminal for better '
                  'progressbar visualization'.format(cols))
            width = 10
        return width

    def start(self):
        if self.task_num > 0:
            sys.stdout.write('[{}] 0/{}, elapsed: 0s, ETA:\n{}\n'.format(' ' * self.bar_width, self.task_num, 'Start...'))
        else:
            sys.stdout.write('completed: 0, elapsed: 0s')
        sys.stdout.flush()
        self.start_time = time.time()

    def update(self, msg='In progress...'):
        self.completed += 1
        elapsed = time.time() - self.start_time
        rate = self.completed / elapsed if elapsed > 0 else float('inf')
        if self.task_num > 0:
            frac = self.completed / float(self.task_num)
            eta = int(elapsed * (1 - frac) / frac + 0.5)
            done = int(self.bar_width * frac)
            sys.stdout.write('\033[2F\033[J')              # two lines up, clear to the end of the screen
            sys.stdout.write('[{}] {}/{}, {:.1f} task/s, elapsed: {}s, ETA: {:5}s\n{}\n'.format(
                '>' * done + '-' * (self.bar_width - done), self.completed, self.task_num, rate, int(elapsed + 0.5),
                eta, msg))
        else:
            sys.stdout.write('completed: {}, elapsed: {}s, {:.1f} tasks/s'.format(self.completed, int(elapsed + 0.5), rate))
        sys.stdout.flush()


# ---- device-side per-frame tail ----------------------------------------------------
def frame_metrics(sr, gt, min_max=(0, 1), need_img=False):
    """PSNR / SSIM of a super-resolved frame against its ground truth, computed on the GPU.

    The per-frame tail of test_dynavsr.py:285-292: ``img = tensor2img(sr, mode='rgb')``,
    ``calculate_psnr(img, hr_image)``, ``calculate_ssim(img, hr_image)`` with ``hr_image = tensor2img(GT)``
    (codes/utils/util.py:112-142, :262-269, :271-313) -- same quantisation, same float64 SSIM -- without
    moving the fp32 frame to the host.  sr, gt: [3,H,W] or [1,3,H,W] fp32 tensors on the GPU (there is no
    CPU path: calculate_psnr / calculate_ssim above are the host-side helpers).  Returns (psnr, ssim) or, with
    need_img, (psnr, ssim, uint8 HWC RGB numpy image of sr for the PNG writer); need_img=None returns the DEVICE
    tensor [mse, ssim] (float64) without synchronising -- psnr = 20 log10(255 / sqrt(mse)) (validation loops collect
    these per frame and convert once)."""
    import torch
    from dynavsr_amd import _lib as L
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
    if need_img is None:        # device result, no host synchronisation: [mse of the uint8 images, mean SSIM] (float64)
        return out
    mse, ssim_v = out.tolist()
    psnr = float('inf') if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))
    if need_img:
        a = img.cpu().numpy()
        return psnr, ssim_v, (a[:, :, 0] if c == 1 else a)
    return psnr, ssim_v
