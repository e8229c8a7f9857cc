"""ORACLE (test infrastructure only -- never imported by dynavsr_amd/).

CPU restatement of Degradation (codes/data/random_kernel_generator.py): build_kernel :19-48, kernel_shift
:51-76 (with ``int`` for the reference's removed ``np.int`` :72), apply :84-130 (ReflectionPad2d(K // 2) +
conv2d(groups = C, stride = scale) with the same kernel on every channel; one kernel per frame when the kernel
array is 3-D, frame i of a T + 2 clip using kernel (i - 1) mod T), and the 8-bit round trip of
vsrbase.py:185.  Pinned by tests/golden/degradation.npz, produced by running the reference's class on CPU
(oracle/gen_golden.py; ``np.int`` is restored as an alias of ``int`` for that run only)."""
import numpy as np
from scipy import ndimage


def build_kernel(kernel_size, theta, sigma):
    if sigma[0] == 0 and sigma[1] == 0:
        k = np.zeros((kernel_size, kernel_size))
        k[kernel_size // 2, kernel_size // 2] = 1
        return k
    r = kernel_size // 2
    ax = np.linspace(-r, r, kernel_size)
    xx, yy = np.meshgrid(ax, ax)
    c2, s2 = np.cos(theta) ** 2, np.sin(theta) ** 2
    sx2, sy2 = 2.0 * sigma[0] ** 2, 2.0 * sigma[1] ** 2
    a = c2 / sx2 + s2 / sy2
    b = np.sin(theta) * np.cos(theta) * (1.0 / sy2 - 1.0 / sx2)
    c = s2 / sx2 + c2 / sy2
    k = np.exp(-(a * xx ** 2 + 2.0 * b * xx * yy + c * yy ** 2))
    return k / k.sum()


def kernel_shift(kernel, scale):
    com = np.array(ndimage.center_of_mass(kernel))
    want = np.array(kernel.shape) // 2 + 0.5 * (scale - (kernel.shape[0] % 2))
    shift = want - com
    kernel = np.pad(kernel, int(np.ceil(np.max(shift))) + 1, 'constant')
    return ndimage.shift(kernel, shift)


def apply(img, kernel, scale, quantise=False):
    """img: numpy [N,C,H,W] float32; kernel: [K0,K0] or [T,K0,K0] float64 (unshifted).  fp32 weights like the
    reference's `.float()`, accumulation in float64 (the GPU / torch results are compared with a tolerance)."""
    img = np.asarray(img, dtype=np.float32)
    n, c, h, w = img.shape
    out = []
    for i in range(n):
        if kernel.ndim == 2:
            k = kernel
        else:
            t = kernel.shape[0]
            assert n == t or n == t + 2
            k = kernel[i] if n == t else kernel[(i - 1) % t]
        k = kernel_shift(k, scale).astype(np.float32).astype(np.float64)
        kk = k.shape[0]
        p = kk // 2
        x = np.pad(img[i].astype(np.float64), ((0, 0), (p, p), (p, p)), mode='reflect')
        ho, wo = (h + 2 * p - kk) // scale + 1, (w + 2 * p - kk) // scale + 1
        y = np.zeros((c, ho, wo))
        for dy in range(kk):
            for dx in range(kk):
                y += k[dy, dx] * x[:, dy:dy + (ho - 1) * scale + 1:scale, dx:dx + (wo - 1) * scale + 1:scale]
        out.append(y)
    out = np.stack(out, 0)
    if quantise:
        out = np.round(np.clip(out.astype(np.float32) * np.float32(255), 0, 255)) / np.float32(255)
    return out.astype(np.float32)
