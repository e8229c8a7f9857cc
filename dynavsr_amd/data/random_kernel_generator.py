"""Synthetic degradation (blur + down-sampling) of a clip: the reference's ``Degradation`` class
(codes/data/random_kernel_generator.py:7-130) with its tensor part on the GPU.

Host side (tiny, stays numpy/scipy like the reference's): the anisotropic Gaussian of ``build_kernel``
(:19-48) and the centre-of-mass shift of ``kernel_shift`` (:51-76; the reference's ``np.int`` at :72 does not
exist in numpy >= 1.24, ``int`` is what it meant).  Device side: ``apply`` (:84-130) -- reflection padding and
the depth-wise strided correlation -- is one launch of ``dvsr_degrade_apply``; ``apply(img, quantise=True)``
additionally fuses the 8-bit round trip that vsrbase.py:185 performs between the two applications.
There is no CPU path for ``apply``: tensors must live on the GPU."""
import numpy as np
import torch
from scipy import ndimage

from dynavsr_amd import _lib as L


class Degradation:
    def __init__(self, kernel_size, scale_factor, theta=0.0, sigma=[1.0, 1.0]):
        self.kernel_size = kernel_size
        self.scale = scale_factor
        self.theta = theta
        self.sigma = sigma
        self.build_kernel()

    def set_parameters(self, sigma, theta):
        self.sigma = sigma
        self.theta = theta

    def build_kernel(self):
        ks = self.kernel_size
        sx, sy = self.sigma[0], self.sigma[1]
        if sx == 0 and sy == 0:          # delta kernel: plain sub-sampling
            k = np.zeros((ks, ks))
            k[ks // 2, ks // 2] = 1
        else:
            r = ks // 2
            ax = np.linspace(-r, r, ks)
            xx, yy = np.meshgrid(ax, ax)
            ct, st = np.cos(self.theta), np.sin(self.theta)
            ix, iy = 1.0 / (2.0 * sx ** 2), 1.0 / (2.0 * sy ** 2)
            # quadratic form of the rotated covariance: a x^2 + 2 b x y + c y^2
            a = ct ** 2 * ix + st ** 2 * iy
            b = st * ct * (iy - ix)
            c = st ** 2 * ix + ct ** 2 * iy
            k = np.exp(-(a * xx ** 2 + 2.0 * b * xx * yy + c * yy ** 2))
            k = k / k.sum()
        self.kernel = k

    def kernel_shift(self, kernel):
        """Centre of mass -> kernel centre + half a (scale - parity) pixel, so that the down-sampled image stays
        aligned with the ground truth; zero-padded first so the cubic-spline shift loses nothing."""
        com = np.array(ndimage.center_of_mass(kernel))
        want = np.array(kernel.shape) // 2 + 0.5 * (self.scale - (kernel.shape[0] % 2))
        shift = want - com
        kernel = np.pad(kernel, int(np.ceil(np.max(shift))) + 1, 'constant')
        return ndimage.shift(kernel, shift)

    def _shifted_on(self, device):
        """The shifted kernel(s) as a [T,K,K] fp32 device tensor.  The reference redoes the spline shift in every
        apply(); the dataset applies the same kernel twice (HR -> LR -> SLR), so the result is kept until the
        kernel (compared by value) or the scale changes."""
        key = (self.kernel.tobytes(), self.kernel.shape, self.scale, str(device))
        if getattr(self, "_shift_key", None) != key:
            ks = self.kernel_shift(self.kernel)[None] if self.kernel.ndim == 2 else \
                np.stack([self.kernel_shift(k) for k in self.kernel], 0)
            self._shift_val = torch.from_numpy(np.ascontiguousarray(ks)).float().to(device)
            self._shift_key = key
        return self._shift_val

    def get_kernel(self):
        return self.kernel

    def set_kernel_directly(self, kernel):
        self.kernel = kernel

    def apply(self, img, quantise=False):
        if not img.is_cuda:
            raise RuntimeError("Degradation.apply runs on the GPU (libdynavsr_hip); there is no CPU fallback")
        single = img.dim() == 3          # one image C H W
        x = img[None] if single else img
        assert x.dim() == 4
        offset = 0
        if self.kernel.ndim == 3:        # T kernels: one per frame; DUF clips carry two extra frames
            t = self.kernel.shape[0]
            assert x.shape[0] == t or x.shape[0] == t + 2
            offset = 0 if x.shape[0] == t else -1
        kt = self._shifted_on(x.device)
        n, c, h, w = x.shape
        kk, s = kt.shape[-1], int(self.scale)
        pad = kk // 2
        ho, wo = (h + 2 * pad - kk) // s + 1, (w + 2 * pad - kk) // s + 1
        x = x.float().contiguous()
        out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
        L.check(L.lib().dvsr_degrade_apply(L.ptr(x), L.ptr(kt), L.ptr(out), n, c, h, w, kk, s, kt.shape[0], offset,
                                           1 if quantise else 0, L.stream()), "degrade_apply")
        return out[0] if single else out
