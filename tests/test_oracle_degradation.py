"""oracle/degradation.py and the host half of dynavsr_amd.data.random_kernel_generator.Degradation against
tests/golden/degradation.npz (outputs of the reference's class run on CPU, oracle/gen_golden.py)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import degradation as od

CASES = ["s4_aniso", "s2_aniso", "s4_iso", "s4_delta", "s2_k11", "s2_k10_even"]


def frames(seed_shape):
    seed, shape = int(seed_shape[0]), tuple(int(v) for v in seed_shape[1:])
    return np.random.RandomState(seed).rand(*shape).astype(np.float32)


@pytest.mark.parametrize("tag", CASES)
def test_kernel_and_apply_match_reference(tag):
    g = load_golden("degradation")
    ks, scale, theta, sx, sy = g[tag + "__params"]
    k = od.build_kernel(int(ks), theta, [sx, sy])
    assert np.abs(k - g[tag + "__kernel"]).max() < 1e-15
    assert np.abs(od.kernel_shift(k, int(scale)) - g[tag + "__shifted"]).max() < 1e-15
    lr = od.apply(frames(g[tag + "__seed_shape"]), k, int(scale))
    assert lr.shape == g[tag + "__lr"].shape
    assert np.abs(lr - g[tag + "__lr"]).max() < 2e-6


def test_per_frame_kernels_and_chain():
    g = load_golden("degradation")
    kset = g["perframe__kernels"]
    for n in (5, 7):     # T frames, and DUF's T + 2 with kernel (i - 1) mod T
        lr = od.apply(frames(g["perframe%d__seed_shape" % n]), kset, 4)
        assert np.abs(lr - g["perframe%d__lr" % n]).max() < 2e-6
    ks, scale, theta, sx, sy = g["chain__params"]
    k = od.build_kernel(int(ks), theta, [sx, sy])
    lr = od.apply(frames(g["chain__seed_shape"]), k, int(scale), quantise=True)
    # the 8-bit round trip may flip a level where the fp32 sums differ in the last bit: at most a few samples
    d = np.abs(lr - g["chain__lr"])
    assert d.max() <= 1.0 / 255 + 1e-7 and (d > 1e-7).mean() < 1e-3
    slr = od.apply(g["chain__lr"], k, int(scale))
    assert np.abs(slr - g["chain__slr"]).max() < 2e-6


def test_host_class_matches_reference_kernels():
    """The product's Degradation: kernel construction and shift on the host (no GPU needed)."""
    from dynavsr_amd.data.random_kernel_generator import Degradation
    import torch
    g = load_golden("degradation")
    for tag in CASES:
        ks, scale, theta, sx, sy = g[tag + "__params"]
        d = Degradation(int(ks), int(scale), theta=theta, sigma=[sx, sy])
        assert np.abs(d.get_kernel() - g[tag + "__kernel"]).max() < 1e-15
        assert np.abs(d.kernel_shift(d.kernel) - g[tag + "__shifted"]).max() < 1e-15
    d.set_parameters([1.0, 2.0], 0.5)
    d.build_kernel()
    assert np.abs(d.kernel - od.build_kernel(int(ks), 0.5, [1.0, 2.0])).max() < 1e-15
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d.apply(torch.zeros(1, 3, 32, 32))
