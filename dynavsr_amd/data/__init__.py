"""`data` package face for the Level-1 drop-in: with ``dynavsr_amd`` first on PYTHONPATH this
directory is what ``import data`` finds, but the drivers' dataset code -- ``data.data_sampler``,
``data.meta_learner`` (test_dynavsr.py:15-19) and everything those import (``data.common``,
``data.random_kernel_generator`` inside the DataLoader workers, on CPU tensors) -- is the reference's own and
out of scope here, so every other ``data`` directory on the path is searched FIRST and this one last.
The device-side ``Degradation`` is an explicit opt-in: ``dynavsr_amd.data.random_kernel_generator``.
"""
import pkgutil as _pkgutil

_own = list(__path__)
__path__ = [p for p in _pkgutil.extend_path(__path__, __name__) if p not in _own] + _own
