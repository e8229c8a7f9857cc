"""CPU: the C-ABI library loads and exports exactly the symbols include/dynavsr_hip.h declares
(no compute calls -- there is no GPU here), and host-side argument checking works."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dynavsr_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        from dynavsr_amd import build
        build.build()
    return _lib


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dynavsr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dvsr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 15
    out = subprocess.run(["nm", "-D", "--defined-only", lib.SO_PATH], stdout=subprocess.PIPE, text=True,
                         check=True).stdout
    exported = set(re.findall(r" T (dvsr_[a-z0-9_]+)", out))
    missing = [s for s in syms if s not in exported]
    assert not missing, "declared in include/dynavsr_hip.h but not exported: %s" % missing
    undeclared = sorted(exported - set(syms))
    assert not undeclared, "exported but not declared in the header: %s" % undeclared


def test_ctypes_signatures_cover_header(lib):
    l = lib.lib()
    assert l.dvsr_version() >= 100
    assert sorted(l._signatures) == declared_symbols()


def test_plan_create_and_errors(lib):
    l = lib.lib()
    h = ctypes.c_void_p()
    cfg = lib.EdvrConfig(64, 5, 8, 5, 10, 4, 2)
    assert l.dvsr_edvr_plan_create(cfg, 1, 64, 64, ctypes.byref(h)) == 0
    assert l.dvsr_edvr_num_params(h) == 144
    assert l.dvsr_edvr_workspace_bytes(h, 1) > 0
    off, n = ctypes.c_longlong(), ctypes.c_longlong()
    assert l.dvsr_edvr_tensor_info(h, b"aligned", ctypes.byref(off), ctypes.byref(n)) == 0
    assert n.value == 5 * 64 * 64 * 64
    assert l.dvsr_edvr_tensor_info(h, b"nope", ctypes.byref(off), ctypes.byref(n)) == -1
    l.dvsr_edvr_plan_destroy(h)
    assert l.dvsr_edvr_plan_create(cfg, 1, 66, 64, ctypes.byref(h)) == -1
    assert b"multiples of 4" in l.dvsr_last_error()
    bad = lib.EdvrConfig(64, 5, 7, 5, 10, 4, 2)
    assert l.dvsr_edvr_plan_create(bad, 1, 64, 64, ctypes.byref(h)) == -1
    big = lib.EdvrConfig(128, 7, 8, 5, 40, 4, 3)     # EDVR-L (BASELINE.json configs[4])
    assert l.dvsr_edvr_plan_create(big, 1, 64, 64, ctypes.byref(h)) == 0
    assert l.dvsr_edvr_num_params(h) == 264
    l.dvsr_edvr_plan_destroy(h)


def test_conv_desc_validation_without_gpu(lib):
    l = lib.lib()
    d = lib.Conv2dDesc()
    assert l.dvsr_conv2d_forward(ctypes.byref(d), None) == -1
    assert b"null" in l.dvsr_last_error()


def test_product_does_not_import_oracle():
    """The product package must never reach into oracle/ (parity claims depend on it)."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "dynavsr_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
