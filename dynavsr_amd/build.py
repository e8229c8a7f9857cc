"""Builds libdynavsr_hip.so (hipcc, gfx950 only) in-tree next to this file.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored
but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
SO = os.path.join(HERE, "libdynavsr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# per-file flags.  conv2d_wino4.hip: the SLP vectoriser turns the split's residual subtractions into v_pk_add_f32 on register
# pairs it has to assemble with moves (36 v_mov per chunk) -- and packed fp32 is slower beside MFMAs anyway (MI355X guide).
EXTRA = {"conv2d_wino4.hip": ["-fno-slp-vectorize"], "conv2d_wino5.hip": ["-fno-slp-vectorize"], "mdcn_split.hip": ["-fno-slp-vectorize"],
         "conv2d_wgrad_bf16.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "dynavsr_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build_trace():
    """Debug library with the conv pipeline's cycle stamps compiled in (tools/conv_trace.py)."""
    so = os.path.join(HERE, "libdynavsr_hip_trace.so")
    srcs = [os.path.join(CSRC, s) for s in _sources()]
    r = subprocess.run([HIPCC] + FLAGS + ["-fno-slp-vectorize", "-DDVSR_CONV_TRACE", "-shared", "-o", so] + srcs,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise RuntimeError("trace build failed:\n" + r.stdout)
    return so


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _deps_mtime()
    jobs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + EXTRA.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stdout))
        if verbose and r.stdout.strip():
            print(r.stdout)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in _sources()]
    if jobs or not os.path.exists(SO):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError("link failed:\n" + r.stdout)
        # a kernel template whose host stub was not instantiated links fine and fails at dlopen: check here, not on the GPU box
        import ctypes
        try:
            ctypes.CDLL(SO)
        except OSError as e:
            raise RuntimeError("libdynavsr_hip.so does not load: %s" % e)
    return SO


if __name__ == "__main__":
    if "--trace" in sys.argv:
        print(build_trace())
    else:
        print(build(force="--force" in sys.argv, verbose=True))
