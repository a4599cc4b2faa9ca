"""Build libxeno_hip.so (gfx950) in-tree with hipcc.  `python -m xenoverse_amd.build [--force]`.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot; hipcc cross-compiles here
without a GPU.  -ffp-contract=off: the kernels spell out every FMA (fmaf / fma), so that float results are
reproducible and equal the CPU oracle's, which is built with the same flag.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libxeno_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-value"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


KERNELS_END = b"#ifndef XV_KERNELS_ONLY"


def source_hash(names):
    """sha256[:16] of the named csrc files, in order: ties a committed profile to the KERNEL source it measured.  A .hip file
    counts up to its `#ifndef XV_KERNELS_ONLY` line — structs, device functions and kernels (what mixed.hip includes of it);
    the host code behind it (handles, graphs, the C-ABI) does not change what a launch moves."""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(CSRC, n), "rb") as f:
            text = f.read()
        cut = text.find(KERNELS_END) if n.endswith(".hip") else -1
        h.update(text if cut < 0 else text[:cut])
    return h.hexdigest()[:16]


_INC = None


def _includes(path, seen):
    """the csrc / include files `path` pulls in through #include "..." (recursively; .hip files include .hip files:
    mixed.hip is built from the three family sources)"""
    import re
    global _INC
    if _INC is None:
        _INC = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)
    with open(path) as f:
        text = f.read()
    for name in _INC.findall(text):
        for d in (os.path.dirname(path), CSRC, os.path.join(os.path.dirname(HERE), "include")):
            q = os.path.join(d, name)
            if os.path.exists(q):
                if q not in seen:
                    seen.add(q)
                    _includes(q, seen)
                break
    return seen


def _deps_mtime(src_path):
    deps = _includes(src_path, set())
    deps.add(os.path.join(os.path.dirname(HERE), "include", "xeno.h"))
    return max(os.path.getmtime(d) for d in deps)


def _compile(src, force, extra):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    s = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(s)
            and os.path.getmtime(obj) > _deps_mtime(s)):
        return obj, False
    cmd = [_hipcc()] + FLAGS + extra + ["-c", s, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build_lib(force=False, extra_flags=(), verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, list(extra_flags)), srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("linked", LIB)
    return LIB


FAST_SRC = os.path.join(CSRC, "xvfast.c")
FAST_LIB = os.path.join(HERE, "_xvfast.so")


def build_fast(force=False, verbose=False):
    """the CPython trampoline of the eager step loop (csrc/xvfast.c, plain C, gcc): xenoverse_amd/_xvfast.so, in-tree like the
    HIP library.  Optional at run time — without it the calls go through ctypes (same C-ABI, ~3 us more per step)."""
    import sysconfig
    if not force and os.path.exists(FAST_LIB) and os.path.getmtime(FAST_LIB) > os.path.getmtime(FAST_SRC):
        return FAST_LIB
    inc = sysconfig.get_paths()["include"]
    if not os.path.exists(os.path.join(inc, "Python.h")):
        raise RuntimeError("Python.h not found under %s: _xvfast.so not built (the ctypes path serves)" % inc)
    cc = os.environ.get("CC", "gcc")
    cmd = [cc, "-O2", "-fPIC", "-shared", "-Wall", "-I", inc, FAST_SRC, "-o", FAST_LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building _xvfast.so failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", FAST_LIB)
    return FAST_LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    try:
        print(build_fast(force="--force" in sys.argv, verbose=True))
    except Exception as ex:
        sys.stderr.write("note: %s\n" % (ex,))
