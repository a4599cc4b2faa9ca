#!/usr/bin/env python3
"""A/B builds of libxeno_hip.so for kernel experiments: recompile the named sources with extra -D flags and link them
with the unchanged objects of the regular build into scripts/devtools/_build/libxeno_<tag>.so (git-ignored; travels
with gpurun).  Select at run time with XV_LIB_PATH=<that file> (xenoverse_amd/_lib.py honours it; measurement only).

  python scripts/devtools/build_variant.py nt linds.hip -DXV_LINDS_NT_STORES=1
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xenoverse_amd import build as xb   # noqa: E402


def main():
    tag = sys.argv[1]
    srcs = [a for a in sys.argv[2:] if a.endswith(".hip")]
    flags = [a for a in sys.argv[2:] if not a.endswith(".hip")]
    xb.build_lib()                                   # the regular objects
    out_dir = os.path.join(ROOT, "scripts", "devtools", "_build")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for s in xb.sources():
        if s in srcs:
            o = os.path.join(out_dir, "%s_%s.o" % (s[:-4], tag))
            cmd = [xb._hipcc()] + xb.FLAGS + flags + ["-c", os.path.join(xb.CSRC, s), "-o", o]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.exit(r.stderr)
            objs.append(o)
        else:
            objs.append(os.path.join(xb.OBJ, s[:-4] + ".o"))
    lib = os.path.join(out_dir, "libxeno_%s.so" % tag)
    r = subprocess.run([xb._hipcc(), "-shared", "-fPIC", "--offload-arch=" + xb.ARCH, "-o", lib] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr)
    print(lib)


if __name__ == "__main__":
    main()
