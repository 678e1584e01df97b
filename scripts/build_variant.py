#!/usr/bin/env python
"""Link a VARIANT of the kernel library for same-box A/B timing: every object of the current build (rendernet_amd/lib/*.o) except the ones
replaced on the command line, which are compiled from the given source files.
    python scripts/build_variant.py <name> conv3d_wino_bf3.hip=/tmp/candidate.hip [other.hip=/path ...] [-DMACRO ...]
-> scripts/_build/librendernet_hip_<name>.so; run with RN_HIP_LIBRARY=scripts/_build/librendernet_hip_<name>.so.  Development tool."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rendernet_amd import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    repl, defs = {}, []
    for a in sys.argv[2:]:
        if a.startswith("-D"):
            defs.append(a)
        else:
            k, v = a.split("=", 1)
            repl[k] = v
    B.build(verbose=False)
    out = os.path.join(ROOT, "scripts", "_build")
    os.makedirs(out, exist_ok=True)
    objs = []
    for s in B.SOURCES:
        obj = os.path.join(B.LIBDIR, s.replace(".hip", ".o"))
        if s in repl:
            obj = os.path.join(out, "%s_%s.o" % (name, s.replace(".hip", "")))
            subprocess.check_call([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(s, []) + defs +
                                  ["-I", B.CSRC, "-I", os.path.join(ROOT, "include"), "-c", repl[s], "-o", obj])
        objs.append(obj)
    so = os.path.join(out, "librendernet_hip_%s.so" % name)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    print(so)


if __name__ == "__main__":
    main()
