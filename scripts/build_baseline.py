#!/usr/bin/env python
"""Build the kernel library of another git revision into scripts/_build/librendernet_hip_<name>.so for same-box A/B timing
(boxes differ by a few per cent in sustained clock, so a variant is only ever compared with a baseline run in the SAME
gpurun call):   python scripts/build_baseline.py [rev=HEAD] [name=base]   then   RN_HIP_LIBRARY=scripts/_build/... python ..."""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rendernet_amd import build as B  # noqa: E402


def main():
    rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
    name = sys.argv[2] if len(sys.argv) > 2 else "base"
    out = os.path.join(ROOT, "scripts", "_build")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call("git archive %s rendernet_amd/csrc include | tar -x -C %s" % (rev, tmp), shell=True, cwd=ROOT)
        csrc = os.path.join(tmp, "rendernet_amd", "csrc")
        srcs = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))

        def cc(s):
            obj = os.path.join(tmp, s.replace(".hip", ".o"))
            subprocess.check_call([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(s, []) + ["-c", os.path.join(csrc, s), "-o", obj])
            return obj
        with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
            objs = list(ex.map(cc, srcs))
        so = os.path.join(out, "librendernet_hip_%s.so" % name)
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    print(so)


if __name__ == "__main__":
    main()
