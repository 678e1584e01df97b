"""Build librendernet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m rendernet_amd.build [--force]

Objects and the shared library land in rendernet_amd/lib/ (git-ignored, shipped to the GPU box
by gpurun).  A source-hash stamp makes rebuilds incremental.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "librendernet_hip.so"
SOURCES = ["capi.hip", "conv_igemm.hip", "conv_wino.hip", "conv_wino_wgrad.hip", "conv_wino43.hip", "conv_wino_bf3.hip", "conv_wino_bf3_wgrad.hip", "conv3d_wino_bf3.hip", "conv_wino43_wgrad.hip", "conv3d_drun.hip", "conv_direct.hip", "conv_tiled.hip", "resample.hip", "resample_tiled.hip", "misc_kernels.hip",
           "conv_wgrad.hip", "train_kernels.hip", "resample_bwd.hip"]
HEADERS = ["rn_common.h", "wino_mats.h", os.path.join("..", "..", "include", "rendernet_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# the resampler must round after every multiply and add (bit parity with the reference's op-by-op
# TF graph): hipcc's default -ffp-contract=fast would fuse them into FMAs
EXTRA_FLAGS = {"resample.hip": ["-ffp-contract=off"], "resample_tiled.hip": ["-ffp-contract=off"],
               "resample_bwd.hip": ["-ffp-contract=off"]}


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def _digest(paths):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dig = _digest([src] + hdrs) + "|" + " ".join(EXTRA_FLAGS.get(s, []))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((src, obj, stamp, dig, EXTRA_FLAGS.get(s, [])))

    def compile_one(job):
        src, obj, stamp, dig, extra = job
        cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(dig)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, jobs))
    so = lib_path()
    if jobs or not os.path.exists(so):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        # an undefined symbol (e.g. a kernel whose host stub the compiler dropped) must fail the BUILD, not
        # the first call on the GPU box.  Checked in a child process: loading the library here would pull the
        # system libamdhip64 into THIS process before torch loads its own copy (two HIP runtimes, no devices).
        r = subprocess.run([sys.executable, "-c", "import ctypes,sys; ctypes.CDLL(sys.argv[1])", so],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("%s does not load:\n%s" % (so, r.stderr))
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
