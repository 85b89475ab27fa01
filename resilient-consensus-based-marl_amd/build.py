"""Build the HIP shared library (librcmarl_hip.so) in-tree for gfx950.

    python -m rcmarl_amd.build            # or __graft_entry__.build()

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the repo snapshot.
"""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "librcmarl_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stamp():
    h = hashlib.sha256()
    for p in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(CSRC, "*.inc"))):
        h.update(p.encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update((" ".join(FLAGS) + " -no-hip-rt").encode())
    return h.hexdigest()


def generate_sources(verbose=True):
    """The selection networks of the resilient aggregation are generated, not tracked (csrc/gen_selnet.py)."""
    sys.path.insert(0, CSRC)
    try:
        import gen_selnet
        return gen_selnet.ensure_generated(CSRC, verbose)
    finally:
        sys.path.remove(CSRC)


def build_hip(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    generate_sources(verbose)
    stamp_file = os.path.join(LIBDIR, ".stamp")
    stamp = _stamp()
    if not force and os.path.exists(lib_path()) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return lib_path()
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [HIPCC] + FLAGS + ["-I", CSRC, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, _sources()))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt", "-o", lib_path()] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return lib_path()


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv))
