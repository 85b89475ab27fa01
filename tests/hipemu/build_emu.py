"""Compile the product's kernel sources with g++ against the hipemu headers
-> tests/hipemu/librcmarl_emu.so (TEST INFRASTRUCTURE; CPU emulation of the
HIP kernels so their indexing/LDS/barrier logic runs in `-m "not gpu"` tests)."""
import glob
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "resilient-consensus-based-marl_amd", "csrc")
OUT = os.path.join(HERE, "librcmarl_emu.so")
FLAGS = ["-O1", "-g0", "-std=c++17", "-fPIC", "-DRCMARL_EMU", "-x", "c++", "-Wno-attributes", "-Wno-unknown-pragmas",
         "-ffp-contract=off"]


def build_emu(force=False):
    import sys
    sys.path.insert(0, CSRC)
    try:
        import gen_selnet
        gen_selnet.ensure_generated(CSRC)          # generated sources (selection networks) are not tracked
    finally:
        sys.path.remove(CSRC)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    h = hashlib.sha256()
    for p in srcs + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(CSRC, "*.inc"))) + \
            [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "include/hip/hip_runtime.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    stamp_file = OUT + ".stamp"
    if not force and os.path.exists(OUT) and os.path.exists(stamp_file) and open(stamp_file).read() == h.hexdigest():
        return OUT
    objdir = os.path.join(HERE, "obj")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        r = subprocess.run(["g++"] + FLAGS + ["-I", os.path.join(HERE, "include"), "-I", CSRC, "-c", src, "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ (hipemu) failed for %s:\n%s" % (src, r.stderr[-6000:]))
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, srcs + [os.path.join(HERE, "hipemu.cpp")]))
    r = subprocess.run(["g++", "-shared", "-o", OUT] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    with open(stamp_file, "w") as f:
        f.write(h.hexdigest())
    return OUT


if __name__ == "__main__":
    print(build_emu(True))
