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
# RCMARL_EMU_SANITIZE=address|undefined|address,undefined: a sanitizer build beside the plain one (own file name, own objects).  Run as
#   RCMARL_EMU_SANITIZE=address LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
#       python -m pytest tests/test_kernels_emu.py -q -p no:cacheprovider
# (the work-items are ucontext fibers on heap-allocated stacks: ASan follows them as heap memory; GPU ASan is not available on this pool)
SAN = os.environ.get("RCMARL_EMU_SANITIZE", "")
TAG = ("_" + SAN.replace(",", "_")) if SAN else ""
OUT = os.path.join(HERE, "librcmarl_emu%s.so" % TAG)
FLAGS = ["-O1", "-g0", "-std=c++17", "-fPIC", "-DRCMARL_EMU", "-x", "c++", "-Wno-attributes", "-Wno-unknown-pragmas",
         "-ffp-contract=off"]
if SAN:
    FLAGS = [f for f in FLAGS if f != "-g0"] + ["-g1", "-fsanitize=" + SAN, "-fno-omit-frame-pointer", "-fno-sanitize-recover=all"]


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
    import fcntl
    lock = open(OUT + ".lock", "w")                   # pytest-xdist workers: one builds, the others wait and find the stamp
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and os.path.exists(OUT) and os.path.exists(stamp_file) and open(stamp_file).read() == h.hexdigest():
            return OUT
        return _build_locked(srcs, h, stamp_file)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(srcs, h, stamp_file):
    objdir = os.path.join(HERE, "obj" + TAG)
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
    r = subprocess.run(["g++", "-shared", "-o", OUT] + (["-fsanitize=" + SAN] if SAN else []) + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    with open(stamp_file, "w") as f:
        f.write(h.hexdigest())
    return OUT


if __name__ == "__main__":
    print(build_emu(True))
