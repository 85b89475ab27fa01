#!/usr/bin/env python
"""Variant build of the product library for A/B kernel measurements (tools/kbench.py mid_ab):
    python tools/build_variant.py NAME SOURCE.hip -DFOO=1 ...   ->  resilient-consensus-based-marl_amd/lib/variants/libNAME.so
Only SOURCE (one file, a comma-separated list, or "all") is recompiled with the extra flags; the other objects are the product
build's (lib/obj)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcmarl_amd import build as B  # noqa: E402


def main():
    name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build_hip()
    objdir = os.path.join(B.LIBDIR, "obj")
    outdir = os.path.join(B.LIBDIR, "variants")
    os.makedirs(outdir, exist_ok=True)
    srcs = src.split(",")                              # several sources: "a.hip,b.hip" ("all": every source of the library)
    if src == "all":
        srcs = [os.path.basename(o)[:-2] for o in glob.glob(os.path.join(objdir, "*.o"))]
    objs = []
    for sname in srcs:
        obj = os.path.join(outdir, name + "_" + os.path.basename(sname) + ".o")
        subprocess.run([B.HIPCC] + B.FLAGS + extra + ["-I", B.CSRC, "-c", os.path.join(B.CSRC, sname), "-o", obj], check=True,
                       stderr=subprocess.DEVNULL)
        objs.append(obj)
    mine = {os.path.basename(sname) + ".o" for sname in srcs}
    others = [o for o in glob.glob(os.path.join(objdir, "*.o")) if os.path.basename(o) not in mine]
    out = os.path.join(outdir, "lib%s.so" % name)
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt", "-o", out] + objs + others, check=True)
    print(out)


if __name__ == "__main__":
    main()
