#!/usr/bin/env python
"""Register / LDS / scratch use of every kernel in a hipcc -save-temps assembly file (the .s of the gfx950 side).

    hipcc --offload-arch=gfx950 -O3 ... -c x.hip -o /tmp/x.o -save-temps=obj ;  python tools/kernel_regs.py /tmp/x-hip-amdgcn-amd-amdhsa-gfx950.s [substr]
"""
import re
import sys

txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if sub and sub not in name:
        continue
    g = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "?"])[1]
    print("%-100s vgpr %s accum_off %s sgpr %s lds %s scratch %s" % (name[:100], g("next_free_vgpr"), g("accum_offset"), g("next_free_sgpr"),
                                                                  g("group_segment_fixed_size"), g("private_segment_fixed_size")))
