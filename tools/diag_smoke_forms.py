#!/usr/bin/env python
"""Diagnostic (GPU): the smoke run of __graft_entry__ on the lattice path (5 agents, 2 seeds, 2 epochs) under both operand forms --
two f16 pieces (default) and three exact bf16 pieces -- and on other seed pairs: is the team-reward net's 8.8e-5 (of a 1e-4 bar)
the operand form, or one LeakyReLU knife edge of that particular run?   python tools/diag_smoke_forms.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import engine_checks as EC  # noqa: E402
from rcmarl_amd import capi  # noqa: E402

L = capi.load()
args = EC.make_args(["Cooperative"] * 5, H=1, n_episodes=6, max_ep_len=10, n_ep_fixed=3, n_epochs=2, buffer_size=40, seed=1)
for seeds in ((1, 2), (3, 4), (5, 6), (7, 8)):
    for mode, midfit in ((3, None), (3, "5"), (0, None)):
        L.rcmarl_lattice_set_f16_mode(mode)
        os.environ["RCMARL_LAT_F16"] = str(mode)
        if midfit:
            os.environ["RCMARL_MIDFIT"] = midfit
        else:
            os.environ.pop("RCMARL_MIDFIT", None)
        try:
            eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, "device", "cuda", None, seeds=seeds, lattice=True)
            w = EC.compare(eng, logs, o_logs, o_w, rtol_w=1e-3)
            print("seeds %s RCMARL_LAT_F16=%d RCMARL_MIDFIT=%s: critic %.2e  tr %.2e" % (seeds, mode, midfit, w["critic"], w["tr"]), flush=True)
        except AssertionError as e:
            print("seeds %s mode %d midfit %s: FAILED %s" % (seeds, mode, midfit, str(e)[:200]), flush=True)
L.rcmarl_lattice_set_f16_mode(-1)
