"""GPU diagnostic: fused vs unfused local fit inside the engine at the bench shape (per-seed differences)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rcmarl_amd import capi

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w = bench.WORKLOADS["cfg4_shard"]
eng = bench.make_engine(w, S, [1000 + k for k in range(S)], capi.load())
eng.rollout_block(eng.cfg.n_ep_fixed)
B = eng.B
eng._lattice_encode(B)
L = eng.lib
rptr, rstride = eng._x("r")
L.rcmarl_team_reward(rptr, rstride, eng.coop.data_ptr(), max(eng.n_coop, 1), eng.rcoop.data_ptr(), S, eng.N, B, eng.ldb, eng.stream)
L.rcmarl_gather_agent_major(rptr, rstride, eng.rcoop.data_ptr(), eng.fit_mode.data_ptr(), eng.ybuf["r_fit"].data_ptr(), S, eng.N, B, eng.ldb, eng.stream)
for net, xkey, ykey in (("tr", "sa", "r_fit"), ("critic", "s", "r_fit")):
    out = {}
    for fused in (True, False):
        saved = dict(eng.lat_wpf)
        if not fused:
            eng.lat_wpf = {}
        eng.msg[net].copy_(eng.theta[net])
        eng.a1_cached[net] = False
        eng._local_fit(net, xkey, eng.ybuf[ykey], B, eng.coop)
        torch.cuda.synchronize()
        out[fused] = (eng.msg[net].clone(), eng.partials.clone(), eng.lat_dzp_f[xkey].clone())
        eng.lat_wpf = saved
    a, b = out[True][0], out[False][0]
    print(net, "finite fused/unfused:", bool(torch.isfinite(a).all()), bool(torch.isfinite(b).all()))
    d = (a - b).abs().amax(dim=(1, 2)).cpu().numpy()
    print("  per-seed max |msg diff|:", np.array2string(d, precision=3))
    bad = (~torch.isfinite(a)).any(dim=2).cpu().numpy()
    print("  non-finite (seed, agent) count:", int(bad.sum()), "seeds:", np.nonzero(bad.any(1))[0][:16], "agents:", np.nonzero(bad.any(0))[0][:20])
    dz = (out[True][2] != out[False][2]).view(S, -1).sum(1).cpu().numpy()
    print("  dz bytes differing per seed (last step):", dz)
