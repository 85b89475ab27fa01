import sys, torch, numpy as np
sys.path.insert(0, '.')
from rcmarl_amd import capi
L = capi.load()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for S, n, ep, B in ((512, 1, 10, 3000), (512, 3, 10, 3000), (512, 1, 10, 1000), (64, 1, 10, 3000), (512, 1, 1, 1000)):
    seeds = torch.arange(1000, 1000 + S, dtype=torch.int64, device="cuda")
    calls = torch.arange(n, dtype=torch.int32, device="cuda")
    perm = torch.empty(S, n, ep, B, dtype=torch.int32, device="cuda")
    t = timeit(lambda: L.rcmarl_shuffle_perms(seeds.data_ptr(), calls.data_ptr(), n, ep, B, perm.data_ptr(), S, st))
    print("shuffle S=%d n=%d epochs=%d B=%d: %8.1f us  (%d permutations, %.3f us each)" % (S, n, ep, B, t, S * n * ep, t / (S * n * ep)))
