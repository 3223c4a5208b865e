"""ComputePowers latency for 1 vs 4 bundle indices (16M-4096): wall time and per-class kernel times"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from bench import SEED0
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", "16M-4096.json")).read()
ctx = apsu_amd.HeContext(js)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(4)])
rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
sd = torch.from_numpy(src.view(np.int64)).cuda()
for idx in ([0], [0, 1], [0, 1, 2, 3]):
    sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx]
    for _ in range(3): pw = ctx.compute_powers(idx, sp, rk, on_device=True)
    ctx.sync() if hasattr(ctx, "sync") else torch.cuda.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        pw = None                                         # as in a query loop: the previous powers are released first (buffer pool)
        pw = ctx.compute_powers(idx, sp, rk, on_device=True)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 50
    ctx.profile_enable(1); ctx.profile_read()
    for _ in range(5):
        pw = None
        pw = ctx.compute_powers(idx, sp, rk, on_device=True)
    torch.cuda.synchronize(); p = ctx.profile_read(); ctx.profile_enable(0)
    ks = sum(v[0] for v in p.values()) / 5
    print(f"nb={len(idx)}: wall {wall:.3f} ms/call, kernel sum {ks:.3f} ms, launches {sum(v[1] for v in p.values())//5}: " +
          ", ".join(f"{k} {v[0]/5:.3f}/{v[1]//5}" for k, v in p.items() if v[1]), flush=True)
