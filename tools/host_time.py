"""host-side enqueue time of ComputePowers / eval_bundles vs GPU time (is the path host-bound?)"""
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
for idx in ([0], [0, 1, 2, 3]):
    sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx]
    for _ in range(3): pw = ctx.compute_powers(idx, sp, rk, on_device=True)
    torch.cuda.synchronize()
    host, wall = [], []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pw = ctx.compute_powers(idx, sp, rk, on_device=True)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
    print(f"nb={len(idx)}: host enqueue median {sorted(host)[5]:.3f} ms, wall median {sorted(wall)[5]:.3f} ms", flush=True)
