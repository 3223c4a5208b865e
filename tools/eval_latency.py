"""eval_bundles latency vs number of BinBundles of one bundle index (16M-4096)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from bench import SEED0
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", "16M-4096.json")).read()
ctx = apsu_amd.HeContext(js)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
src = np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)])
rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
sd = torch.from_numpy(src.view(np.int64)).cuda()
sp = [[sd.data_ptr() + (s * 2 * Lf * n) * 8 for s in range(ns)]]
bundles = [ctx.random_bundle(0, ci, D, SEED0 + ci) for ci in range(7)]
masks = rng.integers(0, t, (7, n), dtype=np.uint64); md = torch.from_numpy(masks.view(np.int64)).cuda()
out = torch.zeros((7, 2, n), dtype=torch.int64, device="cuda")
pw = ctx.compute_powers([0], sp, rk, on_device=True)
for nbun in (1, 2, 4, 7):
    bl = bundles[:nbun]; mp = [md.data_ptr() + i * n * 8 for i in range(nbun)]
    f = lambda: ctx.eval_bundles(bl, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 50
    ctx.profile_enable(1); ctx.profile_read()
    for _ in range(5): f()
    p = ctx.profile_read(); ctx.profile_enable(0)
    print(f"bundles={nbun}: wall {wall:.3f} ms, kernel sum {sum(v[0] for v in p.values())/5:.3f} ms, launches {sum(v[1] for v in p.values())//5}: " +
          ", ".join(f"{k} {v[0]/5:.3f}/{v[1]//5}" for k, v in p.items() if v[1]), flush=True)
