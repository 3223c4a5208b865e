"""k_mac time of the 16M-4096 evaluation across re-allocations of the database inside ONE process (same kernels, same
bytes): is the run-to-run spread of the MAC a property of where the BinBundles land in HBM?"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from bench import SEED0, WORKLOADS
cfg = "16M-4096"
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
ctx = apsu_amd.HeContext(js)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(ctx.bundle_idx_count)])
rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)
sd = torch.from_numpy(src.view(np.int64)).cuda(); md = torch.from_numpy(masks.view(np.int64)).cuda()
out = torch.zeros((len(units), 2, n), dtype=torch.int64, device="cuda")
idx = list(range(ctx.bundle_idx_count))
sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx]
mp = [md.data_ptr() + i * n * 8 for i in range(len(units))]
junk = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in units]
    for _ in range(3):
        pw = ctx.compute_powers(idx, sp, rk, on_device=True)
        ctx.eval_bundles(bl, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
    ctx.profile_enable(1)
    ctx.profile_read(reset=True)
    for _ in range(4):
        pw = ctx.compute_powers(idx, sp, rk, on_device=True)
        ctx.eval_bundles(bl, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
    p = ctx.profile_read(reset=True)
    ctx.profile_enable(0)
    print("allocation %d: k_mac %.4f ms per query (%.0f GB/s)  first bundle at %#x" %
          (trial, p["dyadic_mac"][0] / 4, p["dyadic_mac"][2] * n / 8 / (p["dyadic_mac"][0] / 4) / 1e6 / 4, 0), flush=True)
    del bl, pw
    gc.collect()
    # perturb the allocator so that the next database does not land on the same pages
    junk.append(torch.empty((trial + 1) * 37 * 1024 * 1024, dtype=torch.uint8, device="cuda"))
