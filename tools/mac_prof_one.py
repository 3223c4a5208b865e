"""a few 16M-4096 queries, ComputePowers on ONE stream and a host wait per query: the in-path k_mac launch (one per query: every
dyadic sum of the 28 BinBundles, 5.46 GB of bit-packed rows) runs with the chip to itself.  For `rocprofv3 --pmc ...` passes
(tools/collect_r05.sh mac) and for kernel traces; argv[1] = queries (default 4), argv[2] = config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from bench import SEED0, WORKLOADS
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = sys.argv[2] if len(sys.argv) > 2 else "16M-4096"
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
ctx = apsu_amd.HeContext(js)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in units]
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(ctx.bundle_idx_count)])
rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)
sd = torch.from_numpy(src.view(np.int64)).cuda(); md = torch.from_numpy(masks.view(np.int64)).cuda()
out = torch.zeros((len(units), 2, n), dtype=torch.int64, device="cuda")
idx = list(range(ctx.bundle_idx_count))
sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx]
mp = [md.data_ptr() + i * n * 8 for i in range(len(units))]
ctx.set_two_stream(0)
ctx.set_async_results(True)
torch.cuda.synchronize()
for _ in range(reps):
    pw = ctx.compute_powers(idx, sp, rk, on_device=True)
    ctx.eval_bundles(bl, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
    ctx.sync()
print("done: %d queries, %d BinBundles, %.2f GB of rows" % (reps, len(bl), sum(b.db_bytes for b in bl) / 1e9))
ctx.close()
