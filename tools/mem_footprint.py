"""HBM footprint of the engine: DB, powers, workspace (torch.cuda.mem_get_info before / after), per parameter set"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from bench import SEED0, WORKLOADS
cfg = sys.argv[1] if len(sys.argv) > 1 else "16M-4096"
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
ctx = apsu_amd.HeContext(js)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
free1, _ = torch.cuda.mem_get_info()
bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in units]
torch.cuda.synchronize(); free2, _ = torch.cuda.mem_get_info()
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(ctx.bundle_idx_count)])
rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)])) if K > 1 else None
masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)
idxs = list(range(ctx.bundle_idx_count))
for _ in range(2):
    pw = ctx.compute_powers(idxs, [[src[b, s] for s in range(ns)] for b in idxs], rk)
    out = ctx.eval_bundles(bl, pw, rk, [masks[i] for i in range(len(units))])
torch.cuda.synchronize(); free3, _ = torch.cuda.mem_get_info()
gib = 1 << 30
print(f"{cfg}: HBM total {total/gib:.0f} GiB; context (tables, arenas) {(free0-free1)/gib:.2f} GiB; DB of {len(units)} BinBundles {(free1-free2)/gib:.2f} GiB "
      f"(engine reports {sum(b.db_bytes for b in bl)/gib:.2f}); powers + grown workspace after two queries {(free2-free3)/gib:.2f} GiB; in use {(free0-free3)/gib:.2f} GiB")
