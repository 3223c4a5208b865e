"""host time to QUEUE one query (ComputePowers + evaluation, device-resident inputs, asynchronous results) against the device time
it takes: is a shard host-bound?  usage: python tools/host_enqueue_time.py [world]   (the engine lets the host run two queued
evaluations ahead of the device; the timed loops below stay within that)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from apsu_amd.sharding import partition
from bench import SEED0, WORKLOADS
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = "16M-4096"
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
ctx = apsu_amd.HeContext(js)
ctx.set_async_results(True); ctx.set_query_overlap(True)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
mine = partition(units, ctx.bundle_idx_count, world, ctx.compute_powers_cost())[0]
bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in mine]
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(ctx.bundle_idx_count)])
rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)
sd = torch.from_numpy(src.view(np.int64)).cuda(); md = torch.from_numpy(masks.view(np.int64)).cuda()
out = torch.zeros((len(mine), 2, n), dtype=torch.int64, device="cuda")
idx = sorted({u[0] for u in mine})
sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx]
mp = [md.data_ptr() + i * n * 8 for i in range(len(mine))]
def step():
    pw = ctx.compute_powers(idx, sp, rk, on_device=True)
    ctx.eval_bundles(bl, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
for _ in range(5): step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(20):                                            # two queued queries per sample: the host is never throttled
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(); step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) / 2 * 1e3); total.append((t2 - t0) / 2 * 1e3)
host.sort(); total.sort()
print(f"world {world}: host queues a query in {host[len(host) // 2]:.3f} ms; two queued queries done after {total[len(total) // 2]:.3f} ms each")
