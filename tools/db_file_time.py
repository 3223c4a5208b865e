"""N2: time to save the synthetic 16M-4096 database (28 BinBundles, 5.8 GiB) into one file and to bring it back -- onto one
device, and as the shard of one rank of 8 -- against rebuilding it; checks one query's results against the original DB.
    python tools/db_file_time.py [config] [dir]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from apsu_amd.sharding import partition
from bench import SEED0, WORKLOADS

cfg = sys.argv[1] if len(sys.argv) > 1 else "16M-4096"
where = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
ctx = apsu_amd.HeContext(js)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
t0 = time.perf_counter()
bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in units]
ctx.sync(); t_build = time.perf_counter() - t0
db_bytes = sum(b.db_bytes for b in bl)
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
src = [[np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)] for _ in range(ctx.bundle_idx_count)]
rk_host = np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)])
rk = ctx.upload_relin_keys(rk_host)
masks = [rng.integers(0, t, n, dtype=np.uint64) for _ in units]
idx = list(range(ctx.bundle_idx_count))
want = ctx.eval_bundles(bl, ctx.compute_powers(idx, src, rk), rk, masks)
path = os.path.join(where, "apsu_he_%s.db" % cfg)
t0 = time.perf_counter(); ctx.save_db_file(path, bl); t_save = time.perf_counter() - t0
size = os.path.getsize(path)
print(f"{cfg}: {len(bl)} BinBundles, {db_bytes / 2**30:.2f} GiB in HBM; synthetic build on the GPU {t_build:.2f} s (a real build adds the host's bin placement)")
print(f"save  : {t_save:.2f} s  ({size / 2**30:.2f} GiB file, {size / t_save / 1e9:.2f} GB/s: device -> host, checksum, write)")
del bl
ctx2 = apsu_amd.HeContext(js)                 # what a restarted process would do
for label in ("load (page cache warm)", "load again"):
    t0 = time.perf_counter(); back = ctx2.load_db_file(path); ctx2.sync(); t_load = time.perf_counter() - t0
    print(f"{label:22s}: {t_load:.2f} s  ({size / t_load / 1e9:.2f} GB/s: mapped file -> checksum -> device)")
    if label == "load again": break
    del back
rk2 = ctx2.upload_relin_keys(rk_host)
got = ctx2.eval_bundles(back, ctx2.compute_powers(idx, src, rk2), rk2, masks)
print("query on the reloaded DB equals the original:", bool((got == want).all()))
del back
for world in (8,):
    assign = partition(units, ctx.bundle_idx_count, world, ctx.compute_powers_cost())
    pos = {u: i for i, u in enumerate(units)}
    for r in (0, world - 1):
        only = [pos[u] for u in assign[r]]
        t0 = time.perf_counter(); shard = ctx2.load_db_file(path, only=only); ctx2.sync(); t_sh = time.perf_counter() - t0
        sb = sum(b.db_bytes for b in shard)
        print(f"shard of rank {r} of {world}: {len(only)} BinBundles, {sb / 2**30:.2f} GiB in {t_sh:.2f} s")
        del shard
os.remove(path)
