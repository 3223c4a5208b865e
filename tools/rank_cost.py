"""what one rank of an N-GPU run does (default 16M-4096; argv[1] = another config of bench.py): per-rank step time on a
single GPU, for N = 1, 2, 4, 8 (the shards of rank 0 and of the last rank are executed here, one after the other).
Per shard BOTH figures of bench.py: the LATENCY of one query with a host wait at its end (bench.py's `value`; the N-GPU
projection quotes this one) and the rate of queued, pipelined queries (`throughput_ms_per_query`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from apsu_amd.sharding import partition
from bench import SEED0, WORKLOADS
cfg = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "16M-4096"
REPEAT = int(os.environ.get("RANK_COST_REPEAT", "1"))            # > 1: that many timed runs of 10 steps per shard, all printed
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
ctx = apsu_amd.HeContext(js)
ctx.set_async_results(True)                                     # bench.py's mode: queries queued back to back
ctx.set_query_overlap(True)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
allb = {(b, ci): ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in units}
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(ctx.bundle_idx_count)])
rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)
sd = torch.from_numpy(src.view(np.int64)).cuda(); md = torch.from_numpy(masks.view(np.int64)).cuda()
out = torch.zeros((len(units), 2, n), dtype=torch.int64, device="cuda")
for world in (1, 2, 4, 8):
    assign = partition(units, ctx.bundle_idx_count, world, ctx.compute_powers_cost())
    worst = worst_lat = 0
    seen = set()
    for r in range(world):                                  # every distinct shard shape once (the worst rank sets the step)
        mine = assign[r]
        sig = (len({u[0] for u in mine}), len(mine), sum(u[2] for u in mine))
        if sig in seen or not mine:
            continue
        seen.add(sig)
        idx = sorted({u[0] for u in mine})
        sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx]
        mp = [md.data_ptr() + i * n * 8 for i in range(len(mine))]
        bl = [allb[(u[0], u[1])] for u in mine]
        def step():
            pw = ctx.compute_powers(idx, sp, rk, on_device=True)
            ctx.eval_bundles(bl, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
        for _ in range(3): step()
        runs = []
        for _ in range(REPEAT):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): step()
            torch.cuda.synchronize(); runs.append((time.perf_counter() - t0) * 100)
        ms = sorted(runs)[len(runs) // 2]
        lat = []
        for _ in range(10 * REPEAT):                            # one query at a time, host wait at its end
            torch.cuda.synchronize(); t0 = time.perf_counter()
            step()
            torch.cuda.synchronize(); lat.append((time.perf_counter() - t0) * 1e3)
        lms = sorted(lat)[len(lat) // 2]
        if REPEAT > 1:
            print(f"   runs of 10 steps: {' '.join('%.3f' % v for v in runs)}  spread {100 * (max(runs) - min(runs)) / ms:.1f} % of the median", flush=True)
        t1 = time.perf_counter()
        for _ in range(10): pw = ctx.compute_powers(idx, sp, rk, on_device=True)
        ctx.eval_bundles(bl[:1], pw, rk, mp[:1], out=out.data_ptr(), masks_on_device=True, out_on_device=True)
        torch.cuda.synchronize(); pms = (time.perf_counter() - t1) * 100
        print(f"world={world} rank={r}: {len(idx)} idx, {len(mine)} bundles (deg sum {sum(u[2] for u in mine)}): latency {lms:.3f} ms, queued {ms:.3f} ms per query (powers ~{pms:.3f} ms)", flush=True)
        worst = max(worst, ms)
        worst_lat = max(worst_lat, lms)
    print(f"  => per-rank compute at N={world}: latency {worst_lat:.3f} ms, queued rate {worst:.3f} ms per query")
