"""Per-step wall time and host-side engine events for the small shards of an N-GPU run (rank 0's shard of N = 8, 4, 1) executed on
one GPU, walking a sequence of scheduling settings (two-stream ComputePowers x queued / synchronised evaluation): which steps are slow, and what the host did in
them (apsu_he_debug_counters: host waits, job-table uploads, arena growths, powers allocations, staging wraps)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from apsu_amd.sharding import partition
from bench import SEED0, WORKLOADS

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="16M-4096")
ap.add_argument("--worlds", default="1,4,8")
ap.add_argument("--splits", default="0,1")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--asyncs", default="0,1")
args = ap.parse_args()
cfg = args.config
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
ctx = apsu_amd.HeContext(js)
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

def delta(a, b):
    return {k: b[k] - a[k] for k in a if b[k] != a[k]}

for world in [int(w) for w in args.worlds.split(",")]:
    assign = partition(units, ctx.bundle_idx_count, world, ctx.compute_powers_cost())
    mine = assign[0]
    idx = sorted({u[0] for u in mine})
    sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx]
    mp = [md.data_ptr() + i * n * 8 for i in range(len(mine))]
    bl = [allb[(u[0], u[1])] for u in mine]
    keep = [None]

    def step():
        keep[0] = ctx.compute_powers(idx, sp, rk, on_device=True)
        ctx.eval_bundles(bl, keep[0], rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)

    for asy in [int(x) for x in args.asyncs.split(",")]:
        ctx.set_async_results(bool(asy))
        for split in [int(x) for x in args.splits.split(",")]:
            for pipe in (1,):
                ctx.set_two_stream(split)
                c0 = ctx.debug_counters()
                for _ in range(3): step()
                torch.cuda.synchronize()
                c1 = ctx.debug_counters()
                # (a) every step followed by a device wait: per-step latency; (b) the steps queued back to back
                per = []
                host = []
                for _ in range(args.steps):
                    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
                    per.append((t2 - t0) * 1e3); host.append((t1 - t0) * 1e3)
                c2 = ctx.debug_counters()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(args.steps): step()
                th = time.perf_counter()
                torch.cuda.synchronize(); t1 = time.perf_counter()
                c3 = ctx.debug_counters()
                per_s = sorted(per)
                print(f"world={world} ({len(idx)} idx, {len(mine)} BinBundles) async={asy} two_stream={split} pipe={pipe}: "
                      f"synced steps min {per_s[0]:.3f} median {per_s[len(per)//2]:.3f} max {per_s[-1]:.3f} ms (host enqueue median {sorted(host)[len(host)//2]:.3f}); "
                      f"queued {(t1 - t0) * 1e3 / args.steps:.3f} ms/step (host {(th - t0) * 1e3 / args.steps:.3f})", flush=True)
                print(f"    warm-up events {delta(c0, c1)}  synced-loop events {delta(c1, c2)}  queued-loop events {delta(c2, c3)}", flush=True)
                slow = [f"{i}:{v:.2f}" for i, v in enumerate(per) if v > 1.5 * per_s[len(per)//2]]
                if slow: print("    slow steps (index:ms):", " ".join(slow), flush=True)
