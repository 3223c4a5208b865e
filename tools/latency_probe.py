import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, apsu_amd
from bench import SEED0, WORKLOADS
cfg = "16M-4096"
js = open(os.path.join("tests", "params", cfg + ".json")).read()
ctx = apsu_amd.HeContext(js)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in units]
ns = ctx.source_power_count
rng = np.random.default_rng(SEED0)
rk = None; sp = []; mp = []; keep = []
for kind in range(2):
    src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(ctx.bundle_idx_count)])
    if kind == 0:
        rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
    masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)
    sd = torch.from_numpy(src.view(np.int64)).cuda(); md = torch.from_numpy(masks.view(np.int64)).cuda(); keep += [sd, md]
    sp.append([[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in range(ctx.bundle_idx_count)])
    mp.append([md.data_ptr() + i * n * 8 for i in range(len(units))])
outs = [torch.zeros((len(units), 2, n), dtype=torch.int64, device="cuda") for _ in range(2)]
idx = list(range(ctx.bundle_idx_count))
ctx.set_async_results(True); ctx.set_query_overlap(1)
def step(kind, hold):
    t0 = time.perf_counter()
    pw = ctx.compute_powers(idx, sp[kind], rk, on_device=True)
    t1 = time.perf_counter()
    ctx.eval_bundles(bl, pw, rk, mp[kind], out=outs[kind].data_ptr(), masks_on_device=True, out_on_device=True)
    t2 = time.perf_counter()
    if not hold: del pw
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, pw if hold else None
for name, kinds, hold in (("same kind, powers freed", [0] * 24, False), ("alternating, powers freed", [0, 1] * 12, False), ("alternating, powers held", [0, 1] * 12, True), ("same kind, powers held", [0] * 24, True)):
    for k in kinds[:4]: held = step(k, hold)
    c0 = ctx.debug_counters(); rows = []
    held = None
    for k in kinds:
        r = step(k, hold); rows.append(r[:4]); held = r[4]
    c1 = ctx.debug_counters()
    a = np.array(rows)
    print("%-28s enqueue ComputePowers %.3f ms, enqueue evaluation %.3f ms, wait %.3f ms, total median %.3f ms (mean %.3f); per query: job uploads %.1f hits %.1f, host syncs %.1f, powers allocs %.2f"
          % (name, np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2]), np.median(a[:, 3]), a[:, 3].mean(), (c1["job_upload"] - c0["job_upload"]) / len(kinds), (c1["job_hit"] - c0["job_hit"]) / len(kinds), (c1["host_sync"] - c0["host_sync"]) / len(kinds), (c1["powers_alloc"] - c0["powers_alloc"]) / len(kinds)), flush=True)
