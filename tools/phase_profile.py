"""per-phase (ComputePowers vs eval_bundles) kernel-class times for the bench workload"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from bench import SEED0, WORKLOADS
cfg = sys.argv[1] if len(sys.argv) > 1 else "16M-4096"
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
ctx = apsu_amd.HeContext(js)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
bundles = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in units]
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(ctx.bundle_idx_count)])
rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)
sd = torch.from_numpy(src.view(np.int64)).cuda(); md = torch.from_numpy(masks.view(np.int64)).cuda()
sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in range(ctx.bundle_idx_count)]
mp = [md.data_ptr() + i * n * 8 for i in range(len(units))]
out = torch.zeros((len(units), 2, n), dtype=torch.int64, device="cuda")
idx = list(range(ctx.bundle_idx_count))
for _ in range(2):
    pw = ctx.compute_powers(idx, sp, rk, on_device=True); ctx.eval_bundles(bundles, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
ctx.profile_enable(1); ctx.profile_read()
R = 5
acc = {}
for _ in range(R):
    t0 = time.perf_counter(); pw = ctx.compute_powers(idx, sp, rk, on_device=True); ctx.profile_read(reset=False); t1 = time.perf_counter()
    a = ctx.profile_read()
    ctx.eval_bundles(bundles, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True); t2 = time.perf_counter()
    b = ctx.profile_read()
    for name, d in (("powers", a), ("eval", b)):
        for k, v in d.items():
            e = acc.setdefault((name, k), [0.0, 0, 0]); e[0] += v[0]; e[1] += v[1]; e[2] += v[2]
    acc.setdefault(("powers", "wall"), [0.0, 0, 0])[0] += (t1 - t0) * 1e3; acc.setdefault(("eval", "wall"), [0.0, 0, 0])[0] += (t2 - t1) * 1e3
for phase in ("powers", "eval"):
    tot = sum(v[0] for (p, k), v in acc.items() if p == phase and k != "wall") / R
    print(phase, "kernel sum %.3f ms  wall %.3f ms" % (tot, acc[(phase, "wall")][0] / R))
    for (p, k), v in sorted(acc.items()):
        if p == phase and k != "wall" and v[1]:
            print("   %-12s %.3f ms  launches %d  units %d" % (k, v[0] / R, v[1] // R, v[2] // R))
