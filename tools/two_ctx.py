"""experiment: does running two halves of the query on two contexts (two sets of streams) in one process beat one context?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from bench import SEED0, WORKLOADS
cfg = "16M-4096"
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
ctxs = [apsu_amd.HeContext(js), apsu_amd.HeContext(js)]
c0 = ctxs[0]
n, t, K, first = c0.n, c0.t, c0.K, c0.first_chain_idx
Lf = first + 1; D = c0.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(c0.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
rng = np.random.default_rng(SEED0); ns = c0.source_power_count
src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in c0.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(c0.bundle_idx_count)])
rkh = np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in c0.q]) for _ in range(2)]) for _ in range(K - 1)])
masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)
sd = torch.from_numpy(src.view(np.int64)).cuda(); md = torch.from_numpy(masks.view(np.int64)).cuda()
out = torch.zeros((len(units), 2, n), dtype=torch.int64, device="cuda")
def setup(ctx, idxs):
    mine = [u for u in units if u[0] in idxs]
    bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in mine]
    rk = ctx.upload_relin_keys(rkh)
    sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idxs]
    pos = [units.index(u) for u in mine]
    mp = [md.data_ptr() + i * n * 8 for i in pos]
    o = out.data_ptr() + pos[0] * 2 * n * 8
    def step():
        pw = ctx.compute_powers(idxs, sp, rk, on_device=True)
        ctx.eval_bundles(bl, pw, rk, mp, out=o, masks_on_device=True, out_on_device=True)
    return step
full = setup(ctxs[0], [0, 1, 2, 3])
for _ in range(3): full()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): full()
torch.cuda.synchronize(); print("one context, 4 indices: %.3f ms" % ((time.perf_counter() - t0) * 100))
ref_out = out.clone()
halves = [setup(ctxs[0], [0, 1]), setup(ctxs[1], [2, 3])]
def run(f, k):
    for _ in range(k): f()
for rep in range(2):
    ths = [threading.Thread(target=run, args=(h, 3 if rep == 0 else 10)) for h in halves]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize()
    if rep: print("two contexts x 2 indices, concurrent threads: %.3f ms per query" % ((time.perf_counter() - t0) * 100))
print("same results:", bool((out == ref_out).all()))
