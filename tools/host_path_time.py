"""Where the host-buffer (PCIe-inclusive) path spends its extra time: the same 16M-4096 query with device-resident inputs,
with host query ciphertexts, with host masks / results, and with both (single context, tier-2 calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from bench import SEED0, WORKLOADS
cfg = "16M-4096"
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
ctx = apsu_amd.HeContext(js)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in units]
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
idx = list(range(ctx.bundle_idx_count))
src = [[np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)] for _ in idx]
rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
masks = [rng.integers(0, t, n, dtype=np.uint64) for _ in units]
sd = torch.from_numpy(np.stack([np.stack(s) for s in src]).view(np.int64)).cuda()
md = torch.from_numpy(np.stack(masks).view(np.int64)).cuda()
out = torch.zeros((len(units), 2, n), dtype=torch.int64, device="cuda")
sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx]
mp = [md.data_ptr() + i * n * 8 for i in range(len(units))]

def run(host_src, host_io, reps=10):
    def step():
        pw = ctx.compute_powers(idx, src if host_src else sp, rk, on_device=not host_src)
        if host_io:
            return ctx.eval_bundles(bl, pw, rk, masks)
        ctx.eval_bundles(bl, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps

for hs, hio in ((False, False), (True, False), (False, True), (True, True)):
    print("query ciphertexts on the %s, masks/results on the %s: %.3f ms per query" % ("host" if hs else "device", "host" if hio else "device", run(hs, hio)), flush=True)
t0 = time.perf_counter()
for _ in range(10): ctx.compute_powers(idx, src, rk, on_device=False)
print("ComputePowers alone, host inputs (synchronous): %.3f ms" % ((time.perf_counter() - t0) * 100))
t0 = time.perf_counter()
for _ in range(10): pw = ctx.compute_powers(idx, sp, rk, on_device=True)
torch.cuda.synchronize()
print("ComputePowers alone, device inputs: %.3f ms" % ((time.perf_counter() - t0) * 100))
