"""kernel timeline of one rank's work at N=8 (one bundle index, 4 BinBundles): run under rocprofv3 --kernel-trace"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from bench import SEED0
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", "16M-4096.json")).read()
ctx = apsu_amd.HeContext(js)
if os.environ.get("TRACE_ASYNC", "1") != "0":
    ctx.set_async_results(True); ctx.set_query_overlap(True)       # bench.py's mode: queued, pipelined queries
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1
rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
src = np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)])
rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
sd = torch.from_numpy(src.view(np.int64)).cuda()
bl = [ctx.random_bundle(0, ci, 1303, 5 + ci) for ci in range(4)]
md = torch.from_numpy(rng.integers(0, t, (4, n), dtype=np.uint64).view(np.int64)).cuda()
out = torch.zeros((4, 2, n), dtype=torch.int64, device="cuda")
sp = [[sd.data_ptr() + (s * 2 * Lf * n) * 8 for s in range(ns)]]
mp = [md.data_ptr() + i * n * 8 for i in range(4)]
for _ in range(int(os.environ.get("TRACE_STEPS", "12"))):
    pw = ctx.compute_powers([0], sp, rk, on_device=True)
    ctx.eval_bundles(bl, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
    pw = None
torch.cuda.synchronize()
