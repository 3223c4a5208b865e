"""A/B of two engine configurations inside ONE process on the same box and thermal state: two contexts, created under
different environment switches (read at apsu_he_create), each with its own copy of the synthetic DB; rounds of `--steps`
queries alternate A B A B ...; prints the per-round means, the paired difference and its standard error.
    python tools/ab_compare.py --a APSU_HE_FUSE_TENSOR=0 --b APSU_HE_FUSE_TENSOR=1 [--config 16M-4096] [--world 1]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from apsu_amd.sharding import partition
from bench import SEED0, WORKLOADS

ap = argparse.ArgumentParser()
ap.add_argument("--a", default="")
ap.add_argument("--b", default="")
ap.add_argument("--config", default="16M-4096")
ap.add_argument("--world", type=int, default=1)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--nokeep", action="store_true", help="free each query's powers right after queueing its evaluation (bench.py's pattern) instead of holding them across the next ComputePowers")
args = ap.parse_args()
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", args.config + ".json")).read()

def make(envs):
    saved = {}
    for kv in [x for x in envs.split(",") if x]:
        k, v = kv.split("=", 1); saved[k] = os.environ.get(k); os.environ[k] = v
    ctx = apsu_amd.HeContext(js)
    for k, v in saved.items():
        if v is None: del os.environ[k]
        else: os.environ[k] = v
    n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
    Lf = first + 1; D = ctx.max_items_per_bin - 1
    units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[args.config]["degrees"](D))]
    mine = partition(units, ctx.bundle_idx_count, args.world, ctx.compute_powers_cost())[0]
    bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in mine]
    rng = np.random.default_rng(SEED0); ns = ctx.source_power_count
    src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(ctx.bundle_idx_count)])
    rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)])) if K > 1 else None
    masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)
    sd = torch.from_numpy(src.view(np.int64)).cuda(); md = torch.from_numpy(masks.view(np.int64)).cuda()
    out = torch.zeros((len(mine), 2, n), dtype=torch.int64, device="cuda")
    idx = sorted({u[0] for u in mine})
    sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx]
    mp = [md.data_ptr() + i * n * 8 for i in range(len(mine))]
    ctx.set_async_results(True)
    ctx.set_query_overlap(True)                                  # static, synchronised inputs
    keep = [None, sd, md, rk]
    def step():
        if args.nokeep:
            pw = ctx.compute_powers(idx, sp, rk, on_device=True)
            ctx.eval_bundles(bl, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
            return
        keep[0] = ctx.compute_powers(idx, sp, rk, on_device=True)
        ctx.eval_bundles(bl, keep[0], rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)
    return ctx, step, out

A = make(args.a); B = make(args.b)
for _, step, _o in (A, B):
    for _ in range(3): step()
torch.cuda.synchronize()
same = bool((A[2] == B[2]).all())
ta, tb = [], []
for r in range(args.rounds):
    for which, (ctx, step, _o) in (("a", A), ("b", B)) if r % 2 == 0 else (("b", B), ("a", A)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps): step()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / args.steps
        (ta if which == "a" else tb).append(ms)
ta, tb = np.array(ta), np.array(tb)
d = tb - ta
print(f"A [{args.a}]: mean {ta.mean():.4f} ms  min {ta.min():.4f}   B [{args.b}]: mean {tb.mean():.4f} ms  min {tb.min():.4f}   "
      f"B - A = {d.mean():+.4f} +- {d.std(ddof=1) / np.sqrt(len(d)):.4f} ms ({100 * d.mean() / ta.mean():+.2f} %)  same_bits={same}  "
      f"[{args.config}, world {args.world}, {args.rounds} rounds x {args.steps} steps]", flush=True)
