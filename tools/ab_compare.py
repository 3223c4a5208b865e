"""A/B of two engine configurations inside ONE process on the same box and thermal state: two contexts, each with its own copy
of the synthetic DB; rounds of `--steps` queries alternate A B A B ...; prints the per-round means, the paired difference and
its standard error -- for the LATENCY of one query at a time (bench.py's `value`) and for the rate of QUEUED queries.

A setting is a comma-separated list of
    APSU_HE_X=v         environment switch, read at apsu_he_create
    overlap=0..3        apsu_he_set_query_overlap mode (default 1)
    two_stream=-1|0|1   apsu_he_set_two_stream
Consecutive steps alternate between two different queries (other sources, other masks) into two result buffers, and `same_bits`
is reported per query kind: identical queries would hide a step that read anything of the step in front of it.

    python tools/ab_compare.py --a overlap=2 --b overlap=1 [--config 16M-4096] [--world 1]"""
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
ap.add_argument("--modes", default="latency,queued")
ap.add_argument("--nokeep", action="store_true", help="free each query's powers right after queueing its evaluation (bench.py's pattern) instead of holding them across the next ComputePowers")
args = ap.parse_args()
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", args.config + ".json")).read()


def make(setting):
    saved, api = {}, {"overlap": 1, "two_stream": -1}
    for kv in [x for x in setting.split(",") if x]:
        k, v = kv.split("=", 1)
        if k in api:
            api[k] = int(v)
        else:
            saved[k] = os.environ.get(k); os.environ[k] = v
    ctx = apsu_amd.HeContext(js)
    for k, v in saved.items():
        if v is None: del os.environ[k]
        else: os.environ[k] = v
    n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
    Lf = first + 1; D = ctx.max_items_per_bin - 1
    units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[args.config]["degrees"](D))]
    mine = partition(units, ctx.bundle_idx_count, args.world, ctx.compute_powers_cost())[0]
    bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in mine]
    ns = ctx.source_power_count
    idx = sorted({u[0] for u in mine})
    rng = np.random.default_rng(SEED0)
    rk = None
    sp, mp, keepalive = [], [], []
    for kind in range(2):                                          # two different queries, identical in both contexts
        src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]]) for _ in range(2)]) for _ in range(ns)]) for _ in range(ctx.bundle_idx_count)])
        if kind == 0 and K > 1:
            rk = ctx.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)]))
        masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)
        sd = torch.from_numpy(src.view(np.int64)).cuda(); md = torch.from_numpy(masks.view(np.int64)).cuda()
        keepalive += [sd, md]
        sp.append([[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx])
        mp.append([md.data_ptr() + i * n * 8 for i in range(len(mine))])
    outs = [torch.zeros((len(mine), 2, n), dtype=torch.int64, device="cuda") for _ in range(2)]
    ctx.set_async_results(True)
    ctx.set_query_overlap(api["overlap"])                          # static, synchronised inputs
    ctx.set_two_stream(api["two_stream"])
    keep = [None, keepalive, rk]
    no = [0]

    def step():
        kind = no[0] & 1
        no[0] += 1
        pw = ctx.compute_powers(idx, sp[kind], rk, on_device=True)
        ctx.eval_bundles(bl, pw, rk, mp[kind], out=outs[kind].data_ptr(), masks_on_device=True, out_on_device=True)
        if not args.nokeep:
            keep[0] = pw
    return ctx, step, outs


A = make(args.a); B = make(args.b)
for _, step, _o in (A, B):
    for _ in range(4): step()
torch.cuda.synchronize()
for mode in [m for m in args.modes.split(",") if m]:
    ta, tb = [], []
    for r in range(args.rounds):
        for which, (ctx, step, _o) in (("a", A), ("b", B)) if r % 2 == 0 else (("b", B), ("a", A)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
                if mode == "latency": torch.cuda.synchronize()
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / args.steps
            (ta if which == "a" else tb).append(ms)
    # the buffers as the LAST steps of this mode wrote them (queued mode: pipelined steps)
    same = [bool((A[2][k] == B[2][k]).all()) for k in range(2)]
    differ = not bool((A[2][0] == A[2][1]).all())
    ta, tb = np.array(ta), np.array(tb)
    d = tb - ta
    print(f"{mode:8s} A [{args.a}]: mean {ta.mean():.4f} ms  min {ta.min():.4f}   B [{args.b}]: mean {tb.mean():.4f} ms  min {tb.min():.4f}   "
          f"B - A = {d.mean():+.4f} +- {d.std(ddof=1) / np.sqrt(len(d)):.4f} ms ({100 * d.mean() / ta.mean():+.2f} %)  "
          f"same_bits query0={same[0]} query1={same[1]} (queries differ: {differ})  "
          f"[{args.config}, world {args.world}, {args.rounds} rounds x {args.steps} steps]", flush=True)
