"""Scheduling sweep on one GPU (16M-4096 unless --config): the whole query (N = 1) and the shard of rank 0 of an
N = 2 / 4 / 8 run, for every combination of
  two-stream ComputePowers (apsu_he_set_two_stream 0 / 1) x pipelined evaluation (apsu_he_set_eval_pipeline 1 / 2 / 4 / 7).
Prints ms per step and checks that every setting returns the same bits."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
from apsu_amd.sharding import partition
from bench import SEED0, WORKLOADS

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="16M-4096")
ap.add_argument("--worlds", default="1,8")
ap.add_argument("--splits", default="0,1")
ap.add_argument("--pipes", default="1,2,4,7")
ap.add_argument("--steps", type=int, default=20)
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
for world in [int(w) for w in args.worlds.split(",")]:
    assign = partition(units, ctx.bundle_idx_count, world)
    mine = assign[0]
    idx = sorted({u[0] for u in mine})
    sp = [[sd.data_ptr() + ((b * ns + s) * 2 * Lf * n) * 8 for s in range(ns)] for b in idx]
    mp = [md.data_ptr() + i * n * 8 for i in range(len(mine))]
    bl = [allb[(u[0], u[1])] for u in mine]

    def step():
        pw = ctx.compute_powers(idx, sp, rk, on_device=True)
        ctx.eval_bundles(bl, pw, rk, mp, out=out.data_ptr(), masks_on_device=True, out_on_device=True)

    ref_out = None
    for split in [int(x) for x in args.splits.split(",")]:
        for pipe in [int(x) for x in args.pipes.split(",")]:
            ctx.set_two_stream(split)
            ctx.set_eval_pipeline(pipe)
            out.zero_()
            for _ in range(3): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(args.steps): step()
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / args.steps
            got = out[:len(mine)].cpu()
            if ref_out is None: ref_out = got
            same = bool((got == ref_out).all())
            print(f"world={world} rank0 ({len(idx)} idx, {len(mine)} BinBundles) two_stream={split} eval_pipe={pipe}: {ms:.3f} ms/step  same_bits={same}", flush=True)
            assert same
