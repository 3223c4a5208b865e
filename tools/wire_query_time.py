"""Receiver::RunQuery from the wire at 16M-4096: a framed QueryRequest (24 seeded ciphertexts + seeded RelinKeys, SEAL objects under
compr none / zlib / zstd) -> apsu_he_run_query_request -> 28 ResultPackages; wall time per query next to the device-only 3.4 ms.
    python tools/wire_query_time.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, apsu_amd
from apsu_amd import seal, wire
from bench import SEED0, WORKLOADS

cfg = sys.argv[1] if len(sys.argv) > 1 else "16M-4096"
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
ctx = apsu_amd.HeContext(js)
sc = seal.SealContext(js)
n, t, K, first = ctx.n, ctx.t, ctx.K, ctx.first_chain_idx
Lf = first + 1; D = ctx.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(ctx.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
bl = [ctx.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in units]
rng = np.random.default_rng(SEED0)
powers = sorted(int(p) for p in __import__("json").loads(js)["query_params"]["query_powers"])
masks = [rng.integers(0, t, n, dtype=np.uint64) for _ in units]
ksk = np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q]) for _ in range(2)]) for _ in range(K - 1)])
for compr, name in ((seal.COMPR_NONE, "none"), (seal.COMPR_ZLIB, "zlib"), (seal.COMPR_ZSTD, "zstd")):
    parts = []
    for e in powers:
        cts = []
        for b in range(ctx.bundle_idx_count):
            seed = [int(x) for x in rng.integers(0, 2**63, 8, dtype=np.uint64)]
            c0 = np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in ctx.q[:Lf]])
            cts.append(sc.ct_save(first, False, np.stack([c0, np.zeros_like(c0)]), seed=seed, compr=compr))
        parts.append((e, cts))
    seeds = rng.integers(0, 2**63, (K - 1, 8), dtype=np.uint64)
    rk_blob = sc.relin_keys_save(ksk, seeds=seeds, compr=compr)
    msg = wire.build_query_request(compr, rk_blob, parts)
    for _ in range(2):
        pk = seal.run_query_request(ctx, sc, msg, bl, masks, compr=compr)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); pk = seal.run_query_request(ctx, sc, msg, bl, masks, compr=compr); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{cfg} compr {name}: request {len(msg) / 1e6:.2f} MB, {len(pk)} packages {sum(len(p) for p in pk) / 1e6:.2f} MB: "
          f"run_query_request {sorted(ts)[2]:.1f} ms per query (min {min(ts):.1f})", flush=True)
