"""Full-size parity run (not collected by pytest; `python tests/full_query_parity.py [config]` on the GPU box):
the bench workload — every bundle index through ComputePowers, every BinBundle through eval/eval_patstock — on the GPU
and on the oracle, compared bit for bit (all powers, all results).  ~1 minute for 16M-4096 on 16 host threads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import apsu_amd
from oracle import ref
from bench import SEED0, WORKLOADS, splitmix_values

cfg = sys.argv[1] if len(sys.argv) > 1 else "16M-4096"
js = open(os.path.join(ROOT, "tests", "params", cfg + ".json")).read()
G = apsu_amd.HeContext(js)
p = ref.load_params(js)
C = ref.RefContext.from_params(p)
n, t, K, first = G.n, G.t, G.K, G.first_chain_idx
Lf = first + 1
ps = p["ps_low_degree"]
D = G.max_items_per_bin - 1
units = [(b, ci, deg) for b in range(G.bundle_idx_count) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
rng = np.random.default_rng(SEED0)
ns = G.source_power_count
src = np.stack([np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in G.q[:Lf]]) for _ in range(2)])
                          for _ in range(ns)]) for _ in range(G.bundle_idx_count)])
rkh = np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in G.q]) for _ in range(2)]) for _ in range(K - 1)]) if K > 1 else None
masks = rng.integers(0, t, (len(units), n), dtype=np.uint64)

t0 = time.time()
rk = G.upload_relin_keys(rkh) if K > 1 else None
idxs = list(range(G.bundle_idx_count))
pw = G.compute_powers(idxs, [[src[b, s] for s in range(ns)] for b in idxs], rk)
bl = [G.random_bundle(b, ci, deg, SEED0 + 1000003 * b + 7919 * ci) for (b, ci, deg) in units]
out = G.eval_bundles(bl, pw, rk, [masks[i] for i in range(len(units))])
print("GPU: %d bundle indices, %d BinBundles in %.1f s (incl. DB generation)" % (len(idxs), len(units), time.time() - t0), flush=True)

targets = ref.create_powers_set(ps, p["max_items_per_bin"])
_, nodes = ref.powers_dag(p["query_powers"], targets)
sources = sorted(p["query_powers"])
from concurrent.futures import ThreadPoolExecutor
threads = max(1, min(64, len(os.sched_getaffinity(0))))
pci = C.plain_chain_idx(ps)
bad = 0
t0 = time.time()
pool = ThreadPoolExecutor(threads)                        # ctypes calls into the oracle release the GIL
for b in idxs:
    ref.set_threads(threads)
    opw = C.compute_powers({e: np.ascontiguousarray(src[b, s]) for s, e in enumerate(sources)}, nodes, rkh, ps)
    ref.set_threads(1)
    for power in targets:
        ct, _, _ = pw.download(b, power)
        if not (ct == opw[power]).all():
            bad += 1; print("MISMATCH power %d of bundle index %d" % (power, b), flush=True)
    plist = [None] * (p["max_items_per_bin"] + 1)
    for k, v in opw.items():
        plist[k] = v

    def check(i):
        bb, ci, deg = units[i]
        seed = SEED0 + 1000003 * bb + 7919 * ci
        coeffs = []
        for d in range(deg + 1):
            raw = splitmix_values(seed, d, n, t)
            coeffs.append(C.plain_lift_ntt(raw, pci) if ref.coeff_is_ntt(ps, d) else raw)
        mask = np.ascontiguousarray(masks[i])
        exp = C.eval_patstock(plist, coeffs, ps, rkh, mask) if (ps > 1 and ps < deg) else C.eval(plist, coeffs, plist[1].shape[1] - 1, mask)
        return i, bool((out[i] == exp).all())

    for i, ok in pool.map(check, [i for i, u in enumerate(units) if u[0] == b]):
        bb, ci, deg = units[i]
        bad += 0 if ok else 1
        print("bundle idx %d cache %d degree %d: %s" % (bb, ci, deg, "bit-exact" if ok else "MISMATCH"), flush=True)
print("oracle (%d threads for ComputePowers): %.1f s" % (threads, time.time() - t0))
print("RESULT %s: %d target powers x %d bundle indices and %d BinBundles compared, %d mismatches"
      % (cfg, len(targets), len(idxs), len(units), bad))
sys.exit(1 if bad else 0)
