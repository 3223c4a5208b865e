"""apsu_he_eval_all on the 16M-4096 synthetic workload: the in-process multi-device entry point with HOST inputs and outputs
(query ciphertexts, masks and results cross PCIe inside the timed call — the PCIe-inclusive figure of DESIGN.md section 5).
Device lists: [0] and [0, 0] (two engines on one GPU: the multi-device code path; on a multi-GPU box pass --devices 0,1,...)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import apsu_amd
from bench import SEED0, WORKLOADS

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="16M-4096")
ap.add_argument("--devices", default="0;0,0")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--pinned", action="store_true", help="keep the query ciphertexts and masks in page-locked host memory")
args = ap.parse_args()
cfg = args.config
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", cfg + ".json")).read()
G = apsu_amd.HeContext(js)
n, t, K, first, nidx, ns = G.n, G.t, G.K, G.first_chain_idx, G.bundle_idx_count, G.source_power_count
Lf = first + 1; D = G.max_items_per_bin - 1
q = G.q
G.close()
units = [(b, ci, deg) for b in range(nidx) for ci, deg in enumerate(WORKLOADS[cfg]["degrees"](D))]
rng = np.random.default_rng(SEED0)
src = [np.stack([np.stack([rng.integers(0, qq, n, dtype=np.uint64) for qq in q[:Lf]]) for _ in range(2)]) for _ in range(nidx * ns)]
rkh = np.stack([np.stack([np.stack([rng.integers(0, qq, n, dtype=np.uint64) for qq in q]) for _ in range(2)]) for _ in range(K - 1)])
masks = [rng.integers(0, t, n, dtype=np.uint64) for _ in units]
if args.pinned:
    import torch
    keep = []
    def pin(a):
        tt = torch.from_numpy(a.view(np.int64)).pin_memory()
        keep.append(tt)
        return tt.numpy().view(np.uint64)
    src = [pin(a) for a in src]
    masks = [pin(a) for a in masks]
ref_out = None
for spec in args.devices.split(";"):
    devs = [int(x) for x in spec.split(",")]
    M = apsu_amd.MultiContext(js, devs)
    M.upload_relin_keys(rkh)
    slots = apsu_amd.partition_bundles(units, nidx, len(devs))
    for (b, ci, deg), s in zip(units, slots):
        M.random_bundle(s, b, ci, deg, SEED0 + 1000003 * b + 7919 * ci)
    for _ in range(3):
        out = M.eval_all(src, masks, n)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = M.eval_all(src, masks, n)
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    if ref_out is None:
        ref_out = out
    print("devices %s%s: %.3f ms per query through apsu_he_eval_all (host inputs/outputs, %d BinBundles, per device %s), same bits as [0]: %s"
          % (devs, " [pinned inputs]" if args.pinned else "", ms, len(units), [slots.count(i) for i in range(len(devs))], bool((out == ref_out).all())), flush=True)
    M.close()
