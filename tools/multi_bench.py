"""apsu_he_eval_all_ex on the 16M-4096 synthetic workload: the in-process multi-device entry point with the caller's query buffers in
pageable host memory, in page-locked host memory (query ciphertexts, masks and results cross PCIe inside the timed call — the
PCIe-inclusive figure of DESIGN.md section 5) and already in HBM (device-resident inputs and gathered output).
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
ap.add_argument("--modes", default="pageable,pinned,device", help="where the caller's query buffers live")
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
ref_out = None
for spec in args.devices.split(";"):
    devs = [int(x) for x in spec.split(",")]
    M = apsu_amd.MultiContext(js, devs)
    M.upload_relin_keys(rkh)
    slots = apsu_amd.partition_bundles(units, nidx, len(devs))
    for (b, ci, deg), s in zip(units, slots):
        M.random_bundle(s, b, ci, deg, SEED0 + 1000003 * b + 7919 * ci)
    for mode in args.modes.split(","):
        kw = {}
        if mode == "pinned":
            ps = [apsu_amd.host_alloc(a.shape) for a in src]; pm = [apsu_amd.host_alloc(a.shape) for a in masks]
            for d_, s_ in zip(ps + pm, src + masks): d_[...] = s_
            po = apsu_amd.host_alloc((len(units), 2, 1, n))
            a_src, a_mask, kw = ps, pm, dict(flags=M.IO_SRC_PINNED | M.IO_MASKS_PINNED | M.IO_OUT_PINNED, out=po)
        elif mode == "device":
            import torch
            sd = [torch.from_numpy(a.view(np.int64)).to("cuda:%d" % devs[0]) for a in src]
            md = [torch.from_numpy(a.view(np.int64)).to("cuda:%d" % devs[0]) for a in masks]
            od = torch.zeros((len(units), 2, n), dtype=torch.int64, device="cuda:%d" % devs[0])
            a_src, a_mask = [t.data_ptr() for t in sd], [t.data_ptr() for t in md]
            kw = dict(flags=M.IO_SRC_ON_DEVICE | M.IO_MASKS_ON_DEVICE | M.IO_GATHER_RCCL, in_device_slot=0, out_device_slot=0, out_ptr=od.data_ptr())
        else:
            a_src, a_mask = src, masks
        for _ in range(3):
            out = M.eval_all(a_src, a_mask, n, **kw)
        M.phase_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = M.eval_all(a_src, a_mask, n, **kw)
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        ph = M.phase_read(); M.phase_enable(False)
        if mode == "device":
            torch.cuda.synchronize(); out = od.cpu().numpy().view(np.uint64).reshape(len(units), 2, 1, n)
        if ref_out is None:
            ref_out = np.array(out)
        print("devices %s, %s query buffers: %.3f ms per query through apsu_he_eval_all_ex (%d BinBundles, per device %s%s), same bits: %s;  phases avg ms: %s"
              % (devs, mode, ms, len(units), [slots.count(i) for i in range(len(devs))], (", gather " + M.last_gather()) if mode == "device" else "",
                 bool((np.array(out) == ref_out).all()), {k.split("::")[1]: round(v[1], 3) for k, v in ph.items()}), flush=True)
    M.close()
