"""N1: time to build one full-size 16M-4096 BinBundle on the GPU (polyn_with_roots for 8190 bins of 1303 roots, encode, lift, NTT)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, apsu_amd
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", "16M-4096.json")).read()
ctx = apsu_amd.HeContext(js)
n, t = ctx.n, ctx.t
bins = ctx.info.items_per_bundle * 5
D = ctx.max_items_per_bin - 1
rng = np.random.default_rng(1)
roots = rng.integers(0, t, (bins, D), dtype=np.uint64)
import ctypes as C
from apsu_amd.engine import load_library, _check, _p, Bundle
counts = np.full(bins, D, dtype=np.uint32)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = C.c_void_p()
    _check(load_library().apsu_he_db_build_bundle(ctx.h, 0, rep, _p(roots), C.c_void_p(counts.ctypes.data), bins, D, C.byref(h)))
    b = Bundle(ctx, h, 0, rep, D)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"build_bundle: {bins} bins x {D} roots -> degree {b.degree}: {dt*1e3:.1f} ms ({bins * D * (D + 1) / 2 / dt / 1e9:.2f} G modmul/s)")
