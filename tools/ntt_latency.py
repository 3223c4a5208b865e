"""Latency of SMALL NTT launches (the regime of per-rank shards and of the partial last round of every launch):
forward / inverse transform of 1..N limb polynomials, device time per launch from HIP events on the engine's stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import apsu_amd
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", (sys.argv[1] if len(sys.argv) > 1 else "16M-4096") + ".json")).read()
G = apsu_amd.HeContext(js)
n, first = G.n, G.first_chain_idx
L = first + 1
rng = np.random.default_rng(3)
for polys in (1, 2, 8, 28, 56, 85, 170, 341, 400, 682):
    ct = np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in G.q[:L]]) for _ in range(polys)])
    ref0 = ct.copy()
    G.transform_to_ntt_inplace(ct, first); G.transform_from_ntt_inplace(ct, first)
    G.profile_enable(2); G.profile_read()
    for _ in range(10):
        G.transform_to_ntt_inplace(ct, first)
        G.transform_from_ntt_inplace(ct, first)
    p = G.profile_read(); G.profile_enable(0)
    assert (ct == ref0).all()
    f = p["ntt_fwd"][0] / p["ntt_fwd"][1] * 1e3; i = p["ntt_inv"][0] / p["ntt_inv"][1] * 1e3
    limbs = polys * L
    print(f"{limbs:5d} limbs per launch: forward {f:7.1f} us ({limbs*16*n/f/1e3:6.0f} GB/s)   inverse {i:7.1f} us ({limbs*16*n/i/1e3:6.0f} GB/s)", flush=True)
