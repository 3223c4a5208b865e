"""Latency of SMALL NTT launches (the regime of per-rank shards and of the partial last round of every launch):
forward / inverse transform of 1..N limb polynomials, device time per launch from HIP events on the engine's stream.
Round 6: both forms of the workgroup side by side -- 16 coefficients per lane (the throughput form) and 8 (the latency form,
twice the waves per limb; APSU_HE_NTT_LATENCY_LIMBS selects by launch size) -- on the same data, results compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import apsu_amd
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", (sys.argv[1] if len(sys.argv) > 1 else "16M-4096") + ".json")).read()
os.environ["APSU_HE_NTT_LATENCY_LIMBS"] = "0"
G16 = apsu_amd.HeContext(js)
os.environ["APSU_HE_NTT_LATENCY_LIMBS"] = "100000000"
G8 = apsu_amd.HeContext(js)
del os.environ["APSU_HE_NTT_LATENCY_LIMBS"]
n, first = G16.n, G16.first_chain_idx
L = first + 1
rng = np.random.default_rng(3)
print("# %s: n = %d, %d limbs per polynomial; us per launch, 16 / 8 coefficients per lane" % (sys.argv[1] if len(sys.argv) > 1 else "16M-4096", n, L))
for polys in (1, 2, 8, 16, 28, 42, 56, 85, 128, 170, 256, 341, 400, 512, 682):
    ct = np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in G16.q[:L]]) for _ in range(polys)])
    ref0 = ct.copy()
    res = {}
    for name, G in (("16", G16), ("8", G8)):
        c = ct.copy()
        G.transform_to_ntt_inplace(c, first)
        fwd = c.copy()
        G.transform_from_ntt_inplace(c, first)
        assert (c == ref0).all()
        G.profile_enable(2); G.profile_read()
        for _ in range(10):
            G.transform_to_ntt_inplace(c, first)
            G.transform_from_ntt_inplace(c, first)
        p = G.profile_read(); G.profile_enable(0)
        assert (c == ref0).all()
        res[name] = (p["ntt_fwd"][0] / p["ntt_fwd"][1] * 1e3, p["ntt_inv"][0] / p["ntt_inv"][1] * 1e3, fwd)
    assert (res["16"][2] == res["8"][2]).all()
    limbs = polys * L
    f16, i16, _ = res["16"]; f8, i8, _ = res["8"]
    print(f"{limbs:5d} limbs per launch: forward {f16:7.1f} / {f8:7.1f} us ({100 * (f8 / f16 - 1):+6.1f} %)   inverse {i16:7.1f} / {i8:7.1f} us ({100 * (i8 / i16 - 1):+6.1f} %)"
          f"   [{limbs*16*n/min(f16, f8)/1e3:6.0f} / {limbs*16*n/min(i16, i8)/1e3:6.0f} GB/s best]", flush=True)
