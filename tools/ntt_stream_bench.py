"""NTT kernel streaming measurement: >= 1 GiB of distinct limbs per launch (HBM, not Infinity Cache).
Reports forward / inverse GB/s (16*n bytes per limb transform) for narrow (coefficient) and wide (61-bit) primes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import apsu_amd
from apsu_amd import engine as E

def primes(n, bits, count):
    out, v = [], ((1 << bits) - 1) // (2 * n) * (2 * n) + 1
    def isp(x):
        if x < 2: return False
        for p in (2,3,5,7,11,13,17,19,23,29,31,37):
            if x % p == 0: return x == p
        d, r = x - 1, 0
        while d % 2 == 0: d //= 2; r += 1
        for a in (2,3,5,7,11,13,17,19,23,29,31,37):
            y = pow(a, d, x)
            if y in (1, x - 1): continue
            for _ in range(r - 1):
                y = y * y % x
                if y == x - 1: break
            else: return False
        return True
    while len(out) < count:
        if isp(v): out.append(v)
        v -= 2 * n
    return out

def run(n, bits, label, gib=1.0):
    q = primes(n, bits, 4)            # 3 data limbs + special
    G = apsu_amd.HeContext(n=n, coeff_modulus=q, plain_modulus=65537)
    L = 3
    polys = int(gib * (1 << 30)) // (n * 8 * L)
    rng = np.random.default_rng(1)
    base = np.stack([rng.integers(0, qq, n, dtype=np.uint64) for qq in q[:L]])
    ct = np.ascontiguousarray(np.broadcast_to(base, (polys, L, n))).copy()
    ct[:, :, 0] += np.arange(polys, dtype=np.uint64)[:, None] % np.uint64(1000)
    ref0 = ct[0].copy()
    G.transform_to_ntt_inplace(ct, 2); G.transform_from_ntt_inplace(ct, 2)     # warm-up (first touch of the arena)
    G.profile_enable(2); G.profile_read()
    for it in range(3):
        G.transform_to_ntt_inplace(ct, 2)
        G.transform_from_ntt_inplace(ct, 2)
    p = G.profile_read()
    assert (ct[0] == ref0).all()
    by = polys * L * 16 * n
    f = by * p["ntt_fwd"][1] / (p["ntt_fwd"][0] * 1e-3) / 1e9
    i = by * p["ntt_inv"][1] / (p["ntt_inv"][0] * 1e-3) / 1e9
    print(f"{label}: n={n} bits={bits} limbs/launch={polys*L} fwd {f:.0f} GB/s ({p['ntt_fwd'][0]/p['ntt_fwd'][1]:.3f} ms)  inv {i:.0f} GB/s ({p['ntt_inv'][0]/p['ntt_inv'][1]:.3f} ms)  both {2*by*p['ntt_fwd'][1]/((p['ntt_fwd'][0]+p['ntt_inv'][0])*1e-3)/1e9:.0f} GB/s", flush=True)
    G.close()

if __name__ == "__main__":
    run(8192, 56, "narrow")
    run(8192, 60, "wide")
    run(4096, 48, "narrow")
    run(2048, 48, "narrow")
