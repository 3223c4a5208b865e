"""one forward + one inverse NTT launch over ~1 GiB (for rocprofv3 --pmc runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import apsu_amd
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ntt_stream_bench import primes
n, bits = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 56
q = primes(n, bits, 4)
G = apsu_amd.HeContext(n=n, coeff_modulus=q, plain_modulus=65537)
polys = (1 << 30) // (n * 8 * 3)
rng = np.random.default_rng(1)
base = np.stack([rng.integers(0, qq, n, dtype=np.uint64) for qq in q[:3]])
ct = np.ascontiguousarray(np.broadcast_to(base, (polys, 3, n))).copy()
for _ in range(2):
    G.transform_to_ntt_inplace(ct, 2)
    G.transform_from_ntt_inplace(ct, 2)
G.close()
