"""GPU tier (pytest -m gpu): every tier-1 entry point of the C ABI (one per seal::Evaluator method)
against the CPU oracle on identical inputs, bit-exact, at every level of every parameter family,
plus the golden vectors of the independent Python model.  Calls go through the C ABI
(libapsu_he_gpu.so) via apsu_amd's ctypes binding."""
import numpy as np
import pytest

import apsu_amd
from golden_util import arr, load
from oracle import ref

pytestmark = pytest.mark.gpu

FAMILIES = [
    (64, [40, 40, 40, 36], 0, 17),          # toy (golden vectors)
    (256, [45, 30, 25], 0, 14),
    (1024, [50, 50], 0, 16),
    (2048, [48], 65537, 0),                 # 100K-1
    (4096, [48, 36, 25], 0, 18),            # 1M-1024-com
    (8192, [56, 56, 56, 50], 0, 22),        # 16M-4096
    (8192, [50, 50, 50, 38, 30], 0, 26),    # 256M-4096
    (16384, [58, 58, 50, 40], 0, 22),       # beyond the shipped sets: one 1024-thread workgroup per limb, 144 KiB of LDS
    (32768, [58, 56, 50, 44], 0, 20),       # SEAL's largest ring: a limb is two LDS-resident halves around one global radix-2 stage
]


def rand_ct(C, rng, polys, lvl, edge=False):
    ct = np.stack([np.stack([rng.integers(0, q, C.n, dtype=np.uint64) for q in C.q[:lvl + 1]]) for _ in range(polys)])
    if edge:                                 # extreme residues in the first coefficients
        for j, q in enumerate(C.q[:lvl + 1]):
            ct[:, j, 0] = q - 1
            ct[:, j, 1] = 0
            ct[:, j, 2] = 1
    return ct


@pytest.fixture(scope="module", params=FAMILIES, ids=lambda f: "n%d_K%d" % (f[0], len(f[1])))
def pair(request):
    n, bits, t, pb = request.param
    C = ref.RefContext(n, bits, t, pb)
    G = apsu_amd.HeContext(n=n, coeff_modulus=C.q, plain_modulus=C.t)
    yield C, G
    G.close()


def test_ntt_roundtrip_and_order(pair):
    C, G = pair
    rng = np.random.default_rng(11)
    for lvl in range(C.first, -1, -1):
        ct = rand_ct(C, rng, 2, lvl, edge=True)
        a, g = ct.copy(), ct.copy()
        C.transform_to_ntt(a, lvl)
        G.transform_to_ntt_inplace(g, lvl)
        assert (a == g).all()
        G.transform_from_ntt_inplace(g, lvl)
        assert (g == ct).all()
        z = np.zeros_like(ct)                # zero operands are legal (bin_bundle.cpp:111-114)
        G.transform_to_ntt_inplace(z, lvl)
        assert not z.any()


def test_plain_ops(pair):
    C, G = pair
    rng = np.random.default_rng(12)
    for lvl in range(C.first, -1, -1):
        ct = rand_ct(C, rng, 2, lvl, edge=True)
        pt = rng.integers(0, C.t, C.n, dtype=np.uint64)
        pt[:3] = [C.t - 1, (C.t + 1) // 2, (C.t + 1) // 2 - 1]           # lift threshold edges
        assert (C.plain_lift_ntt(pt, lvl) == G.transform_plain_to_ntt(pt, lvl)).all()
        short = pt[: C.n // 2].copy()                                     # ragged: fewer coefficients than n
        assert (C.plain_lift_ntt(short, lvl) == G.transform_plain_to_ntt(short, lvl)).all()
        ntt = ct.copy()
        C.transform_to_ntt(ntt, lvl)
        ptn = C.plain_lift_ntt(pt, lvl)
        assert (C.multiply_plain_ntt(ntt, ptn, lvl) == G.multiply_plain_ntt(ntt, ptn, lvl)).all()
        assert (C.multiply_plain_coeff(ct, pt, lvl) == G.multiply_plain(ct, pt, lvl)).all()
        for val in (C.t - 2, 1):                                          # monomial shortcut, both halves
            mono = np.zeros(C.n, dtype=np.uint64)
            mono[C.n - 1] = val
            assert (C.multiply_plain_coeff(ct, mono, lvl) == G.multiply_plain(ct, mono, lvl)).all()
        zero = np.zeros(C.n, dtype=np.uint64)
        assert not G.multiply_plain(ct, zero, lvl).any()
        x, y = ct.copy(), ct.copy()
        C.add_plain(x, pt, lvl)
        G.add_plain_inplace(y, pt, lvl)
        assert (x == y).all()
        ct2 = rand_ct(C, rng, 2, lvl)
        x, y = ct.copy(), ct.copy()
        C.add(x, ct2, lvl)
        G.add_inplace(y, ct2, lvl)
        assert (x == y).all()


def test_mod_switch_and_clear_bits(pair):
    C, G = pair
    rng = np.random.default_rng(13)
    for lvl in range(C.first, 0, -1):
        for polys in (2, 3):
            ct = rand_ct(C, rng, polys, lvl, edge=True)
            assert (C.mod_switch_to_next(ct, lvl) == G.mod_switch_to_next(ct, lvl)).all()
    last = rand_ct(C, rng, 2, 0)
    x, y = last.copy(), last.copy()
    C.clear_irrelevant_bits(x)
    G.clear_irrelevant_bits(y)
    assert (x == y).all()
    with pytest.raises(ValueError):
        G.mod_switch_to_next(last, 0)        # "end of modulus switching chain reached"
    with pytest.raises(ValueError):
        G.transform_to_ntt_inplace(rand_ct(C, rng, 2, C.first), C.first + 1)   # key level is not a data level


def test_multiply_square_relinearize(pair):
    C, G = pair
    rng = np.random.default_rng(14)
    rk = rkh = None
    if C.K > 1:
        rkh = np.stack([np.stack([np.stack([rng.integers(0, q, C.n, dtype=np.uint64) for q in C.q]) for _ in range(2)])
                        for _ in range(C.K - 1)])
        rk = G.upload_relin_keys(rkh)
    for lvl in range(C.first, -1, -1):
        a, b = rand_ct(C, rng, 2, lvl, edge=True), rand_ct(C, rng, 2, lvl)
        assert (C.multiply(a, b, lvl) == G.multiply(a, b, lvl)).all()
        assert (C.square(a, lvl) == G.square(a, lvl)).all()
        assert (C.multiply(a, a.copy(), lvl) == G.square(a, lvl)).all()     # square == multiply(a, a)
        z = np.zeros_like(a)
        assert (C.multiply(a, z, lvl) == G.multiply(a, z, lvl)).all()
        if rk is not None:
            ct3 = rand_ct(C, rng, 3, lvl, edge=True)
            assert (C.relinearize(ct3, rkh, lvl) == G.relinearize(ct3, rk, lvl)).all()


def test_golden_ops_on_gpu():
    ops = load("ops_n64.json")
    q = [int(v, 16) for v in ops["coeff_modulus"]]
    G = apsu_amd.HeContext(n=ops["n"], coeff_modulus=q, plain_modulus=int(ops["plain_modulus"], 16))
    rk = G.upload_relin_keys(arr(ops["rk"]))
    for c in ops["levels"]:
        lvl = c["chain_idx"]
        ct, ct2, ct3, pt, mono = (arr(c[k]) for k in ("ct", "ct2", "ct3", "pt", "mono"))
        g = ct.copy()
        G.transform_to_ntt_inplace(g, lvl)
        assert (g == arr(c["ntt"])).all()
        assert (G.transform_plain_to_ntt(pt, lvl) == arr(c["pt_ntt"])).all()
        assert (G.multiply_plain_ntt(arr(c["ntt"]), arr(c["pt_ntt"]), lvl) == arr(c["multiply_plain_ntt"])).all()
        assert (G.multiply_plain(ct, pt, lvl) == arr(c["multiply_plain"])).all()
        assert (G.multiply_plain(ct, mono, lvl) == arr(c["multiply_plain_mono"])).all()
        y = ct.copy()
        G.add_inplace(y, ct2, lvl)
        assert (y == arr(c["add"])).all()
        y = ct.copy()
        G.add_plain_inplace(y, pt, lvl)
        assert (y == arr(c["add_plain"])).all()
        if lvl > 0:
            assert (G.mod_switch_to_next(ct, lvl) == arr(c["mod_switch"])).all()
        assert (G.multiply(ct, ct2, lvl) == arr(c["multiply"])).all()
        assert (G.square(ct, lvl) == arr(c["square"])).all()
        assert (G.relinearize(ct3, rk, lvl) == arr(c["relinearize"])).all()
    x = arr(ops["clear_in"])
    G.clear_irrelevant_bits(x)
    assert (x == arr(ops["clear_out"])).all()
    G.close()


@pytest.mark.parametrize("form", ["0", "100000000", None])
def test_ntt_large_batch_streams_correctly(form, monkeypatch):
    """>= 256 MiB of distinct limbs through one launch (every workgroup index, every limb position), in both forms of the workgroup and
    with the default selection (None: a large forward launch over the data primes takes the 8-coefficient form built for 8 waves per SIMD)"""
    if form is not None:
        monkeypatch.setenv("APSU_HE_NTT_LATENCY_LIMBS", form)
    C = ref.RefContext(8192, [56, 56, 56, 50], 0, 22)
    G = apsu_amd.HeContext(n=8192, coeff_modulus=C.q, plain_modulus=C.t)
    rng = np.random.default_rng(15)
    polys = 1400                                             # 1400 * 3 limbs * 64 KiB = 262 MiB
    ct = np.stack([rng.integers(0, q, (polys, C.n), dtype=np.uint64) for q in C.q[:3]], axis=1)
    ct = np.ascontiguousarray(ct)
    g = ct.copy()
    G.transform_to_ntt_inplace(g, 2)
    for i in (0, 1, 777, polys - 1):                         # spot-check against the oracle
        e = ct[i:i + 1].copy()
        C.transform_to_ntt(e, 2)
        assert (g[i] == e[0]).all()
    G.transform_from_ntt_inplace(g, 2)
    assert (g == ct).all()                                   # size-independent round-trip property
    G.close()


@pytest.mark.parametrize("n,form", [(2048, None), (4096, "0"), (4096, "100000000"), (8192, "0"), (8192, "100000000"), (16384, None), (32768, None)])
def test_ntt_every_prime_width(n, form, monkeypatch):
    # coefficient primes of every width the engine may meet, across the narrow / wide boundary of the lazy butterflies
    # ((4 log n + 1) q < 2^64 up to 58 bits at n = 8192): forward and inverse transforms against the oracle.
    # form (round 6): every launch in the throughput form of the workgroup (16 coefficients per lane, APSU_HE_NTT_LATENCY_LIMBS=0) or
    # every launch in the latency form (8 per lane), for the ring sizes that have both
    if form is not None:
        monkeypatch.setenv("APSU_HE_NTT_LATENCY_LIMBS", form)
    rng = np.random.default_rng(n)
    for bits in (30, 33, 40, 47, 52, 55, 56, 57, 58, 59, 60):
        C = ref.RefContext(n, [bits, bits], 65537, 0)
        G = apsu_amd.HeContext(n=n, coeff_modulus=C.q, plain_modulus=C.t)
        lvl = C.first
        ct = rand_ct(C, rng, 2, lvl, edge=True)
        a, g = ct.copy(), ct.copy()
        C.transform_to_ntt(a, lvl)
        G.transform_to_ntt_inplace(g, lvl)
        assert (a == g).all(), "forward, %d-bit primes" % bits
        G.transform_from_ntt_inplace(g, lvl)
        assert (g == ct).all(), "inverse, %d-bit primes" % bits
        if C.K > 1:                                  # multiply + relinearize exercise the 61-bit auxiliary primes next to them
            b = rand_ct(C, rng, 2, lvl)
            assert (C.multiply(ct, b, lvl) == G.multiply(ct, b, lvl)).all(), "multiply, %d-bit primes" % bits
        G.close()
