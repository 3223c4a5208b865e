"""CPU tier: parameter sets with ONE coefficient prime and a PowersDag that has products.  The reference then never
relinearises (receiver_osn.cpp:416,430-432 ; bin_bundle.cpp:308-310), so powers, Paterson-Stockmeyer products and results are
ciphertexts of more than two polynomials (SEAL's multiply: size_a + size_b - 1, add_inplace: the longer operand).  The C
oracle's sized drivers are checked here against the independent Python big-int model on the same inputs, and against the
plaintext meaning where the noise budget allows (it rarely does: one 60-bit prime carries one product at most)."""
import numpy as np
import pytest

import common
from oracle import pymodel, ref


def _lists(d):
    return {p: ct.tolist() for p, ct in d.items()}


def _scenario(ps_low, max_items, query_powers, degrees, n=32, bits=(58,), plain_bits=10):
    js = common.toy_json(n=n, coeff_bits=bits, plain_bits=plain_bits, ps_low=ps_low, max_items=max_items,
                         query_powers=query_powers, felts=4)
    return common.make_scenario(js, {0: degrees})


CASES = [
    # eval: 3 = 1+2, 4 = 2+2 (size 3), 5 = 1+4 (size 4)
    dict(ps_low=0, max_items=5, query_powers=(1, 2), degrees=[5, 3, 1, 0], sizes={1: 2, 2: 2, 3: 3, 4: 3, 5: 4}, results=[4, 3, 2, 2]),
    # eval_patstock: 2 = 1+1 and 6 = 3+3 (size 3); inner (size 3) x C^3 (size 2) -> 4, x C^6 (size 3) -> 5
    dict(ps_low=2, max_items=8, query_powers=(1, 3), degrees=[8, 7, 4, 2], sizes={1: 2, 2: 3, 3: 2, 6: 3}, results=[5, 4, 3, 3]),
]


@pytest.mark.parametrize("case", CASES, ids=["eval", "eval_patstock"])
def test_sized_path_matches_python_model(case):
    S = _scenario(case["ps_low"], case["max_items"], case["query_powers"], case["degrees"])
    C = S.C
    assert not C.using_keyswitching and S.depth > 0
    assert C.power_sizes(S.nodes) == case["sizes"]
    M = pymodel.Model(C.n, C.q, C.t)
    pw = common.oracle_powers(S)[0]
    mpw = pymodel.compute_powers(M, _lists(S.src[0]), S.nodes, None, S.ps_low)
    for p, ct in pw.items():
        assert ct.shape[0] == case["sizes"][p]
        assert ct.tolist() == mpw[p], "power %d" % p
    for b, want in zip(S.bundles, case["results"]):
        out = common.oracle_eval(S, {0: pw}, b)
        assert out.shape == (want, 1, C.n), "degree %d" % b["degree"]
        coeffs = [c.tolist() for c in b["coeffs"]]
        mp = [None] * (S.p["max_items_per_bin"] + 1)
        for p, ct in mpw.items():
            mp[p] = ct
        if S.ps_low > 1 and S.ps_low < b["degree"]:
            mout = pymodel.eval_patstock(M, mp, coeffs, S.ps_low, None, b["mask"].tolist())
        else:
            mout = pymodel.eval_plain(M, mp, coeffs, 0, b["mask"].tolist())
        assert out.tolist() == mout, "degree %d" % b["degree"]


def test_sized_multiply_is_the_ring_product_and_size_limit():
    """decrypt(a x b) = decrypt(a) * decrypt(b) for sizes 2 x 3 while the noise budget lasts; 17 polynomials are refused"""
    js = common.toy_json(n=64, coeff_bits=(60,), plain_bits=9, ps_low=0, max_items=2, query_powers=(1, 2), felts=4)
    S = common.make_scenario(js, {0: []})
    C = S.C
    a, b = S.src[0][1], S.src[0][2]
    ab = C.multiply_sized(a, b, 0)
    assert ab.shape[0] == 3 and (ab == C.multiply(a, b, 0)).all()
    pt, budget = C.decrypt(S.sk, ab, 0)
    x = S.x[0].astype(object)
    assert budget > 0 and (C.decode(pt).astype(object) == (x ** 3) % C.t).all()
    wide = np.zeros((9, 1, C.n), dtype=np.uint64)
    assert C.multiply_sized(wide, wide[:8], 0).shape[0] == 16
    with pytest.raises(ValueError):
        C.multiply_sized(wide, wide, 0)


def test_algebraize_item_is_the_little_endian_bit_split():
    """util::algebraize_item (db_encoding.cpp:209-256,360-366): field element j = bits [j*b, (j+1)*b) of the item as one
    little-endian 128-bit integer, b = bits(t) - 1; joining them gives back the first item_bit_count bits"""
    rng = np.random.default_rng(5)
    for n, bits, plain_bits, felts in ((64, (40, 40, 36), 17, 5), (64, (40, 40, 36), 17, 8), (256, (58,), 13, 7), (64, (60,), 9, 10)):
        C = ref.RefContext(n, list(bits), 0, plain_bits)
        b = int(C.t).bit_length() - 1
        items = rng.integers(0, 256, (50, 16), dtype=np.uint8)
        items[0] = 0
        items[1] = 255
        got = C.algebraize_items(items, felts)
        for it, row in zip(items, got):
            v = int.from_bytes(bytes(it), "little")
            assert [int(x) for x in row] == [(v >> (j * b)) & ((1 << b) - 1) for j in range(felts)]
            assert sum(int(x) << (j * b) for j, x in enumerate(row)) == v & ((1 << (felts * b)) - 1)
            assert all(int(x) < C.t for x in row)
    # known answer: the item 0x0123456789abcdef fedcba9876543210 (bytes 10 32 54 ... ef cd ab ...), 16-bit field elements
    C = ref.RefContext(64, [40, 40, 36], 0, 17)
    item = np.frombuffer(bytes.fromhex("1032547698badcfeefcdab8967452301"), dtype=np.uint8)
    assert [hex(int(x)) for x in C.algebraize_items(item, 5)[0]] == ["0x3210", "0x7654", "0xba98", "0xfedc", "0xcdef"]
