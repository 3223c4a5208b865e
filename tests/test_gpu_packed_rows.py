"""GPU tier: BinBundle plaintexts kept bit-packed in HBM (APSU_HE_PACKED_ROWS=1; Bundle::packed, k_mac<.., PACKED>) against the
dense 64-bit rows and against the oracle: every producer (upload of the reference's plaintexts, the synthetic generator, the
N1 build from roots, images), every consumer (the multiply-accumulate of eval and eval_patstock with every limb range the
Paterson-Stockmeyer schedule uses, coefficient download), images crossing between the two formats, and the DB size."""
import os

import numpy as np
import pytest

import apsu_amd
import common

pytestmark = pytest.mark.gpu


def make_ctx(js, packed):
    old = os.environ.get("APSU_HE_PACKED_ROWS")
    os.environ["APSU_HE_PACKED_ROWS"] = "1" if packed else "0"
    try:
        return apsu_amd.HeContext(js)                    # the switch is read by apsu_he_create
    finally:
        if old is None:
            del os.environ["APSU_HE_PACKED_ROWS"]
        else:
            os.environ["APSU_HE_PACKED_ROWS"] = old


@pytest.mark.parametrize("name,js,degrees,kw", [
    ("toy 40-bit primes", None, {0: [11, 3, 8], 1: [10]}, {}),
    ("16M-4096 (56-bit rows: 7 bytes)", "16M-4096", {0: [1303], 3: [170]}, {}),
    ("256M-4096 (50-bit rows: 6.25 bytes, three-product MAC)", "256M-4096", {1: [700]}, {"roots_frac": 0.02}),
    ("1M-1024-com (48- and 36-bit rows)", "1M-1024-com", {0: [124, 60]}, {}),
])
def test_packed_rows_match_dense_rows_and_the_oracle(name, js, degrees, kw):
    js = common.toy_json() if js is None else common.param_json(js)
    S = common.make_scenario(js, degrees, **kw)
    opw = common.oracle_powers(S)
    want = [common.oracle_eval(S, opw, b) for b in S.bundles]
    res = {}
    sizes = {}
    images = {}
    for packed in (False, True):
        G = make_ctx(js, packed)
        rk = G.upload_relin_keys(S.rk)
        pw = G.compute_powers(S.bundle_indices, [[S.src[b][e] for e in S.sources] for b in S.bundle_indices], rk)
        gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
        out = G.eval_bundles(gb, pw, rk, [b["mask"] for b in S.bundles])
        for i, b in enumerate(S.bundles):
            assert (out[i] == want[i]).all(), "%s, packed=%s, BinBundle %d" % (name, packed, i)
            # every stored coefficient comes back as it went in, whatever the row format
            for d in sorted({1, 2, b["degree"] // 2, b["degree"]}):
                if 0 < d <= b["degree"]:
                    got, kind = G.bundle_coeff(gb[i], d)
                    if kind == 1:
                        assert (np.asarray(got).reshape(-1) == np.asarray(b["coeffs"][d]).reshape(-1)).all(), (name, packed, d)
        res[packed] = out
        sizes[packed] = sum(x.db_bytes for x in gb)
        images[packed] = [G.save_bundle(x) for x in gb]
        G.close()
    assert sizes[True] < sizes[False], "packed rows must shrink the database (%d vs %d bytes)" % (sizes[True], sizes[False])
    # images cross over: a dense image loaded by a packed context (and the reverse) is converted and evaluates to the same bits
    for packed in (False, True):
        G = make_ctx(js, packed)
        rk = G.upload_relin_keys(S.rk)
        pw = G.compute_powers(S.bundle_indices, [[S.src[b][e] for e in S.sources] for b in S.bundle_indices], rk)
        gb = [G.load_bundle(img) for img in images[not packed]]
        out = G.eval_bundles(gb, pw, rk, [b["mask"] for b in S.bundles])
        for i in range(len(S.bundles)):
            assert (out[i] == want[i]).all(), "%s: image of the other format, packed=%s" % (name, packed)
        assert sum(x.db_bytes for x in gb) == sizes[packed]
        G.close()


def test_generated_and_built_bundles_in_packed_rows():
    """the synthetic generator and the N1 build from roots end in the same packing step: compared with a dense context's results"""
    js = common.param_json("16M-4096")
    outs = []
    for packed in (False, True):
        G = make_ctx(js, packed)
        rng = np.random.default_rng(5)
        K, n, first = G.K, G.n, G.first_chain_idx
        rk = G.upload_relin_keys(np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in G.q]) for _ in range(2)]) for _ in range(K - 1)]))
        src = [[np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in G.q[:first + 1]]) for _ in range(2)]) for _ in range(G.source_power_count)]]
        pw = G.compute_powers([0], src, rk)
        gb = [G.random_bundle(0, 0, 1303, 77), G.random_bundle(0, 1, 171, 78)]
        mask = [rng.integers(0, G.t, n, dtype=np.uint64) for _ in gb]
        outs.append(G.eval_bundles(gb, pw, rk, mask))
        G.close()
    for a, b in zip(*outs):
        assert (a == b).all()
