"""GPU tier (pytest -m gpu): parameter sets with ONE coefficient prime and ciphertext products.  SEALContext::using_keyswitching()
is false there, so the reference never relinearises (receiver_osn.cpp:416,430-432 ; bin_bundle.cpp:308-310): powers, the
Paterson-Stockmeyer products and the results are ciphertexts of more than two polynomials.  Bit-exact against the CPU oracle
(whose sized drivers are pinned against the Python model in tests/test_oracle_sized.py)."""
import numpy as np
import pytest

import apsu_amd
import common

pytestmark = pytest.mark.gpu


def run_sized(js, degrees, want_result_polys=None):
    S = common.make_scenario(js, degrees)
    C = S.C
    assert not C.using_keyswitching
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    assert G.powers_dag() == S.nodes
    sizes = C.power_sizes(S.nodes)
    srcs = [[S.src[b][e] for e in S.sources] for b in S.bundle_indices]
    pw = G.compute_powers(S.bundle_indices, srcs, None)
    for b in S.bundle_indices:
        for p in S.targets:
            assert G.power_size(p) == sizes[p]
            ct, ci, is_ntt = pw.download(b, p)
            exp = opw[b][p]
            assert ci == 0 and ct.shape == exp.shape and (ct == exp).all(), "power %d of bundle index %d" % (p, b)
            assert is_ntt == (S.ps_low == 0 or p <= S.ps_low)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    out = G.eval_bundles(gb, pw, None, [b["mask"] for b in S.bundles])
    assert out.shape[1] == G.result_polys
    if want_result_polys is not None:
        assert G.result_polys == want_result_polys
    for i, b in enumerate(S.bundles):
        exp = common.oracle_eval(S, opw, b)
        rs = G.result_size(gb[i])
        assert rs == exp.shape[0], "bundle degree %d" % b["degree"]
        assert (out[i][:rs] == exp).all(), "bundle idx=%d degree=%d" % (b["bundle_idx"], b["degree"])
        assert not out[i][rs:].any()
    return S, G, pw, gb, out


def test_eval_with_unrelinearized_powers():
    # 3 = 1+2, 4 = 2+2 (three polynomials), 5 = 1+4 (four): results of 4 / 3 / 2 / 2 polynomials
    js = common.toy_json(n=256, coeff_bits=(58,), plain_bits=13, ps_low=0, max_items=5, query_powers=(1, 2), felts=7)
    S, G, pw, gb, out = run_sized(js, {0: [5, 3, 1, 0], 1: [4, 2]}, want_result_polys=4)
    assert [G.result_size(b) for b in gb] == [4, 3, 2, 2, 3, 2]
    G.close()


def test_eval_patstock_with_unrelinearized_powers():
    # 2 = 1+1 and 6 = 3+3 have three polynomials; inner (3) x C^3 (2) -> 4, inner (3) x C^6 (3) -> 5
    js = common.toy_json(n=256, coeff_bits=(58,), plain_bits=13, ps_low=2, max_items=8, query_powers=(1, 3), felts=7)
    S, G, pw, gb, out = run_sized(js, {0: [8, 7, 6, 4, 3, 2, 1, 0], 1: [8, 5]}, want_result_polys=5)
    assert [G.result_size(b) for b in gb][:8] == [5, 4, 4, 3, 3, 3, 2, 2]
    G.close()


def test_deeper_dag_odd_sizes_and_larger_ring():
    # depth 3 from one source: 2 = 1+1 (3), 3 = 1+2 (4), 4 = 2+2 (5), 5 = 1+4 (6), 6 = 3+3 (7), 7 = 3+4 (8) ... odd and even sizes,
    # a Paterson-Stockmeyer split with r == 0 blocks, n = 2048
    js = common.toy_json(n=2048, coeff_bits=(60,), plain_bits=14, ps_low=3, max_items=8, query_powers=(1,), felts=7)
    S, G, pw, gb, out = run_sized(js, {0: [8, 7, 4], 1: [8]})
    G.close()


def test_patstock_without_products_in_the_dag_keeps_three_polynomials():
    # every target power is a source (depth 0): only eval_patstock's own product is left unrelinearised (bin_bundle.cpp:238-240)
    js = common.toy_json(n=64, coeff_bits=(60,), plain_bits=9, ps_low=2, max_items=7, query_powers=(1, 2, 3, 6), felts=10)
    S, G, pw, gb, out = run_sized(js, {0: [7, 6, 3, 2]}, want_result_polys=3)
    assert [G.result_size(b) for b in gb] == [3, 3, 3, 2]
    # one product of fresh ciphertexts fits the noise budget of a 60-bit prime: the results decrypt to P(x) + mask
    for i, b in enumerate(S.bundles):
        ok, budget = common.check_semantics(S, b, out[i][:G.result_size(gb[i])])
        assert ok and budget > 0
    G.close()


def test_device_resident_inputs_and_outputs():
    torch = pytest.importorskip("torch")
    js = common.toy_json(n=256, coeff_bits=(58,), plain_bits=13, ps_low=2, max_items=8, query_powers=(1, 3), felts=7)
    S = common.make_scenario(js, {0: [8, 5, 2]})
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    src = torch.from_numpy(np.stack([S.src[0][e] for e in S.sources]).view(np.int64)).cuda()
    w = src[0].numel()
    pw = G.compute_powers([0], [[src.data_ptr() + i * w * 8 for i in range(len(S.sources))]], None, on_device=True)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    masks = torch.from_numpy(np.stack([b["mask"] for b in S.bundles]).view(np.int64)).cuda()
    R = G.result_polys
    out = torch.full((len(gb), R, G.n), -1, dtype=torch.int64, device="cuda")
    G.eval_bundles(gb, pw, None, [masks.data_ptr() + i * G.n * 8 for i in range(len(gb))], out=out.data_ptr(),
                   masks_on_device=True, out_on_device=True)
    G.sync()
    got = out.cpu().numpy().view(np.uint64)
    for i, b in enumerate(S.bundles):
        exp = common.oracle_eval(S, opw, b)
        assert (got[i][:exp.shape[0]] == exp[:, 0]).all() and not got[i][exp.shape[0]:].any()
    G.close()


def test_sized_multiply_tier1_and_seal_size_limit():
    N = 256
    C = common.make_scenario(common.toy_json(n=N, coeff_bits=(50, 50, 40), plain_bits=13, felts=7), {0: []}).C
    G = apsu_amd.HeContext(n=N, coeff_modulus=C.q, plain_modulus=C.t)
    rng = np.random.default_rng(7)
    for lvl in (0, 1):
        for sa, sb in ((2, 2), (2, 3), (3, 3), (5, 4), (9, 8)):
            a = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in C.q[:lvl + 1]]) for _ in range(sa)])
            b = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in C.q[:lvl + 1]]) for _ in range(sb)])
            assert (G.multiply_sized(a, b, lvl) == C.multiply_sized(a, b, lvl)).all(), (lvl, sa, sb)
        sq = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in C.q[:lvl + 1]]) for _ in range(3)])
        assert (G.multiply_sized(sq, sq, lvl) == C.multiply_sized(sq, sq.copy(), lvl)).all()
    wide = np.zeros((9, 1, N), dtype=np.uint64)
    with pytest.raises(ValueError):                      # 17 polynomials: Ciphertext::resize throws in SEAL
        G.multiply_sized(wide, wide.copy(), 0)
    G.close()


def test_products_beyond_seals_largest_ciphertext_raise():
    # one source, 17 targets: power 16 = 8 + 8 would have 17 polynomials
    js = common.toy_json(n=64, coeff_bits=(60,), plain_bits=17, ps_low=0, max_items=17, query_powers=(1,))
    S = common.make_scenario(js, {0: []})
    with pytest.raises(ValueError):
        common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    with pytest.raises(ValueError, match="invalid size"):            # APSU_HE_INVALID_ARGUMENT = std::invalid_argument, as from SEAL
        G.compute_powers([0], [[S.src[0][e] for e in S.sources]], None)
    G.close()


def test_patstock_product_beyond_the_limit_raises_for_that_bundle_only():
    # powers up to 10 polynomials are fine; a full BinBundle multiplies an inner polynomial of 9 by C^9 of 10 -> 18 > 16
    js = common.toy_json(n=64, coeff_bits=(60,), plain_bits=17, ps_low=8, max_items=17, query_powers=(1,))
    S = common.make_scenario(js, {0: [17, 9, 12]})
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], None)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    with pytest.raises(ValueError):
        common.oracle_eval(S, opw, S.bundles[0])
    with pytest.raises(ValueError, match="invalid size"):
        G.eval_bundles(gb[:1], pw, None, [S.bundles[0]["mask"]])
    out = G.eval_bundles(gb[1:], pw, None, [b["mask"] for b in S.bundles[1:]])
    for i, b in enumerate(S.bundles[1:]):
        exp = common.oracle_eval(S, opw, b)
        assert G.result_size(gb[1 + i]) == exp.shape[0] and (out[i][:exp.shape[0]] == exp).all() and not out[i][exp.shape[0]:].any()
    G.close()


def test_framed_query_without_relin_keys_returns_longer_ciphertexts():
    """apsu_he_run_query_request for a single-prime set: no RelinKeys in the QueryRequest (sender_osn.cpp:223-227 only creates them
    when key switching exists), ResultPackages whose SEAL ciphertexts carry every polynomial of the unrelinearised result"""
    from apsu_amd import seal, wire
    js = common.toy_json(n=256, coeff_bits=(58,), plain_bits=13, ps_low=2, max_items=8, query_powers=(1, 3), felts=7)
    S = common.make_scenario(js, {0: [8, 4], 1: [7]})
    opw = common.oracle_powers(S)
    sc = seal.SealContext(js)
    parts = [(e, [sc.ct_save(0, False, S.src[b][e], compr=seal.COMPR_ZLIB) for b in range(S.p["bundle_idx_count"])]) for e in S.sources]
    msg = wire.build_query_request(seal.COMPR_ZLIB, None, parts)
    G = apsu_amd.HeContext(js)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    pkgs = seal.run_query_request(G, sc, msg, gb, [b["mask"] for b in S.bundles], compr=seal.COMPR_ZLIB)
    for i, b in enumerate(S.bundles):
        exp = common.oracle_eval(S, opw, b)
        back = wire.parse_result_package(pkgs[i])
        got = sc.ct_load(back["psu_result"])
        assert back["bundle_idx"] == b["bundle_idx"] and got["chain_idx"] == 0 and not got["is_ntt_form"]
        assert got["data"].shape == exp.shape and (got["data"] == exp).all()
    G.close()
    sc.close()


def test_multi_device_handle_carries_longer_rows():
    """apsu_he_eval_all for a single-prime set: rows of result_polys polynomials (devices {0, 0} rehearse the two-device path)"""
    js = common.toy_json(n=256, coeff_bits=(58,), plain_bits=13, ps_low=2, max_items=8, query_powers=(1, 3), felts=7)
    S = common.make_scenario(js, {0: [8, 4, 2], 1: [7, 8]})
    G = apsu_amd.HeContext(js)
    pw = G.compute_powers(S.bundle_indices, [[S.src[b][e] for e in S.sources] for b in S.bundle_indices], None)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    want = G.eval_bundles(gb, pw, None, [b["mask"] for b in S.bundles])
    flat = [S.src[b][e] for b in range(S.p["bundle_idx_count"]) for e in S.sources]
    units = [(b["bundle_idx"], b["cache_idx"], b["degree"]) for b in S.bundles]
    for devs in ([0], [0, 0]):
        M = apsu_amd.MultiContext(js, devs)
        slots = apsu_amd.partition_bundles(units, S.p["bundle_idx_count"], len(devs))
        for i, b in enumerate(S.bundles):
            M.upload_bundle(slots[i], b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"])
        assert M.result_polys == G.result_polys
        got = M.eval_all(flat, [b["mask"] for b in S.bundles], G.n)
        assert got.shape == want.shape and (got == want).all(), devs
        M.close()
    G.close()
