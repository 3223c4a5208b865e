"""Randomised differential test: random small parameter sets (ring size, prime widths incl. the wide-prime kernels,
level counts, plain modulus width, Paterson-Stockmeyer shape, query powers, ragged BinBundle degrees) — the GPU path must
equal the oracle bit for bit on every one, whether or not the noise budget survives (parity is about the arithmetic)."""
import json
import os
import random

import numpy as np
import pytest

import common
from common import ref

pytestmark = pytest.mark.gpu

pytest.importorskip("torch")
import apsu_amd                                            # noqa: E402


def random_params(rng):
    n = rng.choice([64, 256, 1024, 4096])
    K = rng.choice([2, 3, 4, 5])
    widths = [rng.choice([30, 36, 40, 45, 50, 56, 60]) for _ in range(K)]
    felts = rng.choice([2, 3, 4, 5, 6, 7, 8])
    # plain moduli above 2^32 take the long-division branch of add_plain's rounding (the noise budget is gone there,
    # bit-exactness is still checked)
    plain_bits = rng.choice([b for b in (16, 17, 18, 20, 22, 28, 33, 41, 44) if 80 <= (b - 1) * felts <= 128] or [17])
    max_items = rng.randint(3, 24)
    ps_low = rng.choice([0, 0] + list(range(2, max(3, max_items // 2 + 1))))
    targets = ref.create_powers_set(ps_low, max_items)
    # sources: 1, a few low powers, and (with PS) some multiples of ps_low + 1
    low = [p for p in targets if not ps_low or p <= ps_low]
    high = [p for p in targets if ps_low and p > ps_low]
    src = {1} | set(rng.sample(low, min(len(low), rng.randint(1, 3))))
    if high:
        src |= {high[0]} | set(rng.sample(high, min(len(high), rng.randint(0, 2))))
    ipb = n // felts
    return json.dumps({
        "table_params": {"hash_func_count": 3, "table_size": ipb * 2, "max_items_per_bin": max_items},
        "item_params": {"felts_per_item": felts},
        "query_params": {"ps_low_degree": ps_low, "query_powers": sorted(src)},
        "seal_params": {"plain_modulus_bits": plain_bits, "poly_modulus_degree": n, "coeff_modulus_bits": widths},
    }), max_items


# APSU_FUZZ_SEEDS=n widens the sweep for one-off runs (profiles/r02_fuzz_extended.txt: 400 seeds)
@pytest.mark.parametrize("seed", range(int(os.environ.get("APSU_FUZZ_SEEDS", "28"))))
def test_random_parameter_sets(seed):
    rng = random.Random(1000 + seed)
    for _ in range(40):                                     # draw until the reference's own validation accepts the set
        js, max_items = random_params(rng)
        try:
            G = apsu_amd.HeContext(js)
            S = None
            degs = {0: sorted({max_items, rng.randint(0, max_items), rng.randint(0, max_items)}, reverse=True)}
            if rng.random() < 0.5:
                degs[1] = [rng.randint(1, max_items)]
            S = common.make_scenario(js, degs, seed=seed * 17 + 3)
        except (ValueError, apsu_amd.ApsuHeError, RuntimeError, AssertionError, KeyError):
            continue
        break
    else:
        pytest.skip("no valid parameter set drawn")
    assert G.powers_dag() == S.nodes
    opw = common.oracle_powers(S)
    rk = G.upload_relin_keys(S.rk) if S.rk is not None else None
    pw = G.compute_powers(S.bundle_indices, [[S.src[b][e] for e in S.sources] for b in S.bundle_indices], rk)
    for b in S.bundle_indices:
        for p in S.targets:
            ct, _, _ = pw.download(b, p)
            assert (ct == opw[b][p]).all(), "params %s: power %d of index %d" % (js, p, b)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    out = G.eval_bundles(gb, pw, rk, [b["mask"] for b in S.bundles])
    for i, b in enumerate(S.bundles):
        assert (out[i] == common.oracle_eval(S, opw, b)).all(), "params %s: bundle %d degree %d" % (js, b["bundle_idx"], b["degree"])
        ok, budget = common.check_semantics(S, b, out[i])
        if budget >= 8:                                      # (an overflowed noise reads as a budget of 0-2 bits: that is the
            assert ok                                        #  parameter set's problem, not parity's)


def random_single_prime_params(rng):
    n = rng.choice([64, 256, 1024])
    felts = rng.choice([5, 6, 7, 8])
    plain_bits = rng.choice([b for b in (16, 17, 18, 20, 22) if 80 <= (b - 1) * felts <= 128] or [17])
    max_items = rng.randint(3, 14)
    ps_low = rng.choice([0, 0] + list(range(2, max(3, max_items // 2 + 1))))
    targets = ref.create_powers_set(ps_low, max_items)
    low = [p for p in targets if not ps_low or p <= ps_low]
    high = [p for p in targets if ps_low and p > ps_low]
    src = {1} | set(rng.sample(low, min(len(low), rng.randint(1, 4))))
    if high:
        src |= {high[0]} | set(rng.sample(high, min(len(high), rng.randint(0, 2))))
    ipb = n // felts
    return json.dumps({
        "table_params": {"hash_func_count": 3, "table_size": ipb * 2, "max_items_per_bin": max_items},
        "item_params": {"felts_per_item": felts},
        "query_params": {"ps_low_degree": ps_low, "query_powers": sorted(src)},
        "seal_params": {"plain_modulus_bits": plain_bits, "poly_modulus_degree": n,
                        "coeff_modulus_bits": [rng.choice([40, 50, 56, 60])]},
    }), max_items


@pytest.mark.parametrize("seed", range(int(os.environ.get("APSU_FUZZ_SEEDS_1P", "16"))))
def test_random_single_prime_sets(seed):
    """one coefficient prime: nothing is ever relinearised (receiver_osn.cpp:416,430-432), ciphertexts grow up to SEAL's limit of
    16 polynomials, beyond which both sides must raise std::invalid_argument's counterpart"""
    rng = random.Random(5000 + seed)
    for _ in range(40):
        js, max_items = random_single_prime_params(rng)
        try:
            G = apsu_amd.HeContext(js)
            degs = {0: sorted({max_items, rng.randint(0, max_items), rng.randint(0, max_items)}, reverse=True), 1: [rng.randint(1, max_items)]}
            S = common.make_scenario(js, degs, seed=seed * 13 + 1)
        except (ValueError, apsu_amd.ApsuHeError, RuntimeError, AssertionError, KeyError):
            continue
        break
    else:
        pytest.skip("no valid parameter set drawn")
    assert G.powers_dag() == S.nodes
    srcs = [[S.src[b][e] for e in S.sources] for b in S.bundle_indices]
    try:
        opw = common.oracle_powers(S)
    except ValueError:                                       # a product above 16 polynomials
        with pytest.raises(ValueError, match="invalid size"):
            G.compute_powers(S.bundle_indices, srcs, None)
        return
    pw = G.compute_powers(S.bundle_indices, srcs, None)
    for b in S.bundle_indices:
        for p in S.targets:
            ct, _, _ = pw.download(b, p)
            assert ct.shape == opw[b][p].shape and (ct == opw[b][p]).all(), "params %s: power %d of index %d" % (js, p, b)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    for g, b in zip(gb, S.bundles):                          # one call per BinBundle: a result above the limit raises on its own
        try:
            exp = common.oracle_eval(S, opw, b)
        except ValueError:
            with pytest.raises(ValueError, match="invalid size"):
                G.eval_bundles([g], pw, None, [b["mask"]])
            continue
        out = G.eval_bundles([g], pw, None, [b["mask"]])[0]
        assert G.result_size(g) == exp.shape[0]
        assert (out[:exp.shape[0]] == exp).all() and not out[exp.shape[0]:].any(), "params %s: bundle %d degree %d" % (js, b["bundle_idx"], b["degree"])
