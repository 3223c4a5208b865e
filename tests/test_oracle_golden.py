"""CPU tier: the C oracle (oracle/ref_*.c) against the golden vectors produced by the independent
Python big-int model (oracle/make_golden.py -> tests/golden).  The reference itself holds no
fixtures for this path, so these are the pins the build created."""
import numpy as np
import pytest

import common
from golden_util import arr, load
from oracle import ref


@pytest.fixture(scope="module")
def ops():
    return load("ops_n64.json")


@pytest.fixture(scope="module")
def ctx(ops):
    return ref.RefContext(ops["n"], ops["coeff_bits"], 0, ops["plain_bits"])


def test_constants(ops, ctx):
    assert ctx.q == [int(v, 16) for v in ops["coeff_modulus"]]
    assert ctx.t == int(ops["plain_modulus"], 16)
    assert ctx.psi == [int(v, 16) for v in ops["psi"]]
    assert ctx.irrelevant_bit_count() == ops["irrelevant_bit_count"]


def test_ops_all_levels(ops, ctx):
    rk = arr(ops["rk"])
    for c in ops["levels"]:
        lvl = c["chain_idx"]
        ct, ct2, ct3, pt, mono = (arr(c[k]) for k in ("ct", "ct2", "ct3", "pt", "mono"))
        a = ct.copy()
        ctx.transform_to_ntt(a, lvl)
        assert (a == arr(c["ntt"])).all()
        ctx.transform_from_ntt(a, lvl)
        assert (a == ct).all()
        assert (ctx.plain_lift_ntt(pt, lvl) == arr(c["pt_ntt"])).all()
        assert (ctx.multiply_plain_ntt(arr(c["ntt"]), arr(c["pt_ntt"]), lvl) == arr(c["multiply_plain_ntt"])).all()
        assert (ctx.multiply_plain_coeff(ct, pt, lvl) == arr(c["multiply_plain"])).all()
        assert (ctx.multiply_plain_coeff(ct, mono, lvl) == arr(c["multiply_plain_mono"])).all()
        x = ct.copy()
        ctx.add(x, ct2, lvl)
        assert (x == arr(c["add"])).all()
        x = ct.copy()
        ctx.add_plain(x, pt, lvl)
        assert (x == arr(c["add_plain"])).all()
        if lvl > 0:
            assert (ctx.mod_switch_to_next(ct, lvl) == arr(c["mod_switch"])).all()
        assert (ctx.multiply(ct, ct2, lvl) == arr(c["multiply"])).all()
        assert (ctx.square(ct, lvl) == arr(c["square"])).all()
        assert (ctx.relinearize(ct3, rk, lvl) == arr(c["relinearize"])).all()
    x = arr(ops["clear_in"])
    ctx.clear_irrelevant_bits(x)
    assert (x == arr(ops["clear_out"])).all()


def test_path_golden():
    g = load("path_n64.json")
    C = ref.RefContext(g["n"], g["coeff_bits"], 0, g["plain_bits"])
    targets = ref.create_powers_set(g["ps_low_degree"], g["max_items_per_bin"])
    assert targets == g["targets"]
    depth, nodes = ref.powers_dag(g["query_powers"], targets)
    assert depth == g["dag_depth"] and [list(nd) for nd in nodes] == g["dag_nodes"]
    rk = arr(g["rk"])
    srcs = {int(e): arr(ct) for e, ct in g["sources"].items()}
    pw = C.compute_powers(srcs, nodes, rk, g["ps_low_degree"])
    for p, ct in g["powers"].items():
        assert (pw[int(p)] == arr(ct)).all(), "power %s" % p
    plist = [None] * (g["max_items_per_bin"] + 1)
    for p, ct in pw.items():
        plist[p] = ct
    sk = np.array([[(v % q) for v in g["secret"]] for q in C.q], dtype=np.uint64)
    for j in range(C.K):                       # oracle keeps the secret in NTT form
        tmp = sk[j].reshape(1, 1, -1).copy()
        # ntt tables are per limb j: use a 1-limb view through transform at level j trick -> do via encode path
        sk[j] = _ntt_limb(C, tmp.reshape(-1), j)
    for b in g["bundles"]:
        coeffs = [arr(c) for c in b["coeffs"]]
        mask = arr(b["mask"])
        if g["ps_low_degree"] > 1 and g["ps_low_degree"] < b["degree"]:
            out = C.eval_patstock(plist, coeffs, g["ps_low_degree"], rk, mask)
        else:
            out = C.eval(plist, coeffs, plist[1].shape[1] - 1, mask)
        assert (out == arr(b["result"])).all(), "bundle degree %d" % b["degree"]
        pt, budget = C.decrypt(sk, out, 0)
        assert (C.decode(pt) == arr(b["expected_slots"])).all()
        assert budget == b["noise_budget"] or abs(budget - b["noise_budget"]) <= 1


def _ntt_limb(C, limb, j):
    """forward NTT of one limb under modulus q_j using the oracle's multi-limb transform."""
    ct = np.zeros((1, C.first + 1 if j <= C.first else j + 1, C.n), dtype=np.uint64)
    if j <= C.first:
        ct[0, j] = limb
        C.transform_to_ntt(ct, C.first)
        return ct[0, j]
    # special prime (key level): reuse the product of a unit ct via relin-free route: schoolbook through encode is
    # unnecessary -- the oracle exposes key-level NTT only through keygen, so rebuild with a one-limb context
    C1 = ref.RefContext(C.n, coeff_modulus=[C.q[j]], plain_modulus=C.t)
    one = limb.reshape(1, 1, -1).copy()
    C1.transform_to_ntt(one, 0)
    return one.reshape(-1)


def test_scenarios_decrypt_semantics():
    """decrypt(eval(query)) == P_bin(x) + mask for every slot; members of a bin give exactly the mask."""
    for js, degs in ((common.toy_json(), {0: [10, 3, 11], 1: [7]}),
                     (common.toy_json(ps_low=0, max_items=6, query_powers=(1, 2, 3, 5)), {0: [6, 1]}),
                     (common.toy_json(n=256, coeff_bits=(50, 50, 36), plain_bits=18, ps_low=2, max_items=8,
                                      query_powers=(1, 3)), {0: [8, 5]})):
        S = common.make_scenario(js, degs)
        pw = common.oracle_powers(S)
        for b in S.bundles:
            out = common.oracle_eval(S, pw, b)
            ok, budget = common.check_semantics(S, b, out)
            assert ok and budget > 0
            # member slots: polynomial vanishes -> decrypts to the mask alone
            pt, _ = S.C.decrypt(S.sk, out, 0)
            got = S.C.decode(pt)
            nroots = int(S.C.n * 0.5) if b["degree"] >= 1 else 0
            assert (got[:nroots] == b["mask_vals"][:nroots]).all()


def test_vec_to_oc_block_matches_model_and_known_answers():
    # N4 packing (receiver_osn.cpp:53-73): C oracle vs the independent Python model, plus hand-computed answers
    from oracle import pymodel
    C = ref.RefContext(64, coeff_bits=[40, 40, 36], plain_bits=17)          # a 17-bit t -> len = 17
    assert (1 << 16) <= C.t < (1 << 17)
    # even felts: lower collects felts 0,2,.. and higher felts 1,3,.., most recent in the low bits
    v = np.array([0x1ABCD, 0x0F00F, 0x00001, 0x10000], dtype=np.uint64)
    got = C.vec_to_oc_block(v, 4)
    assert int(got[0][0]) == ((0x1ABCD << 17) | 0x00001) and int(got[0][1]) == ((0x0F00F << 17) | 0x10000)
    # odd felts: the last felt is split at len/2 = 8 bits, its upper part shifted down by 7 (sic)
    v = np.array([5, 6, 0x1FFFF], dtype=np.uint64)
    got = C.vec_to_oc_block(v, 3)
    assert int(got[0][0]) == ((0xFF << 17) | 5) and int(got[0][1]) == (((0x1FF00 >> 7) << 17) | 6)
    rng = np.random.default_rng(3)
    for felts in (2, 3, 5, 6, 7, 8):
        vals = rng.integers(0, C.t, 4 * felts, dtype=np.uint64)
        blocks = C.vec_to_oc_block(vals, felts)
        for i in range(4):
            lo, hi = pymodel.vec_to_oc_block([int(x) for x in vals[i * felts:(i + 1) * felts]], felts, C.t)
            assert (int(blocks[i][0]), int(blocks[i][1])) == (lo, hi)
    # a 22-bit plain modulus with 7 felts overflows 64 bits per half: the wrap-around must match
    C2 = ref.RefContext(64, coeff_bits=[50, 50, 40], plain_bits=22)
    vals = rng.integers(0, C2.t, 7, dtype=np.uint64)
    lo, hi = pymodel.vec_to_oc_block([int(x) for x in vals], 7, C2.t)
    got = C2.vec_to_oc_block(vals, 7)
    assert (int(got[0][0]), int(got[0][1])) == (lo, hi)
