"""GPU tier — N3 end to end: a framed QueryRequest as the reference's querier would send it (FlatBuffers framing of rop.fbs around
SEAL-serialised objects: SEEDED symmetric ciphertexts and seeded RelinKeys under zlib, sender/apsu/plaintext_powers.cpp:41-46,
sender_osn.cpp:223-227,488) goes through parse -> SEAL codec (seed expansion) -> apsu_he_relin_upload / apsu_he_compute_powers /
apsu_he_eval_bundles, and every result must equal the oracle's on the same (expanded) inputs.  The SEAL object format is
UNPINNED (apsu_amd/csrc/seal_codec.h); the messages are produced here by the library's writer and cross-checked against the
independent Python model of tests/test_seal_codec.py."""
import numpy as np
import pytest

import apsu_amd
import common
from apsu_amd import seal, wire
from test_seal_codec import obj, ct_members, parms_id as model_parms_id, sample_poly_uniform as model_sample

pytestmark = pytest.mark.gpu


def reseed_ciphertext(C, sk, ct, a_new):
    """(c0, a) -> (c0 + (a - a_new) s, a_new): the same plaintext and noise under another uniform polynomial (sk is in NTT form)"""
    L = ct.shape[1]
    lvl = L - 1
    d = np.stack([(ct[1, j].astype(object) - a_new[j].astype(object)) % C.q[j] for j in range(L)]).astype(np.uint64)[None]
    dn = np.ascontiguousarray(d)
    C.transform_to_ntt(dn, lvl)
    ds = C.multiply_plain_ntt(dn, np.ascontiguousarray(sk[:L]), lvl)
    C.transform_from_ntt(ds, lvl)
    c0 = np.stack([(ct[0, j].astype(object) + ds[0, j].astype(object)) % C.q[j] for j in range(L)]).astype(np.uint64)
    return np.stack([c0, a_new])


@pytest.mark.parametrize("compr", [seal.COMPR_ZLIB, seal.COMPR_NONE, seal.COMPR_ZSTD])
def test_framed_seeded_query_runs_through_the_engine(compr):
    js = common.toy_json()
    S = common.make_scenario(js, {0: [11, 5], 1: [8]})
    C = S.C
    K, n, first = C.K, C.n, C.first
    sc = seal.SealContext(js)
    # ---- the querier's side: every source power re-keyed to a seeded ciphertext, saved the way Serializable<Ciphertext> is
    rng = np.random.default_rng(99)
    parts, expanded = [], {}
    for e in S.sources:
        cts = []
        for b in range(S.p["bundle_idx_count"]):
            base = S.src[b][e] if b in S.src else S.src[S.bundle_indices[0]][e]
            seed = [int(x) for x in rng.integers(0, 2**63, 8, dtype=np.uint64)]
            a = sc.sample_poly_uniform(first, seed, first + 1, n)
            assert (a == model_sample(seed, [int(v) for v in C.q[:first + 1]], n)).all()
            ct = reseed_ciphertext(C, S.sk, base, a)
            expanded[(b, e)] = ct
            blob = sc.ct_save(first, False, ct, seed=seed, compr=compr)
            assert blob == obj(ct_members(model_parms_id(n, [int(v) for v in C.q[:first + 1]], C.t), False, ct, (4, 0), seed=seed), compr)
            cts.append(blob)
        parts.append((e, cts))
    rk_blob = sc.relin_keys_save(S.rk, compr=compr)                   # (keys unseeded: the oracle's own uniform polynomials)
    msg = wire.build_query_request(compr, rk_blob, parts)
    # ---- the DB side: nothing but the message and the parameters
    ctype, rk_in, parts_in = wire.parse_query_request(msg)
    assert ctype == compr
    ksk, used = sc.relin_keys_load(rk_in)
    assert used == len(rk_in) and (ksk.reshape(S.rk.shape) == S.rk).all()
    src = {}
    for e, cts in parts_in:
        for b, blob in enumerate(cts):
            got = sc.ct_load(blob)
            assert got["seeded"] and got["chain_idx"] == first and not got["is_ntt_form"] and got["consumed"] == len(blob)
            assert (got["data"] == expanded[(b, e)]).all()
            src[(b, e)] = got["data"]
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(ksk.reshape(S.rk.shape))
    # the production wiring: c0 from the message, c1 expanded ON THE DEVICE from the stored seed (apsu_he_seed_expand), the query
    # never exists expanded on the host; must give the same powers as the host-expanded ciphertexts
    import torch
    Lf = first + 1
    order = [(b, e) for b in S.bundle_indices for e in S.sources]
    blobs = {(b, e): cts[b] for e, cts in parts_in for b in range(len(cts))}
    dev = torch.zeros((len(order), 2, Lf, n), dtype=torch.int64, device="cuda")
    seeds = []
    for i, key in enumerate(order):
        u = sc.ct_load_unexpanded(blobs[key], Lf, n)
        assert u["seeded"] and u["chain_idx"] == first
        dev[i, 0] = torch.from_numpy(u["data"].view(np.int64)).cuda()
        seeds.append(u["seed"])
    G.seed_expand(first, np.array(seeds, dtype=np.uint64), [dev.data_ptr() + (i * 2 + 1) * Lf * n * 8 for i in range(len(order))])
    assert (dev.cpu().numpy().view(np.uint64) == np.stack([src[k] for k in order])).all()
    ns = len(S.sources)
    pw_dev = G.compute_powers(S.bundle_indices, [[dev.data_ptr() + ((bi * ns + si) * 2 * Lf * n) * 8 for si in range(ns)]
                                                 for bi in range(len(S.bundle_indices))], rk, on_device=True)
    pw = G.compute_powers(S.bundle_indices, [[src[(b, e)] for e in S.sources] for b in S.bundle_indices], rk)
    for b in S.bundle_indices:
        for power in S.targets:
            assert (pw_dev.download(b, power)[0] == pw.download(b, power)[0]).all()
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    out = G.eval_bundles(gb, pw, rk, [b["mask"] for b in S.bundles])
    # ---- the oracle on the same expanded inputs, and the plaintext meaning of every result
    S.src = {b: {e: expanded[(b, e)] for e in S.sources} for b in S.bundle_indices}
    opw = common.oracle_powers(S)
    for i, b in enumerate(S.bundles):
        assert (out[i] == common.oracle_eval(S, opw, b)).all()
        assert common.check_semantics(S, b, out[i])[0]
    # ---- the same in ONE call below the C ABI: request bytes in, ResultPackage bytes out (apsu_he_run_query_request)
    pkgs = seal.run_query_request(G, sc, msg, gb, [b["mask"] for b in S.bundles], compr=compr)
    assert len(pkgs) == len(S.bundles)
    for i, b in enumerate(S.bundles):
        back = wire.parse_result_package(pkgs[i])
        assert back["bundle_idx"] == b["bundle_idx"] and back["cache_idx"] == b["cache_idx"] and back["labels"] == []
        got = sc.ct_load(back["psu_result"])
        assert got["chain_idx"] == 0 and not got["seeded"] and (got["data"] == out[i]).all()
    # ... and through the multi-device handle (query decoded on the first device, the others fetch over xGMI; {0, 0} rehearses two)
    units = [(b["bundle_idx"], b["cache_idx"], b["degree"]) for b in S.bundles]
    for devs in ([0], [0, 0]):
        M = apsu_amd.MultiContext(js, devs)
        slots = apsu_amd.partition_bundles(units, S.p["bundle_idx_count"], len(devs))
        for i, b in enumerate(S.bundles):
            M.upload_bundle(slots[i], b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"])
        assert seal.multi_run_query_request(M, sc, msg, [b["mask"] for b in S.bundles], compr=compr) == pkgs, devs
        M.close()
    # unseeded ciphertexts (a querier that saved plain Ciphertext objects) and SEAL 3.6 objects take the same call
    plain_parts = [(e, [sc.ct_save(first, False, expanded[(b, e)], compr=compr, version=(3, 6)) for b in range(S.p["bundle_idx_count"])])
                   for e in S.sources]
    pk2 = seal.run_query_request(G, sc, wire.build_query_request(compr, rk_blob, plain_parts), gb, [b["mask"] for b in S.bundles], compr=compr)
    assert pk2 == pkgs
    # a ciphertext that names the first data level but claims ONE coefficient prime (a forged coeff_modulus_size): refused, nothing is
    # read behind its short coefficient array
    short = expanded[(0, S.sources[0])][:, :1]
    forged = obj(ct_members(model_parms_id(n, [int(v) for v in C.q[:first + 1]], C.t), False, short, (4, 0)), compr)
    bad_parts = [(e, ([forged] + list(c[1:])) if e == S.sources[0] else c) for e, c in plain_parts]
    with pytest.raises((ValueError, apsu_amd.ApsuHeError), match="coeff_modulus_size|first data level"):
        seal.run_query_request(G, sc, wire.build_query_request(compr, rk_blob, bad_parts), gb, [b["mask"] for b in S.bundles], compr=compr)
    # a coefficient that is not a residue of its prime (SEALObject::extract -> is_valid_for -> is_data_valid_for in the reference): refused --
    # the engine's lazy transforms take source limbs as they are, an out-of-range word must not reach them
    for poly, limb in ((0, 0), (1, first)):
        wrong = expanded[(0, S.sources[0])].copy()
        wrong[poly, limb, 5] = int(C.q[limb])                        # == q: the smallest value outside [0, q)
        bad_ct = sc.ct_save(first, False, wrong, compr=compr)
        bad_parts = [(e, ([bad_ct] + list(c[1:])) if e == S.sources[0] else c) for e, c in plain_parts]
        with pytest.raises((ValueError, apsu_amd.ApsuHeError), match="outside"):
            seal.run_query_request(G, sc, wire.build_query_request(compr, rk_blob, bad_parts), gb, [b["mask"] for b in S.bundles], compr=compr)
    wrong_rk = S.rk.copy()
    wrong_rk[0, 1, K - 1, 7] = np.uint64(0xFFFFFFFFFFFFFFFF)
    with pytest.raises((ValueError, apsu_amd.ApsuHeError), match="outside"):
        seal.run_query_request(G, sc, wire.build_query_request(compr, sc.relin_keys_save(wrong_rk, compr=compr), plain_parts), gb,
                               [b["mask"] for b in S.bundles], compr=compr)
    assert seal.run_query_request(G, sc, wire.build_query_request(compr, rk_blob, plain_parts), gb, [b["mask"] for b in S.bundles], compr=compr) == pkgs
    with pytest.raises(ValueError, match="query powers"):            # a part with a foreign exponent (query.cpp:63-68)
        seal.run_query_request(G, sc, wire.build_query_request(compr, rk_blob, [(e + 1, c) for e, c in parts]), gb, [b["mask"] for b in S.bundles])
    with pytest.raises(ValueError, match="relinearization"):
        seal.run_query_request(G, sc, wire.build_query_request(compr, None, parts), gb, [b["mask"] for b in S.bundles])
    # the response: one ResultPackage per BinBundle, the result ciphertext as a SEAL object at the last level
    for i, b in enumerate(S.bundles):
        body = sc.ct_save(0, False, out[i], compr=compr)
        pkg = wire.build_result_package(b["bundle_idx"], b["cache_idx"], body, 0, 0, [])
        back = wire.parse_result_package(pkg)
        assert back["bundle_idx"] == b["bundle_idx"] and back["cache_idx"] == b["cache_idx"]
        assert (sc.ct_load(back["psu_result"])["data"] == out[i]).all()
    G.close()
    sc.close()


@pytest.mark.parametrize("bits,force_host", [((40, 40, 40, 36), False), ((60, 60, 60, 50), False), ((56, 56, 56, 50), False), ((60, 60, 60, 50), True)])
def test_seed_expansion_on_the_device_equals_the_host_codec(monkeypatch, bits, force_host):
    """apsu_he_seed_expand (k_seed_bulk + k_seed_fix: SEAL's sample_poly_uniform with its in-stream rejection sampling) against
    the host codec's expansion (itself held against the Python model in tests/test_seal_codec.py): data levels and the key
    level, several ciphertexts per call, 60-bit primes for hundreds of rejections per limb.  Last case: the engine's own fallback
    for objects whose rejections overflow the device list -- expansion by the host codec, uploaded -- forced for every object
    (APSU_HE_SEED_EXPAND_HOST=1, read at apsu_he_create)"""
    import torch
    if force_host:
        monkeypatch.setenv("APSU_HE_SEED_EXPAND_HOST", "1")
    n = 1024 if bits[0] == 60 else (8192 if bits[0] == 56 else 64)
    js = common.toy_json(n=n, coeff_bits=bits, plain_bits=17 if n < 8192 else 22)
    G = apsu_amd.HeContext(js)
    sc = seal.SealContext(js)
    rng = np.random.default_rng(bits[0])
    K = G.K
    for chain_idx in (G.first_chain_idx, 0, -1):
        L = K if chain_idx < 0 else chain_idx + 1
        count = 5
        seeds = rng.integers(0, 2**63, (count, 8), dtype=np.uint64)
        out = torch.zeros((count, L, n), dtype=torch.int64, device="cuda")
        G.seed_expand(chain_idx, seeds, [out.data_ptr() + i * L * n * 8 for i in range(count)])
        got = out.cpu().numpy().view(np.uint64)
        for i in range(count):
            want = sc.sample_poly_uniform(chain_idx, [int(v) for v in seeds[i]], L, n)
            assert (got[i] == want).all(), (bits, chain_idx, i)
        for _ in range(2):                                      # the rejection lists are left clean for the next call
            G.seed_expand(chain_idx, seeds[:2], [out.data_ptr() + i * L * n * 8 for i in range(2)])
        assert (out.cpu().numpy().view(np.uint64)[:2] == got[:2]).all()
    G.close()
    sc.close()


def test_seeded_relin_keys_are_expanded_on_the_device():
    """apsu_he_run_query_request with Serializable<RelinKeys> as KeyGenerator::create_relin_keys saves them (sender_osn.cpp:223-227):
    the second polynomial of every key comes as a seed; the single-device path samples it on the GPU, in place in the uploaded
    keys.  Parity is about the arithmetic: the keys' c1 are replaced by the seeds' expansions (the keys then no longer decrypt),
    and every result must equal the oracle's under the very same keys; the multi-device handle (every device samples its own copy) must agree."""
    js = common.toy_json()
    S = common.make_scenario(js, {0: [11, 4], 1: [7]})
    C = S.C
    sc = seal.SealContext(js)
    rng = np.random.default_rng(5)
    seeds = rng.integers(0, 2**63, (C.K - 1, 8), dtype=np.uint64)
    ksk = S.rk.copy()
    for d in range(C.K - 1):
        ksk[d, 1] = sc.sample_poly_uniform(-1, [int(x) for x in seeds[d]], C.K, C.n)
    rk_blob = sc.relin_keys_save(ksk, seeds=seeds, compr=seal.COMPR_ZLIB)
    back, _ = sc.relin_keys_load(rk_blob)
    assert (back.reshape(ksk.shape) == ksk).all() and len(rk_blob) < ksk.nbytes * 0.7
    parts = [(e, [sc.ct_save(C.first, False, S.src[b][e] if b in S.src else S.src[0][e]) for b in range(S.p["bundle_idx_count"])]) for e in S.sources]
    msg = wire.build_query_request(seal.COMPR_ZLIB, rk_blob, parts)
    G = apsu_amd.HeContext(js)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    masks = [b["mask"] for b in S.bundles]
    pkgs = seal.run_query_request(G, sc, msg, gb, masks)
    S.rk = ksk                                                   # the oracle under the same keys
    opw = common.oracle_powers(S)
    for i, b in enumerate(S.bundles):
        got = sc.ct_load(wire.parse_result_package(pkgs[i])["psu_result"])["data"]
        assert (got == common.oracle_eval(S, opw, b)).all(), "bundle %d" % i
    units = [(b["bundle_idx"], b["cache_idx"], b["degree"]) for b in S.bundles]
    M = apsu_amd.MultiContext(js, [0, 0])
    slots = apsu_amd.partition_bundles(units, S.p["bundle_idx_count"], 2)
    for i, b in enumerate(S.bundles):
        M.upload_bundle(slots[i], b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"])
    assert seal.multi_run_query_request(M, sc, msg, masks) == pkgs
    M.close()
    G.close()
    sc.close()
