"""GPU tier: the "next" row N1 — BinBundle::regen_polyns / regen_plaintexts / BatchedPlaintextPolyn
ctor on the GPU (apsu_he_db_build_bundle) against the oracle's polyn_with_roots + BatchEncoder + lift/NTT,
coefficient by coefficient and end to end through eval."""
import numpy as np
import pytest

import apsu_amd
import common
from oracle import ref

pytestmark = pytest.mark.gpu


def oracle_build(C, ps_low, bins):
    """reference semantics: per-bin monic polynomial with the bin's items as roots (interpolate.cpp:63-80),
    coefficient d of every bin batched into one plaintext (bin_bundle.cpp:395-430)"""
    n = C.n
    deg = max([len(b) for b in bins] + [0])
    A = np.zeros((deg + 1, n), dtype=np.uint64)
    for s, b in enumerate(bins):
        p = C.polyn_with_roots(np.array(b, dtype=np.uint64)) if len(b) else np.array([1], dtype=np.uint64)
        A[:len(p), s] = p
    pci = C.plain_chain_idx(ps_low)
    coeffs, flags = [], []
    for d in range(deg + 1):
        enc = C.encode(A[d])
        ntt = ref.coeff_is_ntt(ps_low, d)
        coeffs.append(C.plain_lift_ntt(enc, pci) if ntt else enc)
        flags.append(ntt)
    return A, coeffs, flags


def check_build(js, bins, bundle_idx=0):
    p = ref.load_params(js)
    C = ref.RefContext.from_params(p)
    ps = p["ps_low_degree"]
    A, coeffs, flags = oracle_build(C, ps, bins)
    G = apsu_amd.HeContext(js)
    gb = G.build_bundle(bundle_idx, 0, bins)
    deg = len(coeffs) - 1
    assert gb.degree == deg
    use_ps = ps > 1 and ps < deg
    high = C.clamp(1)
    for d in range(deg + 1):
        if d > 0 and not flags[d] and not use_ps:
            continue
        got, kind = G.bundle_coeff(gb, d)
        if d == 0:
            assert kind == 0 and (got == coeffs[0]).all()
        elif flags[d]:
            assert kind == 1 and (got == coeffs[d]).all(), "NTT-form coefficient %d" % d
        else:
            # what multiply_plain (bin_bundle.cpp:334) derives from the coefficient-form plaintext at the high level
            nz = np.count_nonzero(coeffs[d])
            exp = C.plain_lift_ntt(coeffs[d], high)
            if nz == 1:                                   # monomial shortcut: no lift
                raw = np.tile(coeffs[d], (high + 1, 1))
                exp = raw.reshape(1, high + 1, C.n).copy()
                C.transform_to_ntt(exp, high)
                exp = exp[0]
            assert kind == 2 and (got == exp).all(), "coefficient-form coefficient %d" % d
    return C, G, gb, A, coeffs, flags


def rand_bins(rng, t, n_bins, max_count, full_frac=0.3):
    bins = []
    for s in range(n_bins):
        c = max_count if rng.random() < full_frac else int(rng.integers(0, max_count + 1))
        bins.append([int(v) for v in rng.choice(t - 1, size=c, replace=False) + 1])
    return bins


def test_build_toy_ragged_bins():
    rng = np.random.default_rng(5)
    js = common.toy_json()
    t = ref.RefContext.from_params(ref.load_params(js)).t
    bins = rand_bins(rng, t, 60, 10)                      # 60 of 64 slots are bins, 4 stay unused
    bins[3] = []                                          # empty bin -> polynomial 1
    bins[7] = [0]                                         # root 0
    check_build(js, bins)


def test_build_all_bins_same_size_monomial_leading_coefficient():
    # every slot holds exactly h = 4 items: a_4 is the all-ones vector -> constant plaintext (monomial)
    rng = np.random.default_rng(6)
    js = common.toy_json()
    t = ref.RefContext.from_params(ref.load_params(js)).t
    bins = [[int(v) for v in rng.choice(t - 1, size=4, replace=False) + 1] for _ in range(64)]
    bins2 = [b + [int(rng.integers(1, t))] * 0 for b in bins]
    check_build(js, bins2)
    # degree 8 = 2h: a_8 all ones again, a_4 generic
    bins3 = [[int(v) for v in rng.choice(t - 1, size=8, replace=False) + 1] for _ in range(64)]
    check_build(js, bins3)


def test_build_without_paterson_stockmeyer():
    rng = np.random.default_rng(7)
    js = common.toy_json(ps_low=0, max_items=6, query_powers=(1, 2, 3, 5))
    t = ref.RefContext.from_params(ref.load_params(js)).t
    check_build(js, rand_bins(rng, t, 64, 6))


def test_build_1M_params_and_evaluate():
    """GPU-built BinBundle evaluates to the same ciphertext as the oracle-built one, and members decrypt to the mask"""
    rng = np.random.default_rng(8)
    js = common.param_json("1M-1024-com")
    S = common.make_scenario(js, {0: []})
    C = S.C
    n_bins = S.p["items_per_bundle"] * S.p["felts_per_item"]
    bins = rand_bins(rng, C.t, n_bins, 60, full_frac=0.1)
    for s in range(0, n_bins, 2):                         # make the query value a member of every other bin
        if bins[s]:
            bins[s][0] = int(S.x[0][s])
    C2, G, gb, A, coeffs, flags = check_build(js, bins)
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
    mask_vals = ref.fill_uniform(9, C.t, C.n)
    mask = C.encode(mask_vals)
    out = G.eval_bundles([gb], pw, rk, [mask])
    opw = common.oracle_powers(S)
    bundle = dict(bundle_idx=0, cache_idx=0, degree=len(coeffs) - 1, A=A, coeffs=coeffs, flags=flags, mask_vals=mask_vals, mask=mask)
    assert (out[0] == common.oracle_eval(S, opw, bundle)).all()
    ok, budget = common.check_semantics(S, bundle, out[0])
    assert ok and budget > 0
    got = C.decode(C.decrypt(S.sk, out[0], 0)[0])
    members = [s for s in range(0, n_bins, 2) if bins[s]]
    assert (got[members] == mask_vals[members]).all()
    G.close()


def test_build_rejects_bad_input():
    js = common.toy_json()
    G = apsu_amd.HeContext(js)
    with pytest.raises(ValueError):
        G.build_bundle(0, 0, [[1, 2]] * 65)               # more bins than slots
    with pytest.raises(ValueError):
        G.build_bundle(0, 0, [[G.t]])                     # unreduced field element
    with pytest.raises(ValueError):
        G.build_bundle(0, 0, [list(range(1, 14))])        # bin larger than max_items_per_bin
    G.close()


def test_bundle_image_roundtrip(tmp_path):
    """N2: save -> file -> mmap -> load gives a BinBundle that evaluates bit-identically; bad images are rejected"""
    js = common.toy_json()
    S = common.make_scenario(js, {0: [11, 3]})
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
    masks = [b["mask"] for b in S.bundles]
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    ref_out = G.eval_bundles(gb, pw, rk, masks)
    loaded = []
    for i, b in enumerate(gb):
        img = G.save_bundle(b)
        f = tmp_path / ("bundle%d.img" % i)
        f.write_bytes(img.tobytes())
        mm = np.memmap(str(f), dtype=np.uint8, mode="r")
        loaded.append(G.load_bundle(mm))
        assert loaded[-1].degree == b.degree and loaded[-1].db_bytes == b.db_bytes
    assert (G.eval_bundles(loaded, pw, rk, masks) == ref_out).all()
    img = G.save_bundle(gb[0]).copy()
    bad = img.copy(); bad[300] ^= 1
    with pytest.raises(ValueError):
        G.load_bundle(bad)                                    # checksum
    with pytest.raises(ValueError):
        G.load_bundle(img[:-8])                               # truncated
    bad = img.copy(); bad[0] = ord("X")
    with pytest.raises(ValueError):
        G.load_bundle(bad)                                    # magic
    G2 = apsu_amd.HeContext(common.toy_json(ps_low=0, max_items=6, query_powers=(1, 2, 3, 5)))
    with pytest.raises(ValueError):
        G2.load_bundle(img)                                   # built for different parameters
    G.close(); G2.close()


def test_build_spans_several_register_slots_and_the_serial_fallback():
    # the wave-per-bin kernel keeps coefficient i in lane i % 64, slot i / 64: degrees that cross slot boundaries
    # (63, 64, 65, 200), a degree served by its largest instance (128 slots: up to 8191 coefficients, the 256M-1 parameter
    # set has bins of up to 8099 items), and one beyond that capacity, which takes the thread-per-bin fallback
    rng = np.random.default_rng(11)
    for max_items, counts in ((210, [63, 64, 65, 200, 0, 1, 129]), (6200, [6199, 70, 0]), (8100, [8099, 4000, 1]), (8300, [8299, 3])):
        js = common.toy_json(ps_low=0, max_items=max_items, query_powers=(1,))
        p = ref.load_params(js)
        C = ref.RefContext.from_params(p)
        bins = [[int(v) for v in rng.choice(C.t - 1, size=c, replace=False) + 1] for c in counts]
        A, coeffs, flags = oracle_build(C, 0, bins)
        G = apsu_amd.HeContext(js)
        gb = G.build_bundle(0, 0, bins)
        assert gb.degree == max(counts)
        for d in sorted({0, 1, 2, 62, 63, 64, 65, 128, 199, 200, 4000, 6143, 6144, 8098, max(counts) - 1, max(counts)} & set(range(max(counts) + 1))):
            got, kind = G.bundle_coeff(gb, d)
            assert (got == coeffs[d]).all(), "max_items %d coefficient %d" % (max_items, d)


@pytest.mark.parametrize("cfg", ["toy", "16M-4096", "1M-1024-com", "single-prime"])
def test_algebraize_items_matches_oracle(cfg):
    """N1, one step before the bins: util::algebraize_item (db_encoding.cpp:209-256,360-366) on the GPU"""
    import json
    js = {"toy": common.toy_json(), "single-prime": common.toy_json(n=64, coeff_bits=(60,), plain_bits=9, felts=10, ps_low=0, max_items=4)}.get(cfg) \
        or common.param_json(cfg)
    felts = json.loads(js)["item_params"]["felts_per_item"]
    G = apsu_amd.HeContext(js)
    C = ref.RefContext.from_params(ref.load_params(js))
    rng = np.random.default_rng(11)
    items = rng.integers(0, 256, (1000, 16), dtype=np.uint8)
    items[0], items[1] = 0, 255
    got = G.algebraize_items(items)
    assert got.shape == (1000, felts) and (got == C.algebraize_items(items, felts)).all()
    assert G.algebraize_items(items[:0]).shape == (0, felts)
    G.close()


def test_database_file_round_trip_and_rejections(tmp_path):
    """N2 for the whole DB (ReceiverDB::save / Load counterpart): one mmap-able file, BinBundles loaded one by one, by shard, or
    onto the devices of a multi-device handle; same evaluation results; damaged / foreign files are refused"""
    js = common.toy_json()
    S = common.make_scenario(js, {0: [11, 5, 3], 1: [8, 0]})
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers(S.bundle_indices, [[S.src[b][e] for e in S.sources] for b in S.bundle_indices], rk)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    masks = [b["mask"] for b in S.bundles]
    want = G.eval_bundles(gb, pw, rk, masks)
    path = str(tmp_path / "db.apsuhe")
    G.save_db_file(path, gb)
    with apsu_amd.DbFile(path) as f:
        assert len(f) == len(gb) and f.file_bytes % 4096 == 0
        assert [f.entry(i)[:3] for i in range(len(f))] == [(b["bundle_idx"], b["cache_idx"], b["degree"]) for b in S.bundles]
        assert all(f.entry(i)[3] == len(G.save_bundle(gb[i])) for i in range(len(f)))
    # a fresh context (another process would do the same): everything, then only a shard
    G2 = apsu_amd.HeContext(js)
    rk2 = G2.upload_relin_keys(S.rk)
    pw2 = G2.compute_powers(S.bundle_indices, [[S.src[b][e] for e in S.sources] for b in S.bundle_indices], rk2)
    back = G2.load_db_file(path)
    assert [(b.bundle_idx, b.cache_idx, b.degree) for b in back] == [(b.bundle_idx, b.cache_idx, b.degree) for b in gb]
    assert (G2.eval_bundles(back, pw2, rk2, masks) == want).all()
    shard = G2.load_db_file(path, only=[1, 3])
    assert (G2.eval_bundles(shard, pw2, rk2, [masks[1], masks[3]]) == want[[1, 3]]).all()
    # the multi-device handle places the file's BinBundles itself ({0, 0} rehearses two devices); ids = table order
    flat = [S.src[b][e] for b in range(S.p["bundle_idx_count"]) for e in S.sources]
    for devs in ([0], [0, 0]):
        M = apsu_amd.MultiContext(js, devs)
        M.upload_relin_keys(S.rk)
        assert M.load_db_file(path) == len(gb)
        assert (M.eval_all(flat, masks, G.n) == want).all(), devs
        again = str(tmp_path / ("again%d.apsuhe" % len(devs)))
        M.save_db_file(again)
        assert open(again, "rb").read() == open(path, "rb").read()
        M.close()
    # refusals
    raw = bytearray(open(path, "rb").read())
    def refused(mutated, name):
        p = str(tmp_path / name)
        open(p, "wb").write(mutated)
        with pytest.raises(ValueError):
            G2.load_db_file(p)
    refused(raw[:len(raw) - 4096], "truncated")
    refused(b"NOTADBFL" + bytes(raw[8:]), "magic")
    bad = bytearray(raw); bad[256 + 8] ^= 1                       # table entry: degree
    refused(bad, "table")
    bad = bytearray(raw); bad[len(raw) - 5000] ^= 0x40            # payload of the last image (or its padding: then the header's size check)
    with apsu_amd.DbFile(path) as f:
        off = 4096 * ((256 + 32 * len(f) + 4095) // 4096)
    bad = bytearray(raw); bad[off + 300] ^= 0x40                  # first image's payload: the image checksum
    refused(bad, "payload")
    other = apsu_amd.HeContext(common.toy_json(max_items=12))
    with pytest.raises(ValueError, match="different parameters"):
        other.load_db_file(path)
    with pytest.raises(apsu_amd.ApsuHeError):
        G2.load_db_file(str(tmp_path / "missing"))
    for c in (G, G2, other):
        c.close()


@pytest.mark.parametrize("compr", [0, 2])
def test_bundle_upload_from_serialized_plaintexts(compr):
    """apsu_he_db_upload_bundle_serialized: the BinBundle cache as the reference holds it -- one SEAL-serialised Plaintext per
    coefficient (bin_bundle.cpp:421-428, compr_mode none, or zstd for the "-com" parameter sets) -- gives the same device
    BinBundle as the raw-pointer upload; form and level come from each object's parms_id"""
    from apsu_amd import seal
    for js, degrees in ((common.toy_json(), [11, 3]), (common.toy_json(ps_low=0, max_items=6, query_powers=(1, 2, 3, 5)), [6])):
        S = common.make_scenario(js, {0: degrees})
        G = apsu_amd.HeContext(js)
        sc = seal.SealContext(js)
        rk = G.upload_relin_keys(S.rk)
        pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
        pci = S.C.plain_chain_idx(S.ps_low)
        for b in S.bundles:
            blobs = [sc.pt_save(pci if f else -1, c, compr=compr) for c, f in zip(b["coeffs"], b["flags"])]
            ser = seal.upload_bundle_serialized(G, sc, b["bundle_idx"], b["cache_idx"], blobs)
            raw = G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"])
            assert G.save_bundle(ser).tobytes() == G.save_bundle(raw).tobytes()
            out = G.eval_bundles([ser], pw, rk, [b["mask"]])
            assert (out[0] == common.oracle_eval(S, common.oracle_powers(S), b)).all()
            # an NTT-form plaintext at another level than the BinBundle rule prescribes is refused
            if any(b["flags"]) and pci > 0:
                d = b["flags"].index(True)
                wrong = list(blobs)
                wrong[d] = sc.pt_save(pci - 1, b["coeffs"][d][:pci], compr=compr)
                with pytest.raises(ValueError):
                    seal.upload_bundle_serialized(G, sc, b["bundle_idx"], b["cache_idx"], wrong)
        G.close()
        sc.close()


def test_upload_of_a_bin_bundle_as_the_reference_persists_it():
    """apsu_he_db_upload_saved_bundle: the FlatBuffers buffer BinBundle::save appends to a saved ReceiverDB (bin_bundle.fbs) -- with its
    cache (SEAL Plaintext objects -> the serialized upload) and without (item bins -> rebuilt on the GPU) -- against the direct
    uploads; BinBundle::load's checks"""
    from apsu_amd import seal, wire
    from test_wire_framing import build_bin_bundle
    js = common.toy_json()
    p = ref.load_params(js)
    C = ref.RefContext.from_params(p)
    n, t = C.n, C.t
    G = apsu_amd.HeContext(js)
    sc = seal.SealContext(js)
    rng = np.random.default_rng(17)
    bins_per_bundle = (n // p["felts_per_item"]) * p["felts_per_item"]
    bins = [sorted(set(int(v) for v in rng.integers(1, t, int(rng.integers(0, p["max_items_per_bin"] + 1))))) for _ in range(bins_per_bundle)]
    bins[0] = sorted(set(int(v) for v in rng.integers(1, t, p["max_items_per_bin"])))          # one bin as full as the draw allows
    A, coeffs, flags = oracle_build(C, p["ps_low_degree"], bins)
    pci = C.plain_chain_idx(p["ps_low_degree"])
    direct = G.build_bundle(1, 0, bins)
    # (a) without the cache: rebuilt from the item bins
    buf = build_bin_bundle(1, t, bins)
    got, used = seal.upload_saved_bundle(G, None, buf + b"next")
    assert used == len(buf) and (got.bundle_idx, got.degree) == (1, direct.degree)
    assert G.save_bundle(got).tobytes() == G.save_bundle(direct).tobytes()
    # (b) with the cache: the serialized plaintexts, zstd like the "-com" sets when the library is there
    for compr in (0, 2):
        blobs = [sc.pt_save(pci if f else -1, c, compr=compr) for c, f in zip(coeffs, flags)]
        buf = build_bin_bundle(1, t, bins, blobs)
        info = wire.bin_bundle_info(buf)
        assert info["cache_coeffs"] == len(coeffs) and info["n_bins"] == bins_per_bundle and info["largest_bin"] == max(len(b) for b in bins)
        got, used = seal.upload_saved_bundle(G, sc, buf)
        assert used == len(buf) and G.save_bundle(got).tobytes() == G.save_bundle(G.upload_bundle(1, 0, coeffs, flags)).tobytes()
        with pytest.raises(ValueError, match="SEAL context"):
            seal.upload_saved_bundle(G, None, buf)
    # (c) stripped with cache: loads; stripped without: nothing to evaluate
    got, _ = seal.upload_saved_bundle(G, sc, build_bin_bundle(1, t, [], blobs, stripped=True))
    assert got.degree == direct.degree
    for bad, why in ((build_bin_bundle(1, t, [], None, stripped=True), "stripped"),
                     (build_bin_bundle(1, t + 2, bins), "field modulus"),
                     (build_bin_bundle(1, t, bins[:-1]), "number of item bins"),
                     (build_bin_bundle(1, t, [list(range(1, p["max_items_per_bin"] + 2))] + bins[1:]), "max_items_per_bin"),
                     (build_bin_bundle(p["bundle_idx_count"], t, bins), "bundle index")):
        with pytest.raises(apsu_amd.ApsuHeError, match=why):
            seal.upload_saved_bundle(G, sc, bad)
    G.close()
    sc.close()


def test_a_database_saved_by_the_reference_loads_and_answers_a_query():
    """ReceiverDB::save's file (receiver_db.fbs header + bin_bundle.fbs buffers; here written by the tests' FlatBuffers model, one
    BinBundle with its cache, one without) -> apsu_amd.seal.load_reference_db -> the query results of the directly uploaded DB"""
    import struct
    from apsu_amd import seal, wire
    from test_wire_framing import FbBuilder, build_bin_bundle
    js = common.toy_json()
    S = common.make_scenario(js, {0: [11], 1: [7]})
    C = S.C
    sc0 = seal.SealContext(js)
    pci = C.plain_chain_idx(S.ps_low)
    bins_per_bundle = (C.n // S.p["felts_per_item"]) * S.p["felts_per_item"]
    rng = np.random.default_rng(3)
    # bundle index 0: the scenario's BinBundle, saved WITH its cache; bundle index 1: item bins only (rebuilt on the GPU)
    b0 = S.bundles[0]
    blobs = [sc0.pt_save(pci if f else -1, c) for c, f in zip(b0["coeffs"], b0["flags"])]
    bins1 = [sorted(set(int(v) for v in rng.integers(1, C.t, int(rng.integers(0, 8))))) for _ in range(bins_per_bundle)]
    B = FbBuilder()
    hashed = B.struct_vector([struct.pack("<QQ", i, i) for i in range(4)], 8)
    key = B.byte_vector(bytes(32))
    pv = B.byte_vector(wire.psu_params_save(js))
    hdr = B.finish_size_prefixed(B.table([("off", pv), ("struct", struct.pack("<IIQ??", 0, 16, 4, False, False) + bytes(6), 8), ("off", key),
                                          ("off", hashed), ("u32", 2)]))
    blob = hdr + build_bin_bundle(0, C.t, [[] for _ in range(bins_per_bundle)], blobs) + build_bin_bundle(1, C.t, bins1)
    G, sc, loaded = seal.load_reference_db(blob)
    assert [(b.bundle_idx, b.cache_idx) for b in loaded] == [(0, 0), (1, 0)] and loaded[0].degree == 11
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers([0, 1], [[S.src[b][e] for e in S.sources] for b in (0, 1)], rk)
    direct = [G.upload_bundle(0, 0, b0["coeffs"], b0["flags"]), G.build_bundle(1, 0, bins1)]
    masks = [b0["mask"], S.bundles[1]["mask"]]
    got = G.eval_bundles(loaded, pw, rk, masks)
    assert (got == G.eval_bundles(direct, pw, rk, masks)).all()
    assert (got[0] == common.oracle_eval(S, common.oracle_powers(S), b0)).all()
    with pytest.raises(ValueError, match="trailing"):
        seal.load_reference_db(blob + b"\0\0\0\0")
    for c in (G, sc, sc0):
        c.close()
