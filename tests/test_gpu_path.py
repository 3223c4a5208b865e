"""GPU tier (pytest -m gpu): the fused tier-2 path — Receiver::ComputePowers and
BatchedPlaintextPolyn::eval / eval_patstock for batches of BinBundles — against the CPU oracle on the
same seeded inputs (bit-exact), the golden path vectors, the reference's error behaviour, and
size-independent properties at BASELINE.json's full 16M-4096 size."""
import numpy as np
import pytest

import apsu_amd
import common
from golden_util import arr, load
from oracle import ref

pytestmark = pytest.mark.gpu


def run_scenario(js, degrees, roots_frac=0.5):
    S = common.make_scenario(js, degrees, roots_frac=roots_frac)
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    assert G.powers_dag() == S.nodes
    rk = G.upload_relin_keys(S.rk) if S.rk is not None else None
    srcs = [[S.src[b][e] for e in S.sources] for b in S.bundle_indices]
    pw = G.compute_powers(S.bundle_indices, srcs, rk)
    for b in S.bundle_indices:
        for p in S.targets:
            ct, ci, is_ntt = pw.download(b, p)
            exp = opw[b][p]
            assert ct.shape == exp.shape and (ct == exp).all(), "power %d of bundle index %d" % (p, b)
            assert is_ntt == (S.ps_low == 0 or p <= S.ps_low)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    out = G.eval_bundles(gb, pw, rk, [b["mask"] for b in S.bundles])
    for i, b in enumerate(S.bundles):
        exp = common.oracle_eval(S, opw, b)
        assert (out[i] == exp).all(), "bundle idx=%d degree=%d" % (b["bundle_idx"], b["degree"])
        ok, budget = common.check_semantics(S, b, out[i])
        assert ok and budget > 0
    return S, G, pw, rk, gb, out


def test_toy_paterson_stockmeyer_ragged_degrees():
    # degrees: full, == ps_low (plain eval at the low level), multiples of h (r == 0), h exactly, h+1, 1, 0
    run_scenario(common.toy_json(), {0: [11, 10, 3, 8, 4, 5, 1, 0], 1: [7, 2]})


def test_toy_without_paterson_stockmeyer():
    run_scenario(common.toy_json(ps_low=0, max_items=6, query_powers=(1, 2, 3, 5)), {0: [6, 2, 0], 1: [5]})


def test_toy_two_limb_key_level():
    # K = 2: first = last level, relinearisation at chain index 0
    # (all target powers are sources, so the only ct x ct product is eval_patstock's: one 60-bit limb has budget for it)
    run_scenario(common.toy_json(n=256, coeff_bits=(60, 40), plain_bits=14, ps_low=2, max_items=5, query_powers=(1, 2, 3),
                                 felts=7), {0: [5, 4, 2]})


def test_toy_wide_primes_many_low_powers_fallback_path():
    # 60-bit coefficient primes with 16 low powers: (l+1)*q_last >= 2^64, so the i = 0 block cannot use the
    # summed-last-limb shortcut and takes the per-term INTT + drop-limb path
    run_scenario(common.toy_json(coeff_bits=(60, 60, 60, 40), plain_bits=17, ps_low=16, max_items=40, query_powers=(1, 17)),
                 {0: [40, 35], 1: [17]})


def test_ring_size_16384():
    # n = 16384 is outside the shipped parameter files (max 8192) but inside SEAL's range: whole path, small degrees
    run_scenario(common.toy_json(n=16384, coeff_bits=(56, 56, 56, 50), plain_bits=22, ps_low=3, max_items=9, query_powers=(1, 4),
                                 felts=5), {0: [9, 4], 1: [7]}, roots_frac=0.01)


def test_ring_size_32768():
    # n = 32768, the largest poly_modulus_degree SEAL (and psu_params.cpp:95-180) accepts: the whole path on the split transform
    # (two 16384-point halves per limb around one radix-2 stage over global memory), incl. the key switch's gather and RAW outputs
    run_scenario(common.toy_json(n=32768, coeff_bits=(56, 56, 56, 50), plain_bits=20, ps_low=2, max_items=8, query_powers=(1, 3),
                                 felts=5), {0: [7, 2], 1: [5]}, roots_frac=0.01)


def test_config_100K_1():
    run_scenario(common.param_json("100K-1"), {0: [19, 7, 1]})


def test_config_1M_1024_com():
    run_scenario(common.param_json("1M-1024-com"), {0: [124, 30, 6], 1: [124, 5]})


def test_config_16M_4096_reduced():
    # one full BinBundle (D = 1303, H = 28, r = 43) and the short one of the synthetic DB
    run_scenario(common.param_json("16M-4096"), {0: [1303], 3: [170]})


def test_config_256M_4096_reduced():
    # first level has 4 limbs: low powers ARE mod-switched once (4 -> 3), high powers twice (4 -> 2)
    run_scenario(common.param_json("256M-4096"), {1: [700]}, roots_frac=0.02)


def test_golden_path_on_gpu():
    g = load("path_n64.json")
    js = common.toy_json(n=g["n"], coeff_bits=g["coeff_bits"], plain_bits=g["plain_bits"], ps_low=g["ps_low_degree"],
                         max_items=g["max_items_per_bin"], query_powers=g["query_powers"])
    G = apsu_amd.HeContext(js)
    assert [list(nd) for nd in G.powers_dag()] == g["dag_nodes"]
    rk = G.upload_relin_keys(arr(g["rk"]))
    srcs = [[arr(g["sources"][str(e)]) for e in sorted(g["query_powers"])]]
    pw = G.compute_powers([0], srcs, rk)
    for p, ct in g["powers"].items():
        got, _, _ = pw.download(0, int(p))
        assert (got == arr(ct)).all()
    gb = [G.upload_bundle(0, i, [arr(c) for c in b["coeffs"]], b["is_ntt"]) for i, b in enumerate(g["bundles"])]
    out = G.eval_bundles(gb, pw, rk, [arr(b["mask"]) for b in g["bundles"]])
    for i, b in enumerate(g["bundles"]):
        assert (out[i] == arr(b["result"])).all()
    G.close()


def test_special_plaintexts_zero_and_monomial():
    """all-zero coefficients are legal (bin_bundle.cpp:111-114); a coefficient-form a_{i*h} whose
    encoding is a monomial takes SEAL's no-lift shortcut"""
    js = common.toy_json()
    S = common.make_scenario(js, {0: [11]})
    b = S.bundles[0]
    C = S.C
    pci = C.plain_chain_idx(S.ps_low)
    b["A"][5] = 0                                            # NTT-form coefficient identically zero
    b["coeffs"][5] = C.plain_lift_ntt(C.encode(b["A"][5]), pci)
    b["A"][8] = C.t - 1                                      # a_{2h}: all bins equal -> constant polynomial -> monomial
    b["coeffs"][8] = C.encode(b["A"][8])
    assert np.count_nonzero(b["coeffs"][8]) == 1 and b["coeffs"][8][0] >= (C.t + 1) // 2
    b["A"][4] = 0                                            # a_h identically zero (coefficient form)
    b["coeffs"][4] = C.encode(b["A"][4])
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
    gb = G.upload_bundle(0, 0, b["coeffs"], b["flags"])
    out = G.eval_bundles([gb], pw, rk, [b["mask"]])
    assert (out[0] == common.oracle_eval(S, opw, b)).all()
    assert common.check_semantics(S, b, out[0])[0]
    G.close()


def test_error_behaviour_matches_reference():
    js = common.toy_json()
    S = common.make_scenario(js, {0: [10], 1: [3]})
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    b0, b1 = S.bundles
    with pytest.raises(ValueError):                          # layout rule of the ctor violated (bin_bundle.cpp:418-420)
        G.upload_bundle(0, 0, b0["coeffs"], [not f for f in b0["flags"]])
    with pytest.raises(ValueError):                          # degree beyond max_items_per_bin
        G.upload_bundle(0, 0, b0["coeffs"] * 2, b0["flags"] * 2)
    with pytest.raises(ValueError):
        G.upload_bundle(9, 0, b0["coeffs"], b0["flags"])     # bundle index out of range
    pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
    gb1 = G.upload_bundle(1, 0, b1["coeffs"], b1["flags"])
    with pytest.raises(ValueError):                          # no powers for bundle index 1
        G.eval_bundles([gb1], pw, rk, [b1["mask"]])
    gb0 = G.upload_bundle(0, 0, b0["coeffs"], b0["flags"])
    with pytest.raises(ValueError):                          # relin keys are required for eval_patstock
        G.eval_bundles([gb0], pw, None, [b0["mask"]])
    assert G.eval_bundles([], pw, rk, []).shape[0] == 0      # empty batch
    G.close()


def test_device_resident_io_and_profile_hooks():
    """inputs/outputs as device pointers (bench.py's mode) give the same bits as host buffers"""
    import torch
    js = common.param_json("1M-1024-com")
    S = common.make_scenario(js, {0: [124, 17]})
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    src = np.stack([S.src[0][e] for e in S.sources])
    src_d = torch.from_numpy(src.view(np.int64)).cuda()
    w = src[0].size
    ptrs = [[src_d.data_ptr() + i * w * 8 for i in range(len(S.sources))]]
    G.profile_enable(1)
    pw_h = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
    pw_d = G.compute_powers([0], ptrs, rk, on_device=True)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    masks = np.stack([b["mask"] for b in S.bundles])
    mask_d = torch.from_numpy(masks.view(np.int64)).cuda()
    out_d = torch.zeros((len(gb), 2, G.n), dtype=torch.int64, device="cuda")
    out_h = G.eval_bundles(gb, pw_h, rk, list(masks))
    G.eval_bundles(gb, pw_d, rk, [mask_d.data_ptr() + i * G.n * 8 for i in range(len(gb))], out=out_d.data_ptr(),
                   masks_on_device=True, out_on_device=True)
    assert (out_d.cpu().numpy().view(np.uint64).reshape(out_h.shape) == out_h).all()
    prof = G.profile_read()
    assert prof["ntt_fwd"][1] > 0 and prof["ntt_fwd"][2] > 0 and prof["ntt_fwd"][0] > 0
    assert prof["dyadic_mac"][1] > 0
    G.close()


def test_device_resident_sources_outside_their_prime_are_reported():
    """seal::is_data_valid_for for sources handed over as device pointers (tier 2, `src_on_device`): the engine's lazy transforms take source limbs
    as they are, so a word >= q_limb must not pass silently.  The kernel that gathers the sources checks every word; the violation is
    reported by the call that next waits for that work -- a synchronous eval_bundles, or apsu_he_sync in the queued mode -- and the context
    stays usable"""
    import torch
    js = common.param_json("1M-1024-com")
    S = common.make_scenario(js, {0: [124, 17]})
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    masks = [b["mask"] for b in S.bundles]
    src = np.stack([S.src[0][e] for e in S.sources])
    w = src[0].size
    good = torch.from_numpy(src.view(np.int64)).cuda()
    wrong = src.copy()
    wrong[3, 1, 1, 77] = np.uint64(int(S.C.q[1]))                    # source 3, polynomial 1, limb 1: == q, the smallest invalid value
    bad = torch.from_numpy(wrong.view(np.int64)).cuda()
    ptrs = lambda d: [[d.data_ptr() + i * w * 8 for i in range(len(S.sources))]]
    want = np.stack([common.oracle_eval(S, opw, b) for b in S.bundles])
    with pytest.raises(ValueError, match="outside"):                 # synchronous evaluation: reported right there
        G.eval_bundles(gb, G.compute_powers([0], ptrs(bad), rk, on_device=True), rk, masks)
    assert (G.eval_bundles(gb, G.compute_powers([0], ptrs(good), rk, on_device=True), rk, masks) == want).all()
    # queued mode: nothing waits inside the calls, apsu_he_sync reports it
    G.set_async_results(True)
    md = torch.from_numpy(np.stack(masks).view(np.int64)).cuda()
    od = torch.zeros((len(gb), 2, G.n), dtype=torch.int64, device="cuda")
    mp = [md.data_ptr() + i * G.n * 8 for i in range(len(gb))]
    pw = G.compute_powers([0], ptrs(bad), rk, on_device=True)
    G.eval_bundles(gb, pw, rk, mp, out=od.data_ptr(), masks_on_device=True, out_on_device=True)
    with pytest.raises(ValueError, match="outside"):
        G.sync()
    pw = G.compute_powers([0], ptrs(good), rk, on_device=True)
    G.eval_bundles(gb, pw, rk, mp, out=od.data_ptr(), masks_on_device=True, out_on_device=True)
    G.sync()
    assert (od.cpu().numpy().view(np.uint64).reshape(want.shape) == want).all()
    G.close()


def test_host_sources_outside_their_prime_are_reported_per_query():
    """round 6 (advisor): sources handed over as HOST pointers (what both integration adapters do) went through a plain copy unchecked;
    they are now held against their primes on the device behind that copy, and the report names its query: a synchronous evaluation of
    the bad query fails, one of a good query queued right behind it does not, and downloading a power of the bad query fails too"""
    js = common.param_json("1M-1024-com")
    S = common.make_scenario(js, {0: [124, 17]})
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    masks = [b["mask"] for b in S.bundles]
    good = [[S.src[0][e] for e in S.sources]]
    wrong = [[a.copy() for a in good[0]]]
    wrong[0][2][0, 1, 4000] = np.uint64(int(S.C.q[1]) + 5)
    want = np.stack([common.oracle_eval(S, opw, b) for b in S.bundles])
    pw_bad = G.compute_powers([0], wrong, rk)
    pw_good = G.compute_powers([0], good, rk)                        # queued behind it, another powers buffer
    assert (G.eval_bundles(gb, pw_good, rk, masks) == want).all()    # the good query's evaluation does not inherit the report
    with pytest.raises(ValueError, match="outside"):
        pw_bad.download(0, 1)
    pw_bad2 = G.compute_powers([0], wrong, rk)
    with pytest.raises(ValueError, match="outside"):
        G.eval_bundles(gb, pw_bad2, rk, masks)
    G.sync()                                                         # nothing left to report
    assert (G.eval_bundles(gb, G.compute_powers([0], good, rk), rk, masks) == want).all()
    G.close()


@pytest.mark.parametrize("overlap", [0, 1, 2, 3])
def test_async_device_results(overlap):
    """apsu_he_set_query_overlap (modes 1-3): the next query's high-power chain / whole ComputePowers may start before the query in front has finished --
    the inputs below are complete when compute_powers is called (torch's blocking uploads), as that mode requires.
    apsu_he_set_async_results: with device-resident sources, masks and results the calls return with their work queued;
    back-to-back queries without a host synchronisation give the bits of the synchronous path, results are complete
    after apsu_he_sync and in stream order on apsu_he_stream"""
    import torch
    js = common.param_json("1M-1024-com")
    S = common.make_scenario(js, {0: [124, 17, 60], 1: [99, 3]})
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    src = np.stack([np.stack([S.src[b][e] for e in S.sources]) for b in S.bundle_indices])
    w = src[0, 0].size
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    masks = np.stack([b["mask"] for b in S.bundles])
    mask_d = torch.from_numpy(masks.view(np.int64)).cuda()
    mp = [mask_d.data_ptr() + i * G.n * 8 for i in range(len(gb))]
    want = np.stack([common.oracle_eval(S, opw, b) for b in S.bundles])
    G.set_async_results(True)
    G.set_query_overlap(overlap)
    torch.cuda.synchronize()
    outs = [torch.zeros((len(gb), 2, G.n), dtype=torch.int64, device="cuda") for _ in range(3)]
    srcs = []
    for k in range(6):                                     # queries back to back; every second one on zeroed sources
        sk = src if k % 2 == 0 else np.zeros_like(src)
        sd = torch.from_numpy(sk.view(np.int64)).cuda()
        srcs.append(sd)                                     # device inputs must outlive the queued work
        ptrs = [[sd.data_ptr() + ((bi * len(S.sources) + i) * w) * 8 for i in range(len(S.sources))] for bi in range(len(S.bundle_indices))]
        pw = G.compute_powers(S.bundle_indices, ptrs, rk, on_device=True)
        G.eval_bundles(gb, pw, rk, mp, out=outs[k % 3].data_ptr(), masks_on_device=True, out_on_device=True)
        del pw                                              # recycled while its evaluation is still queued
    # mode 3 takes the pipelined walk for every query; mode 1 only for a query that finds the device busy, which these short queries
    # (0.6 ms) leave to timing -- tests/test_gpu_pipelined.py pins that mode down at 16M-4096 size
    piped = G.debug_counters()["pipelined"]
    assert piped == (6 if overlap == 3 else piped if overlap == 1 else 0)
    ext = torch.cuda.ExternalStream(G.stream)
    torch.cuda.current_stream().wait_stream(ext)            # consumer ordered after the context's stream, no host wait
    copy = outs[1].clone()                                  # k = 4: real sources
    G.sync()
    got = outs[1].cpu().numpy().view(np.uint64).reshape(want.shape)
    assert (got == want).all()
    torch.cuda.synchronize()
    assert (copy.cpu().numpy().view(np.uint64).reshape(want.shape) == want).all()
    zero_q = outs[2].cpu().numpy().view(np.uint64).reshape(want.shape)          # k = 5: all-zero query ciphertexts
    assert not (zero_q == want).all()
    # host-memory results still synchronise in this mode
    pw = G.compute_powers(S.bundle_indices, [[S.src[b][e] for e in S.sources] for b in S.bundle_indices], rk)
    assert (G.eval_bundles(gb, pw, rk, list(masks)) == want).all()
    G.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_async_results_soak_with_changing_shapes(monkeypatch, overlap):
    """queued-back-to-back queries whose batch shape changes every call (different BinBundle subsets and bundle-index sets:
    the job-table cache misses, the workspace arena grows, pooled powers buffers change size) under both stream policies
    give the bits of the synchronous path; nothing is waited for until the end"""
    import torch
    js = common.param_json("1M-1024-com")
    S = common.make_scenario(js, {0: [124, 17, 60, 124], 1: [99, 3, 124]})
    monkeypatch.setenv("APSU_HE_ARENA_BYTES", "1048576")              # a 1 MiB initial arena: overflow -> grow -> retry on the way
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    src = np.stack([np.stack([S.src[b][e] for e in S.sources]) for b in S.bundle_indices])
    sd = torch.from_numpy(src.view(np.int64)).cuda()
    w = src[0, 0].size
    ns = len(S.sources)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    masks = np.stack([b["mask"] for b in S.bundles])
    mask_d = torch.from_numpy(masks.view(np.int64)).cuda()
    rng = np.random.default_rng(5)
    subsets = []
    for _ in range(24):
        k = int(rng.integers(1, len(gb) + 1))
        subsets.append(sorted(int(v) for v in rng.choice(len(gb), size=k, replace=False)))

    def run(sub, out):
        idx = sorted({S.bundles[i]["bundle_idx"] for i in sub})
        ptrs = [[sd.data_ptr() + ((S.bundle_indices.index(b) * ns + i) * w) * 8 for i in range(ns)] for b in idx]
        pw = G.compute_powers(idx, ptrs, rk, on_device=True)
        G.eval_bundles([gb[i] for i in sub], pw, rk, [mask_d.data_ptr() + i * G.n * 8 for i in sub], out=out.data_ptr(),
                       masks_on_device=True, out_on_device=True)

    want = []
    for sub in subsets:                                                   # synchronous reference, one query at a time
        o = torch.zeros((len(sub), 2, G.n), dtype=torch.int64, device="cuda")
        run(sub, o)
        want.append(o.cpu().numpy())
    G.set_async_results(True)
    G.set_query_overlap(overlap)                                          # (sd / mask_d were uploaded by blocking copies)
    for split in (1, 0):
        G.set_two_stream(split)
        outs = [torch.zeros((len(sub), 2, G.n), dtype=torch.int64, device="cuda") for sub in subsets]
        for sub, o in zip(subsets, outs):
            run(sub, o)
        G.sync()
        for q, (o, wnt) in enumerate(zip(outs, want)):
            assert (o.cpu().numpy() == wnt).all(), (split, q)
    G.close()


def test_full_size_16M_properties():
    """BASELINE.json size (n = 8192, D = 1303, 241 MB per BinBundle) on the synthetic GPU-generated DB:
    (1) a second evaluation is bit-identical (determinism / no stale workspace);
    (2) decrypt(result) == P(x) + mask per slot with P rebuilt on the host from the documented generator;
    (3) changing only the mask shifts the decrypted slots by exactly the mask difference."""
    from bench import SEED0, splitmix_values
    js = common.param_json("16M-4096")
    S = common.make_scenario(js, {2: []})                    # keys + one encrypted query, no host-built bundles
    C = S.C
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers([2], [[S.src[2][e] for e in S.sources]], rk)
    seed = SEED0 + 4242
    D = G.max_items_per_bin - 1
    gb = G.random_bundle(2, 0, D, seed)
    assert gb.db_bytes > 210 * 2**20                       # bit-packed rows: 7 bytes per coefficient of a 56-bit prime (240.9 MiB as dense words)
    m1 = ref.fill_uniform(1, C.t, C.n)
    m2 = ref.fill_uniform(2, C.t, C.n)
    o1 = G.eval_bundles([gb], pw, rk, [C.encode(m1)])
    o1b = G.eval_bundles([gb], pw, rk, [C.encode(m1)])
    assert (o1 == o1b).all()
    o2 = G.eval_bundles([gb], pw, rk, [C.encode(m2)])
    A = np.stack([C.decode(splitmix_values(seed, d, C.n, C.t)) for d in range(D + 1)])
    x = S.x[2].astype(object)
    acc = np.zeros(C.n, dtype=object)
    for d in range(D, -1, -1):
        acc = (acc * x + A[d].astype(object)) % C.t
    s1 = C.decode(C.decrypt(S.sk, o1[0], 0)[0]).astype(object)
    s2 = C.decode(C.decrypt(S.sk, o2[0], 0)[0]).astype(object)
    assert (s1 == (acc + m1.astype(object)) % C.t).all()
    assert ((s2 - s1) % C.t == (m2.astype(object) - m1.astype(object)) % C.t).all()
    G.close()


def test_concurrent_callers_on_one_context():
    """the reference calls the Evaluator from a thread pool (receiver_osn.cpp:334-364): ABI calls on one context
    must be thread-safe.  Several host threads evaluate different BinBundles / run tier-1 ops concurrently."""
    import threading
    js = common.toy_json()
    S = common.make_scenario(js, {0: [11, 10, 7, 3, 5, 8]})
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    expect = [common.oracle_eval(S, opw, b) for b in S.bundles]
    errors = []

    def worker(i):
        try:
            for _ in range(5):
                out = G.eval_bundles([gb[i]], pw, rk, [S.bundles[i]["mask"]])
                if not (out[0] == expect[i]).all():
                    errors.append("bundle %d mismatch" % i)
                ct = S.src[0][1].copy()
                G.transform_to_ntt_inplace(ct, S.C.first)
                G.transform_from_ntt_inplace(ct, S.C.first)
                if not (ct == S.src[0][1]).all():
                    errors.append("ntt roundtrip %d" % i)
        except Exception as e:                                   # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(gb))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    G.close()


def test_fallback_paths_match():
    # The environment switches the engine still reads (engine.cpp, constructor; round 5 retired the switches of decided A/B
    # experiments together with their losing code paths): the per-term finish of eval_patstock's products instead of the summed-Bsk
    # one (APSU_HE_EVAL_PER_TERM, the form the engine falls back to by itself when terms * q >= 2^63); ComputePowers forced onto
    # one / two streams (APSU_HE_SPLIT=0/1); the three-product k_mac forced on / off (APSU_HE_MAC_KARA); the evaluation's side work
    # on the main stream (APSU_HE_EVAL_SIDE=0); the database rows as dense 64-bit words instead of bit-packed
    # (APSU_HE_PACKED_ROWS=0); a 1 MiB initial arena exercises overflow -> grow -> retry (APSU_HE_ARENA_BYTES), and a 1-byte workspace
    # budget evaluates one BinBundle per chunk (APSU_HE_EVAL_WS_BYTES); round 6: every transform launch in the throughput form (16
    # coefficients per lane, APSU_HE_NTT_LATENCY_LIMBS=0) / every one in the latency form (8 per lane; by default the launch size
    # decides); eval_patstock's last mod-down as its own launch instead of inside the epilogue kernel (APSU_HE_FUSE_TAIL=0).
    # (APSU_HE_SEED_EXPAND_HOST: tests/test_gpu_wire_query.py.)
    # All must give the same bits; scenarios run in child processes.
    import subprocess, sys, os
    head = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\nimport test_gpu_path as t\n"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    small = "t.test_toy_paterson_stockmeyer_ragged_degrees(); t.test_toy_without_paterson_stockmeyer(); t.test_config_1M_1024_com()\n"
    big = ("t.test_toy_wide_primes_many_low_powers_fallback_path(); t.test_config_256M_4096_reduced(); "
           "t.test_config_16M_4096_reduced()\n")
    for switches, code in (({"APSU_HE_EVAL_PER_TERM": "1"}, small + big),
                           ({"APSU_HE_SPLIT": "0", "APSU_HE_MAC_KARA": "1"}, small + big),
                           ({"APSU_HE_SPLIT": "1", "APSU_HE_EVAL_SIDE": "0", "APSU_HE_MAC_KARA": "0"}, small + big),
                           ({"APSU_HE_PACKED_ROWS": "0"}, small + big),
                           ({"APSU_HE_NTT_LATENCY_LIMBS": "0", "APSU_HE_FUSE_TAIL": "0"}, small + big),
                           ({"APSU_HE_NTT_LATENCY_LIMBS": "100000000"}, small + big),
                           ({"APSU_HE_ARENA_BYTES": "1048576", "APSU_HE_EVAL_WS_BYTES": "1"}, small)):
        env = dict(os.environ, **switches)
        r = subprocess.run([sys.executable, "-c", head + code], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, repr(switches) + "\n" + r.stdout + r.stderr


def test_powers_outlive_their_context():
    """apsu_he_powers_free after apsu_he_destroy (the Python binding's garbage collector does this routinely): the
    context orphans its live powers handles, so the late free neither touches the destroyed context nor leaks"""
    js = common.toy_json()
    S = common.make_scenario(js, {0: [5]})
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    pws = [G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk) for _ in range(3)]
    pws[0] = None                                            # one goes back to the pool while the context is alive
    G.close()
    with pytest.raises(ValueError):                          # its buffers went with the context
        pws[1].download(0, 1)
    del pws                                                  # frees after destroy
    G2 = apsu_amd.HeContext(js)                              # the library is still healthy
    rk2 = G2.upload_relin_keys(S.rk)
    pw = G2.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk2)
    opw = common.oracle_powers(S)
    assert (pw.download(0, 1)[0] == opw[0][1]).all()
    G2.close()


def test_summed_finish_guard_uses_the_widest_limb():
    """30-bit first prime next to 60-bit ones with 16 products per BinBundle: 16 * q_1 >= 2^64, so the summed-Bsk finish
    (integer sums of per-term residues of EVERY limb) must not be taken; compared with the oracle's per-term order"""
    js = common.toy_json(n=64, coeff_bits=(30, 60, 60, 40), plain_bits=17, ps_low=2, max_items=50, query_powers=(1, 3))
    S = common.make_scenario(js, {0: [50, 49]})
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
    for p in S.targets:
        assert (pw.download(0, p)[0] == opw[0][p]).all()
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    out = G.eval_bundles(gb, pw, rk, [b["mask"] for b in S.bundles])
    for i, b in enumerate(S.bundles):
        assert (out[i] == common.oracle_eval(S, opw, b)).all()
    G.close()


def test_calls_from_a_thread_on_another_device():
    """HIP's current device is per host thread: a worker thread of the reference's pool may have another device
    current than the context's (several contexts on different GPUs in one process).  Needs two GPUs."""
    import threading
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    js = common.toy_json()
    S = common.make_scenario(js, {0: [11]})
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js, device=1)
    res = {}

    def worker():
        torch.cuda.set_device(0)                             # this thread's current device is NOT the context's
        rk = G.upload_relin_keys(S.rk)
        pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
        b = S.bundles[0]
        gb = G.upload_bundle(0, 0, b["coeffs"], b["flags"])
        res["out"] = G.eval_bundles([gb], pw, rk, [b["mask"]])
        res["dev"] = torch.cuda.current_device()

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert res["dev"] == 0
    assert (res["out"][0] == common.oracle_eval(S, opw, S.bundles[0])).all()
    G.close()


def test_scheduling_options_do_not_change_bits():
    """apsu_he_set_two_stream only moves launches between streams, the host run-ahead bound (two queued evaluations) only
    delays the host: same kernels, same operands, same results (every setting against the oracle-checked default)"""
    import torch
    js = common.param_json("1M-1024-com")
    S = common.make_scenario(js, {0: [124, 77, 30, 124, 9], 1: [124, 5, 60]})
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    src = np.stack([np.stack([S.src[b][e] for e in S.sources]) for b in S.bundle_indices])
    src_d = torch.from_numpy(src.view(np.int64)).cuda()
    w = src[0, 0].size
    ptrs = [[src_d.data_ptr() + ((bi * len(S.sources) + i) * w) * 8 for i in range(len(S.sources))] for bi in range(len(S.bundle_indices))]
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    masks = [b["mask"] for b in S.bundles]
    want = None
    for split in (0, 1):
        G.set_two_stream(split)
        for _ in range(2):
            pw = G.compute_powers(S.bundle_indices, ptrs, rk, on_device=True)   # device-resident inputs: the two-stream walk
            out = G.eval_bundles(gb, pw, rk, masks)
        if want is None:
            want = out
            for i, b in enumerate(S.bundles):
                assert (out[i] == common.oracle_eval(S, opw, b)).all()
        assert (out == want).all(), split
    # queued evaluations (device-resident masks and results): six queries behind each other, the host at most two ahead
    G.set_async_results(True)
    md = torch.from_numpy(np.stack(masks).view(np.int64)).cuda()
    od = torch.zeros((len(gb), 2, G.n), dtype=torch.int64, device="cuda")
    keep = []
    for _ in range(6):
        pw = G.compute_powers(S.bundle_indices, ptrs, rk, on_device=True)
        keep.append(pw)                                                          # the caller holds earlier powers: two job-table versions alternate
        if len(keep) > 2:
            keep.pop(0)
        G.eval_bundles(gb, pw, rk, [md.data_ptr() + i * G.n * 8 for i in range(len(gb))], out=od.data_ptr(), masks_on_device=True, out_on_device=True)
    G.sync()
    assert (od.cpu().numpy().view(np.uint64).reshape(len(gb), 2, 1, G.n) == want).all()
    c = G.debug_counters()
    assert c["job_hit"] > 0 and c["arena_grow"] <= 4
    G.close()


def test_phase_timers_and_counters():
    """apsu_he_phase_*: device time under the reference's STOPWATCH names (receiver_osn.cpp:167,403,504) — one span per
    compute_powers / eval_bundles call, RunQuery covers both; they only observe (same bits)"""
    js = common.toy_json()
    S = common.make_scenario(js, {0: [11, 3]})
    opw = common.oracle_powers(S)
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(S.rk)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    G.phase_enable(True)
    for _ in range(3):
        pw = G.compute_powers([0], [[S.src[0][e] for e in S.sources]], rk)
        out = G.eval_bundles(gb, pw, rk, [b["mask"] for b in S.bundles])
    ph = G.phase_read()
    assert set(ph) == {"Receiver::RunQuery", "Receiver::ComputePowers", "Receiver::ProcessBinBundleCache"}
    for name, (cnt, avg, mn, mx) in ph.items():
        assert cnt == 3 and 0 < mn <= avg <= mx, (name, cnt, avg, mn, mx)
    assert ph["Receiver::RunQuery"][1] >= max(ph["Receiver::ComputePowers"][1], ph["Receiver::ProcessBinBundleCache"][1]) * 0.999
    assert all(v[0] == 0 for v in G.phase_read().values())          # reset
    G.phase_enable(False)
    for i, b in enumerate(S.bundles):
        assert (out[i] == common.oracle_eval(S, opw, b)).all()
    c = G.debug_counters()
    assert set(c) == set(G.COUNTERS) and c["job_upload"] > 0
    G.close()
