"""GPU tier: every parameter file the reference ships (/root/reference/parameters/*.json, copied as data under
tests/params/) through the whole hot path, bit for bit against the CPU oracle:

  * apsu_he_create accepts the file (PSUParams::Load rules, psu_params.cpp:95-180,290-374) and configures the same
    PowersDag (powers.cpp:22-107);
  * Receiver::ComputePowers for one bundle index: every target power (receiver_osn.cpp:395-488);
  * one SHORT BinBundle built like the reference's DB (polyn_with_roots, BatchEncoder, NTT) — compared with the oracle
    and, when the parameters leave noise budget, decrypted and checked per slot;
  * one FULL-DEGREE BinBundle (degree max_items_per_bin - 1: 8099 for 256M-1, 3999 for 256M-*, 1303 for 16M-*) with
    uniformly random stored limbs — compared with the oracle.  This is the shape the engine's size-dependent branches
    see: the serial polyn fallback is not involved here, but the per-term i = 0 path (l * q_last >= 2^64), the 4-limb
    first level, two-prime sets with 24-bit primes next to 56-bit ones and unsorted prime widths all are.
"""
import glob
import os

import numpy as np
import pytest

import apsu_amd
import common
from oracle import ref

pytestmark = pytest.mark.gpu

NAMES = sorted(os.path.basename(f)[:-5] for f in glob.glob(os.path.join(common.PARAM_DIR, "*.json")))


def random_full_bundle(C, ps_low, degree, seed):
    """stored form of a BinBundle with uniformly random contents: NTT-form coefficients are (plain_level+1) limbs of
    residues, coefficient-form ones n values mod t (bin_bundle.cpp:385-420 decides which is which)"""
    rng = np.random.default_rng(seed)
    pci = C.plain_chain_idx(ps_low)
    qs = C.q[:pci + 1]
    coeffs, flags = [], []
    for d in range(degree + 1):
        ntt = ref.coeff_is_ntt(ps_low, d)
        if ntt:
            coeffs.append(np.stack([rng.integers(0, q, C.n, dtype=np.uint64) for q in qs]))
        else:
            coeffs.append(rng.integers(0, C.t, C.n, dtype=np.uint64))
        flags.append(ntt)
    return coeffs, flags


def short_degree(ps_low, D):
    if ps_low > 1 and ps_low + 4 <= D:
        return ps_low + 4                     # Paterson-Stockmeyer with H = 1, r = 3
    return min(D, 5)


@pytest.mark.parametrize("name", NAMES)
def test_reference_parameter_file(name):
    js = common.param_json(name)
    p = ref.load_params(js)
    D = p["max_items_per_bin"] - 1            # a bin never reaches max_items_per_bin (receiver_db.cpp:388-389)
    b = p["bundle_idx_count"] - 1             # the last bundle index
    ref.set_threads(len(os.sched_getaffinity(0)))
    try:
        S = common.make_scenario(js, {b: [short_degree(p["ps_low_degree"], D)]}, roots_frac=0.02)
        opw = common.oracle_powers(S)
    finally:
        ref.set_threads(1)
    C = S.C
    G = apsu_amd.HeContext(js)
    assert G.powers_dag() == S.nodes
    assert G.q == C.q and G.t == C.t and G.first_chain_idx == C.first
    rk = G.upload_relin_keys(S.rk) if S.rk is not None else None
    pw = G.compute_powers([b], [[S.src[b][e] for e in S.sources]], rk)
    for pwr in S.targets:
        ct, _, is_ntt = pw.download(b, pwr)
        assert (ct == opw[b][pwr]).all(), "%s: power %d" % (name, pwr)
        assert is_ntt == (S.ps_low == 0 or pwr <= S.ps_low)
    # short BinBundle with real contents
    sb = S.bundles[0]
    gsb = G.upload_bundle(b, 0, sb["coeffs"], sb["flags"])
    out = G.eval_bundles([gsb], pw, rk, [sb["mask"]])
    assert (out[0] == common.oracle_eval(S, opw, sb)).all(), "%s: short BinBundle" % name
    ok, budget = common.check_semantics(S, sb, out[0])
    assert ok or budget == 0, "%s: decrypts wrongly with %d bits of budget" % (name, budget)
    # full-degree BinBundle, random contents
    coeffs, flags = random_full_bundle(C, S.ps_low, D, common.SEED0 + 17)
    mask = ref.fill_uniform(common.SEED0 + 5, C.t, C.n)
    fb = dict(bundle_idx=b, cache_idx=1, degree=D, coeffs=coeffs, flags=flags, mask=mask)
    gfb = G.upload_bundle(b, 1, coeffs, flags)
    out = G.eval_bundles([gfb], pw, rk, [mask])
    exp = common.oracle_eval(S, opw, fb)
    assert (out[0] == exp).all(), "%s: full-degree BinBundle (D = %d)" % (name, D)
    del gfb, gsb, pw
    G.close()
