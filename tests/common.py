"""Shared scenario builders for the parity tests.

A scenario = everything one query evaluation needs, manufactured with the CPU oracle's harness
pieces (keygen / encrypt / BatchEncoder; oracle/ref_harness.c): per bundle index a slot vector x,
the encrypted source powers, relinearisation keys, BinBundles (batched polynomials in the layout of
BatchedPlaintextPolyn's ctor, bin_bundle.cpp:366-430) and masks.  Synthetic inputs follow
SURVEY.md §8d (seed 0x41505355 + config index).
"""
import json
import os

import numpy as np

from oracle import ref

SEED0 = 0x41505355
PARAM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "params")


def param_json(name):
    """PSUParams JSON text: tests/params/<name>.json (parameter sets are data, copied values)."""
    with open(os.path.join(PARAM_DIR, name + ".json")) as f:
        return f.read()


def toy_json(n=64, coeff_bits=(40, 40, 40, 36), plain_bits=17, ps_low=3, max_items=11, query_powers=(1, 4),
             felts=5, table_mult=2):
    ipb = n // felts
    return json.dumps({
        "table_params": {"hash_func_count": 3, "table_size": ipb * table_mult, "max_items_per_bin": max_items},
        "item_params": {"felts_per_item": felts},
        "query_params": {"ps_low_degree": ps_low, "query_powers": list(query_powers)},
        "seal_params": {"plain_modulus_bits": plain_bits, "poly_modulus_degree": n,
                        "coeff_modulus_bits": list(coeff_bits)},
    })


class Scenario:
    pass


def make_bundle_coeffs(C, ps_low, degree, seed, roots_frac=0.5, x=None):
    """-> (A [degree+1][n] slot values, coeffs list in BinBundle layout, is_ntt flags).

    Each bin (slot) holds a monic degree-`degree` polynomial.  For `roots_frac` of the slots the
    polynomial is built from roots that include x[slot] (a member: evaluates to 0), like
    polyn_with_roots (common/apsu/util/interpolate.cpp:63-80); the rest have random coefficients.
    """
    n, t = C.n, C.t
    rng = np.random.default_rng(seed)
    A = rng.integers(0, t, (degree + 1, n), dtype=np.uint64)
    A[degree] = 1
    if x is not None and degree >= 1:
        nroots = int(n * roots_frac)
        for s in range(nroots):
            roots = rng.integers(0, t, degree, dtype=np.uint64)
            roots[int(rng.integers(0, degree))] = x[s]
            A[:, s] = C.polyn_with_roots(roots)
    pci = C.plain_chain_idx(ps_low)
    coeffs, flags = [], []
    for d in range(degree + 1):
        enc = C.encode(A[d])
        ntt = ref.coeff_is_ntt(ps_low, d)
        coeffs.append(C.plain_lift_ntt(enc, pci) if ntt else enc)
        flags.append(ntt)
    return A, coeffs, flags


def make_scenario(params_json, bundle_degrees, seed=SEED0, roots_frac=0.5):
    """bundle_degrees: {bundle_idx: [degree, ...]} -> Scenario with oracle-side data."""
    p = ref.load_params(params_json)
    C = ref.RefContext.from_params(p)
    S = Scenario()
    S.json, S.p, S.C = params_json, p, C
    S.ps_low = p["ps_low_degree"]
    S.targets = ref.create_powers_set(S.ps_low, p["max_items_per_bin"])
    S.depth, S.nodes = ref.powers_dag(p["query_powers"], S.targets)
    S.sources = sorted(p["query_powers"])
    S.sk = C.keygen(seed)
    S.rk = C.gen_relin_keys(S.sk, seed + 1) if C.K > 1 else None
    S.bundle_indices = sorted(bundle_degrees)
    S.x, S.src = {}, {}
    for b in S.bundle_indices:
        x = ref.fill_uniform(seed + 100 + b, C.t, C.n)
        S.x[b] = x
        S.src[b] = {}
        for e in S.sources:
            xe = np.array([pow(int(v), e, C.t) for v in x], dtype=np.uint64)
            S.src[b][e] = C.encrypt(S.sk, C.encode(xe), seed + 1000 * (b + 1) + e)
    S.bundles = []   # dicts: bundle_idx, cache_idx, degree, A, coeffs, flags, mask_vals, mask
    for b in S.bundle_indices:
        for ci, deg in enumerate(bundle_degrees[b]):
            A, coeffs, flags = make_bundle_coeffs(C, S.ps_low, deg, seed + 7919 * (b + 1) + ci, roots_frac, S.x[b])
            mv = ref.fill_uniform(seed + 31 * (b + 1) + ci + 5, C.t, C.n)
            S.bundles.append(dict(bundle_idx=b, cache_idx=ci, degree=deg, A=A, coeffs=coeffs, flags=flags,
                                  mask_vals=mv, mask=C.encode(mv)))
    return S


def oracle_powers(S):
    """Receiver::ComputePowers on the oracle for every bundle index -> {b: {power: ct}}"""
    return {b: S.C.compute_powers(S.src[b], S.nodes, S.rk, S.ps_low) for b in S.bundle_indices}


def oracle_eval(S, powers, bundle):
    C = S.C
    plist = [None] * (S.p["max_items_per_bin"] + 1)
    for pw, ct in powers[bundle["bundle_idx"]].items():
        plist[pw] = ct
    deg = bundle["degree"]
    if S.ps_low > 1 and S.ps_low < deg:
        return C.eval_patstock(plist, bundle["coeffs"], S.ps_low, S.rk, bundle["mask"])
    return C.eval(plist, bundle["coeffs"], plist[1].shape[1] - 1, bundle["mask"])


def expected_slots(S, bundle):
    """plaintext meaning of the result: P_bin(x_slot) + mask_slot mod t."""
    t = S.C.t
    x = S.x[bundle["bundle_idx"]].astype(object)
    acc = np.zeros(S.C.n, dtype=object)
    for d in range(bundle["degree"], -1, -1):
        acc = (acc * x + bundle["A"][d].astype(object)) % t
    return (acc + bundle["mask_vals"].astype(object)) % t


def check_semantics(S, bundle, out_ct):
    pt, budget = S.C.decrypt(S.sk, np.ascontiguousarray(out_ct), 0)
    got = S.C.decode(pt).astype(object)
    return bool((got == expected_slots(S, bundle)).all()), budget
