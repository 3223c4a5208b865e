#!/usr/bin/env python3
"""Generates tests/golden/*.json from the independent Python big-int model (oracle/pymodel.py).

ORACLE — TEST INFRASTRUCTURE ONLY.  The reference holds no golden vectors for this path and
Microsoft SEAL is not available (PARITY UNPINNED); these vectors pin the C oracle
and the HIP kernels against a second, independent statement of SURVEY.md App. B.
Inputs are drawn from Python's `random` with a fixed seed; the secret key / encryption randomness is
generated here too (no code shared with oracle/ref_harness.c).

    python oracle/make_golden.py            # rewrites tests/golden/ops_n64.json, path_n64.json
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pymodel as pm  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def enc(x):
    """nested lists of ints -> nested lists of hex strings (u64 do not survive JSON floats)."""
    if isinstance(x, int):
        return "%x" % x
    return [enc(v) for v in x]


def rand_ct(M, rnd, polys, lvl):
    return [[[rnd.randrange(q) for _ in range(M.n)] for q in M.base(lvl)] for _ in range(polys)]


def encrypt(M, rnd, s, pt):
    """symmetric BFV encryption at the first data level (model-side; semantics only)."""
    lvl = M.first
    q = M.base(lvl)
    e = [sum(rnd.getrandbits(1) for _ in range(8)) - 4 for _ in range(M.n)]
    c0, c1 = [], []
    for qj in q:
        a = [rnd.randrange(qj) for _ in range(M.n)]
        as_ = M.negacyclic_mul(a, [v % qj for v in s], qj)
        c0.append([(-(x + y)) % qj for x, y in zip(as_, e)])
        c1.append(a)
    return M.add_plain([c0, c1], pt, lvl)


def relin_keys(M, rnd, s):
    K = M.K
    p = M.key_q[K - 1]
    rk = []
    for i in range(K - 1):
        e = [sum(rnd.getrandbits(1) for _ in range(8)) - 4 for _ in range(M.n)]
        c0, c1 = [], []
        for j, qj in enumerate(M.key_q):
            a = [rnd.randrange(qj) for _ in range(M.n)]           # a in the NTT domain
            a_coeff = M.intt(a, qj)
            sj = [v % qj for v in s]
            as_ = M.negacyclic_mul(a_coeff, sj, qj)
            b = [(-(x + y)) % qj for x, y in zip(as_, e)]
            if j == i:
                s2 = M.negacyclic_mul(sj, sj, qj)
                b = [(x + (p % qj) * y) % qj for x, y in zip(b, s2)]
            c0.append(M.ntt(b, qj))
            c1.append(a)
        rk.append([c0, c1])
    return rk


def gen_ops():
    rnd = random.Random(0x41505355)
    n, bits, pb = 64, [40, 40, 40, 36], 17
    M = pm.Model.from_bits(n, bits, 0, pb)
    cases = {"n": n, "coeff_modulus": enc(M.key_q), "plain_modulus": enc(M.t), "coeff_bits": bits, "plain_bits": pb,
             "psi": enc([M.psi[q] for q in M.key_q]), "levels": []}
    rk = [[[[rnd.randrange(q) for _ in range(n)] for q in M.key_q] for _ in range(2)] for _ in range(M.K - 1)]
    cases["rk"] = enc(rk)
    for lvl in range(M.first, -1, -1):
        ct, ct2, ct3 = rand_ct(M, rnd, 2, lvl), rand_ct(M, rnd, 2, lvl), rand_ct(M, rnd, 3, lvl)
        pt = [rnd.randrange(M.t) for _ in range(n)]
        mono = [0] * n
        mono[5] = M.t - 3
        c = {"chain_idx": lvl, "ct": enc(ct), "ct2": enc(ct2), "ct3": enc(ct3), "pt": enc(pt), "mono": enc(mono)}
        ntt = M.transform_to_ntt(ct, lvl)
        c["ntt"] = enc(ntt)
        ptn = M.plain_lift_ntt(pt, lvl)
        c["pt_ntt"] = enc(ptn)
        c["multiply_plain_ntt"] = enc(M.multiply_plain_ntt(ntt, ptn, lvl))
        c["multiply_plain"] = enc(M.multiply_plain_coeff(ct, pt, lvl))
        c["multiply_plain_mono"] = enc(M.multiply_plain_coeff(ct, mono, lvl))
        c["add"] = enc(M.add(ct, ct2, lvl))
        c["add_plain"] = enc(M.add_plain(ct, pt, lvl))
        if lvl > 0:
            c["mod_switch"] = enc(M.mod_switch_to_next(ct, lvl))
        c["multiply"] = enc(M.multiply(ct, ct2, lvl))
        c["square"] = enc(M.square(ct, lvl))
        c["relinearize"] = enc(M.relinearize(ct3, rk, lvl))
        cases["levels"].append(c)
    last = rand_ct(M, rnd, 2, 0)
    cases["clear_in"] = enc(last)
    cases["clear_out"] = enc(M.clear_irrelevant_bits(last))
    cases["irrelevant_bit_count"] = M.irrelevant_bit_count()
    return cases


def gen_path():
    rnd = random.Random(0x41505356)
    n, bits, pb = 64, [40, 40, 40, 36], 17
    ps_low, max_items, qpowers = 3, 11, [1, 4]
    M = pm.Model.from_bits(n, bits, 0, pb)
    targets = pm.create_powers_set(ps_low, max_items)
    depth, nodes = pm.powers_dag(qpowers, targets)
    s = [rnd.randrange(3) - 1 for _ in range(n)]
    rk = relin_keys(M, rnd, s)
    x = [rnd.randrange(M.t) for _ in range(n)]
    srcs = {e: encrypt(M, rnd, s, M.encode([pow(v, e, M.t) for v in x])) for e in qpowers}
    pw = pm.compute_powers(M, srcs, nodes, rk, ps_low)
    out = {"n": n, "coeff_bits": bits, "plain_bits": pb, "ps_low_degree": ps_low, "max_items_per_bin": max_items,
           "query_powers": qpowers, "targets": targets, "dag_depth": depth, "dag_nodes": [list(nd) for nd in nodes],
           "secret": s, "x": enc(x), "rk": enc(rk), "sources": {str(e): enc(ct) for e, ct in srcs.items()},
           "powers": {str(p): enc(ct) for p, ct in pw.items()}, "bundles": []}
    low = min(M.first, 2)
    for degree in (10, 8, 3, 11):
        A = [[rnd.randrange(M.t) for _ in range(n)] for _ in range(degree + 1)]
        A[degree] = [1] * n
        coeffs, flags = [], []
        for d in range(degree + 1):
            e = M.encode(A[d])
            is_ntt = pm.coeff_is_ntt(ps_low, d)
            coeffs.append(M.plain_lift_ntt(e, low) if is_ntt else e)
            flags.append(bool(is_ntt))
        mask_vals = [rnd.randrange(M.t) for _ in range(n)]
        mask = M.encode(mask_vals)
        if ps_low > 1 and ps_low < degree:
            res = pm.eval_patstock(M, pw, coeffs, ps_low, rk, mask)
        else:
            res = pm.eval_plain(M, pw, coeffs, low, mask)
        dec, budget = M.decrypt(s, res, 0)
        slots = M.decode(dec)
        exp = []
        for k in range(n):
            acc = 0
            for d in range(degree, -1, -1):
                acc = (acc * x[k] + A[d][k]) % M.t
            exp.append((acc + mask_vals[k]) % M.t)
        assert slots == exp, "model self-check: decrypt(eval) != P(x) + mask"
        out["bundles"].append({"degree": degree, "coeffs": enc(coeffs), "is_ntt": flags, "mask": enc(mask),
                               "expected_slots": enc(exp), "result": enc(res), "noise_budget": budget})
    return out


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, gen in (("ops_n64.json", gen_ops), ("path_n64.json", gen_path)):
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(gen(), f, separators=(",", ":"))
        print("wrote", name, os.path.getsize(os.path.join(OUT, name)), "bytes")
