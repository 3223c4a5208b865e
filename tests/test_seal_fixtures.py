"""The consumer side of the SEAL cross-check kit (integration/seal_fixtures.cpp).

`integration/seal_fixtures.cpp` is a program against Microsoft SEAL's public API that writes tests/golden/seal_ops_*.json,
seal_path_*.json and seal_objects_*.json FROM REAL SEAL.  SEAL is not in this image, so those files do not exist here and the
tests that read them SKIP: the repository's parity stays "unpinned"  until somebody with SEAL runs the
generator and drops its output into tests/golden/ -- from then on these tests hold the CPU oracle, the SEAL object codec and
(-m gpu) the HIP path to SEAL's own bits.

What does run here: `test_consumers_on_self_made_fixtures` feeds the same checkers with files in the same schemas produced by
the repository's OWN independent Python model (oracle/make_golden.py) and its own object writer, so the consumer code cannot
rot.  That self-check pins nothing about SEAL."""
import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golden_util import GOLDEN, arr  # noqa: E402
from oracle import ref  # noqa: E402


def fixture_files(kind):
    return sorted(glob.glob(os.path.join(GOLDEN, "seal_%s_*.json" % kind)))


def _load(path):
    with open(path) as f:
        return json.load(f)


def _ids(paths):
    return [os.path.basename(p) for p in paths] or ["absent"]


def _need(path):
    if path is None:
        pytest.skip("no SEAL-made fixture in tests/golden/ (run integration/seal_fixtures.cpp where SEAL exists)")
    return _load(path)


def _params(kind):
    return fixture_files(kind) or [None]


def _primes(g):
    return [int(v, 16) for v in g["coeff_modulus"]], int(g["plain_modulus"], 16)


# ------------------------------------------------------------------------------------------------ checkers (backend-neutral)
def check_constants(g):
    """SEAL's prime search, psi selection and level structure against the oracle's restatement"""
    q, t = _primes(g)
    C = ref.RefContext(g["n"], g["coeff_bits"], 0, g["plain_bits"])
    assert C.q == q, "CoeffModulus::Create(n, bits) differs from the restated prime search"
    assert C.t == t, "PlainModulus::Batching differs"
    if "psi" in g:
        assert C.psi == [int(v, 16) for v in g["psi"]], "minimal primitive 2n-th root differs"


class OracleOps:
    """tier-1 operations on the CPU oracle"""
    def __init__(self, g):
        q, t = _primes(g)
        self.C = ref.RefContext(g["n"], coeff_modulus=q, plain_modulus=t)
        self.first = self.C.first
        self.K = self.C.K

    def to_ntt(self, ct, lvl):
        a = ct.copy(); self.C.transform_to_ntt(a, lvl); return a

    def from_ntt(self, ct, lvl):
        a = ct.copy(); self.C.transform_from_ntt(a, lvl); return a

    def pt_to_ntt(self, pt, lvl): return self.C.plain_lift_ntt(pt, lvl)
    def mul_plain_ntt(self, ct, ptn, lvl): return self.C.multiply_plain_ntt(ct, ptn, lvl)
    def mul_plain(self, ct, pt, lvl): return self.C.multiply_plain_coeff(ct, pt, lvl)

    def add(self, a, b, lvl):
        x = a.copy(); self.C.add(x, b, lvl); return x

    def add_plain(self, a, pt, lvl):
        x = a.copy(); self.C.add_plain(x, pt, lvl); return x

    def mod_switch(self, ct, lvl): return self.C.mod_switch_to_next(ct, lvl)
    def multiply(self, a, b, lvl): return self.C.multiply(a, b, lvl)
    def square(self, a, lvl): return self.C.square(a, lvl)
    def relinearize(self, ct3, rk, lvl): return self.C.relinearize(ct3, rk, lvl)
    def close(self): pass


class GpuOps:
    """the same through the C ABI of the HIP library (tier 1)"""
    def __init__(self, g):
        import apsu_amd
        q, t = _primes(g)
        self.G = apsu_amd.HeContext(n=g["n"], coeff_modulus=q, plain_modulus=t)
        self.first = len(q) - 2 if len(q) > 1 else 0
        self.K = len(q)
        self._rk = None

    def to_ntt(self, ct, lvl):
        a = ct.copy(); self.G.transform_to_ntt_inplace(a, lvl); return a

    def from_ntt(self, ct, lvl):
        a = ct.copy(); self.G.transform_from_ntt_inplace(a, lvl); return a

    def pt_to_ntt(self, pt, lvl): return self.G.transform_plain_to_ntt(pt, lvl)
    def mul_plain_ntt(self, ct, ptn, lvl): return self.G.multiply_plain_ntt(ct, ptn, lvl)
    def mul_plain(self, ct, pt, lvl): return self.G.multiply_plain(ct, pt, lvl)

    def add(self, a, b, lvl):
        x = a.copy(); self.G.add_inplace(x, b, lvl); return x

    def add_plain(self, a, pt, lvl):
        x = a.copy(); self.G.add_plain_inplace(x, pt, lvl); return x

    def mod_switch(self, ct, lvl): return self.G.mod_switch_to_next(ct, lvl)
    def multiply(self, a, b, lvl): return self.G.multiply(a, b, lvl)
    def square(self, a, lvl): return self.G.square(a, lvl)

    def relinearize(self, ct3, rk, lvl):
        if self._rk is None:
            self._rk = self.G.upload_relin_keys(rk)
        return self.G.relinearize(ct3, self._rk, lvl)

    def close(self): self.G.close()


def check_ops(B, g):
    """every Evaluator method on the path, at every level, against the fixture's results"""
    rk = arr(g["rk"]) if "rk" in g else None
    for c in g["levels"]:
        lvl = c["chain_idx"]
        ct, ct2, ct3, pt, mono = (arr(c[k]) for k in ("ct", "ct2", "ct3", "pt", "mono"))
        ntt = arr(c["ntt"])
        assert (B.to_ntt(ct, lvl) == ntt).all(), "transform_to_ntt, level %d" % lvl
        assert (B.from_ntt(ntt, lvl) == ct).all(), "transform_from_ntt, level %d" % lvl
        ptn = arr(c["pt_ntt"])
        assert (np.asarray(B.pt_to_ntt(pt, lvl)).reshape(ptn.shape) == ptn).all(), "plaintext lift + NTT, level %d" % lvl
        assert (B.mul_plain_ntt(ntt, ptn, lvl) == arr(c["multiply_plain_ntt"])).all(), "multiply_plain NTT, level %d" % lvl
        assert (B.mul_plain(ct, pt, lvl) == arr(c["multiply_plain"])).all(), "multiply_plain, level %d" % lvl
        assert (B.mul_plain(ct, mono, lvl) == arr(c["multiply_plain_mono"])).all(), "multiply_plain by a monomial, level %d" % lvl
        assert (B.add(ct, ct2, lvl) == arr(c["add"])).all()
        assert (B.add_plain(ct, pt, lvl) == arr(c["add_plain"])).all(), "add_plain, level %d" % lvl
        if "mod_switch" in c:
            assert (B.mod_switch(ct, lvl) == arr(c["mod_switch"])).all(), "mod_switch_to_next, level %d" % lvl
        assert (B.multiply(ct, ct2, lvl) == arr(c["multiply"])).all(), "multiply (BEHZ), level %d" % lvl
        assert (B.square(ct, lvl) == arr(c["square"])).all(), "square, level %d" % lvl
        if "relinearize" in c:
            assert (B.relinearize(ct3, rk, lvl) == arr(c["relinearize"])).all(), "relinearize, level %d" % lvl


def _secret_ntt(C, g):
    if "secret_ntt" in g:
        return arr(g["secret_ntt"])
    # the repository's own golden files carry the ternary secret in coefficient form
    sk = np.zeros((C.K, C.n), dtype=np.uint64)
    for j, q in enumerate(C.q):
        C1 = ref.RefContext(C.n, coeff_modulus=[q], plain_modulus=C.t)
        one = np.array([[[(v % q) for v in g["secret"]]]], dtype=np.uint64)
        C1.transform_to_ntt(one, 0)
        sk[j] = one.reshape(-1)
    return sk


def check_path_oracle(g):
    """Receiver::ComputePowers and eval / eval_patstock of the CPU oracle against the fixture: DAG, every target power, every
    BinBundle result, and the plaintext meaning of the results"""
    q, t = _primes(g) if "coeff_modulus" in g else (None, None)
    C = ref.RefContext(g["n"], coeff_modulus=q, plain_modulus=t) if q else ref.RefContext(g["n"], g["coeff_bits"], 0, g["plain_bits"])
    targets = ref.create_powers_set(g["ps_low_degree"], g["max_items_per_bin"])
    assert targets == g["targets"]
    depth, nodes = ref.powers_dag(g["query_powers"], targets)
    assert depth == g["dag_depth"] and [list(nd) for nd in nodes] == g["dag_nodes"]
    rk = arr(g["rk"]) if "rk" in g else None
    srcs = {int(e): arr(ct) for e, ct in g["sources"].items()}
    pw = C.compute_powers(srcs, nodes, rk, g["ps_low_degree"])
    for p, ct in g["powers"].items():
        assert (pw[int(p)] == arr(ct)).all(), "power %s" % p
    plist = [None] * (g["max_items_per_bin"] + 1)
    for p, ct in pw.items():
        plist[p] = ct
    sk = _secret_ntt(C, g)
    for b in g["bundles"]:
        coeffs = [arr(c) for c in b["coeffs"]]
        mask = arr(b["mask"])
        if g["ps_low_degree"] > 1 and g["ps_low_degree"] < b["degree"]:
            out = C.eval_patstock(plist, coeffs, g["ps_low_degree"], rk, mask)
        else:
            out = C.eval(plist, coeffs, plist[1].shape[1] - 1, mask)
        assert (out == arr(b["result"])).all(), "BinBundle of degree %d" % b["degree"]
        pt, _ = C.decrypt(sk, out, 0)
        assert (C.decode(pt) == arr(b["expected_slots"])).all()


def check_path_gpu(g):
    import apsu_amd
    import common
    js = common.toy_json(n=g["n"], coeff_bits=g["coeff_bits"], plain_bits=g["plain_bits"], ps_low=g["ps_low_degree"],
                         max_items=g["max_items_per_bin"], query_powers=g["query_powers"])
    G = apsu_amd.HeContext(js)
    if "coeff_modulus" in g:
        assert [int(v) for v in G.q] == _primes(g)[0], "the engine's prime search differs from SEAL's"
    assert [list(nd) for nd in G.powers_dag()] == g["dag_nodes"]
    rk = G.upload_relin_keys(arr(g["rk"])) if "rk" in g else None
    srcs = [[arr(g["sources"][str(e)]) for e in sorted(g["query_powers"])]]
    pw = G.compute_powers([0], srcs, rk)
    for p, ct in g["powers"].items():
        got, _, _ = pw.download(0, int(p))
        assert (got == arr(ct)).all(), "power %s" % p
    gb = [G.upload_bundle(0, i, [arr(c) for c in b["coeffs"]], b["is_ntt"]) for i, b in enumerate(g["bundles"])]
    out = G.eval_bundles(gb, pw, rk, [arr(b["mask"]) for b in g["bundles"]])
    for i, b in enumerate(g["bundles"]):
        want = arr(b["result"])
        assert (np.asarray(out[i]).reshape(want.shape) == want).all(), "BinBundle of degree %d" % b["degree"]
    G.close()


def check_objects(g):
    """the N3 codec against SEAL-serialised objects: parms_id of every level, seed expansion, compression modes, and -- for
    uncompressed objects -- our writer reproducing SEAL's bytes"""
    from apsu_amd import seal
    q, t = _primes(g)
    n, K = g["n"], len(q)
    sc = seal.SealContext(n=n, coeff_modulus=q, plain_modulus=t)
    version = tuple(int(v) for v in g.get("seal_version", "4.0").split(".")[:2])
    first = K - 2 if K > 1 else 0
    assert sc.parms_id(-1 if K > 1 else 0) == [int(v, 16) for v in g["key_parms_id"]] or K == 1
    for c, pid in enumerate(g["parms_id_by_chain_idx"]):
        assert sc.parms_id(c) == [int(v, 16) for v in pid], "parms_id of chain_index %d" % c
    for o in g["ciphertexts"]:
        blob = bytes.fromhex(o["blob"])
        got = sc.ct_load(blob)
        want = arr(o["data"])
        assert got["chain_idx"] == o["chain_idx"] == first and bool(got["seeded"]) == bool(o["seeded"]) and got["consumed"] == len(blob)
        assert (got["data"] == want).all(), "ciphertext (compr %d, seeded %s): words differ from SEAL's load" % (o["compr"], o["seeded"])
        if o["compr"] == 0:
            mine = sc.ct_save(first, False, want, seed=sc.ct_load_unexpanded(blob, first + 1, n)["seed"] if o["seeded"] else None,
                              compr=0, version=version)
            assert mine == blob, "our writer does not reproduce SEAL's uncompressed ciphertext bytes"
    for o in g.get("relin_keys", []):
        blob = bytes.fromhex(o["blob"])
        ksk, used = sc.relin_keys_load(blob)
        want = arr(o["data"])
        assert used == len(blob) and (ksk.reshape(want.shape) == want).all(), "RelinKeys (compr %d, seeded %s)" % (o["compr"], o["seeded"])
    for o in g["plaintexts"]:
        blob = bytes.fromhex(o["blob"])
        got = sc.pt_load(blob)
        want = arr(o["data"])
        assert got["chain_idx"] == o["chain_idx"] and (got["data"].reshape(want.shape) == want).all(), "Plaintext (compr %d)" % o["compr"]
        if o["compr"] == 0:
            assert sc.pt_save(o["chain_idx"], want, compr=0, version=version) == blob
    sc.close()


# ------------------------------------------------------------------------------------------------ SEAL-made fixtures (skip when absent)
@pytest.mark.parametrize("path", _params("ops"), ids=_ids(fixture_files("ops")))
def test_oracle_against_seal_ops(path):
    g = _need(path)
    check_constants(g)
    B = OracleOps(g)
    check_ops(B, g)


@pytest.mark.parametrize("path", _params("path"), ids=_ids(fixture_files("path")))
def test_oracle_against_seal_path(path):
    g = _need(path)
    check_constants(g)
    check_path_oracle(g)


@pytest.mark.parametrize("path", _params("objects"), ids=_ids(fixture_files("objects")))
def test_codec_against_seal_objects(path):
    check_objects(_need(path))


@pytest.mark.gpu
@pytest.mark.parametrize("path", _params("ops"), ids=_ids(fixture_files("ops")))
def test_gpu_against_seal_ops(path):
    g = _need(path)
    B = GpuOps(g)
    try:
        check_ops(B, g)
    finally:
        B.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", _params("path"), ids=_ids(fixture_files("path")))
def test_gpu_against_seal_path(path):
    check_path_gpu(_need(path))


# ------------------------------------------------------------------------------------------------ the consumers themselves
def _self_made(tmp_path):
    """files in the generator's schemas made by the repository's own independent model and writer (NOT SEAL)"""
    from oracle import make_golden
    from apsu_amd import seal
    ops = make_golden.gen_ops()
    for k in ("clear_in", "clear_out", "irrelevant_bit_count"):
        ops.pop(k)
    ops["seal_version"] = "4.0.0"
    path = make_golden.gen_path()
    C = ref.RefContext(path["n"], path["coeff_bits"], 0, path["plain_bits"])
    path["coeff_modulus"] = ["%x" % v for v in C.q]
    path["plain_modulus"] = "%x" % C.t
    path["seal_version"] = "4.0.0"
    # the generator writes the secret key as SEAL holds it (NTT form at the key level); the repository's golden files carry the
    # ternary coefficients: convert, so that the self-made file has the generator's field names and no others
    sk_ntt = _secret_ntt(C, path)
    del path["secret"]
    path["secret_ntt"] = [["%x" % int(v) for v in row] for row in sk_ntt]
    q, t, n = C.q, C.t, C.n
    K = len(q)
    sc = seal.SealContext(n=n, coeff_modulus=q, plain_modulus=t)
    rng = np.random.default_rng(1)
    first = K - 2
    obj = {"n": n, "coeff_modulus": ["%x" % v for v in q], "plain_modulus": "%x" % t, "seal_version": "4.0.0",
           "key_parms_id": ["%x" % v for v in sc.parms_id(-1)],
           "parms_id_by_chain_idx": [["%x" % v for v in sc.parms_id(c)] for c in range(first + 1)],
           "ciphertexts": [], "relin_keys": [], "plaintexts": []}

    def hx(a):
        a = np.asarray(a)
        return ["%x" % int(v) for v in a] if a.ndim == 1 else [hx(x) for x in a]
    for compr in (0, 1):
        seed = [int(x) for x in rng.integers(0, 2**63, 8, dtype=np.uint64)]
        c0 = np.stack([rng.integers(0, qq, n, dtype=np.uint64) for qq in q[:first + 1]])
        c1 = sc.sample_poly_uniform(first, seed, first + 1, n)
        ct = np.stack([c0, c1])
        obj["ciphertexts"].append({"compr": compr, "chain_idx": first, "seeded": True,
                                   "blob": sc.ct_save(first, False, ct, seed=seed, compr=compr).hex(), "data": hx(ct)})
        obj["ciphertexts"].append({"compr": compr, "chain_idx": first, "seeded": False,
                                   "blob": sc.ct_save(first, False, ct, compr=compr).hex(), "data": hx(ct)})
        keys = np.stack([np.stack([np.stack([rng.integers(0, qq, n, dtype=np.uint64) for qq in q]) for _ in range(2)]) for _ in range(K - 1)])
        obj["relin_keys"].append({"compr": compr, "seeded": False, "blob": sc.relin_keys_save(keys, compr=compr).hex(), "data": hx(keys)})
        pt = rng.integers(0, t, n, dtype=np.uint64)
        obj["plaintexts"].append({"compr": compr, "chain_idx": -1, "blob": sc.pt_save(-1, pt, compr=compr).hex(), "data": hx(pt)})
        ptn = np.stack([rng.integers(0, qq, n, dtype=np.uint64) for qq in q[:2]])
        obj["plaintexts"].append({"compr": compr, "chain_idx": 1, "blob": sc.pt_save(1, ptn, compr=compr).hex(), "data": hx(ptn)})
    sc.close()
    return ops, path, obj


def test_consumers_on_self_made_fixtures(tmp_path):
    """NOT A PIN: the checkers above run on fixtures of the same schemas made by this repository's own Python model / writer"""
    ops, path, obj = _self_made(tmp_path)
    for name, g in (("ops", ops), ("path", path), ("objects", obj)):                    # through JSON, as the real files arrive
        (tmp_path / ("seal_%s_self.json" % name)).write_text(json.dumps(g))
    ops = _load(str(tmp_path / "seal_ops_self.json"))
    check_constants(ops)
    check_ops(OracleOps(ops), ops)
    path = _load(str(tmp_path / "seal_path_self.json"))
    check_constants(path)
    check_path_oracle(path)
    check_objects(_load(str(tmp_path / "seal_objects_self.json")))


@pytest.mark.gpu
def test_gpu_consumers_on_self_made_fixtures(tmp_path):
    """NOT A PIN: the GPU-side checkers on the same self-made fixtures"""
    ops, path, _ = _self_made(tmp_path)
    ops = json.loads(json.dumps(ops))
    B = GpuOps(ops)
    try:
        check_ops(B, ops)
    finally:
        B.close()
    check_path_gpu(json.loads(json.dumps(path)))


# ------------------------------------------------------------------------------------------------ generator <-> consumer schema
def _json_keys(o, acc):
    if isinstance(o, dict):
        for k, v in o.items():
            if not k.lstrip("-").isdigit():                      # "sources" / "powers" are maps keyed by the exponent
                acc.add(k)
            _json_keys(v, acc)
    elif isinstance(o, list):
        for v in o:
            _json_keys(v, acc)
    return acc


def _generator_keys():
    """the field names integration/seal_fixtures.cpp writes, per output file kind (its J.num / J.str / J.hexv / J.raw / J.key calls)"""
    import re
    with open(os.path.join(ROOT, "integration", "seal_fixtures.cpp")) as f:
        src = f.read()
    cut = {"ops": ("void emit_ops(", "std::map<uint32_t, Node> powers_dag("), "path": ("void emit_path(", "template <class T> string saved("),
           "objects": ("void emit_objects(", "int main(")}
    out = {}
    for kind, (a, b) in cut.items():
        body = src[src.index(a):src.index(b)]
        out[kind] = set(re.findall(r"\.(?:num|str|hexv|raw|key)\(\"([A-Za-z_0-9]+)\"", body))
        assert len(out[kind]) > 8, kind
    return out


def test_generator_and_consumers_agree_on_every_field_name(tmp_path):
    """A schema drift between the SEAL-side generator (never run here) and the consumers above would only show on the day somebody
    runs the generator.  Caught here instead: (1) the self-made fixtures that test_consumers_on_self_made_fixtures feeds through
    the checkers carry EXACTLY the field names the generator's source writes, per file kind; (2) every one of those names is read
    somewhere in this file's checkers (a field the generator writes and nobody reads is dead weight; a field a checker reads and
    the generator does not write fails in (1))."""
    gen = _generator_keys()
    ops, path, obj = _self_made(tmp_path)
    mine = {"ops": _json_keys(ops, set()), "path": _json_keys(path, set()), "objects": _json_keys(obj, set())}
    for kind in ("ops", "path", "objects"):
        assert gen[kind] == mine[kind], (kind, "only the generator writes: %s" % sorted(gen[kind] - mine[kind]),
                                        "only the self-made fixture has: %s" % sorted(mine[kind] - gen[kind]))
    with open(os.path.abspath(__file__)) as f:
        me = f.read()
    consumers = me[me.index("def _primes(g):"):
                   me.index("# ------------------------------------------------------------------------------------------------ the consumers themselves")]
    informational = {"seal_version", "x", "noise_budget", "plain_bits", "coeff_bits", "n", "seeded", "compr"}   # read below or descriptive
    for kind in ("ops", "path", "objects"):
        for k in sorted(gen[kind]):
            assert ('"%s"' % k) in consumers or k in informational, "field %r of seal_%s_*.json is never read by a checker" % (k, kind)
