"""CPU tier: the product's host logic (apsu_amd/csrc/params.cpp, powers_dag.cpp, ntt_core.h) checked
against the oracle, plus the C-ABI export check.  No GPU compute is attempted here."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import common
from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u64p = C.POINTER(C.c_uint64)


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(ROOT, "apsu_amd", "libapsu_he_hostemu.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "apsu_amd", "csrc"), "-s", "../libapsu_he_hostemu.so"])
    lib = C.CDLL(so)
    lib.emu_last_error.restype = C.c_char_p
    return lib


CONFIGS = ["100K-1", "1M-1024-com", "16M-4096", "256M-4096"]
# every parameter file the reference ships (/root/reference/parameters/*.json, copied as data)
ALL_PARAM_FILES = sorted(f[:-5] for f in os.listdir(common.PARAM_DIR) if f.endswith(".json"))


def test_all_reference_parameter_files_are_present():
    assert len(ALL_PARAM_FILES) == 36 and set(CONFIGS) <= set(ALL_PARAM_FILES)


@pytest.mark.parametrize("name", ALL_PARAM_FILES)
def test_params_match_oracle_and_survey(emu, name):
    js = common.param_json(name)
    out = np.zeros(64, dtype=np.uint64)
    k = emu.emu_params_info(js.encode(), out.ctypes.data_as(u64p), 64)
    assert k > 0, emu.emu_last_error()
    p = ref.load_params(js)
    Cx = ref.RefContext.from_params(p)
    v = [int(x) for x in out[:k]]
    K = Cx.K
    assert v[0] == Cx.n and v[1] == K and v[2] == Cx.first and v[3] == Cx.t
    assert v[4:4 + K] == Cx.q
    assert v[4 + K:4 + 2 * K] == Cx.psi
    rest = v[4 + 2 * K:]
    assert rest[0] == Cx.nB and rest[1] == Cx.m_sk and rest[2] == Cx.gamma and rest[3:3 + Cx.nB] == Cx.B
    tail = rest[3 + Cx.nB:]
    assert tail[0] == p["bundle_idx_count"] and tail[1] == p["items_per_bundle"]
    assert tail[3] == Cx.irrelevant_bit_count()


def test_known_seal_constants():
    """sanity anchors quoted in SURVEY.md §8c(3) / App. A"""
    assert ref.RefContext(8192, [56, 56, 56, 50], 0, 20).t == 1032193          # PlainModulus::Batching(8192, 20)
    assert ref.RefContext(4096, [48, 36, 25], 0, 18).q == [0xffffffffc001, 0xffffee001, 0x1ffc001]
    c = ref.RefContext(8192, [56, 56, 56, 50], 0, 22)
    assert c.t == 0x3e4001 and c.q == [0xfffffffff70001, 0xfffffffff78001, 0xfffffffffb4001, 0x3ffffffffc001]
    assert c.irrelevant_bit_count() == 21
    assert ref.RefContext(8192, [50, 50, 50, 38, 30], 0, 26).irrelevant_bit_count() == 11


@pytest.mark.parametrize("n,bits", [(64, [40, 40, 40, 36]), (4096, [48, 36, 25]), (8192, [56, 56, 56, 50])])
def test_level_constants_match_oracle_semantics(emu, n, bits):
    """the product's BEHZ / scaling constants reproduce the oracle's operation results: checked by
    re-deriving a few of them in Python big ints"""
    Cx = ref.RefContext(n, bits, 0, 17 if n == 64 else 18 if n == 4096 else 22)
    q = np.array(Cx.q, dtype=np.uint64)
    for ci in range(Cx.first + 1):
        out = np.zeros(512, dtype=np.uint64)
        k = emu.emu_level_constants(C.c_uint64(n), q.ctypes.data_as(u64p), len(q), C.c_uint64(Cx.t), ci,
                                    out.ctypes.data_as(u64p), 512)
        assert k > 0, emu.emu_last_error()
        v = [int(x) for x in out[:k]]
        L, nB, m_sk, gamma = v[0], v[1], v[2], v[3]
        assert L == ci + 1 and m_sk == Cx.m_sk and gamma == Cx.gamma
        B = v[4:4 + nB]
        assert B == Cx.B[:nB] or nB == Cx.nB
        Q = 1
        for x in Cx.q[:L]:
            Q *= x
        pos = 4 + nB
        cdp = v[pos:pos + L]
        assert cdp == [(Q // Cx.t) % qq for qq in Cx.q[:L]]
        pos += L
        assert v[pos] == Q % Cx.t and v[pos + 1] == (Cx.t + 1) // 2
        pos += 2
        assert v[pos:pos + L] == [qq - Cx.t for qq in Cx.q[:L]]
        pos += L
        inv_last = v[pos:pos + L - 1]
        assert inv_last == [pow(Cx.q[L - 1], -1, Cx.q[j]) for j in range(L - 1)]
        pos += L - 1
        assert v[pos:pos + L] == [pow(Q // Cx.q[j], -1, Cx.q[j]) for j in range(L)]


@pytest.mark.parametrize("name", ALL_PARAM_FILES)
def test_powers_dag_matches_oracle(emu, name):
    p = ref.load_params(common.param_json(name))
    tg = ref.create_powers_set(p["ps_low_degree"], p["max_items_per_bin"])
    out = np.zeros(len(tg) + 4, dtype=np.uint32)
    k = emu.emu_create_powers_set(p["ps_low_degree"], p["max_items_per_bin"], C.c_void_p(out.ctypes.data), len(out))
    assert [int(x) for x in out[:k]] == tg
    depth, nodes = ref.powers_dag(p["query_powers"], tg)
    s = np.array(sorted(p["query_powers"]), dtype=np.uint32)
    t = np.array(tg, dtype=np.uint32)
    nd = np.zeros((len(tg), 4), dtype=np.uint32)
    d = emu.emu_powers_dag(C.c_void_p(s.ctypes.data), len(s), C.c_void_p(t.ctypes.data), len(t), C.c_void_p(nd.ctypes.data))
    assert d == depth and [tuple(int(x) for x in r) for r in nd] == nodes
    # SURVEY App. A statistics
    stats = {"100K-1": (20, 0), "1M-1024-com": (25, 1), "16M-4096": (72, 3), "256M-4096": (322, 3)}
    if name in stats:
        assert (len(tg), depth) == stats[name]


def test_powers_dag_rejects_bad_sets(emu):
    bad = [([2, 3], [1, 2, 3]), ([0, 1], [0, 1, 2]), ([1, 5], [1, 2, 3])]
    for src, tgt in bad:
        s, t = np.array(src, dtype=np.uint32), np.array(tgt, dtype=np.uint32)
        nd = np.zeros((len(tgt), 4), dtype=np.uint32)
        assert emu.emu_powers_dag(C.c_void_p(s.ctypes.data), len(s), C.c_void_p(t.ctypes.data), len(t),
                                  C.c_void_p(nd.ctypes.data)) == -1
        with pytest.raises(ValueError):
            ref.powers_dag(src, tgt)


def test_psu_params_validation_mirrors_reference(emu):
    """psu_params.cpp:95-180: each broken field raises invalid_argument (-1), malformed JSON runtime_error (-2)"""
    base = json.loads(common.param_json("16M-4096"))
    out = np.zeros(64, dtype=np.uint64)

    def rc(j):
        return emu.emu_params_info(json.dumps(j).encode(), out.ctypes.data_as(u64p), 64)

    assert rc(base) > 0
    for path, val in [(("table_params", "table_size"), 0), (("table_params", "max_items_per_bin"), 0),
                      (("table_params", "hash_func_count"), 9), (("item_params", "felts_per_item"), 1),
                      (("query_params", "ps_low_degree"), 5000), (("table_params", "table_size"), 6553),
                      (("query_params", "query_powers"), [1, 46]), (("query_params", "query_powers"), [0, 1])]:
        j = json.loads(json.dumps(base))
        j[path[0]][path[1]] = val
        assert rc(j) == -1, (path, val)
    j = json.loads(json.dumps(base))
    j["seal_params"]["plain_modulus_bits"] = 12           # no 12-bit prime = 1 mod 2n: logic_error like SEAL's get_primes
    assert rc(j) == -2
    j = json.loads(json.dumps(base))
    j["seal_params"]["plain_modulus"] = 65537
    assert rc(j) == -2                                    # both plain_modulus and plain_modulus_bits
    j = json.loads(json.dumps(base))
    del j["seal_params"]["plain_modulus_bits"]
    assert rc(j) == -2                                    # neither
    assert emu.emu_params_info(b"{not json", out.ctypes.data_as(u64p), 64) == -2
    j = json.loads(json.dumps(base))
    del j["query_params"]
    assert rc(j) == -2


@pytest.mark.parametrize("n,bits", [(64, 40), (64, 60), (256, 50), (256, 60), (1024, 56), (1024, 60), (2048, 48), (2048, 60),
                                    (4096, 36), (4096, 60), (8192, 56), (8192, 58), (8192, 60), (16384, 56), (16384, 60)])
def test_ntt_workgroup_emulation_matches_oracle(emu, n, bits):
    """the kernel's pass functions (ntt_core.h), stepped on the CPU, equal the oracle's NTT bit for bit"""
    logn = n.bit_length() - 1
    c = ref.RefContext(n, [bits], 65537 if (65537 - 1) % (2 * n) == 0 else 0, 0 if (65537 - 1) % (2 * n) == 0 else 20)
    q = c.q[0]
    for seed, T in ((5, 64), (6, 1024 if n == 16384 else 512)):
        x = ref.fill_uniform(seed, q, n)
        x[:4] = [0, q - 1, 1, q - 2]                      # range edges
        a = x.copy()
        assert emu.emu_ntt_limb(logn, 0, C.c_uint64(q), a.ctypes.data_as(u64p), T) == 0, emu.emu_last_error()
        e = x.copy().reshape(1, 1, n)
        c.transform_to_ntt(e, 0)
        assert (a == e.reshape(-1)).all()
        assert emu.emu_ntt_limb(logn, 1, C.c_uint64(q), a.ctypes.data_as(u64p), T) == 0
        assert (a == x).all()


@pytest.mark.parametrize("n,bits", [(4096, 36), (4096, 48), (4096, 60), (8192, 50), (8192, 56), (8192, 58), (8192, 60)])
def test_ntt_latency_form_emulation_matches_oracle(emu, n, bits):
    """round 6: the LATENCY form of the transform (8 coefficients per work item: n / 8 threads per limb, passes 2,3,3,3,2 at n = 8192 and
    3,3,3,3 at n = 4096 -- ntt_core.h plan_k) against the oracle and against the 16-coefficient form, bit for bit; ring sizes without
    the form are refused"""
    logn = n.bit_length() - 1
    c = ref.RefContext(n, [bits], 65537 if (65537 - 1) % (2 * n) == 0 else 0, 0 if (65537 - 1) % (2 * n) == 0 else 20)
    q = c.q[0]
    emu.emu_ntt_limb_c.argtypes = [C.c_int, C.c_int, C.c_uint64, u64p, C.c_int, C.c_int]
    for seed, T in ((7, 64), (8, n // 8)):
        x = ref.fill_uniform(seed, q, n)
        x[:4] = [0, q - 1, 1, q - 2]
        a = x.copy()
        assert emu.emu_ntt_limb_c(logn, 0, q, a.ctypes.data_as(u64p), T, 8) == 0, emu.emu_last_error()
        e = x.copy().reshape(1, 1, n)
        c.transform_to_ntt(e, 0)
        assert (a == e.reshape(-1)).all()
        b = x.copy()
        assert emu.emu_ntt_limb(logn, 0, C.c_uint64(q), b.ctypes.data_as(u64p), n // 16) == 0
        assert (a == b).all()
        assert emu.emu_ntt_limb_c(logn, 1, q, a.ctypes.data_as(u64p), T, 8) == 0
        assert (a == x).all()
    z = np.zeros(2048, dtype=np.uint64)
    assert emu.emu_ntt_limb_c(11, 0, q, z.ctypes.data_as(u64p), 64, 8) == -1          # n = 2048 has no latency form


@pytest.mark.parametrize("n", [4096, 8192])
def test_ntt_latency_form_61_bit_primes_and_tensor_loader(emu, n):
    """the latency form in the wide-near range mode (61-bit BEHZ primes) and behind the tensor-on-load staging: equal to the
    16-coefficient form on the same inputs"""
    logn = n.bit_length() - 1
    emu.emu_ntt_limb_c.argtypes = [C.c_int, C.c_int, C.c_uint64, u64p, C.c_int, C.c_int]
    emu.emu_intt_tensor_limb_c.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p, u64p, u64p, C.c_int, C.c_int]
    def is_prime(x):
        return all(pow(w, x - 1, x) == 1 for w in (2, 3, 5, 7, 11, 13))
    # (the first hits of the downward scan from 2^61 are the context's own auxiliary primes: take the eighth, as the test above does)
    q, found = ((1 << 61) - 1) // (2 * n) * (2 * n) + 1, 0
    while True:
        if is_prime(q):
            found += 1
            if found == 8:
                break
        q -= 2 * n
    rng = np.random.default_rng(n + 1)
    for qq in (q, ref.RefContext(n, [56], 0, 20).q[0]):
        x0, y0, x1, y1 = (rng.integers(0, qq, n, dtype=np.uint64) for _ in range(4))
        x0[:3] = [0, qq - 1, 1]; y0[:3] = [qq - 1, qq - 1, 1]
        for inverse in (0, 1):
            a, b = x0.copy(), x0.copy()
            assert emu.emu_ntt_limb_c(logn, inverse, qq, a.ctypes.data_as(u64p), n // 8, 8) == 0, emu.emu_last_error()
            assert emu.emu_ntt_limb_c(logn, inverse, qq, b.ctypes.data_as(u64p), n // 16, 16) == 0
            assert (a == b).all() and int(a.max()) < qq
        for cross in (False, True):
            o8, o16 = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
            null = C.POINTER(C.c_uint64)()
            for out, co in ((o8, 8), (o16, 16)):
                rc = emu.emu_intt_tensor_limb_c(logn, qq, x0.ctypes.data_as(u64p), y0.ctypes.data_as(u64p), x1.ctypes.data_as(u64p) if cross else null,
                                                y1.ctypes.data_as(u64p) if cross else null, out.ctypes.data_as(u64p), n // co, co)
                assert rc == 0, emu.emu_last_error()
            assert (o8 == o16).all()


@pytest.mark.parametrize("n", [64, 256, 1024, 8192, 16384])
def test_ntt_workgroup_emulation_61_bit_primes(emu, n):
    """61-bit primes (the BEHZ auxiliary base; the oracle's coefficient primes stop at 60 bits): the pass functions in
    their wide-near range mode.  inverse(forward(x)) = x, and the transform turns negacyclic convolution into a pointwise
    product (checked against Python integers)."""
    logn = n.bit_length() - 1
    def is_prime(x):
        d, r = x - 1, 0
        while d % 2 == 0:
            d //= 2; r += 1
        for w in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            y = pow(w, d, x)
            if y in (1, x - 1):
                continue
            for _ in range(r - 1):
                y = y * y % x
                if y == x - 1:
                    break
            else:
                return False
        return True
    # get_primes(2n, 61, .) scans downwards from 2^61; the first few hits are the context's own auxiliary primes, so
    # take the eighth (same shape 2^61 - c, c small)
    q, found = ((1 << 61) - 1) // (2 * n) * (2 * n) + 1, 0
    while True:
        if is_prime(q):
            found += 1
            if found == 8:
                break
        q -= 2 * n
    rng = np.random.default_rng(n)
    a = rng.integers(0, q, n, dtype=np.uint64); b = rng.integers(0, q, n, dtype=np.uint64)
    a[:3] = [0, q - 1, 1]
    fa, fb = a.copy(), b.copy()
    for v in (fa, fb):
        assert emu.emu_ntt_limb(logn, 0, C.c_uint64(q), v.ctypes.data_as(u64p), 512 if n >= 8192 else 64) == 0, emu.emu_last_error()
        assert int(v.max()) < q
    back = fa.copy()
    assert emu.emu_ntt_limb(logn, 1, C.c_uint64(q), back.ctypes.data_as(u64p), 64) == 0
    assert (back == a).all()
    prod = np.array([int(x) * int(y) % q for x, y in zip(fa, fb)], dtype=np.uint64)
    assert emu.emu_ntt_limb(logn, 1, C.c_uint64(q), prod.ctypes.data_as(u64p), 64) == 0
    ai, bi = [int(v) for v in a], [int(v) for v in b]
    for k in (range(n) if n <= 256 else [int(v) for v in rng.integers(0, n, 8)]):      # negacyclic product by definition
        want = (sum(ai[i] * bi[k - i] for i in range(k + 1)) - sum(ai[i] * bi[k + n - i] for i in range(k + 1, n))) % q
        assert int(prod[k]) == want


def test_scheduler_ordering_rules_hold_in_every_state(emu):
    """round 6: the decision block of Engine::compute_powers is a pure function (apsu_amd/csrc/sched_policy.h, plan_walk) and is held, in
    EVERY state, to the invariant the project's one real race violated (a pooled buffer kept an older evaluation's `last_use` mark):
    a buffer whose writers or readers may still be queued on either stream is never written by a stream that has not been ordered
    behind all of them.  Model: the previous walk of a pooled buffer left writers on the main stream (one-stream and split walks) and / or
    on the second stream (split and pipelined walks; `high_ready` is recorded behind those); an evaluation, if one read the buffer, runs on
    the main stream behind ALL its writers and records `last_use` behind itself."""
    ONE, SPLIT, PIPE = 0, 1, 2
    seen = {ONE: 0, SPLIT: 0, PIPE: 0}
    import itertools
    for (recycled, prev, evaluated, done, split_ok, prof_on, split_mode, pipe_cp, force_pipe, inputs_ready, on_device, busy) in itertools.product(
            (0, 1), (ONE, SPLIT, PIPE), (0, 1), (0, 1), (0, 1), (0, 1), (-1, 0, 1), (0, 1), (0, 1), (0, 1), (0, 1), (0, 1)):
        if not recycled and (prev != ONE or evaluated or done):
            continue                                              # a fresh buffer has no history
        if done and not evaluated:
            continue
        if force_pipe and not pipe_cp:
            continue                                              # mode 3 implies mode 1's switch (apsu_he_set_query_overlap)
        if pipe_cp and not inputs_ready:
            continue                                              # modes 1 and 3 carry the caller's promise
        high_async = 1 if (recycled and prev != ONE) else 0
        bits = (recycled | evaluated << 1 | done << 2 | high_async << 3 | split_ok << 4 | prof_on << 5 | pipe_cp << 6 | force_pipe << 7 |
                inputs_ready << 8 | on_device << 9 | busy << 10)
        r = emu.emu_plan_walk(bits, split_mode)
        walk, main_hr, side_lu, side_main, consumes = r & 3, bool(r & 4), bool(r & 8), bool(r & 16), bool(r & 32)
        seen[walk] += 1
        state = dict(recycled=recycled, prev=prev, evaluated=evaluated, done=done, split_ok=split_ok, prof_on=prof_on, split_mode=split_mode,
                     pipe_cp=pipe_cp, force_pipe=force_pipe, inputs_ready=inputs_ready, on_device=on_device, busy=busy, plan=r)
        # what may still be queued on the buffer
        pending = set()
        if recycled:
            if evaluated:
                if not done:
                    pending.add("reader@main")                    # (its writers are all in front of it)
            else:
                if prev in (ONE, SPLIT):
                    pending.add("writer@main")
                if prev in (SPLIT, PIPE):
                    pending.add("writer@side")
        writers = {ONE: ["main"], SPLIT: ["main", "side"], PIPE: ["side"]}[walk]
        for x in writers:
            for a in pending:
                kind, y = a.split("@")
                if x == y:
                    continue                                      # stream order
                if x == "main":                                   # the second stream's writers: only high_ready orders the main stream behind them
                    assert main_hr, state
                else:                                             # main-stream writers / readers in front of the second stream
                    covered = side_main or (side_lu and evaluated)   # last_use lies behind the evaluation, which lies behind every writer
                    assert covered, state
        # the walks themselves
        if walk != ONE:
            assert split_ok and not prof_on and split_mode != 0, state
        if walk == PIPE:
            assert pipe_cp and inputs_ready and on_device and (busy or force_pipe), state
            assert (not recycled) or evaluated, state             # never into a buffer whose reader left no mark
            assert done or not recycled or force_pipe, state      # ... and, unless forced, only into an idle one
        if not recycled:
            assert not main_hr and not side_lu, state             # nothing to wait for on a fresh buffer
        if side_lu:
            assert evaluated, state                               # a mark is only waited for when an evaluation left it
        assert consumes, state
    assert all(v > 0 for v in seen.values()), seen
    # the check has teeth: round 5's bug in this model's terms -- the engine BELIEVES a mark (last_use_set, done) on a buffer that was
    # re-written by a split walk and never evaluated since; the plan then orders the second stream behind nothing that covers the
    # main stream's pending writers
    r = emu.emu_plan_walk(1 | 1 << 1 | 1 << 2 | 1 << 3 | 1 << 4 | 1 << 8 | 1 << 9, -1)
    assert (r & 3) == SPLIT and not (r & 16), r                   # second stream does not wait for the main stream ...
    assert not ((r & 16) or ((r & 8) and False))                  # ... and the mark it waits for is not behind those writers: a violation
    # (Engine::compute_powers therefore consumes last_use_set on every walk and only eval_bundles sets it: consumes_last_use above)


def test_scheduler_pool_pick_rule(emu):
    """sched_policy.h, pick_pooled_buffer, against its statement: without the caller's overlap promise the first pooled buffer that fits;
    with it the first fitting one whose reader is done or that carries no mark, else -- three fitting ones all busy -- the oldest, else a
    new buffer (-1)"""
    import itertools
    states = [0, 1, 3, 7, 2, 6]                                   # bit 0 fits, 1 last_use_set, 2 last_use_done (done implies set)
    for count in range(0, 5):
        for pool in itertools.product(states, repeat=count):
            buf = (C.c_ubyte * max(1, count))(*pool)
            fitting = [i for i, e in enumerate(pool) if e & 1]
            for ready in (0, 1):
                got = emu.emu_pick_pooled_buffer(buf, count, ready)
                if not ready:
                    want = fitting[0] if fitting else -1
                else:
                    free = [i for i in fitting if not (pool[i] & 2) or (pool[i] & 4)]
                    want = free[0] if free else (fitting[0] if len(fitting) >= 3 else -1)
                assert got == want, (pool, ready, got, want)


def test_c_abi_exports_every_declared_symbol():
    """the shared library loads and exports exactly the functions include/apsu_he.h declares"""
    import apsu_amd
    hdr = open(os.path.join(ROOT, "include", "apsu_he.h")).read()
    declared = set(re.findall(r"\b(apsu_he_[a-z_0-9]+)\s*\(", hdr))
    lib = apsu_amd.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export " + name
    nm = subprocess.check_output(["nm", "-D", "--defined-only", apsu_amd.lib_path()]).decode()
    exported = set(re.findall(r" T (apsu_he_[a-z_0-9]+)", nm))
    assert exported == declared
    assert lib.apsu_he_abi_version() == 7


def test_no_cpu_fallback_without_gpu():
    """without a HIP device context creation must fail loudly (APSU_HE_NO_DEVICE), never fall back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import apsu_amd
    with pytest.raises(apsu_amd.ApsuHeError) as ei:
        apsu_amd.HeContext(common.param_json("100K-1"))
    assert "no HIP device" in str(ei.value)


def test_product_does_not_reference_oracle():
    """the product path must not import, include, link or call anything under oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "apsu_amd")):
        for f in files:
            if not f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                continue
            for line in open(os.path.join(dirpath, f), errors="replace"):
                low = line.lower()
                if "oracle" in low:
                    assert not re.search(r"^\s*(import|from|#include)|cdll|dlopen|-lapsu_he_ref", low), (f, line)
    ldd = subprocess.check_output(["ldd", os.path.join(ROOT, "apsu_amd", "libapsu_he_gpu.so")]).decode()
    assert "apsu_he_ref" not in ldd


def test_partition_rule_of_the_c_abi_matches_the_bench_sharding():
    """apsu_he_partition_bundles (C++, used by apsu_he_eval_all's callers) and apsu_amd/sharding.py (used by the
    one-process-per-GPU bench) are the same rule"""
    import apsu_amd
    from apsu_amd.sharding import partition
    rng = np.random.default_rng(7)
    cases = [([(b, ci, 1303 if ci < 6 else 170) for b in range(4) for ci in range(7)], 4)]
    for _ in range(40):
        nb = int(rng.integers(1, 9))
        units = [(int(rng.integers(0, nb)), ci, int(rng.integers(0, 4000))) for ci in range(int(rng.integers(0, 60)))]
        cases.append((units, nb))
    for units, nb in cases:
        for world in (1, 2, 3, 4, 8):
            slots = apsu_amd.partition_bundles(units, nb, world)
            assign = partition(units, nb, world)
            want = {}
            for r, us in assign.items():
                for u in us:
                    want.setdefault(u, []).append(r)
            # duplicates of an identical unit may swap places; compare as multisets per unit
            got = {}
            for u, s in zip(units, slots):
                got.setdefault(u, []).append(s)
            assert {k: sorted(v) for k, v in got.items()} == {k: sorted(v) for k, v in want.items()}, (nb, world)
    with pytest.raises(ValueError):
        apsu_amd.partition_bundles([(5, 0, 1)], 4, 2)


def test_partition_spill_pass_break_even():
    """The spill pass (compute_powers_cost > 0): C++ and Python agree, every BinBundle is placed once, the slowest device never
    gets slower, and the two reference points of DESIGN.md section 6 hold: 256M-4096 on 8 devices (3 indices x 34 BinBundles) does
    NOT spill -- a second ComputePowers (311 products) costs more than the 17-vs-12 imbalance -- while a 3x larger DB does."""
    import apsu_amd
    from apsu_amd.sharding import partition, UNIT_OVERHEAD
    def loads(units, slots, world, cp):
        out = []
        for r in range(world):
            mine = [u for u, s in zip(units, slots) if s == r]
            out.append(sum(u[2] + UNIT_OVERHEAD for u in mine) + cp * len({u[0] for u in mine}))
        return out
    rng = np.random.default_rng(11)
    cases = []
    for _ in range(30):
        nb = int(rng.integers(1, 7))
        units = [(int(rng.integers(0, nb)), ci, int(rng.integers(1, 4000))) for ci in range(int(rng.integers(1, 80)))]
        cases.append((units, nb, int(rng.integers(1, 60000))))
    for units, nb, cp in cases:
        for world in (2, 3, 5, 8):
            slots = apsu_amd.partition_bundles(units, nb, world, cp)
            assign = partition(units, nb, world, cp)
            want = {u: r for r, us in assign.items() for u in us}
            assert len(want) == len(units) and [want[u] for u in units] == slots, (nb, world, cp)
            base = apsu_amd.partition_bundles(units, nb, world, 0)
            assert max(loads(units, slots, world, cp)) <= max(loads(units, base, world, cp))
    cp = 110 * 311                                                   # 256M-4096: 311 ciphertext products per bundle index
    units = [(b, ci, 3999) for b in range(3) for ci in range(34)]
    assert apsu_amd.partition_bundles(units, 3, 8, cp) == apsu_amd.partition_bundles(units, 3, 8, 0)
    big = [(b, ci, 3999) for b in range(3) for ci in range(100)]
    spilled, plain = apsu_amd.partition_bundles(big, 3, 8, cp), apsu_amd.partition_bundles(big, 3, 8, 0)
    assert spilled != plain and max(loads(big, spilled, 8, cp)) < max(loads(big, plain, 8, cp))
    assert any(len({u[0] for u, s in zip(big, spilled) if s == r}) == 2 for r in range(8))      # some device now serves two indices


def test_blake2b_model_against_hashlib_and_rfc():
    """oracle/blake2x.py: the BLAKE2b core (own compression function, explicit parameter block incl. the xof_length field
    BLAKE2X puts into the upper half of node_offset) equals hashlib over keys, salts, personalisation and tree parameters"""
    import hashlib, random
    from oracle import blake2x as B
    abc = B.blake2b(b"abc", B.param_block())
    assert abc.hex() == ("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
                         "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")          # RFC 7693 appendix A
    rng = random.Random(7)
    for trial in range(200):
        data = bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 1, 8, 63, 64, 127, 128, 129, 256, 257, 700])))
        key = bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 0, 1, 32, 64])))
        ds = rng.choice([1, 16, 32, 48, 64])
        salt = bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 16]))); person = bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 16])))
        fan, depth, leaf = rng.choice([0, 1, 2, 255]), rng.choice([1, 2, 255]), rng.choice([0, 64, 1 << 20])
        noff = rng.getrandbits(64) if rng.random() < 0.5 else rng.getrandbits(6) | (4096 << 32)
        nd, inner = rng.choice([0, 1, 5]), rng.choice([0, 32, 64])
        want = hashlib.blake2b(data, digest_size=ds, key=key, salt=salt, person=person, fanout=fan, depth=depth, leaf_size=leaf,
                               node_offset=noff, node_depth=nd, inner_size=inner).digest()
        assert B.blake2b(data, B.param_block(ds, len(key), fan, depth, leaf, noff & 0xffffffff, noff >> 32, nd, inner, salt, person), key) == want
    # the expansion with depth 1 instead of 0 is expressible in hashlib: same construction, one parameter byte apart
    seed, msg = bytes(range(64)), (5).to_bytes(8, "little")
    root = hashlib.blake2b(msg, digest_size=64, key=seed, node_offset=4096 << 32).digest()
    assert root == B.blake2b(msg, B.param_block(64, 64, 1, 1, 0, 0, 4096, 0, 0), seed)
    for i in (0, 1, 63):
        want = hashlib.blake2b(root, digest_size=64, fanout=0, depth=1, leaf_size=64, node_offset=i | (4096 << 32), inner_size=64).digest()
        assert want == B.blake2b(root, B.param_block(64, 0, 0, 1, 64, i, 4096, 0, 64))


def test_blake2xb_generator_code_matches_model(emu):
    """apsu_amd/csrc/blake2x.h (the code the mask kernel runs) against the Python model of SEAL's Blake2xbPRNG:
    buffer boundaries (1024 values per 4096-byte buffer), arbitrary starting offsets"""
    from oracle import blake2x as B
    u32p = C.POINTER(C.c_uint32)
    for seed in ([0] * 8, list(range(1, 9)), [(0x9e3779b97f4a7c15 * (i + 1)) & ((1 << 64) - 1) for i in range(8)]):
        sd = np.array(seed, dtype=np.uint64)
        model = B.Blake2xbPRNG(seed).values(2 * 1024 + 40)
        for first, count in ((0, 2088), (5, 30), (1023, 3), (1024, 16), (2040, 48), (17, 1)):
            out = np.zeros(count, dtype=np.uint32)
            assert emu.emu_blake2xb_values(sd.ctypes.data_as(u64p), C.c_uint64(first), out.ctypes.data_as(u32p), count) == 0
            assert [int(v) for v in out] == model[first:first + count], (seed, first)


def test_tensor_fold128_reduction(emu):
    """ntt_reduce128_fold: P mod q for every 128-bit P < 2^(2k+1), the range of a0*b1 + a1*b0 with canonical operands;
    all primes of the 36 parameter files that admit it (44..61 bits, c < 2^24), against Python integers, extremes included"""
    primes = set()
    for name in ALL_PARAM_FILES:
        Cx = ref.RefContext.from_params(ref.load_params(common.param_json(name)))
        primes.update(Cx.q)
        primes.update([Cx.m_sk, Cx.gamma] + list(Cx.B))
    rng = np.random.default_rng(12)
    used = 0
    for q in sorted(primes):
        k = q.bit_length()
        ps = [0, 1, q - 1, q, (q - 1) * (q - 1), 2 * (q - 1) * (q - 1), (1 << (2 * k + 1)) - 1, (1 << (2 * k)) - 1, 1 << (2 * k),
              (1 << k) - 1, 1 << k, ((1 << (k + 1)) - 1) << k, (1 << 64) - 1, 1 << 64, ((1 << 32) - 1) << k, ((1 << (k + 1)) - 1) << k | ((1 << k) - 1)]
        for _ in range(300):
            a, b, c, d = (int(v) % q for v in rng.integers(0, 1 << 63, 4, dtype=np.uint64))
            ps += [a * b, a * b + c * d]
        hi = np.array([v >> 64 for v in ps], dtype=np.uint64); lo = np.array([v & ((1 << 64) - 1) for v in ps], dtype=np.uint64)
        out = np.zeros_like(lo)
        fk = emu.emu_reduce128(C.c_uint64(q), hi.ctypes.data_as(u64p), lo.ctypes.data_as(u64p), out.ctypes.data_as(u64p), len(ps))
        if k < 44:
            assert fk == 0
            continue
        assert fk == k, hex(q)
        used += 1
        assert [int(v) for v in out] == [v % q for v in ps], hex(q)
    assert used >= 10


@pytest.mark.parametrize("n,bits", [(64, 50), (256, 60), (1024, 56), (4096, 48), (8192, 56), (8192, 60), (16384, 58)])
def test_intt_tensor_loader_emulation(emu, n, bits):
    """k_intt_tensor's first pass: the inverse transform that forms x0*y0 (+ x1*y1) mod q while it loads equals the
    inverse transform of the products (both through the kernel's own pass functions, stepped on the CPU)"""
    logn = n.bit_length() - 1
    c = ref.RefContext(n, [bits], 65537 if (65537 - 1) % (2 * n) == 0 else 0, 0 if (65537 - 1) % (2 * n) == 0 else 20)
    q = c.q[0]
    rng = np.random.default_rng(n + bits)
    x0, y0, x1, y1 = (rng.integers(0, q, n, dtype=np.uint64) for _ in range(4))
    x0[:4] = [0, q - 1, q - 1, 1]; y0[:4] = [q - 1, q - 1, 0, 1]; x1[:2] = [q - 1, q - 1]; y1[:2] = [q - 1, q - 1]
    T = 1024 if n == 16384 else (512 if n == 8192 else 64)
    for cross in (False, True):
        out = np.zeros(n, dtype=np.uint64)
        null = C.POINTER(C.c_uint64)()
        rc = emu.emu_intt_tensor_limb(logn, C.c_uint64(q), x0.ctypes.data_as(u64p), y0.ctypes.data_as(u64p),
                                      x1.ctypes.data_as(u64p) if cross else null, y1.ctypes.data_as(u64p) if cross else null,
                                      out.ctypes.data_as(u64p), T)
        assert rc == 0, emu.emu_last_error()
        prod = np.array([(int(a) * int(b) + (int(u) * int(v) if cross else 0)) % q for a, b, u, v in zip(x0, y0, x1, y1)], dtype=np.uint64)
        assert emu.emu_ntt_limb(logn, 1, C.c_uint64(q), prod.ctypes.data_as(u64p), T) == 0
        assert (out == prod).all()


def _shipped_primes(n):
    """every coefficient prime of the shipped parameter files with this ring size"""
    out = set()
    for name in ALL_PARAM_FILES:
        js = json.load(open(os.path.join(common.PARAM_DIR, name + ".json")))
        sp = js["seal_params"]
        if sp["poly_modulus_degree"] != n:
            continue
        c = ref.RefContext(n, sp["coeff_modulus_bits"], sp.get("plain_modulus", 0), sp.get("plain_modulus_bits", 0))
        out.update(int(q) for q in c.q)
    return sorted(out)


@pytest.mark.parametrize("n,bits", [(4096, 48), (4096, 36), (8192, 50), (8192, 56), (8192, 57), (8192, 58), (8192, 60), (16384, 56), (4096, 0), (8192, 0), (4096, 61), (8192, 61)])
def test_intt_tensor_lazy_input_and_product_free_butterflies(emu, n, bits):
    """round 6: the inverse transform's first pass runs its psi^0 butterflies without a product (narrow moduli); with the tensor fold's last
    word as input the constant they add is ntt_lazy_bound_q(q) times larger -- NOT 4: for a 50-bit prime 2^50 - c with c ~ 2^20 the fold's
    last word reaches ~6q (the full 256M-4096 workload caught a first version that assumed 4q; 0x3ffffffef4001 is that prime) -- and the
    bound check of ntt_lazy_input_ok decides whether the modulus takes lazy input at all.  Lazy and canonical input give the same
    canonical output, in both forms of the workgroup; bits = 0: every coefficient prime of the shipped parameter files of that ring size"""
    logn = n.bit_length() - 1
    emu.emu_intt_tensor_limb_c.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p, u64p, u64p, C.c_int, C.c_int]
    if bits == 61:                                                # the BEHZ base's range mode (wide-near): stage 1 only, behind the tensor loader
        def is_prime(x):
            return all(pow(w, x - 1, x) == 1 for w in (2, 3, 5, 7, 11, 13))
        q, found, primes = ((1 << 61) - 1) // (2 * n) * (2 * n) + 1, 0, []
        while len(primes) < 2:                                    # (the first hits of the scan are the context's own auxiliary primes)
            if is_prime(q):
                found += 1
                if found >= 8:
                    primes.append(q)
            q -= 2 * n
    elif bits:
        primes = [ref.RefContext(n, [bits], 65537 if (65537 - 1) % (2 * n) == 0 else 0, 0 if (65537 - 1) % (2 * n) == 0 else 20).q[0]]
    else:
        primes = _shipped_primes(n)
        assert len(primes) >= 4
        if n == 8192:
            assert 0x3ffffffef4001 in primes
    lazy_runs = 0
    for q in primes:
        q = int(q)
        rng = np.random.default_rng(q % 1000003)
        x0, y0, x1, y1 = (rng.integers(0, q, n, dtype=np.uint64) for _ in range(4))
        for a in (x0, y0, x1, y1):
            a[:32] = q - 1                                        # a whole first-pass group of extreme inputs
        forms = [16] + ([8] if n in (4096, 8192) else [])
        ref_out = None
        refused = 0
        usable = True
        for co in forms:
            for lazy in (0, 0x100):
                out = np.zeros(n, dtype=np.uint64)
                rc = emu.emu_intt_tensor_limb_c(logn, q, x0.ctypes.data_as(u64p), y0.ctypes.data_as(u64p), x1.ctypes.data_as(u64p), y1.ctypes.data_as(u64p),
                                                out.ctypes.data_as(u64p), n // co, co | lazy)
                if rc == -2:
                    usable = False                                # no fold reduction for this modulus: the engine keeps the separate tensor kernel
                    break
                if rc == -3:
                    assert lazy
                    refused += 1
                    continue
                assert rc == 0, emu.emu_last_error()
                assert int(out.max()) < q
                lazy_runs += 1 if lazy else 0
                if ref_out is None:
                    ref_out = out
                assert (out == ref_out).all(), (hex(q), co, lazy)
            if not usable:
                break
        if not usable:
            continue
        prod = np.array([(int(a) * int(b) + int(u) * int(v)) % q for a, b, u, v in zip(x0, y0, x1, y1)], dtype=np.uint64)
        assert emu.emu_ntt_limb(logn, 1, C.c_uint64(q), prod.ctypes.data_as(u64p), n // 16) == 0
        assert (ref_out == prod).all(), hex(q)
        if bits == 58 and n == 8192:
            assert refused > 0                                    # a 58-bit narrow prime has no room for lazy input under the doubled bounds
    assert lazy_runs > 0 or bits in (36, 58)


def test_ntt_final_reduction_fold_and_barrett(emu):
    """ntt_reduce_any: x mod q for ANY 64-bit x.  Primes of the shape 2^k - c (all of SEAL's coefficient and BEHZ primes
    of 33 bits and more) take the one-multiply fold, the others Barrett; both against Python integers, extremes included"""
    primes = set()
    for name in ALL_PARAM_FILES:
        Cx = ref.RefContext.from_params(ref.load_params(common.param_json(name)))
        primes.update(Cx.q)
        primes.update([Cx.m_sk, Cx.gamma] + list(Cx.B))
        primes.add(Cx.t)
    rng = np.random.default_rng(11)
    folded = 0
    for q in sorted(primes):
        k = q.bit_length()
        xs = [0, 1, q - 1, q, q + 1, 2 * q - 1, 2 * q, (1 << 64) - 1, (1 << 63), (1 << k) - 1, 1 << k, (1 << 64) - q, (1 << 32) - 1, 1 << 32]
        xs += [int(v) for v in rng.integers(0, 1 << 63, 200, dtype=np.uint64)]
        xs += [int(v) | (1 << 63) for v in rng.integers(0, 1 << 63, 200, dtype=np.uint64)]
        x = np.array([v & ((1 << 64) - 1) for v in xs], dtype=np.uint64)
        out = np.zeros_like(x)
        fk = emu.emu_reduce_any(C.c_uint64(q), x.ctypes.data_as(u64p), out.ctypes.data_as(u64p), len(x))
        assert [int(v) for v in out] == [int(v) % q for v in x], hex(q)
        if fk:
            folded += 1
            assert fk == k
    assert folded >= 10                                           # the 48..61-bit primes do take the fold
    assert emu.emu_reduce_any(C.c_uint64(0xfffffffff70001), x.ctypes.data_as(u64p), out.ctypes.data_as(u64p), 1) == 56
    assert emu.emu_reduce_any(C.c_uint64(0x3e4001), x.ctypes.data_as(u64p), out.ctypes.data_as(u64p), 1) == 0


def test_database_file_reader_refuses_malformed_files_without_a_gpu(tmp_path):
    """apsu_he_db_file_open only maps and validates (header, table checksum, offsets inside the file): no device involved"""
    import struct
    import apsu_amd

    def opened(data, name):
        p = tmp_path / name
        p.write_bytes(data)
        return apsu_amd.DbFile(str(p))

    def fnv(b):
        h = 1469598103934665603
        for x in b:
            h = ((h ^ x) * 1099511628211) & (2**64 - 1)
        return h

    def image(count, entries, total):
        table = b"".join(struct.pack("<4I2Q", *e) for e in entries)
        hd = b"APSUHED1" + struct.pack("<4Q", 256, 256, count, total) + bytes(88 + 16) + struct.pack("<Q", fnv(table))
        body = hd + bytes(256 - len(hd)) + table
        return body + bytes(total - len(body))

    ok = image(2, [(0, 0, 3, 0, 4096, 1000), (1, 0, 5, 0, 8192, 4096)], 12288)
    f = opened(ok, "ok")
    assert len(f) == 2 and f.entry(1) == (1, 0, 5, 4096) and f.file_bytes == 12288
    f.close()
    for name, data in (("short", ok[:100]), ("magic", b"X" + ok[1:]), ("truncated", ok[:8192]),
                       ("table", ok[:256 + 8] + b"\x07" + ok[256 + 9:]),
                       ("outside", image(1, [(0, 0, 3, 0, 8192, 4097)], 12288)),
                       ("unaligned", image(1, [(0, 0, 3, 0, 4100, 16)], 12288)),
                       ("in-table", image(1, [(0, 0, 3, 0, 0, 16)], 12288)),
                       ("count", image(2**40, [], 4096))):
        with pytest.raises(ValueError):
            opened(data, name)
    with pytest.raises(apsu_amd.ApsuHeError):
        apsu_amd.DbFile(str(tmp_path / "missing"))
