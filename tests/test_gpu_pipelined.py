"""GPU tier: the queued-query scheduler against the oracle, deterministically.

bench.py's queued rate and every multi-query caller run ComputePowers of query k+1 next to the evaluation of query k
(apsu_he_set_async_results + apsu_he_set_query_overlap: three streams, up to three pooled powers buffers, event-chained).
The reference gets per-query isolation for free -- a fresh `all_powers` per Receiver::RunQuery
(receiver/apsu/receiver_osn.cpp:286-364) -- so the engine must give the bits of the one-query-at-a-time path whatever it
overlaps.  What these tests pin down, at BASELINE.json's sizes -- 16M-4096 (five full-degree BinBundles and a short one: an
evaluation lasts ~0.6 ms, longer than the host needs to queue the next query) and 256M-4096 (a BinBundle of degree 3999 and one
of degree 1000; its powers change level on their way out of the DAG, on the second stream when pipelined):

* K queued queries that ALTERNATE IRREGULARLY between two source sets and two mask sets (every lag 1, 2, 3 holds both an
  equal and a different pair of kinds, so a query that reads the powers, the workspace or the job tables of query k-1,
  k-2 or k-3 -- the pool cycles through up to three buffers -- produces wrong bits for some k), every query into its own
  output buffer, EVERY output compared with the oracle's result for its (sources, masks) pair;
* the pipelined walk forced (mode 3: `pipelined` == K, independent of timing), natural (mode 1), the split walk with the
  early high-power chain only (mode 2), everything serialised (mode 0), one-stream ComputePowers, and the evaluation's
  side lane on and off (APSU_HE_EVAL_SIDE);
* a query whose powers are computed and DROPPED without an evaluation in front of the next one (the pooled buffer's
  writers may still be queued on either stream: round 4's advisor finding);
* a caller that HOLDS the previous query's powers across the next ComputePowers.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import apsu_amd
import common
from oracle import ref

pytestmark = pytest.mark.gpu

# (parameter set, bundle index of every BinBundle of the test, degrees: full-degree BinBundles of max_items_per_bin - 1 plus a short one).
# 16M-4096: the low powers stay at the first data level (gathered forward transform straight out of the DAG's slots); 256M-4096: four
# data primes, the low powers are switched down one level and the high ones two on the way out (modulus-switch kernels on the second
# stream), 322 target powers per bundle index
WORLDS = [("16M-4096", 1, lambda D: [D] * 5 + [170]), ("256M-4096", 2, lambda D: [D, 1000])]
KINDS = [0, 1, 1, 0, 0, 0, 1, 0, 1, 1, 0, 1]      # source set of query k   (lags 1, 2, 3: equal and different pairs)
MASKS = [0, 0, 1, 0, 1, 1, 0, 1, 1, 0, 0, 1]      # mask set of query k
K = len(KINDS)
SEED = 0x41505355 + 0x5150


def _bundle_seed(ci):
    return SEED + 7919 * ci


@pytest.fixture(scope="module", params=WORLDS, ids=[w[0] for w in WORLDS])
def world(request):
    """sources A / B, masks M0 / M1, relin keys, and the ORACLE's results for every (sources, masks) pair and BinBundle"""
    from bench import splitmix_values
    CFG, IDX, degrees_of = request.param
    js = common.param_json(CFG)
    p = ref.load_params(js)
    C = ref.RefContext.from_params(p)
    n, t = C.n, C.t
    ps = p["ps_low_degree"]
    D = p["max_items_per_bin"] - 1
    degrees = degrees_of(D)
    targets = ref.create_powers_set(ps, p["max_items_per_bin"])
    _, nodes = ref.powers_dag(p["query_powers"], targets)
    sources = sorted(p["query_powers"])
    ns = len(sources)
    q = [int(v) for v in C.q]
    Lf = C.first + 1
    K_ = C.K
    rng = np.random.default_rng(SEED)
    src = [np.stack([np.stack([np.stack([rng.integers(0, q[j], n, dtype=np.uint64) for j in range(Lf)]) for _ in range(2)])
                     for _ in range(ns)]) for _ in range(2)]                                   # [kind][source][2][Lf][n]
    rkh = np.stack([np.stack([np.stack([rng.integers(0, q[j], n, dtype=np.uint64) for j in range(K_)]) for _ in range(2)])
                    for _ in range(K_ - 1)])
    masks = rng.integers(0, t, (2, len(degrees), n), dtype=np.uint64)                          # [mask set][BinBundle][n]
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    pci = C.plain_chain_idx(ps)
    pool = ThreadPoolExecutor(threads)                                                         # ctypes calls release the GIL

    def coeff(a):
        seed, d = a
        raw = splitmix_values(seed, d, n, t)
        return C.plain_lift_ntt(raw, pci) if ref.coeff_is_ntt(ps, d) else raw

    ref.set_threads(1)
    db = [list(pool.map(coeff, [(_bundle_seed(ci), d) for d in range(deg + 1)], chunksize=16)) for ci, deg in enumerate(degrees)]
    opw, plist = [], []
    for kind in range(2):
        ref.set_threads(threads)
        pw = C.compute_powers({e: np.ascontiguousarray(src[kind][s]) for s, e in enumerate(sources)}, nodes, rkh, ps)
        ref.set_threads(1)
        opw.append(pw)
        pl = [None] * (p["max_items_per_bin"] + 1)
        for kk, v in pw.items():
            pl[kk] = v
        plist.append(pl)

    def ev(a):
        kind, mk, ci = a
        mask = np.ascontiguousarray(masks[mk, ci])
        deg = degrees[ci]
        if ps > 1 and ps < deg:
            return C.eval_patstock(plist[kind], db[ci], ps, rkh, mask)
        return C.eval(plist[kind], db[ci], plist[kind][1].shape[1] - 1, mask)

    combos = [(kind, mk, ci) for kind in range(2) for mk in range(2) for ci in range(len(degrees))]
    want = dict(zip(combos, pool.map(ev, combos)))
    pool.shutdown()

    class W:
        pass
    w = W()
    w.js, w.n, w.t, w.ns, w.Lf, w.degrees, w.targets, w.idx, w.cfg = js, n, t, ns, Lf, degrees, targets, IDX, CFG
    w.src, w.rkh, w.masks, w.want, w.opw = src, rkh, masks, want, opw
    # the two kinds must not agree by accident anywhere a hazard could hide
    assert (want[(0, 0, 0)] != want[(1, 0, 0)]).mean() > 0.99 and (want[(0, 0, 0)][0] != want[(0, 1, 0)][0]).mean() > 0.99    # (a mask changes c0 only)
    return w


class Rig:
    """one context with the six BinBundles generated on the GPU and both source / mask sets resident in HBM"""

    def __init__(self, w):
        import torch
        self.torch, self.w = torch, w
        G = self.G = apsu_amd.HeContext(w.js)
        self.rk = G.upload_relin_keys(w.rkh)
        self.bundles = [G.random_bundle(w.idx, ci, deg, _bundle_seed(ci)) for ci, deg in enumerate(w.degrees)]
        self.src_d = [torch.from_numpy(s.view(np.int64)).cuda() for s in w.src]
        words = 2 * w.Lf * w.n
        self.src_ptrs = [[[sd.data_ptr() + s * words * 8 for s in range(w.ns)]] for sd in self.src_d]
        self.mask_d = torch.from_numpy(w.masks.view(np.int64)).cuda()
        nb = len(w.degrees)
        self.mask_ptrs = [[self.mask_d.data_ptr() + ((mk * nb + ci) * w.n) * 8 for ci in range(nb)] for mk in range(2)]
        torch.cuda.synchronize()                           # the overlap promise: inputs complete before any compute_powers

    def out_buffers(self, count):
        return [self.torch.full((len(self.w.degrees), 2, self.w.n), -1, dtype=self.torch.int64, device="cuda") for _ in range(count)]

    def query(self, kind, mk, out, hold=None):
        pw = self.G.compute_powers([self.w.idx], self.src_ptrs[kind], self.rk, on_device=True)
        self.G.eval_bundles(self.bundles, pw, self.rk, self.mask_ptrs[mk], out=out.data_ptr(), masks_on_device=True, out_on_device=True)
        return pw

    def check(self, out, kind, mk, what):
        got = out.cpu().numpy().view(np.uint64)
        for ci in range(len(self.w.degrees)):
            exp = self.w.want[(kind, mk, ci)].reshape(2, self.w.n)
            assert (got[ci] == exp).all(), "%s: BinBundle %d differs from the oracle's result for sources %d / masks %d" % (what, ci, kind, mk)

    def close(self):
        self.G.close()


def _queue(rig, hold_previous=False):
    """K queries back to back, nothing waited for until the end; -> the output buffers"""
    outs = rig.out_buffers(K)
    held = None
    for k in range(K):
        pw = rig.query(KINDS[k], MASKS[k], outs[k])
        if hold_previous:
            held = pw                                       # the previous query's powers are released only now
        else:
            del pw                                          # recycled while its evaluation is still queued (RunQuery's pattern)
    del held
    rig.G.sync()
    return outs


@pytest.mark.parametrize("mode,side,two_stream,hold", [
    (3, "1", -1, False),      # every ComputePowers pipelined (forced), side lane as shipped
    (3, "0", -1, False),      # ... the evaluation's side work on the main stream
    (3, "1", -1, True),       # ... the caller holds the previous query's powers across the next call
    (1, "1", -1, False),      # the shipped policy: pipelined when the device is busy
    (2, "1", -1, False),      # split walk, high-power chain behind the last reader of its buffer only
    (2, "0", -1, True),
    (0, "1", -1, False),      # no overlap promise: the second stream waits for the main stream's queue
    (1, "1", 0, False),       # ComputePowers on one stream
])
def test_queued_queries_are_isolated(world, monkeypatch, mode, side, two_stream, hold):
    monkeypatch.setenv("APSU_HE_EVAL_SIDE", side)
    rig = Rig(world)
    G = rig.G
    G.set_async_results(True)
    G.set_query_overlap(mode)
    G.set_two_stream(two_stream)
    c0 = G.debug_counters()
    outs = _queue(rig, hold_previous=hold)
    c1 = G.debug_counters()
    piped = c1["pipelined"] - c0["pipelined"]
    if mode == 3:
        assert piped == K, "forced mode: every ComputePowers takes the pipelined walk"
    elif mode == 1 and two_stream != 0:
        # How MANY walks the shipped policy pipelines depends on the host's speed relative to the device's (measured: 8 of 12 on a context's
        # first, cold pass -- allocations, job-table uploads --, 11 of 12 from then on, at both sizes): a scheduling heuristic, not a
        # correctness property -- a slower or loaded host, a faster GPU or a profiler may lower it with the engine fully correct (round-5
        # advisor).  The count is bounded here and reported by bench.py (`pipelined_steps`); mode 3 above pins the walk itself.
        assert 0 <= piped <= K
    else:
        assert piped == 0
    for k in range(K):
        rig.check(outs[k], KINDS[k], MASKS[k], "query %d of %d (mode %d)" % (k, K, mode))
    # the queue reached a steady state: a second pass uploads no job table and allocates nothing
    if mode == 3 and not hold:
        c1 = G.debug_counters()
        outs2 = _queue(rig)
        c2 = G.debug_counters()
        assert c2["powers_alloc"] == c1["powers_alloc"] and c2["arena_grow"] == c1["arena_grow"]
        for k in (0, K // 2, K - 1):
            rig.check(outs2[k], KINDS[k], MASKS[k], "second pass, query %d" % k)
    rig.close()


@pytest.mark.parametrize("mode", [1, 3, 2])
def test_dropped_powers_do_not_leak_into_the_next_query(world, mode):
    """compute_powers(A) given back WITHOUT an evaluation, then compute_powers(B) + evaluation, with a long evaluation queued in
    front: the pooled buffer's writers (query A's chains) may still be queued on either stream when query B starts writing it"""
    rig = Rig(world)
    G = rig.G
    G.set_async_results(True)
    G.set_query_overlap(mode)
    outs = rig.out_buffers(8)
    pw = rig.query(0, 0, outs[0])                          # fills the pool with one evaluated buffer
    del pw
    for r in range(3):
        pw = rig.query(1, 1, outs[1 + 2 * r])              # a long evaluation in front ...
        del pw
        kind_drop, kind_keep = (0, 1) if r % 2 == 0 else (1, 0)
        dropped = G.compute_powers([world.idx], rig.src_ptrs[kind_drop], rig.rk, on_device=True)
        del dropped                                         # ... a query that is dropped before its evaluation ...
        if r == 1:                                          # (twice in a row)
            dropped = G.compute_powers([world.idx], rig.src_ptrs[kind_drop], rig.rk, on_device=True)
            del dropped
        pw = rig.query(kind_keep, r % 2, outs[2 + 2 * r])   # ... and the query whose result is checked
        del pw
    G.sync()
    rig.check(outs[0], 0, 0, "first query")
    for r in range(3):
        rig.check(outs[1 + 2 * r], 1, 1, "round %d, the evaluation in front" % r)
        rig.check(outs[2 + 2 * r], 1 if r % 2 == 0 else 0, r % 2, "round %d, the query behind a dropped one (mode %d)" % (r, mode))
    rig.close()


def test_powers_of_a_pipelined_query_match_the_oracle(world):
    """the target powers themselves, downloaded from a query that was computed on the second stream next to an evaluation"""
    rig = Rig(world)
    G = rig.G
    G.set_async_results(True)
    G.set_query_overlap(3)
    outs = rig.out_buffers(3)
    pw = rig.query(0, 0, outs[0]); del pw
    pw = rig.query(1, 1, outs[1]); del pw
    pw = rig.query(0, 1, outs[2])
    for power in world.targets:
        ct, _, _ = pw.download(world.idx, power)
        assert (ct == world.opw[0][power]).all(), "power %d" % power
    rig.check(outs[2], 0, 1, "third query")
    rig.check(outs[1], 1, 1, "second query")
    del pw
    rig.close()


@pytest.mark.parametrize("walk_seed", [2026, 7, 99])
def test_scheduler_fuzz_with_changing_policies(walk_seed):
    """150 queued queries at 1M-1024-com (two bundle indices) under a seeded random walk over everything a caller can change between
    queries: overlap mode 0-3, one / two streams, phase timers on / off, synchronous or queued results, powers freed / held / dropped
    without an evaluation, subsets of the BinBundles (job tables and workspace change shape), an occasional host wait.  Every output
    is compared with the oracle's result for its (sources, masks) pair."""
    import torch
    js = common.param_json("1M-1024-com")
    SA = common.make_scenario(js, {0: [124, 17, 60], 1: [99, 3, 124]}, seed=common.SEED0)
    SB = common.make_scenario(js, {0: [124, 17, 60], 1: [99, 3, 124]}, seed=common.SEED0 + 77)     # other query, other keys are NOT used: see below
    # one key set for both queries: query B is re-encrypted under A's secret (make_scenario derives x and the sources from the seed)
    C = SA.C
    SB.sk, SB.rk = SA.sk, SA.rk
    for b in SB.bundle_indices:
        for e in SB.sources:
            xe = np.array([pow(int(v), e, C.t) for v in SB.x[b]], dtype=np.uint64)
            SB.src[b][e] = C.encrypt(SA.sk, C.encode(xe), common.SEED0 + 5000 * (b + 1) + e)
    S = [SA, SB]
    opw = [common.oracle_powers(s) for s in S]
    bundles = SA.bundles                                           # the database is SA's; masks: SA's and SB's
    masks = [np.stack([b["mask"] for b in SA.bundles]), np.stack([b["mask"] for b in SB.bundles])]
    want = {}
    for kind in range(2):
        for mk in range(2):
            for i, b in enumerate(bundles):
                bb = dict(b, mask=masks[mk][i])
                want[(kind, mk, i)] = common.oracle_eval(S[kind], opw[kind], bb)
    G = apsu_amd.HeContext(js)
    rk = G.upload_relin_keys(SA.rk)
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in bundles]
    ns = len(SA.sources)
    src_d, ptrs = [], []
    for s in S:
        a = np.stack([np.stack([s.src[b][e] for e in s.sources]) for b in s.bundle_indices])
        d = torch.from_numpy(a.view(np.int64)).cuda()
        w = a[0, 0].size
        src_d.append(d)
        ptrs.append({b: [d.data_ptr() + ((bi * ns + i) * w) * 8 for i in range(ns)] for bi, b in enumerate(s.bundle_indices)})
    mask_d = [torch.from_numpy(m.view(np.int64)).cuda() for m in masks]
    torch.cuda.synchronize()
    rng = np.random.default_rng(walk_seed)
    pending, held = [], []
    G.set_async_results(True)
    for q in range(150):
        r = rng.random()
        if r < 0.25:
            G.set_query_overlap(int(rng.integers(0, 4)))
        elif r < 0.35:
            G.set_two_stream(int(rng.integers(-1, 2)))
        elif r < 0.40:
            G.phase_enable(bool(rng.integers(0, 2)))
        elif r < 0.45:
            G.set_async_results(bool(rng.integers(0, 2)))
        elif r < 0.50:
            G.sync()
        kind, mk = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        sub = sorted(int(v) for v in rng.choice(len(gb), size=int(rng.integers(1, len(gb) + 1)), replace=False))
        idx = sorted({bundles[i]["bundle_idx"] for i in sub})
        if rng.random() < 0.15:                                    # a query that is dropped before its evaluation
            dropped = G.compute_powers(idx, [ptrs[1 - kind][b] for b in idx], rk, on_device=True)
            del dropped
        pw = G.compute_powers(idx, [ptrs[kind][b] for b in idx], rk, on_device=True)
        out = torch.full((len(sub), 2, G.n), -1, dtype=torch.int64, device="cuda")
        G.eval_bundles([gb[i] for i in sub], pw, rk, [mask_d[mk].data_ptr() + i * G.n * 8 for i in sub], out=out.data_ptr(),
                       masks_on_device=True, out_on_device=True)
        pending.append((q, kind, mk, sub, out))
        if rng.random() < 0.3:
            held.append(pw)                                        # the caller keeps some powers alive for a while
            if len(held) > 3:
                held.pop(0)
        del pw
    G.sync()
    G.phase_enable(False)
    for q, kind, mk, sub, out in pending:
        got = out.cpu().numpy().view(np.uint64)
        for row, i in enumerate(sub):
            assert (got[row] == want[(kind, mk, i)].reshape(2, G.n)).all(), "query %d (sources %d, masks %d), BinBundle %d" % (q, kind, mk, i)
    G.close()
