"""Developer check run on the GPU box (`python tests/dev_parity_sweep.py`, not collected by pytest): every tier-1 op and
the tier-2 path against the oracle.  Prints one line per check and keeps going on mismatch (maximises information
per gpurun call).  Lives under tests/ because it uses the oracle."""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import apsu_amd
from oracle import ref
import common

fails = 0
def check(name, ok, extra=""):
    global fails
    print(("PASS " if ok else "FAIL ") + name + " " + extra, flush=True)
    if not ok: fails += 1

def rand_ct(C, rng, polys, lvl):
    return np.stack([np.stack([rng.integers(0, q, C.n, dtype=np.uint64) for q in C.q[:lvl+1]]) for _ in range(polys)])

def tier1(n, bits, t, pb):
    C = ref.RefContext(n, bits, t, pb)
    G = apsu_amd.HeContext(n=n, coeff_modulus=C.q, plain_modulus=C.t)
    rng = np.random.default_rng(3)
    tag = f"n={n} bits={bits}"
    rk = None
    if C.K > 1:
        rkh = np.stack([np.stack([np.stack([rng.integers(0,q,n,dtype=np.uint64) for q in C.q]) for _ in range(2)]) for _ in range(C.K-1)])
        rk = G.upload_relin_keys(rkh)
    for lvl in range(C.first, -1, -1):
        try:
            ct = rand_ct(C, rng, 2, lvl); ct2 = rand_ct(C, rng, 2, lvl)
            a = ct.copy(); C.transform_to_ntt(a, lvl); g = ct.copy(); G.transform_to_ntt_inplace(g, lvl)
            check(f"ntt_fwd {tag} lvl={lvl}", (a==g).all())
            G.transform_from_ntt_inplace(g, lvl); check(f"ntt_inv {tag} lvl={lvl}", (g==ct).all())
            pt = rng.integers(0, C.t, n, dtype=np.uint64)
            check(f"plain_to_ntt {tag} lvl={lvl}", (C.plain_lift_ntt(pt, lvl) == G.transform_plain_to_ntt(pt, lvl)).all())
            ptn = C.plain_lift_ntt(pt, lvl)
            check(f"multiply_plain_ntt {tag} lvl={lvl}", (C.multiply_plain_ntt(a, ptn, lvl) == G.multiply_plain_ntt(a, ptn, lvl)).all())
            check(f"multiply_plain {tag} lvl={lvl}", (C.multiply_plain_coeff(ct, pt, lvl) == G.multiply_plain(ct, pt, lvl)).all())
            mono = np.zeros(n, dtype=np.uint64); mono[3] = C.t - 2
            check(f"multiply_plain_mono {tag} lvl={lvl}", (C.multiply_plain_coeff(ct, mono, lvl) == G.multiply_plain(ct, mono, lvl)).all())
            x = ct.copy(); C.add(x, ct2, lvl); y = ct.copy(); G.add_inplace(y, ct2, lvl); check(f"add {tag} lvl={lvl}", (x==y).all())
            x = ct.copy(); C.add_plain(x, pt, lvl); y = ct.copy(); G.add_plain_inplace(y, pt, lvl); check(f"add_plain {tag} lvl={lvl}", (x==y).all())
            if lvl > 0:
                check(f"mod_switch {tag} lvl={lvl}", (C.mod_switch_to_next(ct, lvl) == G.mod_switch_to_next(ct, lvl)).all())
            check(f"multiply {tag} lvl={lvl}", (C.multiply(ct, ct2, lvl) == G.multiply(ct, ct2, lvl)).all())
            check(f"square {tag} lvl={lvl}", (C.square(ct, lvl) == G.square(ct, lvl)).all())
            z = np.zeros_like(ct)
            check(f"multiply_zero {tag} lvl={lvl}", (C.multiply(ct, z, lvl) == G.multiply(ct, z, lvl)).all())
            if rk is not None:
                ct3 = rand_ct(C, rng, 3, lvl)
                check(f"relinearize {tag} lvl={lvl}", (C.relinearize(ct3, rkh, lvl) == G.relinearize(ct3, rk, lvl)).all())
        except Exception as e:
            traceback.print_exc(); check(f"EXC {tag} lvl={lvl} {e}", False)
    last = rand_ct(C, rng, 2, 0); x = last.copy(); C.clear_irrelevant_bits(x); y = last.copy(); G.clear_irrelevant_bits(y)
    check(f"clear_bits {tag}", (x==y).all())
    G.close()

def tier2(name, js, degrees, seed=common.SEED0):
    t0 = time.time()
    S = common.make_scenario(js, degrees, seed)
    opw = common.oracle_powers(S)
    t1 = time.time()
    G = apsu_amd.HeContext(js)
    check(f"{name} dag", G.powers_dag() == S.nodes)
    rk = G.upload_relin_keys(S.rk) if S.rk is not None else None
    srcs = [[S.src[b][e] for e in S.sources] for b in S.bundle_indices]
    tg = time.time()
    pw = G.compute_powers(S.bundle_indices, srcs, rk)
    tg = time.time() - tg
    bad = []
    for b in S.bundle_indices:
        for p in S.targets:
            ct, ci, ntt = pw.download(b, p)
            if ct.shape != opw[b][p].shape or not (ct == opw[b][p]).all(): bad.append((b, p))
    check(f"{name} compute_powers ({len(S.targets)} targets x {len(S.bundle_indices)} idx, gpu {tg*1e3:.1f} ms, oracle+setup {t1-t0:.1f}s)", not bad, f"bad={bad[:8]}")
    gb = [G.upload_bundle(b["bundle_idx"], b["cache_idx"], b["coeffs"], b["flags"]) for b in S.bundles]
    tg = time.time()
    out = G.eval_bundles(gb, pw, rk, [b["mask"] for b in S.bundles])
    tg = time.time() - tg
    for i, b in enumerate(S.bundles):
        exp = common.oracle_eval(S, opw, b)
        ok = (out[i] == exp).all()
        sem, bud = common.check_semantics(S, b, out[i])
        check(f"{name} eval bundle idx={b['bundle_idx']} deg={b['degree']}", ok, f"semantic={sem} budget={bud}")
    print(f"   eval gpu {tg*1e3:.1f} ms for {len(gb)} bundles", flush=True)
    G.close()

if __name__ == "__main__":
    which = sys.argv[1:] or ["t1", "toy", "100K", "1M"]
    if "t1" in which:
        tier1(64, [40,40,40,36], 0, 17)
        tier1(256, [45,30,25], 0, 14)
        tier1(1024, [50,50], 0, 16)
        tier1(2048, [48], 65537, 0)
        tier1(4096, [48,36,25], 0, 18)
        tier1(8192, [56,56,56,50], 0, 22)
    if "toy" in which:
        tier2("toy", common.toy_json(), {0: [10, 3, 7, 11], 1: [11, 4]})
        tier2("toy-nops", common.toy_json(ps_low=0, max_items=6, query_powers=(1, 2, 3, 5)), {0: [6, 2], 1: [5]})
    if "100K" in which:
        tier2("100K-1", common.param_json("100K-1"), {0: [19, 7]})
    if "1M" in which:
        tier2("1M-1024-com", common.param_json("1M-1024-com"), {0: [124, 30], 1: [124]})
    if "16M" in which:
        tier2("16M-4096", common.param_json("16M-4096"), {0: [1303], 2: [170]})
    print("FAILS", fails)
    sys.exit(1 if fails else 0)
