"""GPU tier: tier 1 on device-resident operands (apsu_he_set_tier1_on_device, ABI 5).  A caller that replaces seal::Evaluator
methods one by one (receiver/apsu/receiver_osn.cpp:422-478, bin_bundle.cpp:143-170) keeps its ciphertexts in HBM and issues the
per-method calls with device pointers; nothing is copied to the host or waited for between calls.  The chain below -- multiply,
relinearize, mod_switch_to_next, transform_to_ntt, multiply_plain (NTT), transform_from_ntt, multiply_plain (coefficient form,
incl. SEAL's monomial case), add_plain, add -- must give the bits of the same chain run through host buffers, which the rest of
the suite holds against the oracle."""
import ctypes as C

import numpy as np
import pytest

import apsu_amd
from apsu_amd.engine import load_library
from oracle import ref

pytestmark = pytest.mark.gpu
u64p = C.POINTER(C.c_uint64)


def chain_host(G, a, b, rk, pt, mono, lvl):
    x = G.relinearize(G.multiply(a, b, lvl), rk, lvl)
    x = G.mod_switch_to_next(x, lvl)
    lvl -= 1
    G.transform_to_ntt_inplace(x, lvl)
    x = G.multiply_plain_ntt(x, G.transform_plain_to_ntt(pt, lvl), lvl)
    G.transform_from_ntt_inplace(x, lvl)
    y = G.multiply_plain(x, mono, lvl)
    z = G.multiply_plain(x, pt, lvl)
    G.add_plain_inplace(y, pt, lvl)
    G.add_inplace(y, z, lvl)
    return y


def test_tier1_chain_on_device_pointers():
    import torch
    n, bits = 4096, [48, 36, 25]
    Cx = ref.RefContext(n, bits, 0, 18)
    G = apsu_amd.HeContext(n=n, coeff_modulus=Cx.q, plain_modulus=Cx.t)
    rng = np.random.default_rng(4)
    lvl = Cx.first
    L = lvl + 1
    a = np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in Cx.q[:L]]) for _ in range(2)])
    b = np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in Cx.q[:L]]) for _ in range(2)])
    pt = rng.integers(0, Cx.t, n, dtype=np.uint64)
    mono = np.zeros(n, dtype=np.uint64)
    mono[7] = Cx.t - 2
    rkh = np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in Cx.q]) for _ in range(2)]) for _ in range(Cx.K - 1)])
    rk = G.upload_relin_keys(rkh)
    want = chain_host(G, a, b, rk, pt, mono, lvl)

    lib = load_library()
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).cuda()
    p = lambda t: C.cast(C.c_void_p(t.data_ptr()), u64p)
    da, db, dpt, dmono = dev(a), dev(b), dev(pt), dev(mono)
    d3 = torch.zeros((3, L, n), dtype=torch.int64, device="cuda")
    L2 = L - 1
    dptn = torch.zeros((L2, n), dtype=torch.int64, device="cuda")
    dx = torch.zeros((2, L2, n), dtype=torch.int64, device="cuda")
    dy = torch.zeros((2, L2, n), dtype=torch.int64, device="cuda")
    dz = torch.zeros((2, L2, n), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    G.set_tier1_on_device(True)
    ok = lambda rc: apsu_amd.engine._check(rc)
    ok(lib.apsu_he_multiply(G.h, p(da), p(db), p(d3), lvl))
    ok(lib.apsu_he_relinearize(G.h, p(d3), rk.h, lvl))
    ok(lib.apsu_he_mod_switch_to_next(G.h, p(d3), 2, lvl))                    # result packed [2][L-1][n] at the front
    ok(lib.apsu_he_transform_to_ntt(G.h, p(d3), 2, lvl - 1))
    ok(lib.apsu_he_transform_plain_to_ntt(G.h, p(dpt), C.c_size_t(n), p(dptn), lvl - 1))
    ok(lib.apsu_he_multiply_plain_ntt(G.h, p(d3), p(dptn), p(dx), 2, lvl - 1))
    ok(lib.apsu_he_transform_from_ntt(G.h, p(dx), 2, lvl - 1))
    ok(lib.apsu_he_multiply_plain(G.h, p(dx), p(dmono), C.c_size_t(n), p(dy), 2, lvl - 1))
    ok(lib.apsu_he_multiply_plain(G.h, p(dx), p(dpt), C.c_size_t(n), p(dz), 2, lvl - 1))
    ok(lib.apsu_he_add_plain(G.h, p(dy), p(dpt), C.c_size_t(n), lvl - 1))
    ok(lib.apsu_he_add(G.h, p(dy), p(dz), 2, lvl - 1))
    G.sync()                                                                   # the one completion point of the chain
    got = dy.cpu().numpy().view(np.uint64)
    assert (got == want).all()
    # back to host operands: the same context serves both
    G.set_tier1_on_device(False)
    assert (chain_host(G, a, b, rk, pt, mono, lvl) == want).all()
    G.close()
