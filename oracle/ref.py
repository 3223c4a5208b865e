"""ctypes binding of the CPU oracle (oracle/libapsu_he_ref.so).

ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (SEAL absent; see ref_core.h header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    so = os.path.join(_HERE, "libapsu_he_ref.so")
    srcs = [os.path.join(_HERE, f) for f in ("ref_core.c", "ref_path.c", "ref_harness.c", "ref_core.h", "ref_path.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libapsu_he_ref.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libapsu_he_ref.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        L = _LIB
        L.ref_ctx_create.restype = C.c_void_p
        L.ref_ctx_create.argtypes = [C.c_int, u64p, C.c_int, C.c_uint64]
        L.ref_ctx_create_bits.restype = C.c_void_p
        L.ref_ctx_create_bits.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_uint64, C.c_int]
        L.ref_ctx_destroy.argtypes = [C.c_void_p]
        L.ref_ctx_info.argtypes = [C.c_void_p, u64p, C.c_int]
        L.ref_rng_below.restype = C.c_uint64
        L.ref_mulmod.restype = C.c_uint64
        L.ref_minimal_primitive_root.restype = C.c_uint64
    return _LIB


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _vp(a):
    return C.c_void_p(a.ctypes.data)


def load_params(path_or_json):
    """PSUParams JSON (common/apsu/psu_params.cpp:290-374) -> dict of the fields the path uses."""
    if os.path.exists(path_or_json):
        with open(path_or_json) as f:
            j = json.load(f)
    else:
        j = json.loads(path_or_json)
    sp = j["seal_params"]
    out = dict(
        n=int(sp["poly_modulus_degree"]),
        coeff_bits=[int(b) for b in sp["coeff_modulus_bits"]],
        plain_modulus=int(sp.get("plain_modulus", 0)),
        plain_bits=int(sp.get("plain_modulus_bits", 0)),
        ps_low_degree=int(j["query_params"]["ps_low_degree"]),
        query_powers=sorted(int(p) for p in j["query_params"]["query_powers"]),
        max_items_per_bin=int(j["table_params"]["max_items_per_bin"]),
        table_size=int(j["table_params"]["table_size"]),
        hash_func_count=int(j["table_params"]["hash_func_count"]),
        felts_per_item=int(j["item_params"]["felts_per_item"]),
    )
    ipb = out["n"] // out["felts_per_item"]
    out["items_per_bundle"] = ipb
    out["bundle_idx_count"] = out["table_size"] // ipb
    return out


class RefContext:
    """CPU oracle context; mirrors the subset of CryptoContext + seal::Evaluator on the path."""

    def __init__(self, n, coeff_bits=None, plain_modulus=0, plain_bits=0, coeff_modulus=None):
        L = lib()
        if coeff_modulus is not None:
            arr = np.array(coeff_modulus, dtype=np.uint64)
            self.h = L.ref_ctx_create(n, _p(arr), len(arr), C.c_uint64(plain_modulus))
        else:
            bits = (C.c_int * len(coeff_bits))(*coeff_bits)
            self.h = L.ref_ctx_create_bits(n, bits, len(coeff_bits), C.c_uint64(plain_modulus), plain_bits)
        if not self.h:
            raise ValueError("invalid parameters")
        self.h = C.c_void_p(self.h)
        info = np.zeros(64, dtype=np.uint64)
        k = L.ref_ctx_info(self.h, _p(info), 64)
        info = [int(v) for v in info[:k]]
        self.n, self.K, self.first, self.t = info[0], info[1], info[2], info[3]
        self.q = info[4:4 + self.K]
        self.psi = info[4 + self.K:4 + 2 * self.K]
        rest = info[4 + 2 * self.K:]
        self.nB, self.m_sk, self.gamma = rest[0], rest[1], rest[2]
        self.B = rest[3:3 + self.nB]
        self.using_keyswitching = self.K > 1

    @classmethod
    def from_params(cls, p):
        return cls(p["n"], p["coeff_bits"], p["plain_modulus"], p["plain_bits"])

    def __del__(self):
        try:
            lib().ref_ctx_destroy(self.h)
        except Exception:
            pass

    # ---- level helpers (common/apsu/util/utils.cpp:179-189)
    def clamp(self, chain_idx):
        return min(chain_idx, self.first)

    def ct_shape(self, polys, chain_idx):
        return (polys, chain_idx + 1, self.n)

    # ---- evaluator ops (in place unless an out is returned)
    def transform_to_ntt(self, ct, chain_idx):
        lib().ref_transform_to_ntt(self.h, _p(ct), ct.shape[0], chain_idx)

    def transform_from_ntt(self, ct, chain_idx):
        lib().ref_transform_from_ntt(self.h, _p(ct), ct.shape[0], chain_idx)

    def multiply_plain_ntt(self, ct, pt_ntt, chain_idx):
        out = np.empty_like(ct)
        lib().ref_multiply_plain_ntt(self.h, _p(ct), _p(pt_ntt), _p(out), ct.shape[0], chain_idx)
        return out

    def plain_lift_ntt(self, pt, chain_idx):
        out = np.empty((chain_idx + 1, self.n), dtype=np.uint64)
        lib().ref_plain_lift_ntt(self.h, _p(pt), C.c_size_t(pt.size), _p(out), chain_idx)
        return out

    def multiply_plain_coeff(self, ct, pt, chain_idx):
        out = np.empty_like(ct)
        lib().ref_multiply_plain_coeff(self.h, _p(ct), _p(pt), C.c_size_t(pt.size), _p(out), ct.shape[0], chain_idx)
        return out

    def add(self, acc, x, chain_idx):
        lib().ref_add(self.h, _p(acc), _p(x), acc.shape[0], chain_idx)

    def add_plain(self, ct, pt, chain_idx):
        lib().ref_add_plain(self.h, _p(ct), _p(pt), C.c_size_t(pt.size), chain_idx)

    def multiply(self, a, b, chain_idx):
        out = np.empty(self.ct_shape(3, chain_idx), dtype=np.uint64)
        lib().ref_multiply(self.h, _p(a), _p(b), _p(out), chain_idx)
        return out

    def square(self, a, chain_idx):
        out = np.empty(self.ct_shape(3, chain_idx), dtype=np.uint64)
        lib().ref_square(self.h, _p(a), _p(out), chain_idx)
        return out

    def multiply_sized(self, a, b, chain_idx):
        """Evaluator::multiply of ciphertexts of any size (no relinearisation in between): size_a + size_b - 1 polynomials"""
        out = np.empty(self.ct_shape(a.shape[0] + b.shape[0] - 1, chain_idx), dtype=np.uint64)
        if lib().ref_multiply_sized(self.h, _p(a), a.shape[0], _p(b), b.shape[0], _p(out), chain_idx):
            raise ValueError("invalid size")              # Ciphertext::resize beyond SEAL_CIPHERTEXT_SIZE_MAX
        return out

    def relinearize(self, ct3, rk, chain_idx):
        """ct3 [3][L][n] -> returns size-2 ct (copy)."""
        work = np.ascontiguousarray(ct3.copy())
        lib().ref_relinearize(self.h, _p(work), _p(rk), chain_idx)
        return np.ascontiguousarray(work[:2])

    def mod_switch_to_next(self, ct, chain_idx):
        """returns the ct at chain_idx-1 (copy)."""
        work = np.ascontiguousarray(ct.copy())
        polys = ct.shape[0]
        lib().ref_mod_switch_to_next(self.h, _p(work), polys, chain_idx)
        return np.ascontiguousarray(work.reshape(-1)[: polys * chain_idx * self.n].reshape(polys, chain_idx, self.n))

    def clear_irrelevant_bits(self, ct):
        lib().ref_clear_irrelevant_bits(self.h, _p(ct), ct.shape[0])

    def irrelevant_bit_count(self):
        return lib().ref_irrelevant_bit_count(self.h)

    # ---- harness
    def keygen(self, seed):
        sk = np.empty((self.K, self.n), dtype=np.uint64)
        lib().ref_keygen(self.h, C.c_uint64(seed), _p(sk))
        return sk

    def encrypt(self, sk, pt, seed):
        ct = np.empty(self.ct_shape(2, self.first), dtype=np.uint64)
        lib().ref_encrypt_symmetric(self.h, _p(sk), _p(pt), C.c_uint64(seed), _p(ct))
        return ct

    def gen_relin_keys(self, sk, seed):
        rk = np.empty((self.K - 1, 2, self.K, self.n), dtype=np.uint64)
        lib().ref_gen_relin_keys(self.h, _p(sk), C.c_uint64(seed), _p(rk))
        return rk

    def decrypt(self, sk, ct, chain_idx):
        pt = np.empty(self.n, dtype=np.uint64)
        budget = lib().ref_decrypt(self.h, _p(sk), _p(np.ascontiguousarray(ct)), ct.shape[0], chain_idx, _p(pt))
        return pt, budget

    def encode(self, values):
        pt = np.empty(self.n, dtype=np.uint64)
        lib().ref_batch_encode(self.h, _p(np.ascontiguousarray(values, dtype=np.uint64)), _p(pt))
        return pt

    def decode(self, pt):
        v = np.empty(self.n, dtype=np.uint64)
        lib().ref_batch_decode(self.h, _p(np.ascontiguousarray(pt)), _p(v))
        return v

    def vec_to_oc_block(self, values, felts_per_item):
        """-> [items][2] (low, high): receiver_osn.cpp:53-73 applied to consecutive groups of felts_per_item slot values"""
        values = np.ascontiguousarray(values, dtype=np.uint64)
        items = values.size // felts_per_item
        out = np.empty((items, 2), dtype=np.uint64)
        for i in range(items):
            lib().ref_vec_to_oc_block(_p(values[i * felts_per_item:]), C.c_size_t(felts_per_item), C.c_uint64(self.t), _p(out[i]))
        return out

    def algebraize_items(self, items, felts_per_item):
        """items [count][16] uint8 -> [count][felts_per_item] (util::algebraize_item with item_bit_count = felts * (bits(t) - 1))"""
        items = np.ascontiguousarray(items, dtype=np.uint8).reshape(-1, 16)
        bits = felts_per_item * (int(self.t).bit_length() - 1)
        out = np.zeros((items.shape[0], felts_per_item), dtype=np.uint64)
        for i in range(items.shape[0]):
            k = lib().ref_algebraize_item(items[i].ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_uint32(bits), C.c_uint64(self.t), _p(out[i]))
            assert k == felts_per_item
        return out

    def polyn_with_roots(self, roots):
        roots = np.ascontiguousarray(roots, dtype=np.uint64)
        out = np.empty(roots.size + 1, dtype=np.uint64)
        lib().ref_polyn_with_roots(self.h, _p(roots), C.c_size_t(roots.size), _p(out))
        return out

    # ---- path drivers
    CT_SIZE_MAX = 16

    def power_sizes(self, dag_nodes):
        """{power: polynomials} after ComputePowers (2 with key switching; products keep growing without)"""
        nodes = np.array(dag_nodes, dtype=np.uint32)
        sizes = np.zeros(max(nd[0] for nd in dag_nodes) + 1, dtype=np.uint32)
        if lib().ref_power_sizes(self.h, _vp(nodes), len(dag_nodes), _vp(sizes)) < 0:
            raise ValueError("invalid size")
        return {nd[0]: int(sizes[nd[0]]) for nd in dag_nodes}

    def compute_powers(self, sources, dag_nodes, rk, ps_low_degree):
        """sources: {power: ct [2][first_L][n] coeff}.  Returns {power: ct} per receiver_osn.cpp:459-487."""
        firstL = self.first + 1
        max_p = max(nd[0] for nd in dag_nodes)
        cap = 3 if self.using_keyswitching else self.CT_SIZE_MAX
        bufs = {}
        ptrs = (u64p * (max_p + 1))()
        for nd in dag_nodes:
            b = np.zeros((cap, firstL, self.n), dtype=np.uint64)
            if nd[0] in sources:
                b[:2] = sources[nd[0]]
            bufs[nd[0]] = b
            ptrs[nd[0]] = _p(b)
        nodes = np.array(dag_nodes, dtype=np.uint32)
        sizes = np.zeros(max_p + 1, dtype=np.uint32)
        rkp = _p(rk) if rk is not None else None
        rc = lib().ref_compute_powers(self.h, ptrs, _vp(nodes), len(dag_nodes), rkp, C.c_uint32(ps_low_degree),
                                      None if self.using_keyswitching else _vp(sizes))
        if rc == -3:
            raise ValueError("invalid size")              # a product beyond SEAL_CIPHERTEXT_SIZE_MAX
        assert rc == 0
        high, low = self.clamp(1), self.clamp(2)
        out = {}
        for nd in dag_nodes:
            p = nd[0]
            lvl = high if (ps_low_degree == 0 or p > ps_low_degree) else low
            sz = 2 if self.using_keyswitching else int(sizes[p])
            out[p] = np.ascontiguousarray(bufs[p].reshape(-1)[: sz * (lvl + 1) * self.n].reshape(sz, lvl + 1, self.n))
        return out

    def _ptr_array(self, arrs):
        ptrs = (u64p * len(arrs))()
        for i, a in enumerate(arrs):
            ptrs[i] = _p(a) if a is not None else None
        return ptrs

    def _sizes_of(self, powers):
        """None with key switching (every power has size 2), else the sizes array ref_eval* take"""
        if self.using_keyswitching:
            return None
        return np.array([0 if a is None else a.shape[0] for a in powers], dtype=np.uint32)

    def eval(self, powers, coeffs, lvl, mask):
        """powers: list indexed by power (0 unused).  coeffs: list of arrays (layout per bin_bundle.cpp ctor)."""
        sizes = self._sizes_of(powers)
        out = np.empty((2 if sizes is None else self.CT_SIZE_MAX, 1, self.n), dtype=np.uint64)
        osz = C.c_uint32(0)
        rc = lib().ref_eval(self.h, self._ptr_array(powers), len(powers), self._ptr_array(coeffs), len(coeffs),
                            lvl, _p(mask), _p(out), None if sizes is None else _vp(sizes), C.byref(osz))
        if rc:
            raise ValueError("not enough ciphertext powers available")
        return np.ascontiguousarray(out[:osz.value])

    def eval_patstock(self, powers, coeffs, ps_low_degree, rk, mask):
        sizes = self._sizes_of(powers)
        out = np.empty((2 if sizes is None else self.CT_SIZE_MAX, 1, self.n), dtype=np.uint64)
        osz = C.c_uint32(0)
        rc = lib().ref_eval_patstock(self.h, self._ptr_array(powers), len(powers), self._ptr_array(coeffs),
                                     len(coeffs), C.c_uint32(ps_low_degree), _p(rk) if rk is not None else None, _p(mask), _p(out),
                                     None if sizes is None else _vp(sizes), C.byref(osz))
        if rc == -1:
            raise ValueError("not enough ciphertext powers available")
        if rc == -2:
            raise ValueError("ps_low_degree must be greater than 1 and less than the size of batched_coeffs")
        if rc == -3:
            raise ValueError("invalid size")              # a product beyond SEAL_CIPHERTEXT_SIZE_MAX
        return np.ascontiguousarray(out[:osz.value])

    def plain_chain_idx(self, ps_low_degree):
        return lib().ref_plain_chain_idx(self.h, C.c_uint32(ps_low_degree))


def set_threads(n):
    """threads used by the oracle's task-parallel drivers (the reference's `-t`, cli/base_clp.h)"""
    import ctypes.util
    try:
        omp = C.CDLL(ctypes.util.find_library("gomp") or "libgomp.so.1")
        omp.omp_set_num_threads(int(n))
    except OSError:
        pass


def create_powers_set(ps_low_degree, target_degree):
    cap = target_degree + 2
    out = np.zeros(cap, dtype=np.uint32)
    k = lib().ref_create_powers_set(C.c_uint32(ps_low_degree), C.c_uint32(target_degree), _vp(out), cap)
    if k < 0:
        raise ValueError("bad powers set arguments")
    return [int(v) for v in out[:k]]


def powers_dag(sources, targets):
    """-> (depth, [(power, depth, p1, p2), ...]) ascending by power (common/apsu/powers.cpp:22-107)."""
    s = np.array(sorted(sources), dtype=np.uint32)
    t = np.array(sorted(targets), dtype=np.uint32)
    nodes = np.zeros((len(t), 4), dtype=np.uint32)
    d = lib().ref_powers_dag_configure(_vp(s), len(s), _vp(t), len(t), _vp(nodes))
    if d < 0:
        raise ValueError("PowersDag configure failed")
    return d, [tuple(int(v) for v in row) for row in nodes]


def coeff_is_ntt(ps_low_degree, i):
    return bool(lib().ref_coeff_is_ntt(C.c_uint32(ps_low_degree), C.c_uint32(i)))


def fill_uniform(seed, bound, count):
    out = np.empty(count, dtype=np.uint64)
    lib().ref_fill_uniform(C.c_uint64(seed), C.c_uint64(bound), _p(out), C.c_size_t(count))
    return out
