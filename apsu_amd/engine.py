"""ctypes binding of the engine's C ABI (include/apsu_he.h).

Names follow the reference interface this path replaces:
  seal::Evaluator::{transform_to_ntt_inplace, transform_from_ntt_inplace, multiply_plain,
  add_inplace, add_plain_inplace, multiply, square, relinearize_inplace,
  mod_switch_to_next_inplace}            receiver/apsu/receiver_osn.cpp:422-478, bin_bundle.cpp:143-357
  Receiver::ComputePowers                receiver/apsu/receiver_osn.cpp:395-488
  BatchedPlaintextPolyn::eval{,_patstock} receiver/apsu/bin_bundle.cpp:106-174,192-360
Errors: APSU_HE_INVALID_ARGUMENT -> ValueError (std::invalid_argument in the reference),
everything else -> ApsuHeError (std::runtime_error / std::logic_error).
"""
import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
u64p = C.POINTER(C.c_uint64)


class ApsuHeError(RuntimeError):
    pass


class _Info(C.Structure):
    _fields_ = [
        ("poly_modulus_degree", C.c_uint64), ("plain_modulus", C.c_uint64),
        ("coeff_modulus_size", C.c_int32), ("first_chain_idx", C.c_int32),
        ("using_keyswitching", C.c_int32), ("irrelevant_bit_count", C.c_int32),
        ("coeff_modulus", C.c_uint64 * 8),
        ("ps_low_degree", C.c_uint32), ("max_items_per_bin", C.c_uint32),
        ("bundle_idx_count", C.c_uint32), ("items_per_bundle", C.c_uint32),
        ("source_power_count", C.c_uint32), ("target_power_count", C.c_uint32),
        ("powers_dag_depth", C.c_uint32), ("result_polys", C.c_uint32),
    ]


def lib_path():
    return os.path.join(_HERE, "libapsu_he_gpu.so")


def _preload_shared_hip_runtime():
    """One process must hold ONE HIP runtime.  The PyTorch-ROCm wheel bundles its own libamdhip64
    (same SONAME as /opt/rocm's); whichever is loaded first wins.  If this library came first, a later
    `import torch` would bring a second runtime that sees no GPU ("No HIP GPUs are available"), and
    device pointers could not be shared with torch tensors.  So when torch is installed but not yet
    imported, load torch's copy first; the engine then binds to it through the SONAME."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """Loads libapsu_he_gpu.so; raises (never falls back) if it has not been built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ApsuHeError(
                "HIP extension %s is missing: build it with `make -C apsu_amd/csrc` "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % path)
        _preload_shared_hip_runtime()
        _LIB = C.CDLL(path)
        _LIB.apsu_he_last_error.restype = C.c_char_p
    return _LIB


def _check(rc):
    if rc == 0:
        return
    msg = load_library().apsu_he_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    raise ApsuHeError("apsu_he status %d: %s" % (rc, msg))


def _p(a):
    assert isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "need contiguous uint64"
    return a.ctypes.data_as(u64p)


def _ptr_array(items):
    """items: numpy arrays (host) or ints (device pointers) -> (const uint64_t* const*)"""
    arr = (C.c_void_p * len(items))()
    for i, it in enumerate(items):
        arr[i] = it.ctypes.data if isinstance(it, np.ndarray) else int(it)
    return arr


class RelinKeys:
    def __init__(self, ctx, handle):
        self._ctx, self.h = ctx, handle

    def __del__(self):
        try:
            if self.h:
                load_library().apsu_he_relin_free(self.h)
        except Exception:
            pass


class Bundle:
    """Device-resident BinBundleCache.batched_matching_polyn (receiver/apsu/bin_bundle.h:52-134)."""

    def __init__(self, ctx, handle, bundle_idx, cache_idx, degree):
        self._ctx, self.h = ctx, handle
        self.bundle_idx, self.cache_idx, self.degree = bundle_idx, cache_idx, degree

    @property
    def db_bytes(self):
        v = C.c_uint64()
        _check(load_library().apsu_he_bundle_bytes(self.h, C.byref(v)))
        return v.value

    def __del__(self):
        try:
            if self.h:
                load_library().apsu_he_bundle_free(self.h)
        except Exception:
            pass


class Powers:
    """Device-resident CiphertextPowers (receiver/apsu/receiver_osn.h:41) for some bundle indices."""

    def __init__(self, ctx, handle, bundle_indices):
        self._ctx, self.h, self.bundle_indices = ctx, handle, list(bundle_indices)

    def download(self, bundle_idx, power):
        """-> (ct [size][L][n], chain_idx, is_ntt) in the form receiver_osn.cpp:459-487 leaves it (size 2 with key switching)."""
        ctx = self._ctx
        sz = ctx.power_size(power)
        buf = np.empty(sz * (ctx.first_chain_idx + 1) * ctx.n, dtype=np.uint64)
        ci, ntt = C.c_int(), C.c_int()
        _check(load_library().apsu_he_powers_download(ctx.h, self.h, bundle_idx, power, _p(buf), C.c_size_t(buf.size),
                                                     C.byref(ci), C.byref(ntt)))
        L = ci.value + 1
        return buf[: sz * L * ctx.n].reshape(sz, L, ctx.n).copy(), ci.value, bool(ntt.value)

    def __del__(self):
        try:
            if self.h:
                load_library().apsu_he_powers_free(self.h)
        except Exception:
            pass


class HeContext:
    """CryptoContext + seal::Evaluator replacement (common/apsu/crypto_context.h:28-125)."""

    def __init__(self, psu_params_json=None, device=0, n=None, coeff_modulus=None, plain_modulus=None):
        L = load_library()
        h = C.c_void_p()
        if psu_params_json is not None:
            if os.path.exists(psu_params_json):
                with open(psu_params_json) as f:
                    psu_params_json = f.read()
            _check(L.apsu_he_create(psu_params_json.encode(), device, C.byref(h)))
            self.felts_per_item = int(json.loads(psu_params_json)["item_params"]["felts_per_item"])
        else:
            self.felts_per_item = 0
            q = np.array(coeff_modulus, dtype=np.uint64)
            _check(L.apsu_he_create_raw(C.c_uint64(n), _p(q), len(q), C.c_uint64(plain_modulus), device, C.byref(h)))
        self.h = h
        info = _Info()
        _check(L.apsu_he_get_info(self.h, C.byref(info)))
        self.info = info
        self.n = int(info.poly_modulus_degree)
        self.t = int(info.plain_modulus)
        self.K = int(info.coeff_modulus_size)
        self.first_chain_idx = int(info.first_chain_idx)
        self.q = [int(info.coeff_modulus[i]) for i in range(self.K)]
        self.ps_low_degree = int(info.ps_low_degree)
        self.max_items_per_bin = int(info.max_items_per_bin)
        self.bundle_idx_count = int(info.bundle_idx_count)
        self.source_power_count = int(info.source_power_count)
        self.irrelevant_bit_count = int(info.irrelevant_bit_count)
        self.result_polys = int(info.result_polys) or 2

    def close(self):
        if getattr(self, "h", None):
            load_library().apsu_he_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def powers_dag(self):
        cnt = C.c_int()
        L = load_library()
        _check(L.apsu_he_get_powers_dag(self.h, None, 0, C.byref(cnt)))
        nodes = np.zeros((cnt.value, 4), dtype=np.uint32)
        _check(L.apsu_he_get_powers_dag(self.h, C.c_void_p(nodes.ctypes.data), cnt.value, C.byref(cnt)))
        return [tuple(int(v) for v in r) for r in nodes]

    PROFILE_CLASSES = ("ntt_fwd", "ntt_inv", "dyadic_mac", "behz_ext", "behz_tensor", "behz_finish", "keyswitch",
                       "modswitch", "other", "ntt_fused")

    def phase_enable(self, on=True):
        """phase timers under the reference's STOPWATCH names (apsu_he_phase_*)"""
        _check(load_library().apsu_he_phase_enable(self.h, 1 if on else 0))

    def phase_read(self, reset=True):
        """-> {"Receiver::RunQuery": (count, avg_ms, min_ms, max_ms), "Receiver::ComputePowers": ..., "Receiver::ProcessBinBundleCache": ...}"""
        L = load_library()
        L.apsu_he_phase_name.restype = C.c_char_p
        out = {}
        for ph in range(3):
            cnt = C.c_uint64(); avg = C.c_double(); mn = C.c_double(); mx = C.c_double()
            _check(L.apsu_he_phase_read(self.h, ph, C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx), 0))
            out[L.apsu_he_phase_name(ph).decode()] = (int(cnt.value), avg.value, mn.value, mx.value)
        if reset:
            _check(L.apsu_he_phase_read(self.h, 0, None, None, None, None, 1))
        return out

    def seed_expand(self, chain_idx, seeds, dst_ptrs):
        """apsu_he_seed_expand: seeds [count][8] uint64, dst_ptrs: device pointers ([L][n] words each)"""
        sd = np.ascontiguousarray(seeds, dtype=np.uint64).reshape(-1, 8)
        _check(load_library().apsu_he_seed_expand(self.h, int(chain_idx), len(sd), _p(sd.reshape(-1)), _ptr_array([int(p) for p in dst_ptrs])))

    def compute_powers_cost(self):
        """ComputePowers for one bundle index in the partition rule's cost unit (apsu_he_compute_powers_cost)"""
        v = C.c_uint64()
        _check(load_library().apsu_he_compute_powers_cost(self.h, C.byref(v)))
        return int(v.value)

    COUNTERS = ("host_sync", "job_upload", "job_hit", "arena_grow", "powers_alloc", "stage_wrap", "job_realloc", "pipelined")

    def debug_counters(self):
        """host-side events inside the engine (monotonic): see apsu_he_debug_counters"""
        out = (C.c_uint64 * len(self.COUNTERS))()
        _check(load_library().apsu_he_debug_counters(self.h, out, len(self.COUNTERS)))
        return {k: int(out[i]) for i, k in enumerate(self.COUNTERS)}

    def profile_enable(self, mode=1):
        """0/False off, 1/True every kernel class, 2 NTT launches only (least intrusive)"""
        _check(load_library().apsu_he_profile_enable(self.h, int(mode)))

    def profile_read(self, reset=True):
        """-> {class: (ms, launches, units)} of device time measured with HIP events on the engine's stream."""
        k = len(self.PROFILE_CLASSES)
        ms = (C.c_double * k)()
        la = (C.c_uint64 * k)()
        un = (C.c_uint64 * k)()
        _check(load_library().apsu_he_profile_read(self.h, ms, la, un, k, 1 if reset else 0))
        return {name: (ms[i], int(la[i]), int(un[i])) for i, name in enumerate(self.PROFILE_CLASSES)}

    # ---- tier 1: Evaluator methods (in place on numpy arrays shaped [polys][L][n])
    def transform_to_ntt_inplace(self, ct, chain_idx):
        _check(load_library().apsu_he_transform_to_ntt(self.h, _p(ct), ct.shape[0], chain_idx))

    def transform_from_ntt_inplace(self, ct, chain_idx):
        _check(load_library().apsu_he_transform_from_ntt(self.h, _p(ct), ct.shape[0], chain_idx))

    def transform_plain_to_ntt(self, pt, chain_idx):
        out = np.empty((chain_idx + 1, self.n), dtype=np.uint64)
        _check(load_library().apsu_he_transform_plain_to_ntt(self.h, _p(pt), C.c_size_t(pt.size), _p(out), chain_idx))
        return out

    def multiply_plain_ntt(self, ct, pt_ntt, chain_idx):
        out = np.empty_like(ct)
        _check(load_library().apsu_he_multiply_plain_ntt(self.h, _p(ct), _p(pt_ntt), _p(out), ct.shape[0], chain_idx))
        return out

    def multiply_plain(self, ct, pt, chain_idx):
        out = np.empty_like(ct)
        _check(load_library().apsu_he_multiply_plain(self.h, _p(ct), _p(pt), C.c_size_t(pt.size), _p(out), ct.shape[0], chain_idx))
        return out

    def add_inplace(self, acc, x, chain_idx):
        _check(load_library().apsu_he_add(self.h, _p(acc), _p(x), acc.shape[0], chain_idx))

    def add_plain_inplace(self, ct, pt, chain_idx):
        _check(load_library().apsu_he_add_plain(self.h, _p(ct), _p(pt), C.c_size_t(pt.size), chain_idx))

    def multiply(self, a, b, chain_idx):
        out = np.empty((3, chain_idx + 1, self.n), dtype=np.uint64)
        _check(load_library().apsu_he_multiply(self.h, _p(a), _p(b), _p(out), chain_idx))
        return out

    def square(self, a, chain_idx):
        out = np.empty((3, chain_idx + 1, self.n), dtype=np.uint64)
        _check(load_library().apsu_he_square(self.h, _p(a), _p(out), chain_idx))
        return out

    def multiply_sized(self, a, b, chain_idx):
        """Evaluator::multiply of ciphertexts that were never relinearised: size_a + size_b - 1 polynomials"""
        out = np.empty((a.shape[0] + b.shape[0] - 1, chain_idx + 1, self.n), dtype=np.uint64)
        _check(load_library().apsu_he_multiply_sized(self.h, _p(a), a.shape[0], _p(b), b.shape[0], _p(out), chain_idx))
        return out

    def power_size(self, power):
        """polynomials of a target power after ComputePowers (2 with key switching)"""
        v = C.c_uint32()
        _check(load_library().apsu_he_power_size(self.h, C.c_uint32(power), C.byref(v)))
        return v.value

    def result_size(self, bundle):
        """polynomials of one BinBundle's result (2 with key switching)"""
        v = C.c_uint32()
        _check(load_library().apsu_he_bundle_result_size(self.h, bundle.h, C.byref(v)))
        return v.value

    def relinearize(self, ct3, rk, chain_idx):
        work = np.ascontiguousarray(ct3.copy())
        _check(load_library().apsu_he_relinearize(self.h, _p(work), rk.h, chain_idx))
        return np.ascontiguousarray(work[:2])

    def mod_switch_to_next(self, ct, chain_idx):
        work = np.ascontiguousarray(ct.copy())
        polys = ct.shape[0]
        _check(load_library().apsu_he_mod_switch_to_next(self.h, _p(work), polys, chain_idx))
        return np.ascontiguousarray(work.reshape(-1)[: polys * chain_idx * self.n].reshape(polys, chain_idx, self.n))

    def clear_irrelevant_bits(self, ct):
        _check(load_library().apsu_he_clear_irrelevant_bits(self.h, _p(ct), ct.shape[0]))

    # ---- tier 2
    def upload_relin_keys(self, rk):
        h = C.c_void_p()
        _check(load_library().apsu_he_relin_upload(self.h, _p(np.ascontiguousarray(rk)), C.byref(h)))
        return RelinKeys(self, h)

    def upload_bundle(self, bundle_idx, cache_idx, coeffs, is_ntt):
        """coeffs: list of uint64 arrays (batched_coeffs as Plaintext.data()); is_ntt: list of bool."""
        h = C.c_void_p()
        flags = (C.c_uint8 * len(coeffs))(*[1 if f else 0 for f in is_ntt])
        keep = [np.ascontiguousarray(c, dtype=np.uint64) for c in coeffs]
        _check(load_library().apsu_he_db_upload_bundle(self.h, bundle_idx, cache_idx, len(keep), _ptr_array(keep), flags,
                                                       C.byref(h)))
        return Bundle(self, h, bundle_idx, cache_idx, len(keep) - 1)

    def build_bundle(self, bundle_idx, cache_idx, bins):
        """BinBundle::regen_cache on the GPU.  bins: list (one per bin / slot) of lists of field elements."""
        nb = len(bins)
        stride = max([len(b) for b in bins] + [1])
        roots = np.zeros((max(nb, 1), stride), dtype=np.uint64)
        counts = np.zeros(max(nb, 1), dtype=np.uint32)
        for i, b in enumerate(bins):
            counts[i] = len(b)
            roots[i, :len(b)] = b
        h = C.c_void_p()
        _check(load_library().apsu_he_db_build_bundle(self.h, bundle_idx, cache_idx, _p(roots), C.c_void_p(counts.ctypes.data),
                                                      nb, stride, C.byref(h)))
        deg = C.c_uint32()
        _check(load_library().apsu_he_bundle_degree(h, C.byref(deg)))
        return Bundle(self, h, bundle_idx, cache_idx, deg.value)

    def algebraize_items(self, items):
        """util::algebraize_item for items [count][16] uint8 -> felts [count][felts_per_item] (db_encoding.cpp:209-256,360-366)"""
        items = np.ascontiguousarray(items, dtype=np.uint8).reshape(-1, 16)
        if not self.felts_per_item:
            raise ApsuHeError("context was created without PSUParams")
        out = np.empty((items.shape[0], self.felts_per_item), dtype=np.uint64)
        _check(load_library().apsu_he_algebraize_items(self.h, items.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(items.shape[0]), 0,
                                                       _p(out), 0))
        return out

    def save_db_file(self, path, bundles):
        """the whole DB (these BinBundles, in this order) into one mmap-able file (ReceiverDB::save counterpart)"""
        hs = (C.c_void_p * len(bundles))(*[b.h for b in bundles])
        _check(load_library().apsu_he_db_file_save(self.h, os.fsencode(path), hs, len(bundles)))

    def load_db_file(self, path, only=None):
        """-> [Bundle] of the file's BinBundles (all, or the table positions in `only`), loaded onto this context's device"""
        with DbFile(path) as f:
            return [f.load(self, i) for i in (range(len(f)) if only is None else only)]

    def save_bundle(self, bundle):
        """-> bytes: engine-native image of the BinBundle cache (ReceiverDB::save counterpart)"""
        size = C.c_uint64()
        _check(load_library().apsu_he_bundle_image_size(self.h, bundle.h, C.byref(size)))
        buf = np.empty(size.value, dtype=np.uint8)
        wr = C.c_uint64()
        _check(load_library().apsu_he_bundle_save(self.h, bundle.h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), size, C.byref(wr)))
        return buf[: wr.value]

    def load_bundle(self, image):
        """image: bytes / uint8 array / np.memmap produced by save_bundle (ReceiverDB::Load counterpart)"""
        arr = np.ascontiguousarray(np.frombuffer(image, dtype=np.uint8) if not isinstance(image, np.ndarray) else image)
        h = C.c_void_p()
        _check(load_library().apsu_he_bundle_load(self.h, arr.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint64(arr.size), C.byref(h)))
        deg = C.c_uint32()
        _check(load_library().apsu_he_bundle_degree(h, C.byref(deg)))
        return Bundle(self, h, -1, -1, deg.value)

    def set_two_stream(self, mode):
        """-1 default policy, 0 off, 1 on: ComputePowers' high-power chain on a second stream"""
        _check(load_library().apsu_he_set_two_stream(self.h, int(mode)))

    def set_async_results(self, on):
        """eval_bundles with device-resident masks and output returns once its work is queued; see sync() / stream"""
        _check(load_library().apsu_he_set_async_results(self.h, int(bool(on))))

    def set_query_overlap(self, mode):
        """apsu_he_set_query_overlap: the device-resident inputs of compute_powers are complete when it is called (not produced by
        work queued on the context's stream), so consecutive queued queries may overlap.  False / 0 off, True / 1 on, 2 on without
        the pipelined walk, 3 on with the pipelined walk forced (tests)"""
        _check(load_library().apsu_he_set_query_overlap(self.h, int(mode)))

    def set_tier1_on_device(self, on):
        """tier-1 calls take device pointers (ints) and only queue their work; see sync() / stream"""
        _check(load_library().apsu_he_set_tier1_on_device(self.h, int(bool(on))))

    def sync(self):
        """wait for everything this context has queued"""
        _check(load_library().apsu_he_sync(self.h))

    @property
    def stream(self):
        """the context's main HIP stream as an integer handle (e.g. for torch.cuda.ExternalStream)"""
        p = C.c_void_p()
        _check(load_library().apsu_he_stream(self.h, C.byref(p)))
        return int(p.value or 0)

    def mask_generate(self, seed, count, masks_dev, want_values=True, want_blocks=True):
        """N4: `count` random masks (receiver_osn.cpp:217-284).  masks_dev: device pointer to count*n words receiving the
        encoded plaintexts.  -> (values [count][n] or None, blocks [count][items_per_bundle][2] (low, high) or None)"""
        vals = np.empty((count, self.n), dtype=np.uint64) if want_values else None
        blks = np.empty((count, self.info.items_per_bundle, 2), dtype=np.uint64) if want_blocks else None
        _check(load_library().apsu_he_mask_generate(self.h, C.c_uint64(seed), C.c_uint32(count), C.c_void_p(int(masks_dev)),
                                                    _p(vals) if want_values else None, _p(blks) if want_blocks else None))
        return vals, blks

    def mask_generate_blake2xb(self, seed, count, masks_dev, first_value=0, want_values=True, want_blocks=True):
        """N4 with the reference's generator: SEAL's Blake2xb PRNG under the eight 64-bit words `seed`
        (receiver_osn.cpp:221-224, 248-251), starting at its first_value-th 32-bit output.  Same outputs as mask_generate."""
        sd = np.ascontiguousarray(seed, dtype=np.uint64)
        if sd.size != 8:
            raise ValueError("seed must be eight 64-bit words")
        vals = np.empty((count, self.n), dtype=np.uint64) if want_values else None
        blks = np.empty((count, self.info.items_per_bundle, 2), dtype=np.uint64) if want_blocks else None
        _check(load_library().apsu_he_mask_generate_blake2xb(self.h, _p(sd), C.c_uint64(first_value), C.c_uint32(count),
                                                             C.c_void_p(int(masks_dev)), _p(vals) if want_values else None,
                                                             _p(blks) if want_blocks else None))
        return vals, blks

    def decrypt_decode(self, sk_ntt, cts, count=None, on_device=False, want_blocks=True):
        """N4: the querier's decrypt + decode + packing of `count` results (result_package.cpp:175-213).
        sk_ntt: secret key mod q_0 in NTT form [n]; cts: [count][2][1][n] array, or a device pointer with on_device."""
        if not on_device:
            cts = np.ascontiguousarray(cts, dtype=np.uint64)
            count = cts.size // (2 * self.n)
            ptr = _p(cts)
        else:
            ptr = C.c_void_p(int(cts))
        vals = np.empty((count, self.n), dtype=np.uint64)
        blks = np.empty((count, self.info.items_per_bundle, 2), dtype=np.uint64) if want_blocks else None
        sk = np.ascontiguousarray(sk_ntt, dtype=np.uint64)
        _check(load_library().apsu_he_decrypt_decode(self.h, _p(sk), ptr, 1 if on_device else 0, C.c_uint32(count), _p(vals),
                                                     _p(blks) if want_blocks else None))
        return vals, blks

    def bundle_coeff(self, bundle, degree):
        """test hook -> (array, kind): kind 0 raw mod t, 1 NTT form [L][n], 2 pre-lifted NTT at the high level"""
        buf = np.empty((self.first_chain_idx + 1) * self.n, dtype=np.uint64)
        words, kind = C.c_size_t(), C.c_int()
        _check(load_library().apsu_he_bundle_download(self.h, bundle.h, degree, _p(buf), C.c_size_t(buf.size), C.byref(words),
                                                      C.byref(kind)))
        w = words.value
        return (buf[:w].copy() if kind.value == 0 else buf[:w].reshape(-1, self.n).copy()), kind.value

    def random_bundle(self, bundle_idx, cache_idx, degree, seed):
        h = C.c_void_p()
        _check(load_library().apsu_he_db_random_bundle(self.h, bundle_idx, cache_idx, degree, C.c_uint64(seed), C.byref(h)))
        return Bundle(self, h, bundle_idx, cache_idx, degree)

    def compute_powers(self, bundle_indices, sources, rk, on_device=False):
        """Receiver::ComputePowers.  sources[b][s]: ct (numpy [2][L][n]) or device pointer (int) of the
        s-th source power (ascending) for bundle index bundle_indices[b]."""
        idx = np.array(bundle_indices, dtype=np.uint32)
        flat = [s for per_b in sources for s in per_b]
        h = C.c_void_p()
        _check(load_library().apsu_he_compute_powers(self.h, C.c_void_p(idx.ctypes.data), len(idx), _ptr_array(flat),
                                                     1 if on_device else 0, rk.h if rk is not None else None, C.byref(h)))
        return Powers(self, h, bundle_indices)

    def eval_bundles(self, bundles, powers, rk, masks, out=None, masks_on_device=False, out_on_device=False):
        """ProcessBinBundleCache for every bundle: returns [count][2][1][n] (host) unless out is a device pointer.
        Without key switching a row has result_polys polynomials, result i being its first result_size(bundle i) ones."""
        count = len(bundles)
        hs = (C.c_void_p * count)(*[b.h for b in bundles])
        if not out_on_device:
            out = np.empty((count, self.result_polys, 1, self.n), dtype=np.uint64)
            outp = _p(out)
        else:
            outp = C.c_void_p(int(out))
        _check(load_library().apsu_he_eval_bundles(self.h, hs, count, powers.h, rk.h if rk is not None else None,
                                                   _ptr_array(list(masks)), 1 if masks_on_device else 0, outp,
                                                   1 if out_on_device else 0))
        return out


class DbFile:
    """a database file opened with mmap: the table, and BinBundles loaded one by one (apsu_he_db_file_*)"""

    def __init__(self, path):
        self.h = C.c_void_p()
        _check(load_library().apsu_he_db_file_open(os.fsencode(path), C.byref(self.h)))
        cnt, size = C.c_int(), C.c_uint64()
        _check(load_library().apsu_he_db_file_count(self.h, C.byref(cnt), C.byref(size)))
        self.count, self.file_bytes = cnt.value, size.value

    def __len__(self):
        return self.count

    def entry(self, i):
        """-> (bundle_idx, cache_idx, degree, image_bytes)"""
        b, c, d, sz = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        _check(load_library().apsu_he_db_file_entry(self.h, int(i), C.byref(b), C.byref(c), C.byref(d), C.byref(sz)))
        return b.value, c.value, d.value, sz.value

    def load(self, ctx, i):
        h = C.c_void_p()
        _check(load_library().apsu_he_db_file_load(ctx.h, self.h, int(i), C.byref(h)))
        b, c, d, _ = self.entry(i)
        return Bundle(ctx, h, b, c, d)

    def close(self):
        if getattr(self, "h", None):
            load_library().apsu_he_db_file_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def partition_bundles(units, bundle_idx_count, n_devices, compute_powers_cost=0):
    """apsu_he_partition_bundles(_ex): units [(bundle_idx, cache_idx, degree)] -> device slot per unit (no GPU needed)"""
    cnt = len(units)
    b = np.array([u[0] for u in units] or [0], dtype=np.uint32)
    c = np.array([u[1] for u in units] or [0], dtype=np.uint32)
    d = np.array([u[2] for u in units] or [0], dtype=np.uint32)
    out = np.zeros(max(cnt, 1), dtype=np.int32)
    _check(load_library().apsu_he_partition_bundles_ex(C.c_uint32(bundle_idx_count), int(n_devices), C.c_void_p(b.ctypes.data),
                                                       C.c_void_p(c.ctypes.data), C.c_void_p(d.ctypes.data), cnt,
                                                       C.c_uint64(int(compute_powers_cost)), C.c_void_p(out.ctypes.data)))
    return [int(x) for x in out[:cnt]]


class MultiContext:
    """Several GPUs of one node behind one handle (include/apsu_he.h: apsu_he_multi_*, apsu_he_eval_all): the
    in-process counterpart of Receiver::RunQuery's fan-out (receiver/apsu/receiver_osn.cpp:320-364)."""

    def __init__(self, psu_params_json, devices):
        L = load_library()
        if os.path.exists(psu_params_json):
            with open(psu_params_json) as f:
                psu_params_json = f.read()
        dv = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(L.apsu_he_multi_create(psu_params_json.encode(), dv, len(devices), C.byref(h)))
        self.h = h
        self.devices = list(devices)
        self.n_bundles = 0
        rp = C.c_uint32()
        _check(L.apsu_he_multi_result_polys(self.h, C.byref(rp)))
        self.result_polys = rp.value

    def close(self):
        if getattr(self, "h", None):
            load_library().apsu_he_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_relin_keys(self, rk):
        _check(load_library().apsu_he_multi_relin_upload(self.h, _p(np.ascontiguousarray(rk))))

    def upload_bundle(self, slot, bundle_idx, cache_idx, coeffs, is_ntt):
        flags = (C.c_uint8 * len(coeffs))(*[1 if f else 0 for f in is_ntt])
        keep = [np.ascontiguousarray(c, dtype=np.uint64) for c in coeffs]
        bid = C.c_int()
        _check(load_library().apsu_he_multi_db_upload_bundle(self.h, int(slot), bundle_idx, cache_idx, len(keep), _ptr_array(keep),
                                                             flags, C.byref(bid)))
        self.n_bundles = max(self.n_bundles, bid.value + 1)
        return bid.value

    def random_bundle(self, slot, bundle_idx, cache_idx, degree, seed):
        bid = C.c_int()
        _check(load_library().apsu_he_multi_db_random_bundle(self.h, int(slot), bundle_idx, cache_idx, degree, C.c_uint64(seed),
                                                             C.byref(bid)))
        self.n_bundles = max(self.n_bundles, bid.value + 1)
        return bid.value

    def clear_bundles(self):
        _check(load_library().apsu_he_multi_db_clear(self.h))
        self.n_bundles = 0

    def load_db_file(self, path):
        """every BinBundle of the file onto the handle's devices (partition rule, each device reads its own shard) -> count"""
        with DbFile(path) as f:
            k = C.c_int()
            _check(load_library().apsu_he_multi_db_load_file(self.h, f.h, C.byref(k)))
        self.n_bundles += k.value
        return k.value

    def save_db_file(self, path):
        _check(load_library().apsu_he_multi_db_save_file(self.h, os.fsencode(path)))

    IO_SRC_PINNED, IO_MASKS_PINNED, IO_OUT_PINNED, IO_SRC_ON_DEVICE, IO_MASKS_ON_DEVICE, IO_GATHER_RCCL = 1, 2, 4, 8, 16, 32

    def eval_all(self, sources, masks, n, out_device_slot=-1, out_ptr=None, flags=0, in_device_slot=0, out=None):
        """sources: flat list [bundle_idx][source] of cts (numpy arrays, or device pointers as ints with IO_SRC_ON_DEVICE);
        masks: per bundle id (likewise).  -> [n_bundles][2][1][n] (host; `out` = a caller-provided, e.g. page-locked, array)
        or writes to the device pointer out_ptr on devices[out_device_slot]"""
        if out_device_slot < 0:
            if out is None:
                out = np.zeros((self.n_bundles, self.result_polys, 1, n), dtype=np.uint64)
            outp = _p(out)
        else:
            out = None
            outp = C.c_void_p(int(out_ptr))
        _check(load_library().apsu_he_eval_all_ex(self.h, _ptr_array(list(sources)), _ptr_array(list(masks)), outp, int(out_device_slot),
                                                  C.c_uint(int(flags)), int(in_device_slot)))
        return out

    def last_gather(self):
        L = load_library()
        L.apsu_he_multi_last_gather.restype = C.c_char_p
        return L.apsu_he_multi_last_gather(self.h).decode()

    def phase_enable(self, on=True):
        _check(load_library().apsu_he_multi_phase_enable(self.h, 1 if on else 0))

    def phase_read(self, reset=True):
        L = load_library()
        L.apsu_he_phase_name.restype = C.c_char_p
        cnt = (C.c_uint64 * 3)(); avg = (C.c_double * 3)(); mn = (C.c_double * 3)(); mx = (C.c_double * 3)()
        _check(L.apsu_he_multi_phase_read(self.h, cnt, avg, mn, mx, 1 if reset else 0))
        return {L.apsu_he_phase_name(i).decode(): (int(cnt[i]), avg[i], mn[i], mx[i]) for i in range(3)}


def host_alloc(shape, dtype=np.uint64):
    """page-locked host array (apsu_he_host_alloc); release with host_free(array)"""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    _check(load_library().apsu_he_host_alloc(C.c_size_t(max(1, nbytes)), C.byref(p)))
    buf = (C.c_uint8 * max(1, nbytes)).from_address(p.value)
    a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    _PINNED[a.ctypes.data] = p.value
    return a


def host_free(a):
    p = _PINNED.pop(a.ctypes.data, None)
    if p is not None:
        _check(load_library().apsu_he_host_free(C.c_void_p(p)))


_PINNED = {}
