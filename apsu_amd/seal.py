"""ctypes binding of the SEAL object codec (include/apsu_he.h: apsu_he_seal_*; apsu_amd/csrc/seal_codec.h): seeded / zlib
ciphertexts, RelinKeys, parms_id.  Host only.  UNPINNED restatement of upstream SEAL (see the header)."""
import ctypes as C
import os

import numpy as np

from .engine import _check, load_library
from .wire import _buf, _take, u8p

u64p = C.POINTER(C.c_uint64)
COMPR_NONE, COMPR_ZLIB, COMPR_ZSTD = 0, 1, 2


class SealContext:
    """the modulus chain of one parameter set (PSUParams JSON, or n / coeff_modulus / plain_modulus)"""

    def __init__(self, psu_params_json=None, n=None, coeff_modulus=None, plain_modulus=None):
        L = load_library()
        h = C.c_void_p()
        if psu_params_json is not None:
            _check(L.apsu_he_seal_ctx_create(psu_params_json.encode(), C.byref(h)))
        else:
            q = (C.c_uint64 * len(coeff_modulus))(*coeff_modulus)
            _check(L.apsu_he_seal_ctx_create_raw(C.c_uint64(n), q, len(coeff_modulus), C.c_uint64(plain_modulus), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            load_library().apsu_he_seal_ctx_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def parms_id(self, chain_idx=-1):
        out = (C.c_uint64 * 4)()
        _check(load_library().apsu_he_seal_parms_id(self.h, int(chain_idx), out))
        return [int(v) for v in out]

    def sample_poly_uniform(self, chain_idx, seed, L, n):
        out = np.empty((L, n), dtype=np.uint64)
        s = (C.c_uint64 * 8)(*[int(w) for w in seed])
        _check(load_library().apsu_he_seal_sample_poly_uniform(self.h, int(chain_idx), s, out.ctypes.data_as(u64p)))
        return out

    def ct_load(self, buf):
        """-> dict(parms_id, chain_idx, is_ntt_form, seeded, data [size][L][n], consumed)"""
        Lb = load_library()
        keep = _buf(buf)
        pid = (C.c_uint64 * 4)()
        ci, ntt, seeded = C.c_int(), C.c_int(), C.c_int()
        sz, n, k = C.c_uint64(), C.c_uint64(), C.c_uint64()
        used = C.c_size_t()
        _check(Lb.apsu_he_seal_ct_load(self.h, keep, C.c_size_t(len(buf)), pid, C.byref(ci), C.byref(ntt), C.byref(sz), C.byref(n),
                                       C.byref(k), C.byref(seeded), None, C.c_size_t(0), C.byref(used)))
        data = np.empty((sz.value, k.value, n.value), dtype=np.uint64)
        _check(Lb.apsu_he_seal_ct_load(self.h, keep, C.c_size_t(len(buf)), pid, C.byref(ci), C.byref(ntt), C.byref(sz), C.byref(n),
                                       C.byref(k), C.byref(seeded), data.ctypes.data_as(u64p), C.c_size_t(data.size), C.byref(used)))
        return dict(parms_id=[int(v) for v in pid], chain_idx=ci.value, is_ntt_form=bool(ntt.value), seeded=bool(seeded.value),
                    data=data, consumed=used.value)

    def ct_load_unexpanded(self, buf, L, n):
        """-> dict(chain_idx, seeded, seed (8 words) or None, data: c0 [L][n] when seeded, else [size][L][n])"""
        Lb = load_library()
        keep = _buf(buf)
        ci, ntt, seeded = C.c_int(), C.c_int(), C.c_int()
        sz, k = C.c_uint64(), C.c_uint64()
        seed = (C.c_uint64 * 8)()
        data = np.zeros(2 * L * n, dtype=np.uint64)
        used = C.c_size_t()
        _check(Lb.apsu_he_seal_ct_load_unexpanded(self.h, keep, C.c_size_t(len(buf)), C.byref(ci), C.byref(ntt), C.byref(sz), C.byref(k),
                                                  C.byref(seeded), seed, data.ctypes.data_as(u64p), C.c_size_t(data.size), C.byref(used)))
        if seeded.value:
            return dict(chain_idx=ci.value, seeded=True, seed=[int(v) for v in seed], data=data[:L * n].reshape(L, n), consumed=used.value)
        return dict(chain_idx=ci.value, seeded=False, seed=None, data=data.reshape(int(sz.value), L, n), consumed=used.value)

    def ct_save(self, chain_idx, is_ntt_form, data, seed=None, compr=COMPR_NONE, version=(4, 0)):
        data = np.ascontiguousarray(data, dtype=np.uint64)
        s = (C.c_uint64 * 8)(*[int(w) for w in seed]) if seed is not None else None
        out, size = u8p(), C.c_size_t()
        _check(load_library().apsu_he_seal_ct_save(self.h, int(chain_idx), int(is_ntt_form), C.c_uint64(data.shape[0]),
                                                   data.ctypes.data_as(u64p), s, int(compr), version[0], version[1],
                                                   C.byref(out), C.byref(size)))
        return _take(out, size)

    def relin_keys_load(self, buf):
        """-> (ksk [K-1][2][K][n] flat uint64 array, consumed)"""
        Lb = load_library()
        keep = _buf(buf)
        words, used = C.c_size_t(), C.c_size_t()
        _check(Lb.apsu_he_seal_relin_keys_load(self.h, keep, C.c_size_t(len(buf)), None, C.c_size_t(0), C.byref(words), C.byref(used)))
        out = np.empty(words.value, dtype=np.uint64)
        _check(Lb.apsu_he_seal_relin_keys_load(self.h, keep, C.c_size_t(len(buf)), out.ctypes.data_as(u64p), C.c_size_t(out.size),
                                               C.byref(words), C.byref(used)))
        return out, used.value

    def pt_load(self, buf):
        """-> dict(chain_idx (-1: coefficient form), data [coeff_count], consumed)"""
        Lb = load_library()
        keep = _buf(buf)
        ci, cnt, used = C.c_int(), C.c_uint64(), C.c_size_t()
        _check(Lb.apsu_he_seal_pt_load(self.h, keep, C.c_size_t(len(buf)), C.byref(ci), C.byref(cnt), None, C.c_size_t(0), C.byref(used)))
        data = np.empty(cnt.value, dtype=np.uint64)
        _check(Lb.apsu_he_seal_pt_load(self.h, keep, C.c_size_t(len(buf)), C.byref(ci), C.byref(cnt), data.ctypes.data_as(u64p), C.c_size_t(data.size),
                                       C.byref(used)))
        return dict(chain_idx=ci.value, data=data, consumed=used.value)

    def pt_save(self, chain_idx, data, compr=COMPR_NONE, version=(4, 0)):
        """Plaintext::save: chain_idx -1 = coefficient form (parms_id zero), else NTT form at that level"""
        data = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1)
        out, size = C.POINTER(C.c_uint8)(), C.c_size_t()
        _check(load_library().apsu_he_seal_pt_save(self.h, int(chain_idx), data.ctypes.data_as(u64p), C.c_uint64(data.size), int(compr), version[0],
                                                   version[1], C.byref(out), C.byref(size)))
        return _take(out, size)

    def relin_keys_save(self, ksk, seeds=None, compr=COMPR_NONE, version=(4, 0)):
        ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        sd = None
        if seeds is not None:
            sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        out, size = u8p(), C.c_size_t()
        _check(load_library().apsu_he_seal_relin_keys_save(self.h, ksk.ctypes.data_as(u64p), sd.ctypes.data_as(u64p) if sd is not None else None,
                                                           int(compr), version[0], version[1], C.byref(out), C.byref(size)))
        return _take(out, size)


def run_query_request(ctx, seal_ctx, request, bundles, masks, compr=COMPR_NONE):
    """apsu_he_run_query_request: framed QueryRequest bytes -> list of framed ResultPackage bytes (one per BinBundle).
    ctx: HeContext, bundles: its Bundle handles, masks: numpy arrays (host)"""
    from .engine import _ptr_array
    L = load_library()
    keep = _buf(request)
    cnt = len(bundles)
    bh = (C.c_void_p * max(1, cnt))(*[b.h for b in bundles])
    pk = (u8p * max(1, cnt))()
    sz = (C.c_size_t * max(1, cnt))()
    mk = [np.ascontiguousarray(m, dtype=np.uint64) for m in masks]
    _check(L.apsu_he_run_query_request(ctx.h, seal_ctx.h, keep, C.c_size_t(len(request)), bh, cnt, _ptr_array(mk), 0, int(compr), pk, sz))
    out = []
    for i in range(cnt):
        out.append(C.string_at(pk[i], sz[i]))
        L.apsu_he_wire_buffer_free(pk[i])
    return out



def upload_bundle_serialized(ctx, seal_ctx, bundle_idx, cache_idx, blobs):
    """apsu_he_db_upload_bundle_serialized: blobs[d] = the SEAL-serialised Plaintext the reference's BinBundle cache holds for coefficient d"""
    from .engine import Bundle
    keeps = [_buf(b) for b in blobs]
    arr = (C.c_void_p * len(blobs))(*[C.cast(k, C.c_void_p) for k in keeps])
    sizes = (C.c_size_t * len(blobs))(*[len(b) for b in blobs])
    h = C.c_void_p()
    _check(load_library().apsu_he_db_upload_bundle_serialized(ctx.h, seal_ctx.h, bundle_idx, cache_idx, len(blobs), arr, sizes, C.byref(h)))
    return Bundle(ctx, h, bundle_idx, cache_idx, len(blobs) - 1)


def multi_run_query_request(multi, seal_ctx, request, masks, compr=COMPR_NONE):
    """apsu_he_multi_run_query_request: QueryRequest bytes in -> one ResultPackage (bytes) per BinBundle registered in the handle"""
    from .engine import _ptr_array
    L = load_library()
    cnt = multi.n_bundles
    keep = _buf(request)
    mk = [np.ascontiguousarray(m, dtype=np.uint64) for m in masks]
    pk = (u8p * max(1, cnt))()
    sz = (C.c_size_t * max(1, cnt))()
    _check(L.apsu_he_multi_run_query_request(multi.h, seal_ctx.h, keep, C.c_size_t(len(request)), _ptr_array(mk), int(compr), pk, sz, cnt))
    out = []
    for i in range(cnt):
        out.append(C.string_at(pk[i], sz[i]))
        L.apsu_he_wire_buffer_free(pk[i])
    return out


def upload_saved_bundle(ctx, seal_ctx, buf, cache_idx=0):
    """apsu_he_db_upload_saved_bundle: a BinBundle as the reference persists it -> (Bundle on the device, bytes consumed)"""
    from .engine import Bundle
    keep = _buf(buf)
    h, used, deg = C.c_void_p(), C.c_size_t(), C.c_uint32()
    _check(load_library().apsu_he_db_upload_saved_bundle(ctx.h, seal_ctx.h if seal_ctx is not None else None, keep, C.c_size_t(len(buf)), int(cache_idx),
                                                        C.byref(h), C.byref(used)))
    _check(load_library().apsu_he_bundle_degree(h, C.byref(deg)))
    bi = C.c_uint32()
    _check(load_library().apsu_he_wire_bin_bundle_info(keep, C.c_size_t(len(buf)), C.byref(bi), None, None, None, None, None, None))
    return Bundle(ctx, h, bi.value, int(cache_idx), deg.value), used.value


def load_reference_db(blob, device=0):
    """A database the reference saved with ReceiverDB::save (receiver_db.fbs header, then one bin_bundle.fbs buffer per BinBundle,
    receiver_db.cpp:1182-1260) -> (HeContext, SealContext, [Bundle]) with every BinBundle on the device: from its saved cache, or
    rebuilt on the GPU from the item bins.  cache_idx = a BinBundle's position among those of its bundle index, the order
    ReceiverDB::Load restores them in (receiver_db.cpp:1380-1420)."""
    from . import wire
    from .engine import HeContext
    if isinstance(blob, (str, os.PathLike)):
        with open(blob, "rb") as f:
            blob = f.read()
    hdr = wire.receiver_db_header(blob)
    ctx = HeContext(hdr["params_json"], device=device)
    sc = SealContext(hdr["params_json"])
    at, seen, bundles = hdr["consumed"], {}, []
    for _ in range(hdr["bin_bundle_count"]):
        info = wire.bin_bundle_info(blob[at:])
        ci = seen.get(info["bundle_idx"], 0)
        seen[info["bundle_idx"]] = ci + 1
        b, used = upload_saved_bundle(ctx, sc, blob[at:at + info["consumed"]], ci)
        bundles.append(b)
        at += used
    if at != len(blob):
        raise ValueError("trailing bytes behind the last BinBundle")
    return ctx, sc, bundles
