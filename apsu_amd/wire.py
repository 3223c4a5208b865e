"""ctypes binding of the N3 framing functions (include/apsu_he.h: apsu_he_wire_*): the reference's FlatBuffers messages
around the query-evaluation path (common/apsu/network/*.fbs, receiver_operation.cpp, result_package.cpp).  Host only."""
import ctypes as C

from .engine import _check, load_library

u8p = C.POINTER(C.c_uint8)


def _take(out, size):
    data = C.string_at(out, size.value)
    load_library().apsu_he_wire_buffer_free(out)
    return data


def _buf(b):
    return (C.c_uint8 * len(b)).from_buffer_copy(b) if len(b) else (C.c_uint8 * 1)()


def build_header(version, rop_type):
    out, size = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_build_header(version, rop_type, C.byref(out), C.byref(size)))
    return _take(out, size)


def parse_header(buf):
    v, t = C.c_uint32(), C.c_uint32()
    _check(load_library().apsu_he_wire_parse_header(_buf(buf), C.c_size_t(len(buf)), C.byref(v), C.byref(t)))
    return v.value, t.value


def build_query_request(compression_type, relin_keys, parts):
    """parts: [(exponent, [ct bytes, ...]), ...]; relin_keys: bytes or None"""
    flat = [ct for _, cts in parts for ct in cts]
    keep = [_buf(ct) for ct in flat]
    ptrs = (C.c_void_p * max(1, len(flat)))(*[C.addressof(k) for k in keep])
    sizes = (C.c_size_t * max(1, len(flat)))(*[len(ct) for ct in flat])
    exps = (C.c_uint32 * max(1, len(parts)))(*[e for e, _ in parts])
    cnts = (C.c_uint32 * max(1, len(parts)))(*[len(c) for _, c in parts])
    rk = _buf(relin_keys) if relin_keys is not None else None
    out, size = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_build_query_request(C.c_uint8(compression_type), rk, C.c_size_t(len(relin_keys) if relin_keys is not None else 0),
                                                           len(parts), exps, cnts, ptrs, sizes, C.byref(out), C.byref(size)))
    return _take(out, size)


def parse_query_request(buf):
    """-> (compression_type, relin_keys bytes or None, [(exponent, [ct bytes])])"""
    L = load_library()
    keep = _buf(buf)
    h = C.c_void_p()
    _check(L.apsu_he_wire_parse_query_request(keep, C.c_size_t(len(buf)), C.byref(h)))
    try:
        ct, has, rk, rkn, nparts = C.c_uint8(), C.c_int(), u8p(), C.c_size_t(), C.c_uint32()
        _check(L.apsu_he_wire_query_info(h, C.byref(ct), C.byref(has), C.byref(rk), C.byref(rkn), C.byref(nparts)))
        relin = C.string_at(rk, rkn.value) if has.value else None
        parts = []
        for i in range(nparts.value):
            e, n = C.c_uint32(), C.c_uint32()
            _check(L.apsu_he_wire_query_part(h, i, C.byref(e), C.byref(n)))
            cts = []
            for j in range(n.value):
                p, sz = u8p(), C.c_size_t()
                _check(L.apsu_he_wire_query_ct(h, i, j, C.byref(p), C.byref(sz)))
                cts.append(C.string_at(p, sz.value))
            parts.append((e.value, cts))
        return ct.value, relin, parts
    finally:
        L.apsu_he_wire_query_free(h)


def build_query_response(package_count, alpha_max_cache_count):
    out, size = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_build_query_response(package_count, alpha_max_cache_count, C.byref(out), C.byref(size)))
    return _take(out, size)


def parse_query_response(buf):
    a, b = C.c_uint32(), C.c_uint32()
    _check(load_library().apsu_he_wire_parse_query_response(_buf(buf), C.c_size_t(len(buf)), C.byref(a), C.byref(b)))
    return a.value, b.value


def build_result_package(bundle_idx, cache_idx, psu_result, label_byte_count=0, nonce_byte_count=0, labels=()):
    keep = [_buf(x) for x in labels]
    ptrs = (C.c_void_p * max(1, len(labels)))(*[C.addressof(k) for k in keep])
    sizes = (C.c_size_t * max(1, len(labels)))(*[len(x) for x in labels])
    out, size = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_build_result_package(bundle_idx, cache_idx, _buf(psu_result), C.c_size_t(len(psu_result)),
                                                            label_byte_count, nonce_byte_count, len(labels), ptrs, sizes,
                                                            C.byref(out), C.byref(size)))
    return _take(out, size)


def parse_result_package(buf):
    """-> dict(bundle_idx, cache_idx, psu_result, label_byte_count, nonce_byte_count, labels)"""
    L = load_library()
    keep = _buf(buf)
    b, c, p, pn, lb, nb, nl = C.c_uint32(), C.c_uint32(), u8p(), C.c_size_t(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    _check(L.apsu_he_wire_parse_result_package(keep, C.c_size_t(len(buf)), C.byref(b), C.byref(c), C.byref(p), C.byref(pn), C.byref(lb),
                                               C.byref(nb), C.byref(nl)))
    labels = []
    for i in range(nl.value):
        q, qn = u8p(), C.c_size_t()
        _check(L.apsu_he_wire_result_label(keep, C.c_size_t(len(buf)), i, C.byref(q), C.byref(qn)))
        labels.append(C.string_at(q, qn.value))
    return dict(bundle_idx=b.value, cache_idx=c.value, psu_result=C.string_at(p, pn.value), label_byte_count=lb.value,
                nonce_byte_count=nb.value, labels=labels)


def seal_ct_save(parms_id, is_ntt_form, ct, correction_factor=1, scale=1.0, version=(4, 0)):
    """UNPINNED envelope (see the header).  ct: uint64 array [size][coeff_modulus_size][n]"""
    import numpy as np
    ct = np.ascontiguousarray(ct, dtype=np.uint64)
    pid = (C.c_uint64 * 4)(*parms_id)
    out, size = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_seal_ct_save(pid, int(is_ntt_form), C.c_uint64(ct.shape[0]), C.c_uint64(ct.shape[2]),
                                                    C.c_uint64(ct.shape[1]), C.c_uint64(correction_factor), C.c_double(scale),
                                                    ct.ctypes.data_as(C.POINTER(C.c_uint64)), version[0], version[1],
                                                    C.byref(out), C.byref(size)))
    return _take(out, size)


def seal_ct_load(buf):
    import numpy as np
    L = load_library()
    keep = _buf(buf)
    pid = (C.c_uint64 * 4)()
    ntt, sz, n, k, cf, sc, vmaj, vmin = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_double(), C.c_int(), C.c_int()
    _check(L.apsu_he_wire_seal_ct_load(keep, C.c_size_t(len(buf)), pid, C.byref(ntt), C.byref(sz), C.byref(n), C.byref(k), C.byref(cf),
                                       C.byref(sc), None, C.c_size_t(0), C.byref(vmaj), C.byref(vmin)))
    data = np.empty((sz.value, k.value, n.value), dtype=np.uint64)
    _check(L.apsu_he_wire_seal_ct_load(keep, C.c_size_t(len(buf)), pid, C.byref(ntt), C.byref(sz), C.byref(n), C.byref(k), C.byref(cf),
                                       C.byref(sc), data.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(data.size), C.byref(vmaj),
                                       C.byref(vmin)))
    return dict(parms_id=list(pid), is_ntt_form=bool(ntt.value), data=data, correction_factor=cf.value, scale=sc.value,
                version=(vmaj.value, vmin.value))


def bin_bundle_info(buf):
    """one BinBundle as ReceiverDB::save wrote it (bin_bundle.fbs) -> dict(bundle_idx, mod, stripped, n_bins, largest_bin, cache_coeffs, consumed)"""
    keep = _buf(buf)
    bi, nb, lb, cc = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    mod, st, used = C.c_uint64(), C.c_int(), C.c_size_t()
    _check(load_library().apsu_he_wire_bin_bundle_info(keep, C.c_size_t(len(buf)), C.byref(bi), C.byref(mod), C.byref(st), C.byref(nb), C.byref(lb),
                                                      C.byref(cc), C.byref(used)))
    return dict(bundle_idx=bi.value, mod=mod.value, stripped=bool(st.value), n_bins=nb.value, largest_bin=lb.value, cache_coeffs=cc.value,
                consumed=used.value)


def peek_type(buf, is_response=False):
    t = C.c_int()
    _check(load_library().apsu_he_wire_peek_type(_buf(buf), C.c_size_t(len(buf)), 1 if is_response else 0, C.byref(t)))
    return t.value


def psu_params_save(psu_params_json):
    """PSUParams::save: the JSON form -> the binary form (psu_params.fbs + SEAL EncryptionParameters)"""
    out, size = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_psu_params_save(psu_params_json.encode(), C.byref(out), C.byref(size)))
    return _take(out, size)


def psu_params_load(buf):
    """PSUParams::Load(binary) -> the JSON text apsu_he_create takes"""
    out, size = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_psu_params_load(_buf(buf), C.c_size_t(len(buf)), C.byref(out), C.byref(size)))
    return _take(out, size).decode()


def build_parms_request():
    out, size = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_build_parms_request(C.byref(out), C.byref(size)))
    return _take(out, size)


def build_parms_response(psu_params):
    out, size = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_build_parms_response(_buf(psu_params), C.c_size_t(len(psu_params)), C.byref(out), C.byref(size)))
    return _take(out, size)


def parse_parms_response(buf):
    keep = _buf(buf)
    p, n = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_parse_parms_response(keep, C.c_size_t(len(buf)), C.byref(p), C.byref(n)))
    return C.string_at(p, n.value) if n.value else b""


def build_plain_response(bundle_idx, cache_idx, psu_result):
    import numpy as np
    v = np.ascontiguousarray(psu_result, dtype=np.uint64)
    out, size = u8p(), C.c_size_t()
    _check(load_library().apsu_he_wire_build_plain_response(bundle_idx, cache_idx, v.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(v.size),
                                                           C.byref(out), C.byref(size)))
    return _take(out, size)


def parse_plain_response(buf):
    import numpy as np
    keep = _buf(buf)
    bi, ci, cnt = C.c_uint32(), C.c_uint32(), C.c_size_t()
    _check(load_library().apsu_he_wire_parse_plain_response(keep, C.c_size_t(len(buf)), C.byref(bi), C.byref(ci), None, C.c_size_t(0), C.byref(cnt)))
    v = np.empty(cnt.value, dtype=np.uint64)
    _check(load_library().apsu_he_wire_parse_plain_response(keep, C.c_size_t(len(buf)), C.byref(bi), C.byref(ci), v.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                           C.c_size_t(v.size), C.byref(cnt)))
    return dict(bundle_idx=bi.value, cache_idx=ci.value, psu_result=v)


def receiver_db_header(buf):
    """the header of a database the reference saved (receiver_db.fbs) -> dict(params_json, item_count, bin_bundle_count, compressed, stripped,
    label_byte_count, consumed)"""
    keep = _buf(buf)
    js, jn = u8p(), C.c_size_t()
    items, bbs, lab, used = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_size_t()
    comp, strip = C.c_int(), C.c_int()
    _check(load_library().apsu_he_wire_receiver_db_header(keep, C.c_size_t(len(buf)), C.byref(js), C.byref(jn), C.byref(items), C.byref(bbs),
                                                         C.byref(comp), C.byref(strip), C.byref(lab), C.byref(used)))
    return dict(params_json=_take(js, jn).decode(), item_count=items.value, bin_bundle_count=bbs.value, compressed=bool(comp.value),
                stripped=bool(strip.value), label_byte_count=lab.value, consumed=used.value)
