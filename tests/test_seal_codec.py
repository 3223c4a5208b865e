"""CPU tier: SEAL's object serialisation as restated in apsu_amd/csrc/seal_codec.{h,cpp} (N3) against an INDEPENDENT Python
model of the same description, written here from the field list alone: struct.pack for the layout, python's zlib for
compressed bodies, hashlib.blake2b for parms_id, oracle/blake2x.py (BLAKE2b core pinned by hashlib) for the seed expansion.
**The format itself is UNPINNED** — nothing in the reference or this image holds a SEAL-produced object; these tests show that
the C++ reader / writer and the Python model agree with each other and with the description, not with SEAL."""
import hashlib
import os
import struct
import sys
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apsu_amd
from apsu_amd import seal
from oracle.blake2x import blake2xb

N = 64
Q = [0xffffffffc001, 0xffffee001, 0x1ffc001]     # 48, 36, 25 bits: three key-level primes (1M-1024-com's)
T = 65537
MAGIC = b"\x5e\xa1\x10"


# ---------------------------------------------------------------------------------------------- the Python model
def _zstd():
    """the system's libzstd through ctypes (python has no zstd module here): the test's own compressor, one-shot and streamed"""
    import ctypes as C
    try:
        z = C.CDLL("libzstd.so.1")
    except OSError:
        return None
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    z.ZSTD_decompress.restype = C.c_size_t
    z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    z.ZSTD_createCStream.restype = C.c_void_p
    z.ZSTD_freeCStream.argtypes = [C.c_void_p]
    z.ZSTD_compressStream2.restype = C.c_size_t
    z.ZSTD_compressStream2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    z.ZSTD_isError.argtypes = [C.c_size_t]
    return z


def zstd_compress(data, streamed=False):
    import ctypes as C
    z = _zstd()
    cap = z.ZSTD_compressBound(len(data)) + 64
    dst = C.create_string_buffer(cap)
    if not streamed:
        k = z.ZSTD_compress(dst, cap, data, len(data), 3)
        assert not z.ZSTD_isError(k)
        return dst.raw[:k]

    class Buf(C.Structure):
        _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]
    # the way a SEAL writer does it (ztools.cpp [SEAL-recall]): a compression stream fed in chunks, no pledged source size, ZSTD_e_end last
    cs = z.ZSTD_createCStream()
    src = C.create_string_buffer(bytes(data), len(data))
    out = Buf(C.cast(dst, C.c_void_p), cap, 0)
    chunk = 1000
    at = 0
    while True:
        take = min(chunk, len(data) - at)
        last = at + take == len(data)
        inb = Buf(C.cast(C.addressof(src) + at, C.c_void_p), take, 0)
        while True:
            rem = z.ZSTD_compressStream2(cs, C.byref(out), C.byref(inb), 2 if last else 0)        # ZSTD_e_end / ZSTD_e_continue
            assert not z.ZSTD_isError(rem)
            if (last and rem == 0) or (not last and inb.pos == inb.size):
                break
        at += take
        if last:
            break
    z.ZSTD_freeCStream(cs)
    return dst.raw[:out.pos]


def zstd_decompress(frame, size):
    import ctypes as C
    z = _zstd()
    dst = C.create_string_buffer(size)
    k = z.ZSTD_decompress(dst, size, frame, len(frame))
    assert not z.ZSTD_isError(k) and k == size
    return dst.raw


def obj(members, compr=0, version=(4, 0)):
    stored = zlib.compress(members) if compr == 1 else (zstd_compress(members) if compr == 2 else members)
    return MAGIC + bytes([version[0], version[1], compr, 0, 0]) + struct.pack("<Q", 16 + len(stored)) + stored


def dyn_array(words, version):
    return obj(struct.pack("<Q", len(words)) + struct.pack("<%dQ" % len(words), *[int(w) for w in words]), 0, version)


def ct_members(parms_id, is_ntt, data, version, seed=None):
    size, L, n = data.shape
    m = struct.pack("<4Q", *parms_id) + bytes([1 if is_ntt else 0]) + struct.pack("<3Q", size, n, L)
    if version[0] >= 4:
        m += struct.pack("<Q", 1)
    m += struct.pack("<d", 1.0)
    if seed is None:
        return m + dyn_array(data.reshape(-1), version)
    return m + dyn_array(data[0].reshape(-1), version) + obj(bytes([1]) + struct.pack("<8Q", *seed), 0, version)


def parms_id(n, q, t):
    d = hashlib.blake2b(struct.pack("<%dQ" % (3 + len(q)), 1, n, *q, t), digest_size=32).digest()
    return list(struct.unpack("<4Q", d))


def prng_words(seed):
    """SEAL's Blake2xb generator as a stream of 64-bit words"""
    key = struct.pack("<8Q", *seed)
    counter = 0
    while True:
        buf = blake2xb(4096, struct.pack("<Q", counter), key)
        counter += 1
        yield from struct.unpack("<512Q", buf)


def shake_words(seed):
    """SEAL's Shake256 generator [SEAL-recall]: 4096-byte buffer k = SHAKE256(seed || k as u64), python's hashlib as the pin of SHAKE256"""
    key = struct.pack("<8Q", *seed)
    counter = 0
    while True:
        buf = hashlib.shake_256(key + struct.pack("<Q", counter)).digest(4096)
        counter += 1
        yield from struct.unpack("<512Q", buf)


def sample_poly_uniform(seed, q, n, shake=False):
    g = shake_words(seed) if shake else prng_words(seed)
    L = len(q)
    out = np.array([next(g) for _ in range(L * n)], dtype=object).reshape(L, n)
    for j, qj in enumerate(q):
        max_multiple = (2**64 - 1) - ((2**64 - 1) % qj) - 1
        for k in range(n):
            r = out[j, k]
            while r >= max_multiple:
                r = next(g)
            out[j, k] = r % qj
    return out.astype(np.uint64)


@pytest.fixture(scope="module")
def ctx():
    c = seal.SealContext(n=N, coeff_modulus=Q, plain_modulus=T)
    yield c
    c.close()


# ---------------------------------------------------------------------------------------------- tests
def test_parms_id_every_level(ctx):
    assert ctx.parms_id(-1) == parms_id(N, Q, T) == ctx.parms_id(2)
    assert ctx.parms_id(1) == parms_id(N, Q[:2], T)
    assert ctx.parms_id(0) == parms_id(N, Q[:1], T)
    with pytest.raises(ValueError):
        ctx.parms_id(5)
    js = open(os.path.join(os.path.dirname(__file__), "params", "16M-4096.json")).read()
    c = seal.SealContext(js)
    G_q = [0xfffffffff70001, 0xfffffffff78001, 0xfffffffffb4001, 0x3ffffffffc001]
    assert c.parms_id(-1) == parms_id(8192, G_q, 4079617)
    assert c.parms_id(2) == parms_id(8192, G_q[:3], 4079617)
    c.close()


def test_sample_poly_uniform_matches_model_including_rejections(ctx):
    rng = np.random.default_rng(7)
    # a 64-bit modulus-sized rejection zone is ~q / 2^64: force rejections with a context whose prime is just above 2^62
    big = seal.SealContext(n=N, coeff_modulus=[(1 << 62) + 0x2c01, 0xffffee001], plain_modulus=T)   # not an NTT prime: sampling only needs the value
    hits = 0
    for _ in range(4):
        seed = [int(x) for x in rng.integers(0, 2**63, 8, dtype=np.uint64)]
        assert (ctx.sample_poly_uniform(1, seed, 2, N) == sample_poly_uniform(seed, Q[:2], N)).all()
        want = sample_poly_uniform(seed, [(1 << 62) + 0x2c01], N)
        got = big.sample_poly_uniform(0, seed, 1, N)
        assert (got == want).all()
        g = prng_words(seed)
        hits += sum(1 for _ in range(N) if next(g) >= (2**64 - 1) - ((2**64 - 1) % ((1 << 62) + 0x2c01)) - 1)
    assert hits > 0                                                   # the rejection path really ran
    big.close()


@pytest.mark.parametrize("version", [(4, 0), (3, 6)])
@pytest.mark.parametrize("compr", [0, 1])
def test_ciphertext_plain_and_seeded_both_directions(ctx, version, compr):
    rng = np.random.default_rng(11)
    L = 2
    pid = parms_id(N, Q[:L], T)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in Q[:L]]) for _ in range(2)])
    # unseeded: model -> C++ ; C++ -> model bytes
    blob = obj(ct_members(pid, False, data, version), compr, version)
    got = ctx.ct_load(blob)
    assert got["parms_id"] == pid and got["chain_idx"] == L - 1 and not got["seeded"] and not got["is_ntt_form"]
    assert (got["data"] == data).all() and got["consumed"] == len(blob)
    mine = ctx.ct_save(L - 1, False, data, compr=compr, version=version)
    assert mine == blob                                              # byte-identical (python's zlib == the C library's default level)
    # trailing bytes behind the object are not consumed
    assert ctx.ct_load(blob + b"tail")["consumed"] == len(blob)
    # seeded: c1 is the expansion of the seed
    seed = [int(x) for x in rng.integers(0, 2**63, 8, dtype=np.uint64)]
    c1 = sample_poly_uniform(seed, Q[:L], N)
    sdata = np.stack([data[0], c1])
    sblob = obj(ct_members(pid, False, sdata, version, seed=seed), compr, version)
    got = ctx.ct_load(sblob)
    assert got["seeded"] and (got["data"] == sdata).all()
    assert ctx.ct_save(L - 1, False, sdata, seed=seed, compr=compr, version=version) == sblob
    assert len(sblob) < len(blob) or compr == 1


@pytest.mark.skipif(_zstd() is None, reason="libzstd.so.1 is not on this system")
def test_zstd_objects_both_directions(ctx):
    """compr_mode zstd -- SEAL's default when built with it -- through the system's libzstd (loaded at run time by the codec):
    one-shot and streamed frames in, frames out that the library itself inflates to the model's member bytes"""
    rng = np.random.default_rng(21)
    L = 2
    pid = parms_id(N, Q[:L], T)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in Q[:L]]) for _ in range(2)])
    seed = [int(x) for x in rng.integers(0, 2**63, 8, dtype=np.uint64)]
    sdata = np.stack([data[0], sample_poly_uniform(seed, Q[:L], N)])
    for version in ((4, 0), (3, 6)):
        for members, want, sd in ((ct_members(pid, False, data, version), data, None), (ct_members(pid, False, sdata, version, seed=seed), sdata, seed)):
            for streamed in (False, True):
                stored = zstd_compress(members, streamed)
                blob = MAGIC + bytes([version[0], version[1], 2, 0, 0]) + struct.pack("<Q", 16 + len(stored)) + stored
                got = ctx.ct_load(blob + b"tail")
                assert got["consumed"] == len(blob) and got["seeded"] == (sd is not None) and (got["data"] == want).all()
            mine = ctx.ct_save(L - 1, False, want, seed=sd, compr=2, version=version)
            assert mine[:8] == MAGIC + bytes([version[0], version[1], 2, 0, 0]) and struct.unpack_from("<Q", mine, 8)[0] == len(mine)
            assert zstd_decompress(mine[16:], len(members)) == members
    # relinearisation keys, the big object of a query
    ksk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in Q]) for _ in range(2)]) for _ in range(len(Q) - 1)])
    blob = ctx.relin_keys_save(ksk, compr=2)
    back, used = ctx.relin_keys_load(blob)
    assert used == len(blob) and (back.reshape(ksk.shape) == ksk).all() and blob[5] == 2
    # damage: a flipped byte in the frame, a truncated frame
    stored = zstd_compress(ct_members(pid, False, data, (4, 0)))
    for bad in (stored[:40] + bytes([stored[40] ^ 0x55]) + stored[41:], stored[:len(stored) // 2]):
        with pytest.raises(apsu_amd.ApsuHeError, match="zstd|truncated|SEAL object"):
            ctx.ct_load(MAGIC + bytes([4, 0, 2, 0, 0]) + struct.pack("<Q", 16 + len(bad)) + bad)


def pt_members(parms_id, data, version):
    return struct.pack("<4Q", *parms_id) + struct.pack("<Q", len(data)) + struct.pack("<d", 1.0) + dyn_array(data, version)


@pytest.mark.parametrize("compr", [0, 1, 2])
def test_plaintext_both_directions(ctx, compr):
    """seal::Plaintext as a BinBundle cache stores it per coefficient (bin_bundle.cpp:421-428): coefficient form (parms_id zero)
    and NTT form (the level's parms_id, one polynomial per prime)"""
    if compr == 2 and _zstd() is None:
        pytest.skip("libzstd.so.1 is not on this system")
    rng = np.random.default_rng(31)
    for version in ((4, 0), (3, 6)):
        coeff = rng.integers(0, T, N, dtype=np.uint64)
        blob = obj(pt_members([0, 0, 0, 0], coeff, version), compr, version)
        got = ctx.pt_load(blob + b"x")
        assert got["chain_idx"] == -1 and got["consumed"] == len(blob) and (got["data"] == coeff).all()
        assert ctx.pt_save(-1, coeff, compr=compr, version=version) == blob
        L = 2
        ntt = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in Q[:L]])
        blob = obj(pt_members(parms_id(N, Q[:L], T), ntt.reshape(-1), version), compr, version)
        got = ctx.pt_load(blob)
        assert got["chain_idx"] == L - 1 and (got["data"].reshape(L, N) == ntt).all()
        assert ctx.pt_save(L - 1, ntt, compr=compr, version=version) == blob
    with pytest.raises(apsu_amd.ApsuHeError):
        ctx.pt_load(blob[:-9])
    with pytest.raises(ValueError, match="parms_id"):
        ctx.pt_load(obj(pt_members([5, 5, 5, 5], coeff, (4, 0)), 0))
    with pytest.raises(apsu_amd.ApsuHeError, match="coeff_count"):
        ctx.pt_load(obj(struct.pack("<4Q", 0, 0, 0, 0) + struct.pack("<Q", N + 1) + struct.pack("<d", 1.0) + dyn_array(coeff, (4, 0)), 0))
    with pytest.raises(ValueError):
        ctx.pt_save(1, coeff)                                         # an NTT-form plaintext at level 1 has 2 n words


def test_ciphertext_rejects_malformed(ctx):
    rng = np.random.default_rng(3)
    pid = parms_id(N, Q[:2], T)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in Q[:2]]) for _ in range(2)])
    blob = obj(ct_members(pid, False, data, (4, 0)), 0)
    for bad in (blob[:-1], blob[:40], b"\x00" + blob[1:], blob[:5] + bytes([2]) + blob[6:], blob[:5] + bytes([7]) + blob[6:],
                blob[:3] + bytes([2, 0]) + blob[5:]):
        with pytest.raises(apsu_amd.ApsuHeError):
            ctx.ct_load(bad)
    with pytest.raises(apsu_amd.ApsuHeError, match="zstd"):
        ctx.ct_load(blob[:5] + bytes([2]) + blob[6:])
    # dimensions are the peer's claim: a level's parms_id with another coeff_modulus_size is refused in every load mode (the
    # reference's SEALObject::extract -> is_valid_for does the same, common/apsu/seal_object.h:161-219) ...
    forged = obj(ct_members(pid, False, data[:, :1], (4, 0)), 0)
    with pytest.raises(apsu_amd.ApsuHeError, match="coeff_modulus_size"):
        ctx.ct_load(forged)
    with pytest.raises(apsu_amd.ApsuHeError, match="coeff_modulus_size"):
        ctx.ct_load_unexpanded(forged, 2, N)
    # ... and a ~100-byte object that claims 2^32 words allocates nothing: the array must be the whole ciphertext or its seeded half
    import time
    hdr = struct.pack("<4Q", *pid) + bytes([0]) + struct.pack("<3Q", 64, 1 << 20, 64) + struct.pack("<Q", 1) + struct.pack("<d", 1.0)
    t0 = time.time()
    for cnt_words in ([], [1, 2, 3]):
        with pytest.raises(apsu_amd.ApsuHeError, match="inconsistent coefficient array"):
            ctx.ct_load(obj(hdr + dyn_array(cnt_words, (4, 0)), 0))
    assert time.time() - t0 < 1.0
    # a seeded ciphertext whose parms_id this context does not know
    seed = [1, 2, 3, 4, 5, 6, 7, 8]
    sblob = obj(ct_members([9, 9, 9, 9], False, data, (4, 0), seed=seed), 0)
    with pytest.raises(apsu_amd.ApsuHeError, match="parms_id"):
        ctx.ct_load(sblob)
    # an unknown generator type
    m = ct_members(pid, False, data, (4, 0), seed=seed)
    with pytest.raises(apsu_amd.ApsuHeError, match="generator"):
        ctx.ct_load(obj(m[:-65] + bytes([3]) + m[-64:], 0))
    # corrupt zlib stream, and random corruptions never crash
    z = obj(ct_members(pid, False, data, (4, 0)), 1)
    with pytest.raises(apsu_amd.ApsuHeError):
        ctx.ct_load(z[:30] + bytes([z[30] ^ 0xff]) + z[31:])
    for i in range(300):
        b = bytearray(blob)
        for _ in range(1 + i % 3):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        try:
            ctx.ct_load(bytes(b))
        except apsu_amd.ApsuHeError:
            pass


@pytest.mark.parametrize("compr", [0, 1])
@pytest.mark.parametrize("seeded", [False, True])
def test_relin_keys_both_directions(ctx, compr, seeded):
    rng = np.random.default_rng(5)
    K = len(Q)
    pid = parms_id(N, Q, T)
    keys, seeds = [], []
    for j in range(K - 1):
        seed = [int(x) for x in rng.integers(0, 2**63, 8, dtype=np.uint64)]
        c0 = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in Q])
        c1 = sample_poly_uniform(seed, Q, N) if seeded else np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in Q])
        keys.append(np.stack([c0, c1]))
        seeds.append(seed)
    members = struct.pack("<4Q", *pid) + struct.pack("<Q", 1) + struct.pack("<Q", K - 1)
    for j in range(K - 1):
        # PublicKey::save forwards to its Ciphertext's save (SEAL publickey.h): ONE header per key, no envelope of its own
        members += obj(ct_members(pid, True, keys[j], (4, 0), seed=seeds[j] if seeded else None), 0)
    blob = obj(members, compr)
    ksk, used = ctx.relin_keys_load(blob)
    assert used == len(blob)
    assert (ksk.reshape(K - 1, 2, K, N) == np.stack(keys)).all()
    mine = ctx.relin_keys_save(np.stack(keys), seeds=np.array(seeds, dtype=np.uint64) if seeded else None, compr=compr)
    assert mine == blob
    # keys of another parameter set are refused
    other = seal.SealContext(n=N, coeff_modulus=Q[:2], plain_modulus=T)
    with pytest.raises((ValueError, apsu_amd.ApsuHeError)):          # seeded: unknown parms_id; unseeded: parameter mismatch
        other.relin_keys_load(blob)
    other.close()


def test_seeded_objects_of_a_shake256_seal(ctx):
    """a SEAL built with SEAL_DEFAULT_PRNG=Shake256 writes prng_type 2 into its seeded objects (sender/apsu/plaintext_powers.cpp:41-46
    would produce such queries): expanded by the host codec in every load mode, SHAKE256 itself pinned by hashlib"""
    rng = np.random.default_rng(9)
    pid = parms_id(N, Q[:2], T)
    seed = [int(x) for x in rng.integers(0, 2**63, 8, dtype=np.uint64)]
    c0 = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in Q[:2]])
    c1 = sample_poly_uniform(seed, Q[:2], N, shake=True)
    assert not (c1 == sample_poly_uniform(seed, Q[:2], N)).all()
    m = ct_members(pid, False, np.stack([c0, c1]), (4, 0), seed=seed)
    for compr in (0, 1):
        blob = obj(m[:-65] + bytes([2]) + m[-64:], compr)
        got = ctx.ct_load(blob)
        assert got["seeded"] and got["chain_idx"] == 1 and (got["data"] == np.stack([c0, c1])).all()
        # the load that leaves Blake2xb seeds to the device hands a Shake256 object back complete and marked unseeded
        u = ctx.ct_load_unexpanded(blob, 2, N)
        assert not u["seeded"] and (u["data"] == np.stack([c0, c1])).all()
