"""TEST INFRASTRUCTURE ONLY (see oracle/README.md): Python model of SEAL's default PRNG (Blake2xb) as the reference uses it
to draw the BinBundle masks (receiver/apsu/receiver_osn.cpp:221-224, 248-251).

* `blake2b` — RFC 7693 written out (own compression function, arbitrary 64-byte parameter block).  Pinned: tests compare it
  with `hashlib.blake2b` over keys, salts, personalisation and tree parameters, and with the RFC's "abc" digest.
* `blake2xb` — the BLAKE2X expansion of the BLAKE2 reference implementation (blake2xb.c: `blake2xb_init_key` /
  `blake2xb_final`), which SEAL vendors: root hash with `xof_length` in parameter-block bytes 12..15, then one BLAKE2b call
  per 64 output bytes with {key 0, fanout 0, depth 0, leaf_length 64, node_offset i, node_depth 0, inner_length 64}.
  `hashlib` refuses depth 0, so this layer is restated from the published construction: **unpinned** (no known-answer
  vector for it is available in this environment).
* `Blake2xbPRNG` — seal/randomgen.cpp [SEAL-recall]: 4096-byte buffer = blake2xb(4096, counter as 8 LE bytes, key = the 64
  seed bytes), counter 0, 1, ...; `generate()` returns the next 4 bytes as a little-endian uint32.
"""
import struct

_IV = (0x6a09e667f3bcc908, 0xbb67ae8584caa73b, 0x3c6ef372fe94f82b, 0xa54ff53a5f1d36f1,
       0x510e527fade682d1, 0x9b05688c2b3e6c1f, 0x1f83d9abfb41bd6b, 0x5be0cd19137e2179)
_SIGMA = ((0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), (14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3),
          (11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4), (7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8),
          (9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13), (2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9),
          (12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11), (13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10),
          (6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5), (10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0))
_M = (1 << 64) - 1


def _rotr(x, r):
    return ((x >> r) | (x << (64 - r))) & _M


def _compress(h, block, t, last):
    m = struct.unpack("<16Q", block)
    v = list(h) + list(_IV)
    v[12] ^= t & _M
    v[13] ^= t >> 64
    if last:
        v[14] ^= _M

    def g(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & _M; v[d] = _rotr(v[d] ^ v[a], 32)
        v[c] = (v[c] + v[d]) & _M;     v[b] = _rotr(v[b] ^ v[c], 24)
        v[a] = (v[a] + v[b] + y) & _M; v[d] = _rotr(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & _M;     v[b] = _rotr(v[b] ^ v[c], 63)

    for r in range(12):
        s = _SIGMA[r % 10]
        g(0, 4, 8, 12, m[s[0]], m[s[1]]); g(1, 5, 9, 13, m[s[2]], m[s[3]])
        g(2, 6, 10, 14, m[s[4]], m[s[5]]); g(3, 7, 11, 15, m[s[6]], m[s[7]])
        g(0, 5, 10, 15, m[s[8]], m[s[9]]); g(1, 6, 11, 12, m[s[10]], m[s[11]])
        g(2, 7, 8, 13, m[s[12]], m[s[13]]); g(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def param_block(digest_length=64, key_length=0, fanout=1, depth=1, leaf_length=0, node_offset=0, xof_length=0, node_depth=0,
                inner_length=0, salt=b"", personal=b""):
    """the 64-byte BLAKE2b parameter block (BLAKE2X layout: node_offset 4 bytes + xof_length 4 bytes)"""
    return (struct.pack("<BBBBIIIBB", digest_length, key_length, fanout, depth, leaf_length, node_offset, xof_length, node_depth,
                        inner_length) + bytes(14) + salt.ljust(16, b"\0") + personal.ljust(16, b"\0"))


def blake2b(data, params, key=b""):
    """BLAKE2b of `data` under an explicit parameter block (bytes 0 and 1 give digest and key length)"""
    assert len(params) == 64 and params[1] == len(key)
    h = [iv ^ p for iv, p in zip(_IV, struct.unpack("<8Q", params))]
    if key:
        data = key.ljust(128, b"\0") + data
    t = 0
    while len(data) > 128:
        t += 128
        h = _compress(h, data[:128], t, False)
        data = data[128:]
    t += len(data)
    h = _compress(h, data.ljust(128, b"\0"), t, True)
    return struct.pack("<8Q", *h)[:params[0]]


def blake2xb(outlen, data, key=b""):
    root = blake2b(data, param_block(64, len(key), 1, 1, 0, 0, outlen, 0, 0), key)
    out = b""
    i = 0
    while len(out) < outlen:
        size = min(64, outlen - len(out))
        out += blake2b(root, param_block(size, 0, 0, 0, 64, i, outlen, 0, 64))
        i += 1
    return out


class Blake2xbPRNG:
    BUFFER = 4096

    def __init__(self, seed):
        """seed: 64 bytes, or eight 64-bit words (seal::prng_seed_type)"""
        self.seed = bytes(seed) if isinstance(seed, (bytes, bytearray)) else struct.pack("<8Q", *[int(w) for w in seed])
        assert len(self.seed) == 64
        self.counter = 0
        self.buf = b""
        self.head = 0

    def _refill(self):
        self.buf = blake2xb(self.BUFFER, struct.pack("<Q", self.counter), self.seed)
        self.counter += 1
        self.head = 0

    def generate(self):
        if self.head == len(self.buf):
            self._refill()
        v = struct.unpack_from("<I", self.buf, self.head)[0]
        self.head += 4
        return v

    def values(self, count, skip=0):
        for _ in range(skip):
            self.generate()
        return [self.generate() for _ in range(count)]
