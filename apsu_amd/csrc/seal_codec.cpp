// See seal_codec.h.  UNPINNED restatement of Microsoft SEAL's object serialisation; host code only.
#include "seal_codec.h"

#include <dlfcn.h>
#include <zlib.h>

#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

#include "blake2x.h"
#include "keccak.h"

namespace apsu_he {
namespace sealio {

namespace {

[[noreturn]] void bad(const char *m) { throw std::runtime_error(std::string("failed to load SEAL object: ") + m); }

uint64_t rd64(const uint8_t *p) { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | p[i]; return v; }
void wr64(std::vector<uint8_t> &b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
// coefficient arrays: little-endian words on the wire = this host's memory layout (x86-64): one copy instead of a loop per byte
static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "the coefficient arrays are copied as little-endian words");
void rd64n(uint64_t *dst, const uint8_t *p, size_t count) { if (count) std::memcpy(dst, p, count * 8); }
void wr64n(std::vector<uint8_t> &b, const uint64_t *src, size_t count)
{
    const size_t at = b.size();
    b.resize(at + count * 8);
    if (count) std::memcpy(b.data() + at, src, count * 8);
}

constexpr size_t MAX_BODY = (size_t)1 << 30;                    // inflated size cap (zip bombs)

struct Header { uint8_t vmaj = 0, vmin = 0, compr = 0; uint64_t total = 0; };

// Zstandard, SEAL's default compr_mode when it is built with SEAL_USE_ZSTD (the default of its CMake): loaded at run time from the
// system's libzstd.so.1 -- the image has the library but not its headers, and the codec must not require it.  Only the stable
// one-shot compressor and the streaming decompressor are used (a SEAL writer streams its frame, ztools.cpp; any single frame is
// what its reader takes).  [SEAL-recall: one zstd frame of the member bytes, like the one deflate stream of compr zlib.]
struct Zstd {
    struct InBuf { const void *src; size_t size, pos; };          // ZSTD_inBuffer
    struct OutBuf { void *dst; size_t size, pos; };               // ZSTD_outBuffer
    void *lib = nullptr;
    size_t (*compress_bound)(size_t) = nullptr;
    size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    void *(*create_dstream)() = nullptr;
    size_t (*free_dstream)(void *) = nullptr;
    size_t (*init_dstream)(void *) = nullptr;
    size_t (*decompress_stream)(void *, OutBuf *, InBuf *) = nullptr;
    bool ok = false;
    Zstd()
    {
        for (const char *name : { "libzstd.so.1", "libzstd.so" }) if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) return;
        compress_bound = reinterpret_cast<decltype(compress_bound)>(dlsym(lib, "ZSTD_compressBound"));
        compress = reinterpret_cast<decltype(compress)>(dlsym(lib, "ZSTD_compress"));
        is_error = reinterpret_cast<decltype(is_error)>(dlsym(lib, "ZSTD_isError"));
        create_dstream = reinterpret_cast<decltype(create_dstream)>(dlsym(lib, "ZSTD_createDStream"));
        free_dstream = reinterpret_cast<decltype(free_dstream)>(dlsym(lib, "ZSTD_freeDStream"));
        init_dstream = reinterpret_cast<decltype(init_dstream)>(dlsym(lib, "ZSTD_initDStream"));
        decompress_stream = reinterpret_cast<decltype(decompress_stream)>(dlsym(lib, "ZSTD_decompressStream"));
        ok = compress_bound && compress && is_error && create_dstream && free_dstream && init_dstream && decompress_stream;
    }
};
const Zstd &zstd()
{
    static const Zstd z;                                          // (thread-safe initialisation; the handle stays open)
    return z;
}

Header read_header(const uint8_t *p, size_t avail)
{
    if (avail < 16) bad("truncated header");
    if ((p[0] | (p[1] << 8)) != 0xA15E || p[2] != 0x10) bad("bad magic");
    Header h;
    h.vmaj = p[3]; h.vmin = p[4]; h.compr = p[5];
    h.total = rd64(p + 8);
    if (h.total < 16 || h.total > avail) bad("bad size");
    if (!((h.vmaj == 3 && h.vmin >= 6) || h.vmaj == 4)) bad("unsupported SEAL version (need 3.6+ or 4.x)");
    return h;
}

void write_header(std::vector<uint8_t> &b, uint8_t vmaj, uint8_t vmin, uint8_t compr, uint64_t total)
{
    b.push_back(0x5E); b.push_back(0xA1); b.push_back(0x10); b.push_back(vmaj); b.push_back(vmin); b.push_back(compr);
    b.push_back(0); b.push_back(0);
    wr64(b, total);
}

// A byte range that is either a view into the caller's buffer (compr none) or an owned inflated copy.
struct Body {
    Header h;
    const uint8_t *p = nullptr;
    size_t n = 0;
    std::vector<uint8_t> owned;
};

Body open_object(const uint8_t *buf, size_t size)
{
    Body b;
    b.h = read_header(buf, size);
    const uint8_t *stored = buf + 16;
    const size_t stored_n = (size_t)b.h.total - 16;
    if (b.h.compr == COMPR_NONE) { b.p = stored; b.n = stored_n; return b; }
    if (b.h.compr == COMPR_ZSTD) {
        const Zstd &z = zstd();
        if (!z.ok) bad("zstd-compressed SEAL object, and libzstd.so.1 is not on this system; have the peer use zlib or none");
        void *ds = z.create_dstream();
        if (!ds) bad("zstd initialisation failed");
        struct Free { const Zstd &z; void *ds; ~Free() { z.free_dstream(ds); } } guard{ z, ds };
        if (z.is_error(z.init_dstream(ds))) bad("zstd initialisation failed");
        Zstd::InBuf in{ stored, stored_n, 0 };
        b.owned.resize(std::max<size_t>(4096, stored_n * 4));
        size_t have = 0, rc = 1;
        while (in.pos < in.size || rc != 0) {
            if (have == b.owned.size()) {
                if (b.owned.size() >= MAX_BODY) bad("inflated object too large");
                b.owned.resize(std::min(MAX_BODY, b.owned.size() * 2));
            }
            Zstd::OutBuf out{ b.owned.data() + have, b.owned.size() - have, 0 };
            const size_t in_before = in.pos;
            rc = z.decompress_stream(ds, &out, &in);
            if (z.is_error(rc)) bad("corrupt zstd stream");
            have += out.pos;
            if (rc != 0 && in.pos == in.size && out.pos == 0 && in.pos == in_before) bad("corrupt zstd stream (truncated frame)");
        }
        b.owned.resize(have);
        b.p = b.owned.data(); b.n = have;
        return b;
    }
    if (b.h.compr != COMPR_ZLIB) bad("unknown compression mode");
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) bad("zlib initialisation failed");
    zs.next_in = const_cast<Bytef *>(stored);
    zs.avail_in = (uInt)stored_n;
    if ((size_t)zs.avail_in != stored_n) { inflateEnd(&zs); bad("object too large"); }
    b.owned.resize(std::max<size_t>(4096, stored_n * 4));
    size_t have = 0;
    for (;;) {
        if (have == b.owned.size()) {
            if (b.owned.size() >= MAX_BODY) { inflateEnd(&zs); bad("inflated object too large"); }
            b.owned.resize(std::min(MAX_BODY, b.owned.size() * 2));
        }
        zs.next_out = b.owned.data() + have;
        zs.avail_out = (uInt)std::min<size_t>(b.owned.size() - have, 1u << 30);
        const size_t before = zs.avail_out;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        have += before - zs.avail_out;
        if (rc == Z_STREAM_END) break;
        if (rc != Z_OK || (zs.avail_in == 0 && zs.avail_out != 0)) { inflateEnd(&zs); bad("corrupt zlib stream"); }
    }
    inflateEnd(&zs);
    b.owned.resize(have);
    b.p = b.owned.data(); b.n = have;
    return b;
}

std::vector<uint8_t> close_object(const std::vector<uint8_t> &members, uint8_t vmaj, uint8_t vmin, uint8_t compr)
{
    std::vector<uint8_t> out;
    if (compr == COMPR_NONE) {
        write_header(out, vmaj, vmin, COMPR_NONE, 16 + members.size());
        out.insert(out.end(), members.begin(), members.end());
        return out;
    }
    if (compr == COMPR_ZSTD) {
        const Zstd &z = zstd();
        if (!z.ok) throw std::runtime_error("zstd requested, and libzstd.so.1 is not on this system");
        std::vector<uint8_t> f(z.compress_bound(members.size()));
        const size_t k = z.compress(f.data(), f.size(), members.data(), members.size(), 3 /* ZSTD_CLEVEL_DEFAULT */);
        if (z.is_error(k)) throw std::runtime_error("zstd compression failed");
        write_header(out, vmaj, vmin, COMPR_ZSTD, 16 + k);
        out.insert(out.end(), f.begin(), f.begin() + k);
        return out;
    }
    if (compr != COMPR_ZLIB) throw std::invalid_argument("unsupported compression mode for saving (none, zlib or zstd)");
    uLongf cap = compressBound((uLong)members.size());
    std::vector<uint8_t> z(cap);
    if (compress2(z.data(), &cap, members.data(), (uLong)members.size(), Z_DEFAULT_COMPRESSION) != Z_OK) throw std::runtime_error("zlib deflate failed");
    write_header(out, vmaj, vmin, COMPR_ZLIB, 16 + cap);
    out.insert(out.end(), z.begin(), z.begin() + cap);
    return out;
}

struct Cursor {
    const uint8_t *p; size_t n, at = 0;
    void need(size_t k) const { if (k > n - at) bad("truncated body"); }
    uint64_t u64() { need(8); const uint64_t v = rd64(p + at); at += 8; return v; }
    uint8_t u8() { need(1); return p[at++]; }
    const uint8_t *here() const { return p + at; }
    size_t left() const { return n - at; }
    void skip(size_t k) { need(k); at += k; }
};

const Level *find_level(const std::vector<Level> &chain, const uint64_t id[4])
{
    for (const Level &l : chain) if (!std::memcmp(l.parms_id, id, 32)) return &l;
    return nullptr;
}

void blake2b_256(const uint8_t *msg, size_t len, uint64_t out[4])
{
    u64 h[8];
    blake2b_init(h, (u64)32 | ((u64)1 << 16) | ((u64)1 << 24), 0, 0);      // digest 32, no key, fanout 1, depth 1
    size_t off = 0;
    u64 m[16];
    while (len - off > 128) {
        std::memcpy(m, msg + off, 128);
        off += 128;
        blake2b_compress(h, m, off, false);
    }
    std::memset(m, 0, sizeof(m));
    std::memcpy(m, msg + off, len - off);
    blake2b_compress(h, m, len, true);
    for (int i = 0; i < 4; i++) out[i] = h[i];
}

void parse_ciphertext_members(Cursor &c, uint8_t vmaj, uint8_t vmin, const std::vector<Level> &chain, Ciphertext &ct, bool expand = true)
{
    for (int i = 0; i < 4; i++) ct.parms_id[i] = c.u64();
    ct.is_ntt_form = c.u8();
    ct.size = c.u64(); ct.poly_modulus_degree = c.u64(); ct.coeff_modulus_size = c.u64();
    if (vmaj >= 4) ct.correction_factor = c.u64();
    { const uint64_t s = c.u64(); std::memcpy(&ct.scale, &s, 8); }
    ct.version_major = vmaj; ct.version_minor = vmin;
    if (ct.size > 64 || ct.coeff_modulus_size > 64 || ct.poly_modulus_degree > (1u << 20)) bad("implausible dimensions");
    const uint64_t poly_words = ct.coeff_modulus_size * ct.poly_modulus_degree, total = ct.size * poly_words;
    // the coefficient array: its own object
    Body arr = open_object(c.here(), c.left());
    c.skip((size_t)arr.h.total);
    Cursor a{ arr.p, arr.n };
    const uint64_t count = a.u64();
    // nothing is allocated before the peer's dimensions are backed by bytes that are really there: the array is either the whole
    // ciphertext or (seeded, size 2) its first polynomial, and the level named by parms_id must have coeff_modulus_size primes
    if (!(count == total || (ct.size == 2 && count == poly_words))) bad("inconsistent coefficient array size");
    a.need((size_t)count * 8);
    const Level *lv = find_level(chain, ct.parms_id);
    if (lv && (lv->q.size() != ct.coeff_modulus_size)) bad("coeff_modulus_size does not match the parms_id's level");
    // (total <= 2 * count and count * 8 bytes are present: the allocation below is at most twice the bytes received)
    ct.data.assign((size_t)total, 0);
    rd64n(ct.data.data(), a.here(), (size_t)count);
    if (count == total) { ct.seeded = false; return; }
    // seeded: the generator's description follows; c1 is what it samples
    Body info = open_object(c.here(), c.left());
    c.skip((size_t)info.h.total);
    Cursor ic{ info.p, info.n };
    const uint8_t type = ic.u8();
    if (type != PRNG_BLAKE2XB && type != PRNG_SHAKE256) bad("unknown generator type in a seeded ciphertext");
    ic.need(64);
    for (int i = 0; i < 8; i++) ct.seed[i] = rd64(ic.here() + 8 * i);
    ct.seeded = true;
    ct.prng_type = type;
    if (!expand && type == PRNG_BLAKE2XB) return;                  // left to the device (apsu_he_seed_expand)
    if (!lv) bad("parms_id of a seeded ciphertext is not in this context's modulus chain");
    sample_poly_uniform(ct.seed, lv->q.data(), lv->q.size(), (size_t)ct.poly_modulus_degree, ct.data.data() + poly_words, type);
    if (!expand) ct.seeded = false;                               // a Shake256 object handed to a caller that expands on the device: complete as it is
}

std::vector<uint8_t> ciphertext_members(const Ciphertext &ct)
{
    const uint64_t poly_words = ct.coeff_modulus_size * ct.poly_modulus_degree, total = ct.size * poly_words;
    if (ct.size > 64 || ct.coeff_modulus_size > 64 || ct.poly_modulus_degree > (1u << 20)) throw std::invalid_argument("implausible ciphertext dimensions");
    if (ct.data.size() != total) throw std::invalid_argument("ciphertext data size does not match its dimensions");
    if (ct.seeded && ct.size != 2) throw std::invalid_argument("only size-2 ciphertexts can be saved seeded");
    std::vector<uint8_t> m;
    for (int i = 0; i < 4; i++) wr64(m, ct.parms_id[i]);
    m.push_back(ct.is_ntt_form ? 1 : 0);
    wr64(m, ct.size); wr64(m, ct.poly_modulus_degree); wr64(m, ct.coeff_modulus_size);
    if (ct.version_major >= 4) wr64(m, ct.correction_factor);
    { uint64_t s; std::memcpy(&s, &ct.scale, 8); wr64(m, s); }
    const uint64_t count = ct.seeded ? poly_words : total;
    std::vector<uint8_t> arr;
    wr64(arr, count);
    wr64n(arr, ct.data.data(), (size_t)count);
    const std::vector<uint8_t> ao = close_object(arr, ct.version_major, ct.version_minor, COMPR_NONE);
    m.insert(m.end(), ao.begin(), ao.end());
    if (ct.seeded) {
        std::vector<uint8_t> info;
        info.push_back(ct.prng_type);                             // prng_type::blake2xb (1) or shake256 (2)
        for (int i = 0; i < 8; i++) wr64(info, ct.seed[i]);
        const std::vector<uint8_t> io = close_object(info, ct.version_major, ct.version_minor, COMPR_NONE);
        m.insert(m.end(), io.begin(), io.end());
    }
    return m;
}

} // namespace

void compute_parms_id(uint64_t out[4], uint64_t scheme, uint64_t n, const uint64_t *q, size_t count, uint64_t t)
{
    std::vector<uint8_t> msg;
    wr64(msg, scheme); wr64(msg, n);
    for (size_t i = 0; i < count; i++) wr64(msg, q[i]);
    wr64(msg, t);
    blake2b_256(msg.data(), msg.size(), out);
}

std::vector<Level> modulus_chain(uint64_t n, const std::vector<uint64_t> &key_moduli, uint64_t t)
{
    std::vector<Level> chain;
    const size_t K = key_moduli.size();
    auto add = [&](size_t cnt) {
        Level l;
        l.q.assign(key_moduli.begin(), key_moduli.begin() + cnt);
        compute_parms_id(l.parms_id, 1 /* scheme_type::bfv */, n, l.q.data(), cnt, t);
        chain.push_back(std::move(l));
    };
    add(K);
    for (size_t cnt = K > 1 ? K - 1 : 0; cnt >= 1; cnt--) add(cnt);
    return chain;
}

void sample_poly_uniform(const uint64_t seed[8], const uint64_t *q, size_t L, size_t n, uint64_t *dst, uint8_t prng_type)
{
    if (prng_type == PRNG_SHAKE256) {
        // Shake256PRNG::refill_buffer (randomgen.cpp [SEAL-recall]): 4096-byte buffer k = SHAKE256(seed || k), consumed in order
        std::vector<uint64_t> buf(512);
        uint64_t have = ~(uint64_t)0;
        auto word = [&](uint64_t w) -> uint64_t {
            if ((w >> 9) != have) {
                have = w >> 9;
                uint64_t in[9];
                for (int i = 0; i < 8; i++) in[i] = seed[i];
                in[8] = have;
                keccak::shake256(reinterpret_cast<uint8_t *>(buf.data()), 4096, reinterpret_cast<const uint8_t *>(in), sizeof(in));
            }
            return buf[w & 511];
        };
        const uint64_t bulk = (uint64_t)L * n;
        for (uint64_t w = 0; w < bulk; w++) dst[w] = word(w);
        uint64_t next = bulk;
        for (size_t j = 0; j < L; j++) {
            const uint64_t max_random = ~(uint64_t)0;
            const uint64_t max_multiple = max_random - (max_random % q[j]) - 1;
            uint64_t *p = dst + j * n;
            for (size_t k = 0; k < n; k++) {
                uint64_t r = p[k];
                while (r >= max_multiple) r = word(next++);
                p[k] = r % q[j];
            }
        }
        return;
    }
    if (prng_type != PRNG_BLAKE2XB) throw std::invalid_argument("unknown generator type");
    Blake2xbSeed s;
    for (int i = 0; i < 8; i++) s.w[i] = seed[i];
    uint64_t blk[8];
    uint64_t cur = ~(uint64_t)0;                                  // stream block held in blk
    auto word = [&](uint64_t w) -> uint64_t {                     // the w-th 64-bit word of the generator's output
        if ((w >> 3) != cur) { cur = w >> 3; blake2xb_stream_block(s, cur, blk); }
        return blk[w & 7];
    };
    const uint64_t bulk = (uint64_t)L * n;
    for (uint64_t w = 0; w < bulk; w++) dst[w] = word(w);
    uint64_t next = bulk;                                         // fresh draws continue behind the bulk fill
    for (size_t j = 0; j < L; j++) {
        const uint64_t max_random = ~(uint64_t)0;
        const uint64_t max_multiple = max_random - (max_random % q[j]) - 1;
        uint64_t *p = dst + j * n;
        for (size_t k = 0; k < n; k++) {
            uint64_t r = p[k];
            while (r >= max_multiple) r = word(next++);
            p[k] = r % q[j];
        }
    }
}

Ciphertext load_ciphertext(const uint8_t *buf, size_t size, const std::vector<Level> &chain, size_t *consumed, bool expand)
{
    if (!buf) bad("null buffer");
    Body b = open_object(buf, size);
    Cursor c{ b.p, b.n };
    Ciphertext ct;
    parse_ciphertext_members(c, b.h.vmaj, b.h.vmin, chain, ct, expand);
    if (consumed) *consumed = (size_t)b.h.total;
    return ct;
}

std::vector<uint8_t> save_ciphertext(const Ciphertext &ct, uint8_t compr)
{
    return close_object(ciphertext_members(ct), ct.version_major, ct.version_minor, compr);
}

EncryptionParameters load_encryption_parameters(const uint8_t *buf, size_t size, size_t *consumed)
{
    if (!buf) bad("null buffer");
    Body b = open_object(buf, size);
    Cursor c{ b.p, b.n };
    EncryptionParameters p;
    p.version_major = b.h.vmaj; p.version_minor = b.h.vmin;
    p.scheme = c.u8();
    p.poly_modulus_degree = c.u64();
    const uint64_t count = c.u64();
    if (count > 64 || p.poly_modulus_degree > (1u << 20)) bad("implausible encryption parameters");
    auto modulus = [&]() {
        Body m = open_object(c.here(), c.left());
        c.skip((size_t)m.h.total);
        Cursor mc{ m.p, m.n };
        return mc.u64();
    };
    for (uint64_t i = 0; i < count; i++) p.coeff_modulus.push_back(modulus());
    p.plain_modulus = modulus();
    if (consumed) *consumed = (size_t)b.h.total;
    return p;
}

std::vector<uint8_t> save_encryption_parameters(const EncryptionParameters &p, uint8_t compr)
{
    std::vector<uint8_t> m;
    m.push_back(p.scheme);
    wr64(m, p.poly_modulus_degree);
    wr64(m, p.coeff_modulus.size());
    auto modulus = [&](uint64_t v) {
        std::vector<uint8_t> val;
        wr64(val, v);
        const std::vector<uint8_t> o = close_object(val, p.version_major, p.version_minor, COMPR_NONE);
        m.insert(m.end(), o.begin(), o.end());
    };
    for (uint64_t q : p.coeff_modulus) modulus(q);
    modulus(p.plain_modulus);
    return close_object(m, p.version_major, p.version_minor, compr);
}

Plaintext load_plaintext(const uint8_t *buf, size_t size, size_t *consumed)
{
    if (!buf) bad("null buffer");
    Body b = open_object(buf, size);
    Cursor c{ b.p, b.n };
    Plaintext pt;
    for (int i = 0; i < 4; i++) pt.parms_id[i] = c.u64();
    pt.coeff_count = c.u64();
    { const uint64_t s = c.u64(); std::memcpy(&pt.scale, &s, 8); }
    pt.version_major = b.h.vmaj; pt.version_minor = b.h.vmin;
    if (pt.coeff_count > ((uint64_t)64 << 20)) bad("implausible plaintext size");
    Body arr = open_object(c.here(), c.left());
    Cursor a{ arr.p, arr.n };
    const uint64_t count = a.u64();
    if (count != pt.coeff_count) bad("plaintext coefficient array does not match coeff_count");
    a.need((size_t)count * 8);
    pt.data.resize((size_t)count);
    rd64n(pt.data.data(), a.here(), (size_t)count);
    if (consumed) *consumed = (size_t)b.h.total;
    return pt;
}

std::vector<uint8_t> save_plaintext(const Plaintext &pt, uint8_t compr)
{
    if (pt.data.size() != pt.coeff_count) throw std::invalid_argument("plaintext data size does not match coeff_count");
    std::vector<uint8_t> m;
    for (int i = 0; i < 4; i++) wr64(m, pt.parms_id[i]);
    wr64(m, pt.coeff_count);
    { uint64_t s; std::memcpy(&s, &pt.scale, 8); wr64(m, s); }
    std::vector<uint8_t> arr;
    wr64(arr, pt.coeff_count);
    wr64n(arr, pt.data.data(), pt.data.size());
    const std::vector<uint8_t> ao = close_object(arr, pt.version_major, pt.version_minor, COMPR_NONE);
    m.insert(m.end(), ao.begin(), ao.end());
    return close_object(m, pt.version_major, pt.version_minor, compr);
}

KSwitchKeys load_kswitch_keys(const uint8_t *buf, size_t size, const std::vector<Level> &chain, size_t *consumed, bool expand)
{
    if (!buf) bad("null buffer");
    Body b = open_object(buf, size);
    Cursor c{ b.p, b.n };
    KSwitchKeys k;
    k.version_major = b.h.vmaj; k.version_minor = b.h.vmin;
    for (int i = 0; i < 4; i++) k.parms_id[i] = c.u64();
    const uint64_t dim1 = c.u64();
    if (dim1 > 64) bad("implausible key count");
    k.keys.resize((size_t)dim1);
    for (uint64_t i = 0; i < dim1; i++) {
        const uint64_t dim2 = c.u64();
        if (dim2 > 64) bad("implausible decomposition count");
        for (uint64_t j = 0; j < dim2; j++) {
            // PublicKey::save forwards to its Ciphertext's save: ONE header per key, the ciphertext members right behind it
            Body cb = open_object(c.here(), c.left());
            c.skip((size_t)cb.h.total);
            Cursor cc{ cb.p, cb.n };
            Ciphertext ct;
            parse_ciphertext_members(cc, cb.h.vmaj, cb.h.vmin, chain, ct, expand);
            k.keys[(size_t)i].push_back(std::move(ct));
        }
    }
    if (consumed) *consumed = (size_t)b.h.total;
    return k;
}

std::vector<uint8_t> save_kswitch_keys(const KSwitchKeys &k, uint8_t compr)
{
    std::vector<uint8_t> m;
    for (int i = 0; i < 4; i++) wr64(m, k.parms_id[i]);
    wr64(m, k.keys.size());
    for (const auto &row : k.keys) {
        wr64(m, row.size());
        for (const Ciphertext &ct : row) {
            const std::vector<uint8_t> pk = close_object(ciphertext_members(ct), ct.version_major, ct.version_minor, COMPR_NONE);
            m.insert(m.end(), pk.begin(), pk.end());
        }
    }
    return close_object(m, k.version_major, k.version_minor, compr);
}

std::vector<uint64_t> relin_keys_layout(const KSwitchKeys &k, size_t K, size_t n)
{
    if (k.keys.empty() || k.keys[0].empty()) throw std::invalid_argument("RelinKeys object holds no key");
    const auto &row = k.keys[0];
    if (K < 2 || row.size() != K - 1) throw std::invalid_argument("RelinKeys decomposition count does not match the parameters");
    std::vector<uint64_t> out;
    out.reserve(row.size() * 2 * K * n);
    for (const Ciphertext &ct : row) {
        if (ct.size != 2 || ct.coeff_modulus_size != K || ct.poly_modulus_degree != n || !ct.is_ntt_form)
            throw std::invalid_argument("RelinKeys entry is not a size-2 NTT-form ciphertext at the key level");
        out.insert(out.end(), ct.data.begin(), ct.data.end());
    }
    return out;
}

} // namespace sealio
} // namespace apsu_he
