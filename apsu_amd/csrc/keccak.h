// SHAKE256 (FIPS 202) for SEAL's second generator type (prng_type::shake256): host code only, used by the SEAL object codec
// (seal_codec.cpp) to expand seeded objects written by a SEAL built with SEAL_DEFAULT_PRNG=Shake256.  Pinned by python's
// hashlib.shake_256 in tests/test_seal_codec.py.  SEAL's use of it (randomgen.cpp, restated from memory, UNPINNED like the rest of
// the codec): buffer k of the stream = SHAKE256(seed (64 bytes) || k as u64 little-endian) squeezed to 4096 bytes, k = 0, 1, ...
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace apsu_he {
namespace keccak {

inline uint64_t rotl(uint64_t x, int s) { return (x << s) | (x >> (64 - s)); }

inline void f1600(uint64_t st[25])
{
    static const uint64_t RC[24] = { 0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                                     0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                                     0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                                     0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                     0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL };
    static const int ROT[24] = { 1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44 };
    static const int PIL[24] = { 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1 };
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            const uint64_t t = bc[(i + 4) % 5] ^ rotl(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; i++) {
            const int j = PIL[i];
            const uint64_t b = st[j];
            st[j] = rotl(t, ROT[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
}

// out[0 .. outlen) = SHAKE256(in[0 .. inlen)); little-endian host assumed (as everywhere in this library)
inline void shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen)
{
    constexpr size_t RATE = 136;
    uint64_t st[25];
    std::memset(st, 0, sizeof(st));
    uint8_t *sb = reinterpret_cast<uint8_t *>(st);
    while (inlen >= RATE) {
        for (size_t i = 0; i < RATE; i++) sb[i] ^= in[i];
        f1600(st);
        in += RATE; inlen -= RATE;
    }
    for (size_t i = 0; i < inlen; i++) sb[i] ^= in[i];
    sb[inlen] ^= 0x1f;
    sb[RATE - 1] ^= 0x80;
    f1600(st);
    while (outlen > 0) {
        const size_t take = outlen < RATE ? outlen : RATE;
        std::memcpy(out, sb, take);
        out += take; outlen -= take;
        if (outlen) f1600(st);
    }
}

} // namespace keccak
} // namespace apsu_he
