// BLAKE2b compression (RFC 7693) and the BLAKE2Xb expansion, as SEAL's default PRNG uses them
// (seal/randomgen.cpp Blake2xbPRNG::refill_buffer -> blake2xb(buffer, 4096, &counter, 8, seed, 64); [SEAL-recall]).
// The reference draws the per-BinBundle masks from that generator (receiver/apsu/receiver_osn.cpp:221-224,248-251:
// UniformRandomGeneratorInfo(prng_type::blake2xb, seed).make_prng(), `generate() % plain_modulus` with a 32-bit generate()).
//
// Stream definition restated here: buffer c (c = 0, 1, ...) is 4096 bytes =
//   root   = BLAKE2b-512(key = the 64 seed bytes, message = c as 8 little-endian bytes), parameter block with xof_length = 4096
//   node i = BLAKE2b-512(message = root) under the parameter block {digest 64, key 0, fanout 0, depth 0, leaf_length 64,
//            node_offset i, xof_length 4096, node_depth 0, inner_length 64},  i = 0 .. 63
// and the generator hands out the buffers' bytes in order, four at a time as little-endian 32-bit values.
// __host__ __device__ so that the CPU test tier runs the same code against an independent Python model (oracle/blake2x.py,
// whose BLAKE2b core is checked against hashlib).
#pragma once
#include "modmath.h"

struct Blake2xbSeed { u64 w[8]; };              // seal::prng_seed_type: the 64 key bytes as eight little-endian words

HD u64 b2_rotr(u64 x, int r) { return (x >> r) | (x << (64 - r)); }

HD void b2_g(u64 &a, u64 &b, u64 &c, u64 &d, u64 x, u64 y)
{
    a = a + b + x; d = b2_rotr(d ^ a, 32);
    c = c + d;     b = b2_rotr(b ^ c, 24);
    a = a + b + y; d = b2_rotr(d ^ a, 16);
    c = c + d;     b = b2_rotr(b ^ c, 63);
}

// h <- F(h, m, t, last) for a message shorter than 2^64 bytes (t1 = 0)
HD void blake2b_compress(u64 h[8], const u64 m[16], u64 t0, bool last)
{
    constexpr u64 IV[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                            0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
    constexpr unsigned char S[12][16] = {
        { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
        { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
        { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
        { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
        { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 },
        { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 } };
    u64 v[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = IV[i]; }
    v[12] ^= t0;
    if (last) v[14] = ~v[14];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        b2_g(v[0], v[4], v[8], v[12], m[S[r][0]], m[S[r][1]]);
        b2_g(v[1], v[5], v[9], v[13], m[S[r][2]], m[S[r][3]]);
        b2_g(v[2], v[6], v[10], v[14], m[S[r][4]], m[S[r][5]]);
        b2_g(v[3], v[7], v[11], v[15], m[S[r][6]], m[S[r][7]]);
        b2_g(v[0], v[5], v[10], v[15], m[S[r][8]], m[S[r][9]]);
        b2_g(v[1], v[6], v[11], v[12], m[S[r][10]], m[S[r][11]]);
        b2_g(v[2], v[7], v[8], v[13], m[S[r][12]], m[S[r][13]]);
        b2_g(v[3], v[4], v[9], v[14], m[S[r][14]], m[S[r][15]]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
}

HD void blake2b_init(u64 h[8], u64 p0, u64 p1, u64 p2)
{
    constexpr u64 IV[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                            0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= p0; h[1] ^= p1; h[2] ^= p2;             // parameter-block words 3..7 (reserved, salt, personal) are zero
}

constexpr u64 BLAKE2XB_BUFFER_BYTES = 4096;        // SEAL's UniformRandomGenerator buffer size

// out[0..7] = the eight little-endian words of stream block `sb` (64 bytes: bytes sb*64 .. sb*64+63 of the generator's output)
HD void blake2xb_stream_block(const Blake2xbSeed &seed, u64 sb, u64 out[8])
{
    const u64 counter = sb >> 6, node = sb & 63;
    u64 m[16];
    // root: keyed BLAKE2b-512 of the counter; digest 64 | key 64 | fanout 1 | depth 1; xof_length 4096 in the upper half of word 1
    u64 root[8];
    blake2b_init(root, (u64)64 | ((u64)64 << 8) | ((u64)1 << 16) | ((u64)1 << 24), BLAKE2XB_BUFFER_BYTES << 32, 0);
#pragma unroll
    for (int i = 0; i < 8; i++) { m[i] = seed.w[i]; m[8 + i] = 0; }
    blake2b_compress(root, m, 128, false);          // the key, padded to one block
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = 0;
    m[0] = counter;
    blake2b_compress(root, m, 128 + 8, true);
    // expansion node: digest 64 | key 0 | fanout 0 | depth 0 | leaf_length 64 ; node_offset | xof_length ; node_depth 0 | inner_length 64
    blake2b_init(out, (u64)64 | ((u64)64 << 32), node | (BLAKE2XB_BUFFER_BYTES << 32), (u64)64 << 8);
#pragma unroll
    for (int i = 0; i < 8; i++) { m[i] = root[i]; m[8 + i] = 0; }
    blake2b_compress(out, m, 64, true);
}

// the generator's idx-th 32-bit output (UniformRandomGenerator::generate())
HD u32 blake2xb_stream_u32(const u64 block[8], unsigned idx_in_block) { return (u32)(block[idx_in_block >> 1] >> ((idx_in_block & 1) * 32)); }
