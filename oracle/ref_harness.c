/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ref_core.h header: PARITY UNPINNED).
 *
 * Harness-only pieces: what the OTHER party (query side) and the DB build do, in just enough
 * detail to manufacture valid inputs and to check outputs semantically:
 *   BatchEncoder encode/decode          [SEAL-recall batchencoder.cpp]  (B4)
 *   secret key / symmetric encryption   sender/apsu/plaintext_powers.cpp:41-46 (semantics only;
 *                                       not SEAL's PRNG stream — inputs are arbitrary valid cts)
 *   relinearisation keys                [SEAL-recall keygenerator.cpp generate_one_kswitch_key]
 *   decrypt + invariant noise budget    common/apsu/network/result_package.cpp:175-213
 *   polyn_with_roots                    common/apsu/util/interpolate.cpp:27-80
 * PRNG = splitmix64 -> xoshiro256** (SURVEY.md §8d).
 */
#include "ref_path.h"
#include <stdlib.h>
#include <string.h>

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void ref_rng_seed(ref_rng *r, uint64_t seed)
{
    for (int i = 0; i < 4; i++) {
        uint64_t z = (seed += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        r->s[i] = z ^ (z >> 31);
    }
}

uint64_t ref_rng_next(ref_rng *r)
{
    uint64_t *s = r->s;
    uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}

uint64_t ref_rng_below(ref_rng *r, uint64_t bound)
{
    /* rejection sampling, unbiased */
    uint64_t lim = UINT64_MAX - (UINT64_MAX % bound) - 1;
    uint64_t v;
    do { v = ref_rng_next(r); } while (v > lim);
    return v % bound;
}

void ref_fill_uniform(uint64_t seed, uint64_t bound, uint64_t *out, size_t count)
{
    ref_rng r; ref_rng_seed(&r, seed);
    for (size_t i = 0; i < count; i++) out[i] = ref_rng_below(&r, bound);
}

static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { uint64_t s = a + b; return s >= q ? s - q : s; }
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }

/* ternary secret, stored in NTT form over all K key limbs */
void ref_keygen(const ref_ctx *c, uint64_t seed, uint64_t *sk)
{
    ref_rng r; ref_rng_seed(&r, seed);
    size_t n = c->n;
    int8_t *s = (int8_t *)malloc(n);
    for (size_t k = 0; k < n; k++) s[k] = (int8_t)ref_rng_below(&r, 3) - 1;
    for (int j = 0; j < c->K; j++) {
        uint64_t q = c->key_q[j].value;
        for (size_t k = 0; k < n; k++) sk[(size_t)j * n + k] = s[k] < 0 ? q - 1 : (uint64_t)s[k];
        ref_ntt_fwd(&c->ntt_q[j], c->logn, sk + (size_t)j * n);
    }
    free(s);
}

/* centred binomial noise, sigma ~ 3.2 (21 coin pairs -> variance 10.5) */
static int sample_noise(ref_rng *r)
{
    uint64_t bits = ref_rng_next(r);
    return __builtin_popcountll(bits & 0x1FFFFF) - __builtin_popcountll((bits >> 21) & 0x1FFFFF);
}

/* (c0, c1) = (-(a s + e), a) over limbs [0, L) of the key modulus; is_ntt selects the output form */
static void encrypt_zero(const ref_ctx *c, const uint64_t *sk, ref_rng *r, int L, int is_ntt, uint64_t *ct)
{
    size_t n = c->n;
    int *e = (int *)malloc(sizeof(int) * n);
    for (size_t k = 0; k < n; k++) e[k] = sample_noise(r);
    uint64_t *c0 = ct, *c1 = ct + (size_t)L * n;
    for (int j = 0; j < L; j++) {
        const ref_mod *q = &c->key_q[j];
        uint64_t *a = c1 + (size_t)j * n, *b = c0 + (size_t)j * n;
        for (size_t k = 0; k < n; k++) a[k] = ref_rng_below(r, q->value);      /* a, NTT domain */
        for (size_t k = 0; k < n; k++) b[k] = e[k] < 0 ? q->value - (uint64_t)(-e[k]) : (uint64_t)e[k];
        ref_ntt_fwd(&c->ntt_q[j], c->logn, b);
        for (size_t k = 0; k < n; k++) {
            uint64_t as = ref_mulmod(a[k], sk[(size_t)j * n + k], q);
            b[k] = submod(0, addmod(as, b[k], q->value), q->value);
        }
        if (!is_ntt) { ref_ntt_inv(&c->ntt_q[j], c->logn, a); ref_ntt_inv(&c->ntt_q[j], c->logn, b); }
    }
    free(e);
}

void ref_encrypt_symmetric(const ref_ctx *c, const uint64_t *sk, const uint64_t *pt, uint64_t seed, uint64_t *ct)
{
    ref_rng r; ref_rng_seed(&r, seed);
    int first = c->first_chain_idx;
    encrypt_zero(c, sk, &r, first + 1, 0, ct);
    ref_add_plain(c, ct, pt, c->n, first);
}

void ref_gen_relin_keys(const ref_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *rk)
{
    ref_rng r; ref_rng_seed(&r, seed);
    int K = c->K; size_t n = c->n;
    uint64_t p = c->key_q[K - 1].value;
    for (int i = 0; i < K - 1; i++) {
        uint64_t *key = rk + (size_t)i * 2 * K * n;
        encrypt_zero(c, sk, &r, K, 1, key);
        const ref_mod *q = &c->key_q[i];
        uint64_t factor = p % q->value;
        uint64_t *c0i = key + (size_t)i * n;
        for (size_t k = 0; k < n; k++) {
            uint64_t s = sk[(size_t)i * n + k];
            uint64_t s2 = ref_mulmod(s, s, q);
            c0i[k] = addmod(c0i[k], ref_mulmod(s2, factor, q), q->value);
        }
    }
}

int ref_decrypt(const ref_ctx *c, const uint64_t *sk, const uint64_t *ct, int polys, int chain_idx, uint64_t *pt)
{
    size_t n = c->n;
    int L = chain_idx + 1;
    /* phase = c0 + c1 s + c2 s^2 per limb */
    uint64_t *ph = (uint64_t *)calloc((size_t)L * n, sizeof(uint64_t));
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (int j = 0; j < L; j++) {
        const ref_mod *q = &c->key_q[j];
        uint64_t *acc = ph + (size_t)j * n;
        for (int p = polys - 1; p >= 1; p--) {
            memcpy(tmp, ct + ((size_t)p * L + j) * n, sizeof(uint64_t) * n);
            ref_ntt_fwd(&c->ntt_q[j], c->logn, tmp);
            for (size_t k = 0; k < n; k++)
                acc[k] = ref_mulmod(addmod(acc[k], tmp[k], q->value), sk[(size_t)j * n + k], q);
        }
        ref_ntt_inv(&c->ntt_q[j], c->logn, acc);
        for (size_t k = 0; k < n; k++) acc[k] = addmod(acc[k], ct[(size_t)j * n + k], q->value);
    }
    /* bring to one limb by exact rounding division (adds < 1 bit of noise per step) */
    for (int l = chain_idx; l > 0; l--) ref_mod_switch_to_next(c, ph, 1, l);
    uint64_t q0 = c->key_q[0].value, t = c->t.value;
    u128 worst = 0;
    for (size_t k = 0; k < n; k++) {
        u128 num = (u128)ph[k] * t;
        uint64_t m = (uint64_t)((num + (q0 >> 1)) / q0);
        /* invariant noise ~ |t*x - m*q0| / q0 ; budget = log2(q0 / (2 * |t x mod q0|_centred)) */
        u128 rem = num % q0;
        u128 dist = rem > (q0 >> 1) ? (u128)q0 - rem : rem;
        if (dist > worst) worst = dist;
        pt[k] = m % t;
    }
    free(ph); free(tmp);
    int budget = 0;
    while (worst && ((worst << (budget + 1)) < (u128)q0)) budget++;
    if (!worst) budget = c->key_q[0].bits;
    return budget;
}

void ref_batch_encode(const ref_ctx *c, const uint64_t *values, uint64_t *pt)
{
    size_t n = c->n;
    for (size_t i = 0; i < n; i++) pt[c->slot_map[i]] = values[i];
    ref_ntt_inv(&c->ntt_t, c->logn, pt);
}

void ref_batch_decode(const ref_ctx *c, const uint64_t *pt, uint64_t *values)
{
    size_t n = c->n;
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    memcpy(tmp, pt, sizeof(uint64_t) * n);
    ref_ntt_fwd(&c->ntt_t, c->logn, tmp);
    for (size_t i = 0; i < n; i++) values[i] = tmp[c->slot_map[i]];
    free(tmp);
}

/* receiver/apsu/receiver_osn.cpp:53-73.  The reference computes the masks with `1 << len` on an int and shifts
 * uint64_t accumulators, which wrap; the odd-felt branch shifts the upper half by len/2 - 1 (sic).
 * The masks are formed here with 64-bit shifts: identical to the reference's `int` arithmetic for every plain
 * modulus below 2^31 (len <= 31), which covers all 36 shipped parameter sets (t < 2^27); for larger t the
 * reference's `1 << len` is undefined behaviour on int and no value is pinned. */
void ref_vec_to_oc_block(const uint64_t *in, size_t felts_per_item, uint64_t plain_modulus, uint64_t out[2])
{
    uint32_t len = 1;
    while ((((uint64_t)1 << len) - 1) < plain_modulus) len++;                      /* :54-57 */
    uint64_t mask = ((uint64_t)1 << len) - 1;                                       /* :58 */
    uint64_t mask_lower = ((uint64_t)1 << (len >> 1)) - 1;                          /* :59 */
    uint64_t mask_higher = mask - mask_lower;                                       /* :60 */
    uint64_t lower = 0, higher = 0;
    if (felts_per_item & 1) {                                                       /* :63-66 */
        lower = in[felts_per_item - 1] & mask_lower;
        higher = (in[felts_per_item - 1] & mask_higher) >> ((len >> 1) - 1);
    }
    for (size_t pla = 0; pla + 1 < felts_per_item; pla += 2) {                      /* :67-70 */
        lower = (in[pla] & mask) | (lower << len);
        higher = (in[pla + 1] & mask) | (higher << len);
    }
    out[0] = lower;
    out[1] = higher;
}

void ref_polyn_with_roots(const ref_ctx *c, const uint64_t *roots, size_t count, uint64_t *out)
{
    const ref_mod *t = &c->t;
    size_t len = 1;
    out[0] = 1;
    for (size_t r = 0; r < count; r++) {
        uint64_t neg_a = roots[r] ? t->value - roots[r] : 0;
        out[len] = 0;
        for (size_t i = len; i > 0; i--)
            out[i] = addmod(ref_mulmod(out[i], neg_a, t), out[i - 1], t->value);
        out[0] = ref_mulmod(out[0], neg_a, t);
        len++;
    }
}

/* common/apsu/util/db_encoding.cpp:209-256 (bits_to_field_elts) as called by algebraize_item (:360-366) on the first
   item_bit_count bits of a 16-byte hashed item: every field element takes the next bits_per_felt = bit_count(mod) - 1 bits
   (the last one what is left) of the bit string -- bit k of the string is bit k % 8 of byte k / 8 -- into the low bits of a
   little-endian 8-byte value.  Restated bit by bit (the reference copies byte fragments, copy_with_bit_offset :150-207).
   Returns the number of field elements, ceil(item_bit_count / bits_per_felt). */
int ref_algebraize_item(const unsigned char item[16], uint32_t item_bit_count, uint64_t plain_modulus, uint64_t *felts)
{
    int mod_bits = 0;
    for (uint64_t v = plain_modulus; v; v >>= 1) mod_bits++;
    if (mod_bits < 2 || !item_bit_count || item_bit_count > 128) return -1;
    uint32_t bits_per_felt = (uint32_t)mod_bits - 1;
    uint32_t num_felts = (item_bit_count + bits_per_felt - 1) / bits_per_felt;
    uint32_t left = item_bit_count, src = 0;
    for (uint32_t j = 0; j < num_felts; j++) {
        uint32_t copy = left < bits_per_felt ? left : bits_per_felt;
        uint64_t f = 0;
        for (uint32_t k = 0; k < copy; k++) {
            uint32_t bit = src + k;
            if ((item[bit >> 3] >> (bit & 7)) & 1) f |= (uint64_t)1 << k;
        }
        felts[j] = f;
        src += bits_per_felt;
        left -= copy;
    }
    return (int)num_felts;
}
