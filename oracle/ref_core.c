/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ref_core.h header: PARITY UNPINNED).
 *
 * Small-modulus arithmetic, prime selection, NTT tables, modulus-chain / RNS-tool constants
 * and the per-operation restatements of the seal::Evaluator methods that APSU calls at
 *   receiver/apsu/receiver_osn.cpp:422,424,431,463,467,471,475,478
 *   receiver/apsu/bin_bundle.cpp:143-148,154-162,169,252-273,281-303,309,315-323,329-336,340-346,354-357
 * Algorithms follow SURVEY.md App. B (B1..B10) = published Microsoft SEAL, tagged [SEAL-recall].
 */
#include "ref_core.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ------------------------------------------------------------------ small modulus (B2) */

void ref_mod_init(ref_mod *m, uint64_t value)
{
    m->value = value;
    m->bits = 64 - __builtin_clzll(value);
    /* floor(2^128 / value) : 2^128 = (2^128 - 1) + 1 */
    u128 all = ~(u128)0;
    u128 r = all / value;
    if (all % value == (u128)value - 1) r += 1;   /* only when value | 2^128 (power of two) */
    m->ratio[0] = (uint64_t)r;
    m->ratio[1] = (uint64_t)(r >> 64);
}

/* [SEAL-recall util/uintarithsmallmod.h barrett_reduce_128] */
uint64_t ref_bred128(u128 x, const ref_mod *q)
{
    uint64_t x0 = (uint64_t)x, x1 = (uint64_t)(x >> 64);
    uint64_t r0 = q->ratio[0], r1 = q->ratio[1];
    uint64_t carry = (uint64_t)(((u128)x0 * r0) >> 64);
    u128 t2 = (u128)x0 * r1;
    u128 s = (u128)(uint64_t)t2 + carry;
    uint64_t tmp1 = (uint64_t)s;
    uint64_t tmp3 = (uint64_t)(t2 >> 64) + (uint64_t)(s >> 64);
    t2 = (u128)x1 * r0;
    s = (u128)tmp1 + (uint64_t)t2;
    carry = (uint64_t)(t2 >> 64) + (uint64_t)(s >> 64);
    tmp1 = x1 * r1 + tmp3 + carry;
    tmp3 = x0 - tmp1 * q->value;
    return tmp3 >= q->value ? tmp3 - q->value : tmp3;
}

/* [SEAL-recall barrett_reduce_64] */
uint64_t ref_bred64(uint64_t x, const ref_mod *q)
{
    uint64_t hi = (uint64_t)(((u128)x * q->ratio[1]) >> 64);
    uint64_t r = x - hi * q->value;
    return r >= q->value ? r - q->value : r;
}

uint64_t ref_mulmod(uint64_t a, uint64_t b, const ref_mod *q)
{
    return ref_bred128((u128)a * b, q);
}

static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q)
{
    uint64_t s = a + b;
    return s >= q ? s - q : s;
}
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q)
{
    return a >= b ? a - b : a + q - b;
}
static inline uint64_t negmod(uint64_t a, uint64_t q) { return a ? q - a : 0; }

uint64_t ref_powmod(uint64_t a, uint64_t e, const ref_mod *q)
{
    uint64_t r = 1 % q->value;
    a %= q->value;
    while (e) {
        if (e & 1) r = (uint64_t)(((u128)r * a) % q->value);
        a = (uint64_t)(((u128)a * a) % q->value);
        e >>= 1;
    }
    return r;
}

/* generic modular inverse (extended Euclid); m may be composite (m_tilde = 2^32) */
int ref_invmod(uint64_t a, uint64_t m, uint64_t *out)
{
    __int128 t0 = 0, t1 = 1;
    __int128 r0 = m, r1 = a % m;
    while (r1 != 0) {
        __int128 qq = r0 / r1;
        __int128 tt = t0 - qq * t1; t0 = t1; t1 = tt;
        __int128 rr = r0 - qq * r1; r0 = r1; r1 = rr;
    }
    if (r0 != 1) return -1;
    if (t0 < 0) t0 += m;
    *out = (uint64_t)t0;
    return 0;
}

/* Shoup quotient floor(w * 2^64 / q)  [SEAL-recall MultiplyUIntModOperand] */
static inline uint64_t shoup(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }
static inline uint64_t mulmod_shoup(uint64_t x, uint64_t w, uint64_t wq, uint64_t q)
{
    uint64_t h = (uint64_t)(((u128)x * wq) >> 64);
    uint64_t r = x * w - h * q;
    return r >= q ? r - q : r;
}

/* ------------------------------------------------------------------ primes (B1) */

/* deterministic Miller-Rabin for 64-bit integers */
int ref_is_prime(uint64_t v)
{
    if (v < 2) return 0;
    static const uint64_t small[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++) {
        if (v == small[i]) return 1;
        if (v % small[i] == 0) return 0;
    }
    uint64_t d = v - 1;
    int r = 0;
    while (!(d & 1)) { d >>= 1; r++; }
    ref_mod m; ref_mod_init(&m, v);
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++) {
        uint64_t x = ref_powmod(small[i], d, &m);
        if (x == 1 || x == v - 1) continue;
        int comp = 1;
        for (int j = 1; j < r; j++) {
            x = (uint64_t)(((u128)x * x) % v);
            if (x == v - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

/* [SEAL-recall util/numth.cpp get_primes]: descending search over v == 1 (mod factor) */
int ref_get_primes(uint64_t factor, int bits, int count, uint64_t *out)
{
    uint64_t lower = (uint64_t)1 << (bits - 1);
    uint64_t v = (((uint64_t)1 << bits) - 1) / factor * factor + 1;
    int got = 0;
    while (got < count && v > lower) {
        if (ref_is_prime(v)) out[got++] = v;
        v -= factor;
    }
    return got == count ? 0 : -1;
}

/* [SEAL-recall modulus.cpp CoeffModulus::Create]: per bit-size list, entries take the BACK
   (smallest remaining) of their size's list, in the order given. */
int ref_coeff_modulus_create(int n, const int *bits, int count, uint64_t *out)
{
    int cnt[64] = { 0 };
    uint64_t *lists[64] = { 0 };
    for (int i = 0; i < count; i++) {
        if (bits[i] < 2 || bits[i] > 60) return -1;
        cnt[bits[i]]++;
    }
    int rc = 0;
    for (int b = 0; b < 64; b++) {
        if (!cnt[b]) continue;
        lists[b] = (uint64_t *)malloc(sizeof(uint64_t) * cnt[b]);
        if (ref_get_primes(2 * (uint64_t)n, b, cnt[b], lists[b])) rc = -1;
    }
    if (!rc) {
        for (int i = 0; i < count; i++) out[i] = lists[bits[i]][--cnt[bits[i]]];
    }
    for (int b = 0; b < 64; b++) free(lists[b]);
    return rc;
}

/* [SEAL-recall numth.cpp try_minimal_primitive_root]: smallest among all primitive
   degree-th roots of unity mod q (degree a power of two dividing q-1). */
uint64_t ref_minimal_primitive_root(uint64_t degree, const ref_mod *q)
{
    uint64_t qv = q->value;
    uint64_t exp = (qv - 1) / degree;
    uint64_t root = 0;
    for (uint64_t g = 2; g < qv; g++) {
        uint64_t r = ref_powmod(g, exp, q);
        /* primitive iff r^(degree/2) == -1 */
        if (ref_powmod(r, degree / 2, q) == qv - 1) { root = r; break; }
    }
    uint64_t gsq = ref_mulmod(root, root, q);
    uint64_t cur = root, best = root;
    for (uint64_t i = 0; i < degree / 2; i++) {
        if (cur < best) best = cur;
        cur = ref_mulmod(cur, gsq, q);
    }
    return best;
}

/* ------------------------------------------------------------------ NTT (B3) */

static inline uint32_t brv(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

static void ntt_tables_init(ref_ntt *t, uint64_t qv, int n, int logn)
{
    ref_mod_init(&t->q, qv);
    t->psi = ref_minimal_primitive_root(2 * (uint64_t)n, &t->q);
    t->w = (uint64_t *)malloc(sizeof(uint64_t) * n * 4);
    t->wq = t->w + n; t->iw = t->w + 2 * n; t->iwq = t->w + 3 * n;
    uint64_t ipsi = 0;
    ref_invmod(t->psi, qv, &ipsi);
    uint64_t p = 1, ip = 1;
    for (int i = 0; i < n; i++) {
        uint32_t k = brv((uint32_t)i, logn);
        t->w[k] = p;   t->wq[k] = shoup(p, qv);
        t->iw[k] = ip; t->iwq[k] = shoup(ip, qv);
        p = ref_mulmod(p, t->psi, &t->q);
        ip = ref_mulmod(ip, ipsi, &t->q);
    }
    ref_invmod((uint64_t)n % qv, qv, &t->inv_n);
    t->inv_n_q = shoup(t->inv_n, qv);
}

/* Cooley-Tukey, natural in -> bit-reversed out: out[i] = a(psi^(2*brv(i)+1)) */
void ref_ntt_fwd(const ref_ntt *t, int logn, uint64_t *a)
{
    size_t n = (size_t)1 << logn;
    uint64_t q = t->q.value;
    size_t gap = n >> 1;
    for (size_t m = 1; m < n; m <<= 1, gap >>= 1) {
        for (size_t i = 0; i < m; i++) {
            uint64_t w = t->w[m + i], wq = t->wq[m + i];
            uint64_t *x = a + 2 * i * gap, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                uint64_t u = x[j];
                uint64_t v = mulmod_shoup(y[j], w, wq, q);
                x[j] = addmod(u, v, q);
                y[j] = submod(u, v, q);
            }
        }
    }
}

/* Gentleman-Sande inverse of the above, incl. n^{-1}: bit-reversed in -> natural out */
void ref_ntt_inv(const ref_ntt *t, int logn, uint64_t *a)
{
    size_t n = (size_t)1 << logn;
    uint64_t q = t->q.value;
    size_t gap = 1;
    for (size_t m = n >> 1; m >= 1; m >>= 1, gap <<= 1) {
        for (size_t i = 0; i < m; i++) {
            uint64_t w = t->iw[m + i], wq = t->iwq[m + i];
            uint64_t *x = a + 2 * i * gap, *y = x + gap;
            for (size_t j = 0; j < gap; j++) {
                uint64_t u = x[j], v = y[j];
                x[j] = addmod(u, v, q);
                y[j] = mulmod_shoup(submod(u, v, q), w, wq, q);
            }
        }
    }
    for (size_t j = 0; j < n; j++) a[j] = mulmod_shoup(a[j], t->inv_n, t->inv_n_q, q);
}

/* ------------------------------------------------------------------ tiny bignum */

#define BN 12
typedef struct { uint64_t w[BN]; } bn_t;
static void bn_set(bn_t *a, uint64_t v) { memset(a, 0, sizeof(*a)); a->w[0] = v; }
static void bn_mul64(bn_t *a, uint64_t m)
{
    uint64_t carry = 0;
    for (int i = 0; i < BN; i++) {
        u128 p = (u128)a->w[i] * m + carry;
        a->w[i] = (uint64_t)p; carry = (uint64_t)(p >> 64);
    }
}
static uint64_t bn_divmod64(bn_t *a, uint64_t d)   /* a /= d, returns remainder */
{
    u128 rem = 0;
    for (int i = BN - 1; i >= 0; i--) {
        u128 cur = (rem << 64) | a->w[i];
        a->w[i] = (uint64_t)(cur / d);
        rem = cur % d;
    }
    return (uint64_t)rem;
}
static uint64_t bn_mod64(const bn_t *a, uint64_t d) { bn_t c = *a; return bn_divmod64(&c, d); }
static int bn_bits(const bn_t *a)
{
    for (int i = BN - 1; i >= 0; i--)
        if (a->w[i]) return 64 * i + 64 - __builtin_clzll(a->w[i]);
    return 0;
}

/* ------------------------------------------------------------------ base converter (B9) */

static uint64_t prod_mod_except(const ref_mod *base, int nb, int skip, const ref_mod *m)
{
    uint64_t r = 1 % m->value;
    for (int k = 0; k < nb; k++) {
        if (k == skip) continue;
        r = (uint64_t)(((u128)r * (base[k].value % m->value)) % m->value);
    }
    return r;
}

static void bconv_init(ref_bconv *bc, const ref_mod *ib, int ni, const ref_mod *ob, int no)
{
    bc->ni = ni; bc->no = no;
    for (int i = 0; i < ni; i++) bc->ib[i] = ib[i];
    for (int o = 0; o < no; o++) bc->ob[o] = ob[o];
    for (int i = 0; i < ni; i++) {
        uint64_t punct = prod_mod_except(ib, ni, i, &ib[i]);
        ref_invmod(punct, ib[i].value, &bc->inv_punct[i]);
        for (int o = 0; o < no; o++) bc->matrix[o][i] = prod_mod_except(ib, ni, i, &ob[o]);
    }
}

/* [SEAL-recall BaseConverter::fast_convert_array]: in [ni][n] -> out [no][n],
   out_o = sum_i [x_i * (Q/q_i)^-1]_{q_i} * ((Q/q_i) mod p_o)  mod p_o  (no correction) */
static void bconv_apply(const ref_bconv *bc, const uint64_t *in, uint64_t *out, size_t n)
{
    uint64_t tmp[REF_MAXB];
    for (size_t k = 0; k < n; k++) {
        for (int i = 0; i < bc->ni; i++)
            tmp[i] = ref_mulmod(in[(size_t)i * n + k], bc->inv_punct[i], &bc->ib[i]);
        for (int o = 0; o < bc->no; o++) {
            /* dot_product_mod: lazy 128-bit accumulate, reduce at the end (ni <= 10 terms of
               < 2^125 would overflow for 61x64 bit; reduce each product instead -- same value) */
            uint64_t acc = 0;
            for (int i = 0; i < bc->ni; i++)
                acc = addmod(acc, ref_mulmod(tmp[i], bc->matrix[o][i], &bc->ob[o]), bc->ob[o].value);
            out[(size_t)o * n + k] = acc;
        }
    }
}

/* ------------------------------------------------------------------ context */

static int level_init(ref_ctx *c, ref_level *lv, int L, const uint64_t *baseconv_primes)
{
    int n = c->n;
    uint64_t t = c->t.value;
    lv->L = L;
    for (int j = 0; j < L; j++) lv->q[j] = c->key_q[j];

    /* Q as bignum */
    bn_t Q; bn_set(&Q, 1);
    for (int j = 0; j < L; j++) bn_mul64(&Q, lv->q[j].value);
    int q_bits = bn_bits(&Q);
    bn_t Qdiv = Q;
    lv->q_mod_t = bn_divmod64(&Qdiv, t);                 /* Qdiv = floor(Q/t) */
    lv->upper_half_threshold = (t + 1) >> 1;
    for (int j = 0; j < L; j++) {
        lv->coeff_div_plain[j] = bn_mod64(&Qdiv, lv->q[j].value);
        if (lv->q[j].value <= t) return -1;              /* fast plain lift only */
        lv->upper_half_incr[j] = lv->q[j].value - t;
    }
    for (int j = 0; j + 1 < L; j++)
        ref_invmod(lv->q[L - 1].value % lv->q[j].value, lv->q[j].value, &lv->inv_q_last_mod_q[j]);

    /* RNSTool::initialize [SEAL-recall]: |B| = L, +1 if 32 + bits(t) + bits(Q) >= 61*L + 61 */
    int nB = L;
    if (32 + c->t.bits + q_bits >= 61 * L + 61) nB++;
    if (nB + 2 > REF_MAXB) return -1;
    lv->nB = nB;
    ref_mod_init(&lv->m_sk, baseconv_primes[0]);
    ref_mod_init(&lv->gamma, baseconv_primes[1]);
    for (int i = 0; i < nB; i++) ref_mod_init(&lv->B[i], baseconv_primes[2 + i]);
    ref_mod_init(&lv->m_tilde, (uint64_t)1 << 32);
    for (int i = 0; i < nB; i++) lv->Bsk[i] = lv->B[i];
    lv->Bsk[nB] = lv->m_sk;

    bconv_init(&lv->q_to_Bsk, lv->q, L, lv->Bsk, nB + 1);
    bconv_init(&lv->q_to_mtilde, lv->q, L, &lv->m_tilde, 1);
    bconv_init(&lv->B_to_q, lv->B, nB, lv->q, L);
    bconv_init(&lv->B_to_msk, lv->B, nB, &lv->m_sk, 1);

    for (int i = 0; i <= nB; i++) {
        const ref_mod *m = &lv->Bsk[i];
        lv->prod_q_mod_Bsk[i] = prod_mod_except(lv->q, L, -1, m);
        ref_invmod(lv->prod_q_mod_Bsk[i], m->value, &lv->inv_prod_q_mod_Bsk[i]);
        ref_invmod(lv->m_tilde.value % m->value, m->value, &lv->inv_mtilde_mod_Bsk[i]);
    }
    {
        uint64_t qm = prod_mod_except(lv->q, L, -1, &lv->m_tilde), inv;
        ref_invmod(qm, lv->m_tilde.value, &inv);
        lv->neg_inv_prod_q_mod_mtilde = negmod(inv, lv->m_tilde.value);
    }
    {
        uint64_t bm = prod_mod_except(lv->B, nB, -1, &lv->m_sk);
        ref_invmod(bm, lv->m_sk.value, &lv->inv_prod_B_mod_msk);
    }
    for (int j = 0; j < L; j++) lv->prod_B_mod_q[j] = prod_mod_except(lv->B, nB, -1, &lv->q[j]);
    (void)n;
    return 0;
}

ref_ctx *ref_ctx_create(int n, const uint64_t *coeff_modulus, int K, uint64_t plain_modulus)
{
    if (K < 1 || K > REF_MAXK) return NULL;
    int logn = 0;
    while ((1 << logn) < n) logn++;
    if ((1 << logn) != n) return NULL;
    ref_ctx *c = (ref_ctx *)calloc(1, sizeof(ref_ctx));
    c->n = n; c->logn = logn; c->K = K;
    c->using_keyswitching = K > 1;
    c->first_chain_idx = K > 1 ? K - 2 : 0;
    ref_mod_init(&c->t, plain_modulus);
    for (int j = 0; j < K; j++) {
        ref_mod_init(&c->key_q[j], coeff_modulus[j]);
        ntt_tables_init(&c->ntt_q[j], coeff_modulus[j], n, logn);
    }
    if ((plain_modulus - 1) % (2 * (uint64_t)n) == 0 && ref_is_prime(plain_modulus)) {
        ntt_tables_init(&c->ntt_t, plain_modulus, n, logn);
        /* [SEAL-recall BatchEncoder::populate_matrix_reps_index_map] */
        c->slot_map = (uint64_t *)malloc(sizeof(uint64_t) * n);
        uint64_t gen = 3, pos = 1, m = 2 * (uint64_t)n;
        size_t row = (size_t)n >> 1;
        for (size_t i = 0; i < row; i++) {
            uint64_t i1 = (pos - 1) >> 1, i2 = (m - pos - 1) >> 1;
            c->slot_map[i] = brv((uint32_t)i1, logn);
            c->slot_map[row | i] = brv((uint32_t)i2, logn);
            pos = (pos * gen) & (m - 1);
        }
    }
    /* base-conversion primes: get_primes(2n, 61, maxL + 3) covers |B| = L+1 too */
    int maxL = c->first_chain_idx + 1;
    uint64_t bcp[REF_MAXB + 1];
    int nbcp = maxL + 3;
    if (nbcp > REF_MAXB) nbcp = REF_MAXB;
    if (ref_get_primes(2 * (uint64_t)n, 61, nbcp, bcp)) { ref_ctx_destroy(c); return NULL; }
    for (int i = 0; i < nbcp; i++) ntt_tables_init(&c->ntt_bsk[i], bcp[i], n, logn);
    for (int ci = 0; ci <= c->first_chain_idx; ci++) {
        if (level_init(c, &c->level[ci], ci + 1, bcp)) { ref_ctx_destroy(c); return NULL; }
    }
    if (K > 1) {
        uint64_t p = c->key_q[K - 1].value;
        for (int j = 0; j < K - 1; j++)
            ref_invmod(p % c->key_q[j].value, c->key_q[j].value, &c->inv_p_mod_q[j]);
    }
    return c;
}

ref_ctx *ref_ctx_create_bits(int n, const int *coeff_bits, int K, uint64_t plain_modulus, int plain_bits)
{
    uint64_t q[REF_MAXK];
    if (K < 1 || K > REF_MAXK) return NULL;
    if (ref_coeff_modulus_create(n, coeff_bits, K, q)) return NULL;
    if (!plain_modulus) {
        /* PlainModulus::Batching(n, bits) = CoeffModulus::Create(n, {bits})[0] */
        if (ref_coeff_modulus_create(n, &plain_bits, 1, &plain_modulus)) return NULL;
    }
    return ref_ctx_create(n, q, K, plain_modulus);
}

static void ntt_free(ref_ntt *t) { free(t->w); t->w = NULL; }

void ref_ctx_destroy(ref_ctx *c)
{
    if (!c) return;
    for (int j = 0; j < REF_MAXK; j++) ntt_free(&c->ntt_q[j]);
    for (int j = 0; j < REF_MAXB; j++) ntt_free(&c->ntt_bsk[j]);
    ntt_free(&c->ntt_t);
    free(c->slot_map);
    free(c);
}

/* receiver: common/apsu/util/utils.cpp:179-189 get_parms_id_for_chain_idx (clamp to first) */
int ref_clamp_chain_idx(const ref_ctx *c, int chain_idx)
{
    return chain_idx > c->first_chain_idx ? c->first_chain_idx : chain_idx;
}

int ref_ctx_info(const ref_ctx *c, uint64_t *out, int cap)
{
    int k = 0;
#define PUT(v) do { if (k < cap) out[k] = (uint64_t)(v); k++; } while (0)
    PUT(c->n); PUT(c->K); PUT(c->first_chain_idx); PUT(c->t.value);
    for (int j = 0; j < c->K; j++) PUT(c->key_q[j].value);
    for (int j = 0; j < c->K; j++) PUT(c->ntt_q[j].psi);
    const ref_level *lv = &c->level[c->first_chain_idx];
    PUT(lv->nB); PUT(lv->m_sk.value); PUT(lv->gamma.value);
    for (int i = 0; i < lv->nB; i++) PUT(lv->B[i].value);
#undef PUT
    return k;
}

/* ------------------------------------------------------------------ evaluator ops */

static const ref_level *LV(const ref_ctx *c, int chain_idx) { return &c->level[chain_idx]; }

/* receiver_osn.cpp:467,475  Evaluator::transform_to_ntt_inplace(Ciphertext) */
void ref_transform_to_ntt(const ref_ctx *c, uint64_t *ct, int polys, int chain_idx)
{
    int L = chain_idx + 1; size_t n = c->n;
    for (int p = 0; p < polys; p++)
        for (int j = 0; j < L; j++) ref_ntt_fwd(&c->ntt_q[j], c->logn, ct + ((size_t)p * L + j) * n);
}

/* bin_bundle.cpp:154,268,297,321  Evaluator::transform_from_ntt_inplace */
void ref_transform_from_ntt(const ref_ctx *c, uint64_t *ct, int polys, int chain_idx)
{
    int L = chain_idx + 1; size_t n = c->n;
    for (int p = 0; p < polys; p++)
        for (int j = 0; j < L; j++) ref_ntt_inv(&c->ntt_q[j], c->logn, ct + ((size_t)p * L + j) * n);
}

/* bin_bundle.cpp:147,258,287,320  multiply_plain on NTT ct x NTT pt  (B6) */
void ref_multiply_plain_ntt(const ref_ctx *c, const uint64_t *ct, const uint64_t *pt_ntt,
                            uint64_t *out, int polys, int chain_idx)
{
    int L = chain_idx + 1; size_t n = c->n;
    for (int p = 0; p < polys; p++)
        for (int j = 0; j < L; j++) {
            const ref_mod *q = &c->key_q[j];
            const uint64_t *a = ct + ((size_t)p * L + j) * n, *b = pt_ntt + (size_t)j * n;
            uint64_t *o = out + ((size_t)p * L + j) * n;
            for (size_t k = 0; k < n; k++) o[k] = ref_mulmod(a[k], b[k], q);
        }
}

/* bin_bundle.cpp:419  Evaluator::transform_to_ntt_inplace(Plaintext, parms_id)  (B5) */
void ref_plain_lift_ntt(const ref_ctx *c, const uint64_t *pt, size_t pt_coeffs, uint64_t *out,
                        int chain_idx)
{
    const ref_level *lv = LV(c, chain_idx);
    int L = lv->L; size_t n = c->n;
    for (int j = 0; j < L; j++) {
        uint64_t *o = out + (size_t)j * n;
        for (size_t k = 0; k < n; k++) {
            uint64_t v = k < pt_coeffs ? pt[k] : 0;
            o[k] = v >= lv->upper_half_threshold ? v + lv->upper_half_incr[j] : v;
        }
        ref_ntt_fwd(&c->ntt_q[j], c->logn, o);
    }
}

/* bin_bundle.cpp:334  multiply_plain on coefficient-form ct x coefficient-form pt
   [SEAL-recall Evaluator::multiply_plain_normal].  Monomial shortcut: when the plaintext has
   exactly one non-zero coefficient SEAL multiplies by c*x^e directly and, under fast plain
   lift, does NOT add q_j - t to c even if c >= threshold. */
void ref_multiply_plain_coeff(const ref_ctx *c, const uint64_t *ct, const uint64_t *pt,
                              size_t pt_coeffs, uint64_t *out, int polys, int chain_idx)
{
    const ref_level *lv = LV(c, chain_idx);
    int L = lv->L; size_t n = c->n;
    size_t nonzero = 0, mono = 0;
    for (size_t k = 0; k < pt_coeffs && k < n; k++) if (pt[k]) { nonzero++; mono = k; }
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * n * L);
    if (nonzero == 1) {
        for (int j = 0; j < L; j++) {
            uint64_t *o = tmp + (size_t)j * n;
            memset(o, 0, sizeof(uint64_t) * n);
            o[mono] = pt[mono];
            ref_ntt_fwd(&c->ntt_q[j], c->logn, o);
        }
    } else {
        ref_plain_lift_ntt(c, pt, pt_coeffs, tmp, chain_idx);
    }
    for (int p = 0; p < polys; p++)
        for (int j = 0; j < L; j++) {
            const ref_mod *q = &c->key_q[j];
            uint64_t *o = out + ((size_t)p * L + j) * n;
            if (o != ct + ((size_t)p * L + j) * n) memcpy(o, ct + ((size_t)p * L + j) * n, sizeof(uint64_t) * n);
            ref_ntt_fwd(&c->ntt_q[j], c->logn, o);
            const uint64_t *b = tmp + (size_t)j * n;
            for (size_t k = 0; k < n; k++) o[k] = ref_mulmod(o[k], b[k], q);
            ref_ntt_inv(&c->ntt_q[j], c->logn, o);
        }
    free(tmp);
}

/* bin_bundle.cpp:148,264,273,293,303,323,336  Evaluator::add_inplace */
void ref_add(const ref_ctx *c, uint64_t *acc, const uint64_t *x, int polys, int chain_idx)
{
    int L = chain_idx + 1; size_t n = c->n;
    for (int p = 0; p < polys; p++)
        for (int j = 0; j < L; j++) {
            uint64_t q = c->key_q[j].value;
            uint64_t *a = acc + ((size_t)p * L + j) * n;
            const uint64_t *b = x + ((size_t)p * L + j) * n;
            for (size_t k = 0; k < n; k++) a[k] = addmod(a[k], b[k], q);
        }
}

/* bin_bundle.cpp:159,162,345,346  add_plain_inplace
   [SEAL-recall scalingvariant.cpp multiply_add_plain_with_scaling_variant]  (B7) */
void ref_add_plain(const ref_ctx *c, uint64_t *ct, const uint64_t *pt, size_t pt_coeffs, int chain_idx)
{
    const ref_level *lv = LV(c, chain_idx);
    int L = lv->L; size_t n = c->n;
    uint64_t t = c->t.value;
    for (size_t k = 0; k < pt_coeffs && k < n; k++) {
        u128 num = (u128)pt[k] * lv->q_mod_t + lv->upper_half_threshold;
        uint64_t fix = (uint64_t)(num / t);
        for (int j = 0; j < L; j++) {
            const ref_mod *q = &lv->q[j];
            uint64_t scaled = ref_bred128((u128)pt[k] * lv->coeff_div_plain[j] + fix, q);
            ct[(size_t)j * n + k] = addmod(ct[(size_t)j * n + k], scaled, q->value);
        }
    }
}

/* receiver_osn.cpp:463,471,478; bin_bundle.cpp:169,269,298,322,335,355
   [SEAL-recall RNSTool::divide_and_round_q_last_inplace]  (B8) */
void ref_mod_switch_to_next(const ref_ctx *c, uint64_t *ct, int polys, int chain_idx)
{
    const ref_level *lv = LV(c, chain_idx);
    int L = lv->L; size_t n = c->n;
    const ref_mod *ql = &lv->q[L - 1];
    uint64_t half = ql->value >> 1;
    /* output is packed [poly][L-1][n] in place */
    for (int p = 0; p < polys; p++) {
        uint64_t *src = ct + (size_t)p * L * n;
        uint64_t *dst = ct + (size_t)p * (L - 1) * n;
        uint64_t *last = src + (size_t)(L - 1) * n;
        for (size_t k = 0; k < n; k++) last[k] = addmod(last[k], half, ql->value);
        for (int j = 0; j + 1 < L; j++) {
            const ref_mod *q = &lv->q[j];
            uint64_t half_mod = ref_bred64(half, q);
            for (size_t k = 0; k < n; k++) {
                uint64_t tmp = ref_bred64(last[k], q);
                tmp = submod(tmp, half_mod, q->value);
                uint64_t v = submod(src[(size_t)j * n + k], tmp, q->value);
                dst[(size_t)j * n + k] = ref_mulmod(v, lv->inv_q_last_mod_q[j], q);
            }
        }
    }
}

int ref_irrelevant_bit_count(const ref_ctx *c)
{
    /* bin_bundle.cpp:67-97: bits(q0) - (bits(t) + significant_bits(n) - 1), last level 1 limb */
    int compr = c->t.bits + (c->logn + 1) - 1;
    int irr = c->key_q[0].bits - compr;
    return irr > 0 ? irr : 0;
}

/* bin_bundle.cpp:67-97 try_clear_irrelevant_bits (last level always has one limb here) */
void ref_clear_irrelevant_bits(const ref_ctx *c, uint64_t *ct, int polys)
{
    int irr = ref_irrelevant_bit_count(c);
    if (!irr) return;
    uint64_t mask = ~(((uint64_t)1 << irr) - 1);
    for (size_t k = 0; k < (size_t)polys * c->n; k++) ct[k] &= mask;
}

/* ---- BFV multiply (BEHZ)  [SEAL-recall Evaluator::bfv_multiply + RNSTool]  (B9) ---- */

/* steps (1)-(3) for one input polynomial: in [L][n] (coeff) -> out_q [L][n] NTT, out_bsk [nB+1][n] NTT */
static void behz_extend_ntt(const ref_ctx *c, const ref_level *lv, const uint64_t *in,
                            uint64_t *out_q, uint64_t *out_bsk)
{
    int L = lv->L, nBsk = lv->nB + 1; size_t n = c->n;
    memcpy(out_q, in, sizeof(uint64_t) * n * L);
    for (int j = 0; j < L; j++) ref_ntt_fwd(&c->ntt_q[j], c->logn, out_q + (size_t)j * n);

    /* fastbconv_m_tilde: temp = in * m_tilde mod q ; -> Bsk and -> {m_tilde} */
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n * (L + nBsk + 1));
    uint64_t *ext = temp + (size_t)L * n;                    /* [nBsk + 1][n] */
    for (int j = 0; j < L; j++)
        for (size_t k = 0; k < n; k++)
            temp[(size_t)j * n + k] = ref_mulmod(in[(size_t)j * n + k], lv->m_tilde.value % lv->q[j].value, &lv->q[j]);
    bconv_apply(&lv->q_to_Bsk, temp, ext, n);
    bconv_apply(&lv->q_to_mtilde, temp, ext + (size_t)nBsk * n, n);

    /* sm_mrq */
    const uint64_t *in_mt = ext + (size_t)nBsk * n;
    uint64_t mt = lv->m_tilde.value, mt_half = mt >> 1;
    for (int i = 0; i < nBsk; i++) {
        const ref_mod *m = &lv->Bsk[i];
        for (size_t k = 0; k < n; k++) {
            uint64_t r = ref_mulmod(in_mt[k], lv->neg_inv_prod_q_mod_mtilde, &lv->m_tilde);
            if (r >= mt_half) r += m->value - mt;
            /* (input + q*r) * m_tilde^-1 mod Bsk_i */
            uint64_t v = ref_bred128((u128)r * lv->prod_q_mod_Bsk[i] + ext[(size_t)i * n + k], m);
            out_bsk[(size_t)i * n + k] = ref_mulmod(v, lv->inv_mtilde_mod_Bsk[i], m);
        }
    }
    free(temp);
    /* NTT in Bsk: Bsk[i] = B[i] -> ntt_bsk[2+i]; Bsk[nB] = m_sk -> ntt_bsk[0] */
    for (int i = 0; i < nBsk; i++) {
        const ref_ntt *t = i < lv->nB ? &c->ntt_bsk[2 + i] : &c->ntt_bsk[0];
        ref_ntt_fwd(t, c->logn, out_bsk + (size_t)i * n);
    }
}

/* steps (5)-(8) for one output polynomial (NTT form in q and Bsk) -> dst [L][n] coeff */
static void behz_finish(const ref_ctx *c, const ref_level *lv, uint64_t *dq, uint64_t *dbsk, uint64_t *dst)
{
    int L = lv->L, nB = lv->nB, nBsk = nB + 1; size_t n = c->n;
    uint64_t t = c->t.value;
    for (int j = 0; j < L; j++) ref_ntt_inv(&c->ntt_q[j], c->logn, dq + (size_t)j * n);
    for (int i = 0; i < nBsk; i++) {
        const ref_ntt *tb = i < nB ? &c->ntt_bsk[2 + i] : &c->ntt_bsk[0];
        ref_ntt_inv(tb, c->logn, dbsk + (size_t)i * n);
    }
    /* (6) multiply by t */
    for (int j = 0; j < L; j++)
        for (size_t k = 0; k < n; k++) dq[(size_t)j * n + k] = ref_mulmod(dq[(size_t)j * n + k], t, &lv->q[j]);
    for (int i = 0; i < nBsk; i++)
        for (size_t k = 0; k < n; k++) dbsk[(size_t)i * n + k] = ref_mulmod(dbsk[(size_t)i * n + k], t, &lv->Bsk[i]);
    /* (7) fast_floor: q u Bsk -> Bsk */
    uint64_t *fl = (uint64_t *)malloc(sizeof(uint64_t) * n * (nBsk + 2));
    bconv_apply(&lv->q_to_Bsk, dq, fl, n);
    for (int i = 0; i < nBsk; i++) {
        const ref_mod *m = &lv->Bsk[i];
        for (size_t k = 0; k < n; k++) {
            uint64_t v = dbsk[(size_t)i * n + k] + (m->value - fl[(size_t)i * n + k]);
            fl[(size_t)i * n + k] = ref_mulmod(v, lv->inv_prod_q_mod_Bsk[i], m);
        }
    }
    /* (8) fastbconv_sk: Bsk -> q */
    bconv_apply(&lv->B_to_q, fl, dst, n);
    uint64_t *tmp_sk = fl + (size_t)nBsk * n;
    bconv_apply(&lv->B_to_msk, fl, tmp_sk, n);
    uint64_t msk = lv->m_sk.value, msk_half = msk >> 1;
    const uint64_t *in_sk = fl + (size_t)nB * n;
    for (size_t k = 0; k < n; k++) {
        uint64_t alpha = ref_mulmod(tmp_sk[k] + (msk - in_sk[k]), lv->inv_prod_B_mod_msk, &lv->m_sk);
        for (int j = 0; j < L; j++) {
            const ref_mod *q = &lv->q[j];
            uint64_t cur = dst[(size_t)j * n + k];
            if (alpha > msk_half) {
                cur = ref_bred128((u128)(msk - alpha) * lv->prod_B_mod_q[j] + cur, q);
            } else {
                cur = ref_bred128((u128)alpha * (q->value - lv->prod_B_mod_q[j]) + cur, q);
            }
            dst[(size_t)j * n + k] = cur;
        }
    }
    free(fl);
}

/* Evaluator::bfv_multiply for ciphertexts of any size [SEAL-recall]: dest_size = size_a + size_b - 1 (throws when that exceeds
   SEAL_CIPHERTEXT_SIZE_MAX = 16, Ciphertext::resize); output polynomial I = sum_{i + j = I} a_i b_j formed limb-wise in the NTT
   domain over q and Bsk, every product reduced before the modular add (behz_ciphertext_product).  size 2 x size 2 is the
   case with key switching (d0 = a0 b0 ; d1 = a0 b1 + a1 b0 ; d2 = a1 b1). */
static int behz_multiply(const ref_ctx *c, const uint64_t *a, int sa, const uint64_t *b, int sb, uint64_t *out,
                         int chain_idx, int is_square)
{
    const ref_level *lv = LV(c, chain_idx);
    int L = lv->L, nBsk = lv->nB + 1; size_t n = c->n;
    int so = sa + sb - 1;
    if (sa < 2 || sb < 2 || so > REF_CT_SIZE_MAX) return -1;
    size_t sq = (size_t)L * n, sb_ = (size_t)nBsk * n;
    uint64_t *buf = (uint64_t *)calloc((size_t)(sa + sb + so) * (sq + sb_), sizeof(uint64_t));
    uint64_t *aq = buf, *ab = aq + sa * sq, *bq = ab + sa * sb_, *bb = bq + sb * sq;
    uint64_t *dq = bb + sb * sb_, *db = dq + so * sq;
    for (int p = 0; p < sa; p++) behz_extend_ntt(c, lv, a + p * sq, aq + p * sq, ab + p * sb_);
    if (is_square) { bq = aq; bb = ab; }
    else for (int p = 0; p < sb; p++) behz_extend_ntt(c, lv, b + p * sq, bq + p * sq, bb + p * sb_);
    /* (4) tensor */
    for (int base = 0; base < 2; base++) {
        int nl = base ? nBsk : L;
        size_t sp = base ? sb_ : sq;
        const uint64_t *x = base ? ab : aq, *y = base ? bb : bq;
        uint64_t *d = base ? db : dq;
        for (int I = 0; I < so; I++) {
            int i0 = I - (sb - 1) > 0 ? I - (sb - 1) : 0, i1 = I < sa - 1 ? I : sa - 1;
            for (int i = i0; i <= i1; i++)
                for (int j = 0; j < nl; j++) {
                    const ref_mod *m = base ? &lv->Bsk[j] : &lv->q[j];
                    for (size_t k = 0; k < n; k++) {
                        size_t o = (size_t)j * n + k;
                        d[I * sp + o] = addmod(d[I * sp + o], ref_mulmod(x[i * sp + o], y[(I - i) * sp + o], m), m->value);
                    }
                }
        }
    }
    for (int p = 0; p < so; p++) behz_finish(c, lv, dq + p * sq, db + p * sb_, out + p * sq);
    free(buf);
    return 0;
}

/* receiver_osn.cpp:424 ; bin_bundle.cpp:272,301  Evaluator::multiply / multiply_inplace */
void ref_multiply(const ref_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out3, int chain_idx)
{
    behz_multiply(c, a, 2, b, 2, out3, chain_idx, 0);
}

/* receiver_osn.cpp:422  Evaluator::square  (bfv_square: d1 = 2 a0 a1 — identical mod q) */
void ref_square(const ref_ctx *c, const uint64_t *a, uint64_t *out3, int chain_idx)
{
    behz_multiply(c, a, 2, a, 2, out3, chain_idx, 1);
}

/* the same for operands that were never relinearised (parameter sets with one coefficient prime: receiver_osn.cpp:427-432
   skips relinearize_inplace, so powers and bin_bundle.cpp:272,301's operands grow).  out: size_a + size_b - 1 polynomials.
   Returns -1 where SEAL throws (size above 16).  bfv_square falls back to bfv_multiply for sizes other than 2. */
int ref_multiply_sized(const ref_ctx *c, const uint64_t *a, int size_a, const uint64_t *b, int size_b, uint64_t *out, int chain_idx)
{
    return behz_multiply(c, a, size_a, b, size_b, out, chain_idx, a == b && size_a == size_b);
}

/* receiver_osn.cpp:431 ; bin_bundle.cpp:309  relinearize_inplace
   [SEAL-recall Evaluator::switch_key_inplace, BFV branch]  (B10).
   ct3: [3][L][n] coefficient form; on return polys 0,1 hold the size-2 result. */
void ref_relinearize(const ref_ctx *c, uint64_t *ct3, const uint64_t *rk, int chain_idx)
{
    int L = chain_idx + 1, K = c->K; size_t n = c->n;
    int R = L + 1;                                   /* rns_modulus_size */
    const uint64_t *target = ct3 + (size_t)2 * L * n;
    uint64_t *prod = (uint64_t *)calloc((size_t)2 * R * n, sizeof(uint64_t));   /* [comp][R][n] */
    uint64_t *tntt = (uint64_t *)malloc(sizeof(uint64_t) * n);
    u128 *acc = (u128 *)malloc(sizeof(u128) * 2 * n);
    for (int I = 0; I < R; I++) {
        int key_index = (I == L) ? K - 1 : I;
        const ref_mod *qm = &c->key_q[key_index];
        memset(acc, 0, sizeof(u128) * 2 * n);
        for (int J = 0; J < L; J++) {
            const uint64_t *src = target + (size_t)J * n;
            if (c->key_q[J].value <= qm->value) memcpy(tntt, src, sizeof(uint64_t) * n);
            else for (size_t k = 0; k < n; k++) tntt[k] = ref_bred64(src[k], qm);
            ref_ntt_fwd(&c->ntt_q[key_index], c->logn, tntt);
            for (int comp = 0; comp < 2; comp++) {
                const uint64_t *key = rk + (((size_t)J * 2 + comp) * K + key_index) * n;
                for (size_t k = 0; k < n; k++) acc[(size_t)comp * n + k] += (u128)tntt[k] * key[k];
            }
        }
        for (int comp = 0; comp < 2; comp++)
            for (size_t k = 0; k < n; k++)
                prod[((size_t)comp * R + I) * n + k] = ref_bred128(acc[(size_t)comp * n + k], qm);
    }
    /* mod-down by the special prime with rounding, add to (c0, c1) */
    const ref_mod *pm = &c->key_q[K - 1];
    uint64_t p_half = pm->value >> 1;
    for (int comp = 0; comp < 2; comp++) {
        uint64_t *tl = prod + ((size_t)comp * R + L) * n;
        ref_ntt_inv(&c->ntt_q[K - 1], c->logn, tl);
        for (size_t k = 0; k < n; k++) tl[k] = ref_bred64(tl[k] + p_half, pm);
        for (int j = 0; j < L; j++) {
            const ref_mod *q = &c->key_q[j];
            uint64_t *pj = prod + ((size_t)comp * R + j) * n;
            ref_ntt_inv(&c->ntt_q[j], c->logn, pj);
            uint64_t half_mod = ref_bred64(p_half, q);
            uint64_t *dst = ct3 + ((size_t)comp * L + j) * n;
            for (size_t k = 0; k < n; k++) {
                uint64_t tk = submod(ref_bred64(tl[k], q), half_mod, q->value);
                uint64_t v = ref_mulmod(submod(pj[k], tk, q->value), c->inv_p_mod_q[j], q);
                dst[k] = addmod(dst[k], v, q->value);
            }
        }
    }
    free(prod); free(tntt); free(acc);
}
