/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * CPU restatement (plain C, unsigned __int128) of the Microsoft SEAL BFV operations that
 * APSU's DB-side query evaluation executes (reference call sites:
 * receiver/apsu/receiver_osn.cpp:395-488, receiver/apsu/bin_bundle.cpp:67-174,192-360).
 *
 * PARITY UNPINNED: Microsoft SEAL (>=3.7, cmake/APSUConfig.cmake.in:45) is a third-party
 * dependency that is absent from /root/reference and from this image, and the reference
 * holds no golden vectors / KATs for this path.  The algorithms below are
 * restated from SEAL's published sources (evaluator.cpp, util/rns.cpp, util/ntt.cpp,
 * util/scalingvariant.cpp, util/numth.cpp, modulus.cpp) from memory; every such place is
 * tagged [SEAL-recall].  The oracle is pinned only by (a) an independent Python big-int
 * model (oracle/pymodel.py -> tests/golden), (b) algebraic invariants (decrypt o eval).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 */
#ifndef APSU_REF_CORE_H
#define APSU_REF_CORE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REF_MAXK 8            /* max limbs in the key-level coefficient modulus */
#define REF_MAXB (REF_MAXK + 2)

typedef unsigned __int128 u128;

typedef struct {
    uint64_t value;
    uint64_t ratio[2];        /* floor(2^128 / value), low/high word [SEAL-recall Modulus::const_ratio] */
    int bits;
} ref_mod;

typedef struct {              /* negacyclic NTT tables for one modulus [SEAL-recall util/ntt.cpp] */
    ref_mod q;
    uint64_t psi;             /* minimal primitive 2n-th root of unity */
    uint64_t *w, *wq;         /* w[k] = psi^{brv(k)}, wq = floor(w*2^64/q) */
    uint64_t *iw, *iwq;       /* iw[k] = psi^{-brv(k)} */
    uint64_t inv_n, inv_n_q;
} ref_ntt;

typedef struct {              /* fast base converter [SEAL-recall util/rns.cpp BaseConverter] */
    int ni, no;
    ref_mod ib[REF_MAXB], ob[REF_MAXB];
    uint64_t inv_punct[REF_MAXB];          /* (Q/q_i)^{-1} mod q_i */
    uint64_t matrix[REF_MAXB][REF_MAXB];   /* [o][i] = (Q/q_i) mod p_o */
} ref_bconv;

typedef struct {              /* one level of the modulus chain (chain_idx = L-1) */
    int L;
    ref_mod q[REF_MAXK];
    /* plaintext scaling [SEAL-recall context.cpp / scalingvariant.cpp] */
    uint64_t coeff_div_plain[REF_MAXK];    /* floor(Q/t) mod q_j */
    uint64_t q_mod_t;
    uint64_t upper_half_threshold;         /* (t+1)/2 */
    uint64_t upper_half_incr[REF_MAXK];    /* q_j - t  (fast plain lift) */
    uint64_t inv_q_last_mod_q[REF_MAXK];   /* q_{L-1}^{-1} mod q_j, j<L-1 */
    /* BEHZ RNS tool [SEAL-recall util/rns.cpp RNSTool::initialize] */
    int nB;                                /* |B| ; Bsk = B u {m_sk}, nB+1 moduli, m_sk last */
    ref_mod B[REF_MAXB], m_sk, gamma, m_tilde;
    ref_mod Bsk[REF_MAXB];
    ref_bconv q_to_Bsk, q_to_mtilde, B_to_q, B_to_msk;
    uint64_t inv_prod_q_mod_Bsk[REF_MAXB];
    uint64_t prod_q_mod_Bsk[REF_MAXB];
    uint64_t inv_mtilde_mod_Bsk[REF_MAXB];
    uint64_t neg_inv_prod_q_mod_mtilde;
    uint64_t inv_prod_B_mod_msk;
    uint64_t prod_B_mod_q[REF_MAXK];
} ref_level;

typedef struct {
    int n, logn;
    int K;                                 /* limbs at key level */
    int first_chain_idx;                   /* K-2 if K>1 else 0 */
    int using_keyswitching;                /* K>1 */
    ref_mod t;
    ref_mod key_q[REF_MAXK];
    ref_ntt ntt_q[REF_MAXK];
    ref_ntt ntt_bsk[REF_MAXB];             /* index i <-> baseconv prime list: [0]=m_sk,[1]=gamma,[2..]=B */
    ref_ntt ntt_t;                         /* mod t, for the BatchEncoder (harness) */
    uint64_t *slot_map;                    /* matrix_reps_index_map [SEAL-recall batchencoder.cpp] */
    ref_level level[REF_MAXK];             /* level[c] valid for c <= first_chain_idx */
    uint64_t inv_p_mod_q[REF_MAXK];        /* special prime^{-1} mod q_j (key switching) */
} ref_ctx;

/* ---- small-modulus arithmetic ---- */
void     ref_mod_init(ref_mod *m, uint64_t value);
uint64_t ref_mulmod(uint64_t a, uint64_t b, const ref_mod *q);
uint64_t ref_bred128(u128 x, const ref_mod *q);
uint64_t ref_bred64(uint64_t x, const ref_mod *q);
uint64_t ref_powmod(uint64_t a, uint64_t e, const ref_mod *q);
int      ref_invmod(uint64_t a, uint64_t m, uint64_t *out);   /* generic (m need not be prime) */
int      ref_is_prime(uint64_t v);
int      ref_get_primes(uint64_t factor, int bits, int count, uint64_t *out);
int      ref_coeff_modulus_create(int n, const int *bits, int count, uint64_t *out);
uint64_t ref_minimal_primitive_root(uint64_t degree, const ref_mod *q);

/* ---- context ---- */
ref_ctx *ref_ctx_create(int n, const uint64_t *coeff_modulus, int K, uint64_t plain_modulus);
ref_ctx *ref_ctx_create_bits(int n, const int *coeff_bits, int K, uint64_t plain_modulus, int plain_bits);
void     ref_ctx_destroy(ref_ctx *c);
int      ref_clamp_chain_idx(const ref_ctx *c, int chain_idx);   /* get_parms_id_for_chain_idx */
/* introspection used by tests to diff product constants vs oracle */
int      ref_ctx_info(const ref_ctx *c, uint64_t *out, int cap);

/* ---- NTT on one limb ---- */
void ref_ntt_fwd(const ref_ntt *t, int logn, uint64_t *a);
void ref_ntt_inv(const ref_ntt *t, int logn, uint64_t *a);

/* ---- Evaluator restatements; ct layout [poly][limb][coeff] at level chain_idx ---- */
void ref_transform_to_ntt(const ref_ctx *c, uint64_t *ct, int polys, int chain_idx);
void ref_transform_from_ntt(const ref_ctx *c, uint64_t *ct, int polys, int chain_idx);
void ref_multiply_plain_ntt(const ref_ctx *c, const uint64_t *ct, const uint64_t *pt_ntt,
                            uint64_t *out, int polys, int chain_idx);
void ref_plain_lift_ntt(const ref_ctx *c, const uint64_t *pt_mod_t, size_t pt_coeffs,
                        uint64_t *out, int chain_idx);   /* Evaluator::transform_to_ntt(Plaintext) */
void ref_multiply_plain_coeff(const ref_ctx *c, const uint64_t *ct, const uint64_t *pt_mod_t,
                              size_t pt_coeffs, uint64_t *out, int polys, int chain_idx);
void ref_add(const ref_ctx *c, uint64_t *acc, const uint64_t *x, int polys, int chain_idx);
void ref_add_plain(const ref_ctx *c, uint64_t *ct, const uint64_t *pt_mod_t, size_t pt_coeffs,
                   int chain_idx);
void ref_multiply(const ref_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out3,
                  int chain_idx);                        /* size2 x size2 -> size3 */
void ref_square(const ref_ctx *c, const uint64_t *a, uint64_t *out3, int chain_idx);
#define REF_CT_SIZE_MAX 16   /* SEAL_CIPHERTEXT_SIZE_MAX [SEAL-recall util/defines.h] */
/* any sizes (no relinearisation in between): out has size_a + size_b - 1 polynomials; -1 = SEAL throws (size > 16) */
int  ref_multiply_sized(const ref_ctx *c, const uint64_t *a, int size_a, const uint64_t *b, int size_b, uint64_t *out, int chain_idx);
/* relin key layout: [decomp i < K-1][component 2][limb K][n], NTT form */
void ref_relinearize(const ref_ctx *c, uint64_t *ct3, const uint64_t *rk, int chain_idx);
void ref_mod_switch_to_next(const ref_ctx *c, uint64_t *ct, int polys, int chain_idx);
void ref_clear_irrelevant_bits(const ref_ctx *c, uint64_t *ct_last, int polys);
int  ref_irrelevant_bit_count(const ref_ctx *c);

#ifdef __cplusplus
}
#endif
#endif
