/* ORACLE — TEST INFRASTRUCTURE ONLY (see ref_core.h header: PARITY UNPINNED). */
#ifndef APSU_REF_PATH_H
#define APSU_REF_PATH_H
#include "ref_core.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint32_t power, depth, p1, p2; } ref_dag_node;   /* powers.h:53-77 PowersNode */

int ref_create_powers_set(uint32_t ps_low_degree, uint32_t target_degree, uint32_t *out, int cap);
int ref_powers_dag_configure(const uint32_t *sources, int ns, const uint32_t *targets, int nt,
                             ref_dag_node *nodes);
/* sizes: polynomials per target power, indexed by power; required when the parameters have no key switching and the
   PowersDag has products (buffers then hold REF_CT_SIZE_MAX polynomials), may be NULL otherwise */
int ref_power_sizes(const ref_ctx *c, const ref_dag_node *nodes, int n_nodes, uint32_t *sizes);
int ref_compute_powers(const ref_ctx *c, uint64_t **powers, const ref_dag_node *nodes, int n_nodes,
                       const uint64_t *rk, uint32_t ps_low_degree, uint32_t *sizes);
int ref_plain_chain_idx(const ref_ctx *c, uint32_t ps_low_degree);
int ref_coeff_is_ntt(uint32_t ps_low_degree, uint32_t i);
int ref_eval(const ref_ctx *c, uint64_t *const *powers, int n_powers, const uint64_t *const *coeffs,
             int n_coeffs, int lvl, const uint64_t *mask, uint64_t *out, const uint32_t *sizes, uint32_t *out_size);
int ref_eval_patstock(const ref_ctx *c, uint64_t *const *powers, int n_powers,
                      const uint64_t *const *coeffs, int n_coeffs, uint32_t ps_low_degree,
                      const uint64_t *rk, const uint64_t *mask, uint64_t *out, const uint32_t *sizes, uint32_t *out_size);

/* ---- harness-only pieces (other party / DB build), needed to make inputs & check outputs ---- */
typedef struct { uint64_t s[4]; } ref_rng;
void     ref_rng_seed(ref_rng *r, uint64_t seed);
uint64_t ref_rng_next(ref_rng *r);
uint64_t ref_rng_below(ref_rng *r, uint64_t bound);
void ref_fill_uniform(uint64_t seed, uint64_t bound, uint64_t *out, size_t count);

void ref_keygen(const ref_ctx *c, uint64_t seed, uint64_t *sk_ntt /* [K][n] */);
void ref_encrypt_symmetric(const ref_ctx *c, const uint64_t *sk_ntt, const uint64_t *pt_mod_t,
                           uint64_t seed, uint64_t *ct /* [2][first_L][n] coeff form */);
void ref_gen_relin_keys(const ref_ctx *c, const uint64_t *sk_ntt, uint64_t seed,
                        uint64_t *rk /* [K-1][2][K][n] NTT form */);
/* decrypt a coefficient-form ct (polys 2 or 3) at chain_idx -> n coefficients mod t;
   returns invariant-noise budget in bits (>=0) */
int  ref_decrypt(const ref_ctx *c, const uint64_t *sk_ntt, const uint64_t *ct, int polys,
                 int chain_idx, uint64_t *pt_mod_t);
void ref_batch_encode(const ref_ctx *c, const uint64_t *values, uint64_t *pt_mod_t);
void ref_batch_decode(const ref_ctx *c, const uint64_t *pt_mod_t, uint64_t *values);
/* receiver/apsu/receiver_osn.cpp:53-73 (vec_to_oc_block): the felts of one item packed into a 128-bit block;
 * out[0] = low 64 bits ("lower"), out[1] = high 64 bits ("higher") of oc::toBlock(higher, lower) */
void ref_vec_to_oc_block(const uint64_t *in, size_t felts_per_item, uint64_t plain_modulus, uint64_t out[2]);
/* common/apsu/util/db_encoding.cpp:209-256,360-366 (algebraize_item): felts of one 16-byte item; returns their count */
int ref_algebraize_item(const unsigned char item[16], uint32_t item_bit_count, uint64_t plain_modulus, uint64_t *felts);
/* common/apsu/util/interpolate.cpp:63-80 ; out has count+1 entries, degree ascending */
void ref_polyn_with_roots(const ref_ctx *c, const uint64_t *roots, size_t count, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
