// Device-side constant blocks and kernel launch entry points of the query-evaluation engine.
// Everything here is plain-old-data uploaded once per context; kernels read it through
// wave-uniform (scalar) loads.  Layout of all polynomial data is SEAL's in-memory order
// [poly][limb][coeff] of uint64.
#pragma once
#include <hip/hip_runtime.h>
#include "modmath.h"
#include "ntt_core.h"
#include "blake2x.h"

namespace apsu_he {

constexpr int DMAXL = 8;            // limbs of q at a data level
constexpr int DMAXB = DMAXL + 2;    // |Bsk|
constexpr int DMAXE = DMAXL + DMAXB; // limbs of an extended (q u Bsk) polynomial

struct ShoupConst { u64 w, wq; };

// One level of the modulus chain (chain_idx = L-1).
struct DevLevel {
    int L, nB, nBsk, E;                         // E = L + nBsk limbs of an extended polynomial
    Mod q[DMAXL];
    Mod bsk[DMAXB];                             // B_0..B_{nB-1}, m_sk
    Mod ext[DMAXE];                             // q_0..q_{L-1}, Bsk..   (modulus of each ext limb)
    u64 t;
    u32 mac_shift[DMAXL], mac_chunk[DMAXL];     // k_mac: operand split width s = ceil(bits(q_j)/2) and terms per carry-free chunk
    u32 mac_chunk_k[DMAXL];                     // ... of the three-product form (middle products have 2 s + 2 bits)
    // Bit-packed database rows (round 4; k_mac<.., PACKED>): limb j of a stored NTT-form plaintext takes mac_bits[j] bits per
    // coefficient -- the smallest width >= bits(q_j) for which a lane's 16-byte window covers its two coefficients at every
    // position (48, 49, 50, 52, 56; else 64 = not packed) -- in rows of n * mac_bits[j] / 8 bytes at byte offset mac_row_off[j]
    // inside a plaintext slot.  Rows depend on the limb only, so every level's prefix of them is the same.
    u32 mac_bits[DMAXL], mac_row_off[DMAXL], mac_mask_hi[DMAXL];
    // add_plain (App. B7) and plaintext lift (B5)
    u64 coeff_div_plain[DMAXL];
    u64 q_mod_t, threshold;
    u64 incr[DMAXL];
    // drop-last-limb with rounding (B8)
    u64 half;
    u64 half_mod[DMAXL];
    ShoupConst inv_q_last[DMAXL];
    // BEHZ extension: fastbconv_m_tilde + sm_mrq (B9 steps 1-2)
    ShoupConst ext_scale[DMAXL];                // m_tilde * (Q/q_j)^-1 mod q_j
    u64 q_to_bsk[DMAXB][DMAXL];                 // (Q/q_j) mod Bsk_i
    u32 q_to_mt[DMAXL];                         // (Q/q_j) mod 2^32
    u32 neg_inv_q_mt;
    u64 prod_q_bsk[DMAXB];
    ShoupConst inv_mt_bsk[DMAXB];
    // BEHZ finish: multiply by t, fast_floor, fastbconv_sk (B9 steps 6-8)
    ShoupConst t_inv_punct_q[DMAXL];            // t * (Q/q_j)^-1 mod q_j
    ShoupConst t_bsk[DMAXB];                    // t mod Bsk_i
    ShoupConst inv_prod_q_bsk[DMAXB];
    ShoupConst inv_punct_B[DMAXB];
    u64 B_to_q[DMAXL][DMAXB];                   // (B/b_i) mod q_j
    u64 B_to_msk[DMAXB];
    ShoupConst inv_prod_B_msk;
    u64 prod_B_q[DMAXL], neg_prod_B_q[DMAXL];
    u64 msk_half;
    // the same matrices as Shoup constants for the fully unrolled kernels (L = nB <= 3): every product is a
    // lazy Shoup product (< 2m), sums stay below 8m < 2^64 and are reduced once
    ShoupConst s_q_to_bsk[DMAXB][DMAXL];
    ShoupConst s_prod_q_bsk[DMAXB];
    // the same two with m_tilde^-1 folded in (sm_mrq's closing product becomes a plain reduction: behz_ext2_body)
    ShoupConst s_q_to_bsk_mt[DMAXB][DMAXL];
    ShoupConst s_prod_q_bsk_mt[DMAXB];
    ShoupConst s_fl[DMAXB];                     // i < nB: (Q^-1 * (B/b_i)^-1) mod b_i ; i = nB: Q^-1 mod m_sk
    ShoupConst s_B_to_q[DMAXL][DMAXB];
    ShoupConst s_B_to_msk[DMAXB];
    ShoupConst s_prod_B_q[DMAXL], s_neg_prod_B_q[DMAXL];
    // The unrolled finish kernels (L = nB <= 3) consume the output of an inverse NTT and start by multiplying it with a
    // constant (t (Q/q_j)^-1 for the q limbs, t for the Bsk limbs).  For them the inverse transform leaves out its own final
    // twist (n^-1 psi^-k, one exact Shoup product per coefficient) and writes the raw lazy value; the twist rides on the
    // finish's constant instead: fin_q[j][k] = t (Q/q_j)^-1 n^-1 psi_j^-k mod q_j, fin_b[i][k] = t n^-1 psi_i^-k mod Bsk_i.
    // Same residues, one modular product per coefficient less.  Null for levels that use the generic finish.
    const ShoupConst *fin_q[DMAXL];
    const ShoupConst *fin_b[DMAXB];
    // The same idea for the consumers of an inverse NTT that drop this level's last limb (mod_switch_to_next: the fused
    // drop + extension of eval_patstock's inner polynomials, the i = 0 block's finish): the transform writes raw values and
    // the twist rides on the drop's own constant, drop_tw[j][k] = n^-1 psi_j^-k q_last^-1 mod q_j (j < L - 1);
    // last_tw[k] = n^-1 psi_{L-1}^-k (the dropped limb needs its canonical residue).  Null where unused.
    const ShoupConst *drop_tw[DMAXL];
    const ShoupConst *last_tw;
};

// Key-switching constants (App. B10); moduli indexed by key limb.
struct DevKey {
    int K;
    Mod q[DMAXL + 1];
    u64 p_half;
    u64 p_half_mod[DMAXL];
    ShoupConst inv_p[DMAXL];
    // mod-down behind a RAW inverse transform: md_tw[j][k] = n^-1 psi_j^-k p^-1 mod q_j, p_tw[k] = n^-1 psi_p^-k mod p
    const ShoupConst *md_tw[DMAXL];
    const ShoupConst *p_tw;
};

// Multiply-accumulate job: for g < ng:  out[g][2][L][n] = sum_{j<cnt} PW_j (.) PT_{g,j}   (NTT domain).
// All streams of a job share the ciphertext powers PW (same bundle index) and the term count.
#ifndef APSU_MAC_GMAX
#define APSU_MAC_GMAX 4
#endif
constexpr int MAC_G = APSU_MAC_GMAX;
struct MacJob {
    const u64 *pt[MAC_G]; // first plaintext of stream g; term j at + j*pt_stride ; limb l at + l*n
    u64 *out[MAC_G];      // [2][L][n]
    const u64 *pw;        // first ciphertext; term j at pw + j*pw_stride ; poly p at + p*pw_poly_stride
    u32 cnt, ng;
    u32 pt_stride, pw_stride, pw_poly_stride;        // in u64 words
    u32 out_poly_stride;  // words between the two output polynomials (L*n for a full ciphertext)
    u32 limb0, nl;        // limbs limb0 .. limb0+nl-1 are handled (grid.y >= nl exits); modulus = q[limb]
    u32 packed, pad;      // packed: pt[] point at bit-packed plaintext slots and pt_stride is in BYTES (DevLevel::mac_bits)
};

// ---- launch wrappers (all asynchronous on `st`) --------------------------------------------
// NTT over `count` consecutive limb polynomials of n coefficients; limb g uses
// tabs[modmap[g % period] & NTT_MAP_MASK].  An inverse transform of a limb whose map entry carries NTT_MAP_RAW writes its
// result WITHOUT the final twist n^-1 psi^-k and without the final reduction (consumers: the unrolled BEHZ finish kernels).
constexpr int NTT_MAP_RAW = 1 << 30, NTT_MAP_MASK = NTT_MAP_RAW - 1;
// latency_limbs (round 6): a launch of at most that many limbs takes the LATENCY form of the transform (8 coefficients per lane, twice the
// waves per limb; ntt_core.h plan_k) where the ring size has one (n = 8192, 4096); 0 = always the throughput form; NTT_FORM_AUTO = the
// measured crossover per ring size and kind of launch (kernels.hip, ntt_use_latency_form).  Same bits.
constexpr size_t NTT_FORM_AUTO = ~(size_t)0;
// narrow_only: the caller knows that every modulus of the launch is narrow (ntt_is_narrow: the data primes of every shipped parameter set) --
// large forward launches then take the 8-coefficient form compiled for 8 waves per SIMD (NTT_FORM_AUTO only).
void launch_ntt(int logn, bool inverse, u64 *data, size_t count, const NttTable *tabs, const int *modmap,
                int period, hipStream_t st, size_t latency_limbs = 0, bool narrow_only = false);
// forward NTT of limbs gathered from src[g] (reduced into the table's modulus on load), written to data + g*n.
// nored: the caller has checked ntt_gather_nored_ok for every (source, target) pair of the launch: no reduction on load
void launch_ntt_gather(int logn, const u64 *const *src, u64 *data, size_t count, const NttTable *tabs, const int *modmap, int period,
                       hipStream_t st, bool nored = false, size_t latency_limbs = 0);
// out = a (.) b per limb; a:[batch][polys][L][n], b:[batch][L][n] (b_batch_stride may be 0)
void launch_dyadic_plain(const DevLevel *lv, const u64 *ct, const u64 *pt, u64 *out, int polys, size_t n, int batch,
                         size_t pt_batch_stride, hipStream_t st);
void launch_add(const DevLevel *lv, u64 *acc, const u64 *x, int polys, size_t n, int batch, hipStream_t st);
// acc[b] += sum_{i<terms} x[b][i]   (x: [batch][terms][polys][L][n]; acc: [batch] stride acc_stride words)
void launch_add_many(const DevLevel *lv, u64 *acc, size_t acc_stride, const u64 *x, int terms, int polys, size_t n,
                     int batch, hipStream_t st);
struct SumJob { const u64 *src; u64 *dst; int terms; int pad; };
void launch_sum_jobs(const DevLevel *lv, int L, const SumJob *jobs, int polys, size_t n, int njobs, hipStream_t st);
struct PlainJob { u64 *ct; const u64 *pt; };        // ct: c0 limbs [L][n] ; pt: n coefficients mod t
void launch_add_plain(const DevLevel *lv, const PlainJob *jobs, size_t n, int batch, hipStream_t st);
void launch_lift(const DevLevel *lv, const u64 *pt, u64 *out, size_t n, int batch, const unsigned char *no_lift,
                 hipStream_t st);
// drop last limb: ct c at in + c*in_stride holds `polys` polys [L][n] -> out packed [c][polys][L-1][n]
void launch_modswitch(const DevLevel *lv, const u64 *in, size_t in_stride, int polys, u64 *out, size_t n, int cts,
                      hipStream_t st);
void launch_clear_bits(u64 *ct, size_t words, int bits, hipStream_t st);
struct CtJob { const u64 *src; u64 *dst; };
void launch_copy_jobs(const CtJob *jobs, size_t words, int njobs, hipStream_t st);
// copy of query source ciphertexts ([2][L][n] words each) that flags words outside [0, q_limb) in *bad (device-visible host memory)
void launch_copy_sources(const CtJob *jobs, size_t words, int njobs, const DevLevel *lv, int L, size_t n, unsigned *bad, unsigned seq, hipStream_t st);
// drop last limb of `polys` polynomials per job: src [polys][L][n] -> dst [polys][L-1][n]
void launch_modswitch_jobs(const DevLevel *lv, const CtJob *jobs, int polys, size_t n, int njobs, hipStream_t st);
void launch_fill_random(u64 *out, size_t words, u64 seed, u64 bound, hipStream_t st);
// out[i] = (the (first + i)-th 32-bit output of SEAL's Blake2xb generator under `seed`) % bound
void launch_fill_blake2xb(u64 *out, size_t words, const Blake2xbSeed &seed, u64 first, u64 bound, hipStream_t st);
// N3: c1 of seeded ciphertexts / keys = util::sample_poly_uniform under SEAL's Blake2xb generator (seal_codec.h), dst[L][n] at level lv.
// rej: [njobs][1 + 8192] u32, zeroed once (the kernels leave the counters at zero); *overflow is set when a ciphertext has more
// rejected words than the list holds (a modulus within a factor 3 of 2^64: not a SEAL modulus).
struct SeedJob { Blake2xbSeed seed; u64 *dst; };
void launch_seed_expand(const SeedJob *jobs, int njobs, const DevLevel *lv, int L, const u64 *max_multiple, size_t n, u32 *rej, int *overflow,
                        hipStream_t st);
// N1: BinBundle build (polyn_with_roots per bin, BatchEncoder scatter, monomial detection)
void launch_polyn_with_roots(const u64 *roots, const u32 *counts, u32 bins, u32 stride, u32 max_deg, Mod t, u64 *poly, size_t n,
                             hipStream_t st);
void launch_scatter_slots(const u64 *in, const u32 *slot_map, u64 *out, size_t n, int batch, hipStream_t st);
void launch_gather_slots(const u64 *in, const u32 *slot_map, u64 *out, size_t n, int batch, hipStream_t st);
// N1: algebraize_item for `count` 16-byte items -> out[count][felts]; bpf = bits per field element, item_bits = felts * bpf
void launch_algebraize(const unsigned char *items, size_t count, u32 felts, u32 bpf, u32 item_bits, u64 *out, hipStream_t st);
// N4: item -> 128-bit block packing for PEQT, and the rounding step of the querier's decryption
void launch_pack_blocks(const u64 *values, size_t n, u32 items, u32 felts, u32 len, u64 *out, int batch, hipStream_t st);
void launch_decrypt_round(const u64 *ct, size_t ct_stride, const u64 *v, u64 q0, u64 t, u64 *out, size_t n, int batch, hipStream_t st);
void launch_flag_monomial(const u64 *pt, size_t n, int batch, unsigned char *flag, hipStream_t st);
// BEHZ
// ct c at in + c*in_stride holds `polys` polys [L][n]; out packed [c][polys][E][n]
void launch_behz_ext(const DevLevel *lv, int L, int nB, const u64 *in, size_t in_stride, int polys, u64 *out, size_t n, int cts,
                     hipStream_t st);
// mod_switch_to_next + extension in one pass (input: L + 1 limbs per polynomial at level lv + 1); false = not available for this size
// raw: `in` comes from an inverse NTT that left out its twist (NTT_MAP_RAW); needs lv[1].drop_tw / last_tw
bool launch_drop_behz_ext(const DevLevel *lv, int L, int nB, const u64 *in, size_t in_stride, int polys, u64 *out, size_t n, int cts,
                          hipStream_t st, bool raw = false);
struct TensorJob { const u64 *a, *b; u64 *d; };   // a,b: [2][E][n] ext-NTT ; d: [3][E][n]
void launch_tensor(const DevLevel *lv, const TensorJob *jobs, size_t n, int batch, hipStream_t st);
// operands of any size: a: [sa][E][n], b: [sb][E][n] ext-NTT ; d: [sa + sb - 1][E][n]   (no key switching: products are never relinearised)
struct TensorConvJob { const u64 *a, *b; u64 *d; int sa, sb; };
void launch_tensor_conv(const DevLevel *lv, const TensorConvJob *jobs, size_t n, int njobs, hipStream_t st);
// The same for a sum of products sharing one output (eval_patstock's sum over i): a, b: [terms][2][E][n];
// dq: [terms][3][L][n] per-term q limbs; bs: [3][nBsk][n] Bsk limbs summed over the terms
struct TensorSumJob { const u64 *a, *b; u64 *dq, *bs; int terms; int pad; };
// e0: first ext limb handled (0: everything; L: only the Bsk sums, the per-term q limbs being formed by launch_intt_tensor)
void launch_tensor_sum(const DevLevel *lv, int E, const TensorSumJob *jobs, size_t n, int njobs, int e0, hipStream_t st);
// tensor product + inverse NTT in one launch: njobs products x 3 polys x `limbs` limbs (operand polys src_ps words apart,
// output job.d[3][limbs][n], coefficient form), followed by n_plain limbs at `plain` transformed in place; modmap covers both
// (grid order: the three workgroups of one (product, limb) pair -- they read the same operand limbs -- on one XCD)
void launch_intt_tensor(int logn, const TensorJob *jobs, int njobs, int limbs, size_t src_ps, u64 *plain, size_t n_plain,
                        const NttTable *tabs, const int *modmap, int period, hipStream_t st, size_t latency_limbs = 0);
struct FinishSumJob { const u64 *dq, *bs; u64 *out; int terms; int pad; };   // out: [3][L][n] = sum of the finished terms
void launch_behz_finish_sum(const DevLevel *lv, int L, int nB, const FinishSumJob *jobs, size_t n, int njobs, hipStream_t st);
// finish: out[3][L][n] (+)= sum over `terms` consecutive products d[term][3][E][n] (coeff form)
struct FinishJob { const u64 *d; u64 *out; int terms; int pad; };
void launch_behz_finish(const DevLevel *lv, int L, int nB, const FinishJob *jobs, bool accumulate, size_t n, int njobs, hipStream_t st);
// key switching
void launch_ks_inner(const DevKey *key, int L, const u64 *tdec, const u64 *rk, u64 *acc, size_t n, int batch,
                     hipStream_t st);
// raw: acc comes from an inverse NTT that left out its twist (L <= 4; needs key->md_tw / p_tw)
void launch_ks_moddown(const DevKey *key, int L, const u64 *acc, u64 *ct, size_t ct_stride, size_t n, int batch,
                       hipStream_t st, const DevLevel *lv = nullptr, u64 *ext = nullptr, int n_ext = 0, bool raw = false);
// kara: the three-product accumulation (k_mac<.., true>, lv->mac_chunk_k)
// packed: every job of the launch reads bit-packed plaintexts (MacJob::packed)
void launch_mac(const DevLevel *lv, int nlimbs, const MacJob *jobs, size_t n, int njobs, hipStream_t st, bool kara = false, bool packed = false);
// single products on one limb: out[0][k] = a[k] * pw[limb][k], out[out_poly_stride + k] = a[k] * pw[pw_poly_stride + limb n + k]  (mod q_limb);
// pt: the plaintext's slot as k_mac takes it (dense: [L][n] words; packed: the bit-packed slot, rows per DevLevel::mac_row_off)
struct TermJob { const u64 *pt; const u64 *pw; u64 *out; };
void launch_term_product(const DevLevel *lv, const TermJob *jobs, size_t njobs, size_t n, int limb, u32 pw_poly_stride, u32 out_poly_stride,
                         bool packed, hipStream_t st);
// dense [slots][L][n] u64 <-> bit-packed [slots][slot_bytes] (rows per DevLevel::mac_bits / mac_row_off of `lv`)
void launch_pack_rows(const DevLevel *lv, int L, const u64 *dense, void *packed, size_t slot_bytes, size_t n, size_t slots, hipStream_t st);
void launch_unpack_rows(const DevLevel *lv, int L, const void *packed, size_t slot_bytes, u64 *dense, size_t n, size_t slots, hipStream_t st);
// Fused tail of eval / eval_patstock (bin_bundle.cpp:159-171, 345-357): (c0,c1) (+ optional exact addends) + Delta*a0 +
// Delta*mask, drop limbs down to the last level, clear the irrelevant bits, write the 2n-word result.
struct EpiJob { const u64 *ct; const u64 *add1; const u64 *add2; const u64 *a0; const u64 *mask; u64 *out;
                const u64 *ks_acc = nullptr; };   // round 6: RAW key-switch sums [2][L+1][n] whose mod-down the epilogue performs (launch_eval_epilogue with a key)
void launch_eval_epilogue(const DevLevel *levels, int lvl, const EpiJob *jobs, size_t ct_poly_stride, int clear_bits, size_t n,
                          int njobs, hipStream_t st, const DevKey *key = nullptr);
// i = 0 block of eval_patstock when exactly one limb is dropped (bin_bundle.cpp:314-324, note N1):
// acc[p][m] += (S[p][m] + terms*half - sum_t ((V[t][p] + half) mod q_last)) * q_last^-1  mod q_m
struct I0Job { const u64 *s; const u64 *v; u64 *acc; int terms; int store; };   // s:[2][L-1][n] v:[terms][2][n] acc:[2][L-1][n] (store: = instead of +=)
// raw: s and v come from an inverse NTT that left out its twist (needs lv_low->drop_tw / last_tw)
void launch_i0_finish(const DevLevel *lv_low, const I0Job *jobs, size_t n, int njobs, hipStream_t st, bool raw = false);

} // namespace apsu_he
