/*
 * apsu_he.h — C ABI of the MI355X homomorphic query-evaluation engine for APSU.
 *
 * Drop-in boundary: these entry points replace the seal::Evaluator calls that the DB-holding
 * party (apsu::receiver::Receiver) makes on its hot path.  The reference has no FFI for this
 * path; the replaced interface is the C++ class seal::Evaluator reached through
 * CryptoContext::evaluator() (common/apsu/crypto_context.h:101-104).  Each function below
 * cites the reference call site(s) it replaces.  INTEGRATION.md shows the adapter a maintainer
 * adds to receiver/apsu/receiver_osn.cpp and receiver/apsu/bin_bundle.cpp.
 *
 * Conventions
 *  - Buffers are raw uint64_t limb arrays in SEAL's in-memory order [poly][limb][coeff], i.e.
 *    exactly Ciphertext::data() / Plaintext::data().
 *  - Levels are named by SEAL's chain_index (0 = last level); get_parms_id_for_chain_idx
 *    (common/apsu/util/utils.cpp:179-189) gives the mapping on the SEAL side.
 *  - Every function returns 0 on success or a negative apsu_he_status; the message of the last
 *    failure on the calling thread is available from apsu_he_last_error().  C++ callers map
 *    APSU_HE_INVALID_ARGUMENT -> std::invalid_argument, others -> std::runtime_error, which is
 *    what the reference's callers see from SEAL (bin_bundle.cpp:116-118,204-213).
 *  - All-zero operands are legal (SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT=OFF, bin_bundle.cpp:111-114).
 *  - Calls on one context are thread-safe (serialised internally); the reference calls the
 *    Evaluator from a thread pool (receiver_osn.cpp:334-364).
 *  - There is NO CPU fallback: apsu_he_create fails with APSU_HE_NO_DEVICE without a GPU.
 */
#ifndef APSU_HE_H
#define APSU_HE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    APSU_HE_OK = 0,
    APSU_HE_INVALID_ARGUMENT = -1,   /* std::invalid_argument in the reference */
    APSU_HE_LOGIC_ERROR = -2,        /* std::logic_error */
    APSU_HE_RUNTIME_ERROR = -3,      /* std::runtime_error (incl. HIP failures) */
    APSU_HE_NO_DEVICE = -4,          /* no MI355X / HIP device visible */
    APSU_HE_OUT_OF_MEMORY = -5
} apsu_he_status;

typedef struct apsu_he_ctx apsu_he_ctx;         /* CryptoContext + Evaluator replacement */
typedef struct apsu_he_relin apsu_he_relin;     /* device-resident seal::RelinKeys */
typedef struct apsu_he_bundle apsu_he_bundle;   /* device-resident BinBundleCache::batched_matching_polyn */
typedef struct apsu_he_powers apsu_he_powers;   /* device-resident CiphertextPowers for some bundle indices */

typedef struct {
    uint64_t poly_modulus_degree;
    uint64_t plain_modulus;
    int32_t coeff_modulus_size;      /* K, limbs at key level */
    int32_t first_chain_idx;         /* chain_index of the first data level */
    int32_t using_keyswitching;
    int32_t irrelevant_bit_count;    /* bin_bundle.cpp:67-97 */
    uint64_t coeff_modulus[8];
    /* PSUParams-derived (zero when created without PSUParams) */
    uint32_t ps_low_degree, max_items_per_bin, bundle_idx_count, items_per_bundle;
    uint32_t source_power_count, target_power_count, powers_dag_depth;
    uint32_t result_polys;           /* polynomials per row of apsu_he_eval_bundles' output: 2 with key switching; more only for
                                      * parameter sets with ONE coefficient prime, whose products are never relinearised
                                      * (receiver_osn.cpp:416,430-432 ; bin_bundle.cpp:238-240,308-310)   (ABI 4; was `reserved`) */
} apsu_he_info;

typedef struct { uint32_t power, depth, parent1, parent2; } apsu_he_dag_node;   /* powers.h:53-77 */

const char *apsu_he_last_error(void);
/* 1: tiers 1 and 2, N1, N2, N4.  2 (additive): apsu_he_multi_*, apsu_he_eval_all, apsu_he_partition_bundles, apsu_he_wire_*,
 * apsu_he_set_async_results / apsu_he_sync / apsu_he_stream, apsu_he_mask_generate_blake2xb.
 * 6: apsu_he_set_query_overlap.  7: its modes 2 and 3; the APSU_HE_* environment switches of measured-and-decided A/B experiments are gone.
 * 3: apsu_he_set_eval_pipeline removed (measured-negative scheduling experiment, profiles/r03_eval_pipeline.txt); added
 * apsu_he_debug_counters, apsu_he_phase_* / apsu_he_multi_phase_*, apsu_he_eval_all_ex + apsu_he_host_alloc, apsu_he_partition_bundles_ex,
 * the SEAL object codec apsu_he_seal_*, apsu_he_seed_expand, apsu_he_run_query_request; poly_modulus_degree 32768.
 * 4 (additive): ciphertexts of more than two polynomials for parameter sets without key switching -- apsu_he_multiply_sized,
 * apsu_he_power_size, apsu_he_bundle_result_size, apsu_he_info.result_polys (the former `reserved`); those sets were refused before.
 * apsu_he_algebraize_items (N1: item -> field elements); apsu_he_db_file_* / apsu_he_multi_db_load_file / _save_file (N2: the whole DB
 * in one mmap-able file); apsu_he_seal_pt_load / _save, apsu_he_db_upload_bundle_serialized (BinBundle caches as the reference stores
 * them), apsu_he_db_upload_saved_bundle (a BinBundle as ReceiverDB::save wrote it); zstd bodies in the SEAL codec;
 * apsu_he_multi_run_query_request, apsu_he_multi_result_polys; the parameter exchange, plainResponse, PSUParams in binary form and
 * the header of a saved ReceiverDB (apsu_he_wire_peek_type ... apsu_he_wire_receiver_db_header).
 * 5 (additive): apsu_he_set_tier1_on_device (tier-1 calls on device-resident operands without a host round trip per call);
 * BinBundle images and DB files carry a row format (bit-packed database rows, the default, or dense words) and load into either. */
#define APSU_HE_ABI_VERSION 7   /* what this header describes; compare with apsu_he_abi_version() of the loaded library */
int apsu_he_abi_version(void);

/* ---- lifetime ------------------------------------------------------------------------------ */
/* PSUParams::Load(json) + CryptoContext(params) (psu_params.cpp:290-374, crypto_context.h:32-37) */
int apsu_he_create(const char *psu_params_json, int device, apsu_he_ctx **out);
/* SEALContext from explicit primes (tests / tier-1 use without PSUParams) */
int apsu_he_create_raw(uint64_t poly_modulus_degree, const uint64_t *coeff_modulus, int coeff_modulus_size,
                       uint64_t plain_modulus, int device, apsu_he_ctx **out);
int apsu_he_destroy(apsu_he_ctx *ctx);
int apsu_he_get_info(const apsu_he_ctx *ctx, apsu_he_info *out);
/* PowersDag::configure result (powers.cpp:22-107); nodes ascending by power. Returns count via *n_nodes. */
int apsu_he_get_powers_dag(const apsu_he_ctx *ctx, apsu_he_dag_node *nodes, int capacity, int *n_nodes);

/* ---- tier 1: one call per Evaluator method (host buffers) ---------------------------------- */
/* Evaluator::transform_to_ntt_inplace(Ciphertext)        receiver_osn.cpp:467,475 */
int apsu_he_transform_to_ntt(apsu_he_ctx *ctx, uint64_t *ct, int polys, int chain_idx);
/* Evaluator::transform_from_ntt_inplace                  bin_bundle.cpp:154,268,297,321 */
int apsu_he_transform_from_ntt(apsu_he_ctx *ctx, uint64_t *ct, int polys, int chain_idx);
/* Evaluator::transform_to_ntt_inplace(Plaintext, parms)  bin_bundle.cpp:419 ; out: (chain_idx+1)*n words */
int apsu_he_transform_plain_to_ntt(apsu_he_ctx *ctx, const uint64_t *pt_mod_t, size_t pt_coeff_count, uint64_t *out,
                                   int chain_idx);
/* Evaluator::multiply_plain, NTT ct x NTT plaintext      bin_bundle.cpp:147,258,287,320 */
int apsu_he_multiply_plain_ntt(apsu_he_ctx *ctx, const uint64_t *ct, const uint64_t *pt_ntt, uint64_t *out, int polys,
                               int chain_idx);
/* Evaluator::multiply_plain, coefficient ct x coefficient plaintext   bin_bundle.cpp:334 */
int apsu_he_multiply_plain(apsu_he_ctx *ctx, const uint64_t *ct, const uint64_t *pt_mod_t, size_t pt_coeff_count,
                           uint64_t *out, int polys, int chain_idx);
/* Evaluator::add_inplace                                  bin_bundle.cpp:148,264,273,293,303,323,336 */
int apsu_he_add(apsu_he_ctx *ctx, uint64_t *acc, const uint64_t *x, int polys, int chain_idx);
/* Evaluator::add_plain_inplace (adds to c0)               bin_bundle.cpp:159,162,345,346 */
int apsu_he_add_plain(apsu_he_ctx *ctx, uint64_t *ct, const uint64_t *pt_mod_t, size_t pt_coeff_count, int chain_idx);
/* Evaluator::multiply / multiply_inplace (size 2 x size 2 -> size 3)   receiver_osn.cpp:424 ; bin_bundle.cpp:272,301 */
int apsu_he_multiply(apsu_he_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out3, int chain_idx);
/* Evaluator::square                                        receiver_osn.cpp:422 */
int apsu_he_square(apsu_he_ctx *ctx, const uint64_t *a, uint64_t *out3, int chain_idx);
/* Evaluator::multiply for operands that were never relinearised (one coefficient prime: receiver_osn.cpp:424 with :430-432
 * skipped ; bin_bundle.cpp:272,301): out has size_a + size_b - 1 polynomials; APSU_HE_INVALID_ARGUMENT beyond SEAL's largest
 * ciphertext (16 polynomials), where SEAL throws */
int apsu_he_multiply_sized(apsu_he_ctx *ctx, const uint64_t *a, int size_a, const uint64_t *b, int size_b, uint64_t *out, int chain_idx);
/* Evaluator::relinearize_inplace (size 3 -> size 2, in place in the first two polys)
 *                                                          receiver_osn.cpp:431 ; bin_bundle.cpp:309 */
int apsu_he_relinearize(apsu_he_ctx *ctx, uint64_t *ct3, const apsu_he_relin *rk, int chain_idx);
/* Evaluator::mod_switch_to_next_inplace; result packed [polys][chain_idx][n] at the front of ct
 *                                                          receiver_osn.cpp:463,471,478 ; bin_bundle.cpp:169,269,298,322,355 */
int apsu_he_mod_switch_to_next(apsu_he_ctx *ctx, uint64_t *ct, int polys, int chain_idx);
/* try_clear_irrelevant_bits (last level, one limb)         bin_bundle.cpp:67-97 */
int apsu_he_clear_irrelevant_bits(apsu_he_ctx *ctx, uint64_t *ct_last_level, int polys);

/* ---- tier 2: fused, HBM-resident ------------------------------------------------------------ */
/* RelinKeys of the query: [decomp K-1][component 2][limb K][n] = key_vector[J].data() of
 * KSwitchKeys::data()[0], NTT form (query.cpp:46-52, crypto_context.h:45-49) */
int apsu_he_relin_upload(apsu_he_ctx *ctx, const uint64_t *ksk, apsu_he_relin **out);
int apsu_he_relin_free(apsu_he_relin *rk);
/* BatchedPlaintextPolyn ctor output (bin_bundle.cpp:366-430): coeff_ptrs[d] = Plaintext::data() of
 * batched_coeffs[d]; is_ntt[d] per the rule at :418-420 (checked).  NTT-form plaintexts have
 * (plain_level+1)*n words, coefficient-form ones n words.  The engine copies; caller may free.
 * Called from ReceiverDB::generate_caches (receiver_db.cpp:808-820). */
int apsu_he_db_upload_bundle(apsu_he_ctx *ctx, uint32_t bundle_idx, uint32_t cache_idx, uint32_t n_coeffs,
                             const uint64_t *const *coeff_ptrs, const uint8_t *is_ntt, apsu_he_bundle **out);
/* Synthetic BinBundle for benchmarks: coefficient d, index k of the batched polynomial (coefficient
 * form, mod t) = splitmix64_mix(seed + (d*n + k + 1) * 0x9e3779b97f4a7c15) % t; generated on the GPU. */
int apsu_he_db_random_bundle(apsu_he_ctx *ctx, uint32_t bundle_idx, uint32_t cache_idx, uint32_t degree, uint64_t seed,
                             apsu_he_bundle **out);
/* "next" row N1: BinBundle::regen_polyns + regen_plaintexts on the GPU
 * (bin_bundle.cpp:934-1026 -> polyn_with_roots, common/apsu/util/interpolate.cpp:63-80 -> BatchedPlaintextPolyn
 * ctor bin_bundle.cpp:366-430: BatchEncoder::encode, transform_to_ntt).  roots[bin*stride + r], r < counts[bin],
 * are the field elements (mod plain_modulus) stored in bin `bin`; bins <= poly_modulus_degree; empty bins give the
 * polynomial 1.  The result is identical to uploading the reference-built cache with apsu_he_db_upload_bundle. */
int apsu_he_db_build_bundle(apsu_he_ctx *ctx, uint32_t bundle_idx, uint32_t cache_idx, const uint64_t *roots,
                            const uint32_t *counts, uint32_t bins, uint32_t stride, apsu_he_bundle **out);
/* "next" row N2: engine-native image of one BinBundle cache (256-byte header with a parameter
 * fingerprint and checksum + the raw limb arrays), the GPU-resident counterpart of ReceiverDB::save / Load
 * (receiver/apsu/receiver_db.cpp:1182-1429, bin_bundle.fbs).  The buffer may be an mmap of a file. */
int apsu_he_bundle_image_size(apsu_he_ctx *ctx, const apsu_he_bundle *b, uint64_t *bytes);
int apsu_he_bundle_save(apsu_he_ctx *ctx, const apsu_he_bundle *b, uint8_t *buf, uint64_t capacity, uint64_t *written);
int apsu_he_bundle_load(apsu_he_ctx *ctx, const uint8_t *buf, uint64_t size, apsu_he_bundle **out);
/* "next" row N4: the per-BinBundle host work either side of the evaluation, on the GPU.
 * apsu_he_mask_generate replaces the "random gen" block of Receiver::RunQuery (receiver/apsu/receiver_osn.cpp:217-284):
 * for each of `count` BinBundles, n slot values uniform mod plain_modulus, BatchEncoder::encode of them
 * (masks_dev: count*n words on the device, the `masks` of apsu_he_eval_bundles with masks_on_device = 1), and
 * vec_to_oc_block of every item (receiver_osn.cpp:53-73; blocks: count*items_per_bundle*2 words, (low, high)
 * halves of the 128-bit block; host; may be NULL).  values (host, count*n; may be NULL) receives the slot values.
 * The reference draws them from SEAL's Blake2xb PRNG under a fresh random seed; here value(c, i) =
 * splitmix64(seed + (c*n + i + 1) * 0x9e3779b97f4a7c15) mod plain_modulus — a different uniform stream, the
 * same encode and packing.
 * apsu_he_decrypt_decode is the querier's side of a loopback check (sender/apsu/sender_osn.cpp:675-700,
 * common/apsu/network/result_package.cpp:175-213): Decryptor::decrypt of `count` results (size 2, last level,
 * 2*n words each), BatchEncoder::decode, and the same block packing.  sk_ntt: the secret key modulo q_0 in NTT
 * form (n words, host).  The rounding is the exact round(t*x/q_0): equal to SEAL's decrypt for every
 * ciphertext with a positive noise budget. */
int apsu_he_mask_generate(apsu_he_ctx *ctx, uint64_t seed, uint32_t count, uint64_t *masks_dev, uint64_t *values, uint64_t *blocks);
/* The same with the reference's own generator (receiver_osn.cpp:221-224: UniformRandomGeneratorInfo(prng_type::blake2xb,
 * seed).make_prng(); :248-251: `generate() % plain_modulus`, generate() being SEAL's 32-bit draw): seed = the eight words of
 * seal::prng_seed_type (the reference fills them with random_bytes); value(c, i) = (output number first_value + c*n + i of that
 * generator) % plain_modulus, so one call with first_value = 0 over the BinBundles in the reference's loop order
 * (cache_idx outer, bundle_idx inner, padded caches skipped) reproduces its masks for the same seed, and several calls
 * can continue one stream.  BLAKE2b is checked against RFC 7693 / hashlib; the BLAKE2X expansion and SEAL's buffering
 * (4096-byte buffers keyed by the seed, message = buffer counter) are restated from their published sources — unpinned,
 * like every other SEAL-derived detail. */
int apsu_he_mask_generate_blake2xb(apsu_he_ctx *ctx, const uint64_t seed[8], uint64_t first_value, uint32_t count, uint64_t *masks_dev,
                                   uint64_t *values, uint64_t *blocks);
int apsu_he_decrypt_decode(apsu_he_ctx *ctx, const uint64_t *sk_ntt, const uint64_t *cts, int cts_on_device, uint32_t count,
                           uint64_t *values, uint64_t *blocks);
/* N1, one step earlier: util::algebraize_item (common/apsu/util/db_encoding.cpp:209-256,360-366; called at
 * receiver_db.cpp:296-298 on every OPRF'd item) for `count` hashed items of 16 bytes each: felts[i * felts_per_item + j] =
 * bits [j*b, (j+1)*b) of item i's first item_bit_count bits, read as a little-endian bit string, b = bit_count(plain_modulus) - 1.
 * These are the roots apsu_he_db_build_bundle takes once the host has placed them into bins.  (ABI 4) */
int apsu_he_algebraize_items(apsu_he_ctx *ctx, const uint8_t *items, size_t count, int items_on_device, uint64_t *felts, int felts_on_device);
/* N2 for the whole database -- the counterpart of ReceiverDB::save / Load (receiver/apsu/receiver_db.cpp:1182-1429) for a DB
 * that lives in HBM as raw limb arrays: ONE file per parameter set = header with the parameter fingerprint, a table
 * (bundle index, cache index, degree, offset, size) and the BinBundle images above at 4096-byte aligned offsets.  The file is
 * mapped (mmap), so a process touches only the BinBundles it loads, each array going to the device with one copy from the
 * mapping: a device of a node reads its shard of a 75 GiB database, not the file.  apsu_he_db_file_save writes to path + ".tmp"
 * and renames.  A file written for other parameters, truncated or with a damaged table / image is refused
 * (APSU_HE_INVALID_ARGUMENT).  (ABI 4) */
typedef struct apsu_he_db_file apsu_he_db_file;
int apsu_he_db_file_save(apsu_he_ctx *ctx, const char *path, const apsu_he_bundle *const *bundles, int count);
int apsu_he_db_file_open(const char *path, apsu_he_db_file **out);
int apsu_he_db_file_close(apsu_he_db_file *f);
int apsu_he_db_file_count(const apsu_he_db_file *f, int *count, uint64_t *file_bytes);
int apsu_he_db_file_entry(const apsu_he_db_file *f, int i, uint32_t *bundle_idx, uint32_t *cache_idx, uint32_t *degree, uint64_t *image_bytes);
int apsu_he_db_file_load(apsu_he_ctx *ctx, const apsu_he_db_file *f, int i, apsu_he_bundle **out);
/* test hooks: degree of the batched polynomial; stored form of coefficient `degree`
 * (kind 0: raw mod t [n]; 1: NTT form [(plain_level+1)*n]; 2: pre-lifted + NTT at the high level [(high+1)*n]) */
int apsu_he_bundle_degree(const apsu_he_bundle *b, uint32_t *degree);
int apsu_he_bundle_download(apsu_he_ctx *ctx, const apsu_he_bundle *b, uint32_t degree, uint64_t *out, size_t capacity_words,
                            size_t *words, int *kind);
/* Polynomials of a target power after ComputePowers and of one BinBundle's result: 2 with key switching.  Without it
 * (one coefficient prime) the reference never relinearises: a product has size(parent1) + size(parent2) - 1 polynomials,
 * eval's result the size of its longest power, eval_patstock's the size of its longest product (at least 3). */
int apsu_he_power_size(const apsu_he_ctx *ctx, uint32_t power, uint32_t *polys);
int apsu_he_bundle_result_size(const apsu_he_ctx *ctx, const apsu_he_bundle *b, uint32_t *polys);
int apsu_he_bundle_free(apsu_he_bundle *b);
int apsu_he_bundle_bytes(const apsu_he_bundle *b, uint64_t *db_bytes);
/* Receiver::ComputePowers for n_bundle_idx bundle indices at once (receiver_osn.cpp:320-328,395-488).
 * src_cts[b * source_power_count + s] = query ciphertext of the s-th source power (ascending) for
 * bundle index bundle_indices[b]: size 2, coefficient form, first data level (receiver_osn.cpp:304-317).
 * src_on_device != 0: the pointers are device pointers (inputs already resident in HBM).
 * Every coefficient must be a canonical residue of its limb's prime, which is what a valid seal::Ciphertext holds
 * (seal::is_data_valid_for, checked by SEALObject::extract in the reference, seal_object.h:161-219): the engine's lazy transforms
 * take source limbs as they are.  apsu_he_run_query_request checks it on the decoded objects and fails like the reference; device-resident
 * sources are checked by the kernel that gathers them, and a violation is reported (APSU_HE_INVALID_ARGUMENT) by the call that next
 * waits for that work: a synchronous apsu_he_eval_bundles, apsu_he_powers_download or apsu_he_sync. */
int apsu_he_compute_powers(apsu_he_ctx *ctx, const uint32_t *bundle_indices, int n_bundle_idx,
                           const uint64_t *const *src_cts, int src_on_device, const apsu_he_relin *rk,
                           apsu_he_powers **out);
int apsu_he_powers_free(apsu_he_powers *p);
/* test hook: copy one computed power back (form/level per receiver_osn.cpp:459-487); words = apsu_he_power_size * (level+1) * n */
int apsu_he_powers_download(apsu_he_ctx *ctx, const apsu_he_powers *p, uint32_t bundle_idx, uint32_t power,
                            uint64_t *out, size_t capacity_words, int *chain_idx, int *is_ntt);
/* Receiver::ProcessBinBundleCache -> BatchedPlaintextPolyn::eval / eval_patstock for `count`
 * BinBundles (receiver_osn.cpp:490-540 ; bin_bundle.cpp:106-174,192-360).  masks[i] = random_plain
 * (n coefficients mod t, receiver_osn.cpp:217-284).  out_cts: count * 2 * n words, result i at
 * out_cts + i*2*n, last level, irrelevant bits cleared.  *_on_device: pointers are device pointers.
 * Parameter sets without key switching: rows of apsu_he_info.result_polys * n words, result i = its first
 * apsu_he_bundle_result_size polynomials, zeros behind them. */
int apsu_he_eval_bundles(apsu_he_ctx *ctx, const apsu_he_bundle *const *bundles, int count, const apsu_he_powers *powers,
                         const apsu_he_relin *rk, const uint64_t *const *masks, int masks_on_device, uint64_t *out_cts,
                         int out_on_device);

/* ---- several GPUs of one node behind one handle ------------------------------------------------------------------
 * The in-process counterpart of Receiver::RunQuery's fan-out (receiver/apsu/receiver_osn.cpp:320-364: ComputePowers per
 * bundle index, then one ProcessBinBundleCache task per BinBundle on the thread pool).  One engine and one host thread per
 * device; the BinBundle is the sharded unit: devices are assigned to bundle indices first (a device then needs the
 * powers of few indices only), an index's BinBundles are split over its devices by cost ~ degree.  A query's only data
 * exchange is the final gather of the fixed-size results.  Devices may repeat (e.g. {0, 0}) to rehearse
 * the multi-device path on one GPU.  (bench.py's one-process-per-GPU launch uses the same partition rule and RCCL.) */
typedef struct apsu_he_multi apsu_he_multi;
/* the partition rule alone (no GPU needed): device slot of each BinBundle (bundle_idx, cache_idx, degree) */
int apsu_he_partition_bundles(uint32_t bundle_idx_count, int n_devices, const uint32_t *bundle_idx, const uint32_t *cache_idx,
                              const uint32_t *degree, int count, int *device_slot);
/* The same with a spill pass: compute_powers_cost = what ComputePowers for ONE bundle index costs in the rule's unit (degree + 64
 * per BinBundle; apsu_he_compute_powers_cost gives the MI355X figure for a context's PowersDag).  BinBundles then move off the
 * slowest device to devices of OTHER bundle indices while that lowers the slowest device's cost including the second
 * ComputePowers the receiving device has to run (3 bundle indices on 8 devices: one index has two devices and 1.5x the load). */
int apsu_he_partition_bundles_ex(uint32_t bundle_idx_count, int n_devices, const uint32_t *bundle_idx, const uint32_t *cache_idx,
                                 const uint32_t *degree, int count, uint64_t compute_powers_cost, int *device_slot);
int apsu_he_compute_powers_cost(const apsu_he_ctx *ctx, uint64_t *cost);
int apsu_he_multi_create(const char *psu_params_json, const int *devices, int n_devices, apsu_he_multi **out);
int apsu_he_multi_destroy(apsu_he_multi *m);
int apsu_he_multi_device_count(const apsu_he_multi *m, int *n_devices);
/* polynomials per output row of apsu_he_eval_all(_ex) = apsu_he_info.result_polys of the parameter set (2 with key switching) */
int apsu_he_multi_result_polys(const apsu_he_multi *m, uint32_t *polys);
/* the query's RelinKeys, replicated on every device (layout as apsu_he_relin_upload) */
int apsu_he_multi_relin_upload(apsu_he_multi *m, const uint64_t *ksk);
/* DB placement (ReceiverDB::generate_caches, receiver_db.cpp:808-820): as apsu_he_db_upload_bundle / _random_bundle on the
 * device in `device_slot` (take it from apsu_he_partition_bundles).  *bundle_id = registration order = the BinBundle's
 * row in apsu_he_eval_all's masks and output. */
int apsu_he_multi_db_upload_bundle(apsu_he_multi *m, int device_slot, uint32_t bundle_idx, uint32_t cache_idx, uint32_t n_coeffs,
                                   const uint64_t *const *coeff_ptrs, const uint8_t *is_ntt, int *bundle_id);
int apsu_he_multi_db_random_bundle(apsu_he_multi *m, int device_slot, uint32_t bundle_idx, uint32_t cache_idx, uint32_t degree,
                                   uint64_t seed, int *bundle_id);
/* the whole file onto the handle's devices: BinBundles placed by apsu_he_partition_bundles_ex's rule (spill pass included), every
 * device reading its own shard from the shared mapping, all devices at once; bundle ids = the file's table order (appended to
 * what is registered already).  apsu_he_multi_db_save_file writes every registered BinBundle, in id order, into one file. */
int apsu_he_multi_db_load_file(apsu_he_multi *m, const apsu_he_db_file *f, int *n_loaded);
int apsu_he_multi_db_save_file(apsu_he_multi *m, const char *path);
int apsu_he_multi_db_clear(apsu_he_multi *m);
/* One query on all devices (receiver_osn.cpp:304-364): src_cts[b * source_power_count + s] = host ciphertext of source
 * power s (ascending) of bundle index b, for EVERY bundle index (each device uploads the ones it needs); masks[id] = n
 * words mod t (host).  out_cts: bundle count * 2n words, row = bundle id; host memory when out_device_slot < 0, else
 * device memory on that device (rows gathered with peer copies over xGMI). */
int apsu_he_eval_all(apsu_he_multi *m, const uint64_t *const *src_cts, const uint64_t *const *masks, uint64_t *out_cts,
                     int out_device_slot);
/* The same with the caller saying where its buffers live (flags) -- the query arrives from the network into host memory
 * (receiver_osn.cpp:290-317), a co-located producer may already hold it in HBM:
 *   APSU_HE_IO_SRC_PINNED / _MASKS_PINNED / _OUT_PINNED  that host buffer is page-locked AND device-visible (apsu_he_host_alloc,
 *       hipHostMalloc): the kernels that consume / produce it anyway (ComputePowers' source gather, the evaluation's epilogue)
 *       read and write it in place over PCIe -- no copy at all.  Unflagged host buffers are pageable and go through each
 *       device's own page-locked area (a few helper threads do the staging copy).
 *   APSU_HE_IO_SRC_ON_DEVICE / _MASKS_ON_DEVICE  the pointers are device pointers on devices[in_device_slot]; every device with
 *       peer access reads them in place over xGMI (the others get copy-engine peer copies).
 *   APSU_HE_IO_GATHER_RCCL  (out_device_slot >= 0) the gather is ONE RCCL all-gather of fixed-size rows (ncclAllGather over
 *       xGMI; librccl.so is loaded on first use) instead of the default, in which every device's epilogue kernel stores its
 *       rows straight into the output device's buffer (peer writes; copy-engine peer copies without peer access).  Falls back
 *       to the default when RCCL is absent or the device list repeats a device (apsu_he_multi_last_gather: "rccl" / "peer").
 * Nothing of a device's share is waited for before its last kernel is queued. */
#define APSU_HE_IO_SRC_PINNED 1u
#define APSU_HE_IO_MASKS_PINNED 2u
#define APSU_HE_IO_OUT_PINNED 4u
#define APSU_HE_IO_SRC_ON_DEVICE 8u
#define APSU_HE_IO_MASKS_ON_DEVICE 16u
#define APSU_HE_IO_GATHER_RCCL 32u
int apsu_he_eval_all_ex(apsu_he_multi *m, const uint64_t *const *src_cts, const uint64_t *const *masks, uint64_t *out_cts,
                        int out_device_slot, unsigned flags, int in_device_slot);
const char *apsu_he_multi_last_gather(const apsu_he_multi *m);
/* page-locked host memory for query buffers (any thread, any device) */
int apsu_he_host_alloc(size_t bytes, void **out);
int apsu_he_host_free(void *p);
/* phase timers of the multi-device entry (see apsu_he_phase_*): arrays of APSU_HE_PHASES; "Receiver::RunQuery" is the HOST
 * wall time of apsu_he_eval_all(_ex) (uploads and downloads included), the other two the slowest device's device time */
int apsu_he_multi_phase_enable(apsu_he_multi *m, int on);
int apsu_he_multi_phase_read(apsu_he_multi *m, uint64_t *count, double *avg_ms, double *min_ms, double *max_ms, int reset);

/* Scheduling option: ComputePowers may walk the high-power half of the PowersDag on a second HIP stream, next to the
 * low-power half and to the BinBundle inner products (bit-identical results).  mode -1 = default policy (on
 * whenever the PowersDag splits into independent halves and the inputs are device resident), 0 = off, 1 = on; the environment variable APSU_HE_SPLIT=0/1
 * (read at apsu_he_create) sets the default for contexts that never call this.  Event profiling (apsu_he_profile_enable) always uses one stream. */
int apsu_he_set_two_stream(apsu_he_ctx *ctx, int mode);

/* Device-resident pipelines: with on != 0, an apsu_he_eval_bundles call whose masks AND results live in device memory
 * returns as soon as its work is queued (apsu_he_compute_powers with device-resident sources always does).  The results
 * are complete after apsu_he_sync(ctx), or for work ordered after the context's main HIP stream (apsu_he_stream: a
 * hipStream_t; e.g. hipEventRecord on it + hipStreamWaitEvent on the consumer's stream).  The caller's device buffers must
 * stay alive and unmodified until then.  Host-memory arguments always synchronise, as does event profiling.
 * Default off. */
int apsu_he_set_async_results(apsu_he_ctx *ctx, int on);
/* Overlap of consecutive queries (ABI 6; modes 2 and 3 ABI 7).  With mode != 0 the caller promises that the device-resident inputs
 * of apsu_he_compute_powers -- the source ciphertexts and the relinearisation keys -- are COMPLETE when the call is made, i.e. not
 * still being produced by work queued on the context's stream (apsu_he_stream).  The engine's second stream then does not wait for
 * the main stream's queue at the start of a ComputePowers (receiver_osn.cpp:395-488) but only for the last evaluation that read the
 * powers buffer it is about to reuse, and a ComputePowers that finds an evaluation still running on the device is queued as ONE
 * chain on the second stream next to it (pipelined queries: -4 % on the rate of queued queries; one query alone takes the same
 * time).  mode 0 (default): off -- a caller that uploads a query on the context's stream and calls apsu_he_compute_powers behind it
 * (apsu_he_run_query_request does) needs the ordering.  1: on.  2: on, without the pipelined walk (only the high-power chain starts
 * early).  3: on, every ComputePowers takes the pipelined walk whether or not the device is busy (for tests: the walk's event chain
 * is exercised deterministically; slower for a query that runs alone).  Results are bit-identical in every mode. */
int apsu_he_set_query_overlap(apsu_he_ctx *ctx, int mode);
/* Tier 1 on device-resident operands (ABI 5): with on != 0 every pointer argument of the tier-1 calls (apsu_he_transform_to_ntt ...
 * apsu_he_clear_irrelevant_bits; relinearisation keys stay handles) is DEVICE memory -- or page-locked host memory, which the device
 * addresses -- and the calls return with their work queued on the context's stream instead of copying in, waiting and copying out:
 * a caller that replaces Evaluator methods one by one (receiver_osn.cpp:422-478, bin_bundle.cpp:143-170) keeps its ciphertexts in
 * HBM across calls and pays no host round trip per method.  Ordering and completion as for apsu_he_set_async_results:
 * calls on one context execute in call order; apsu_he_sync / apsu_he_stream give the completion point.  Default off. */
int apsu_he_set_tier1_on_device(apsu_he_ctx *ctx, int on);
int apsu_he_sync(apsu_he_ctx *ctx);
int apsu_he_stream(apsu_he_ctx *ctx, void **hip_stream);

/* ---- "next" row N3: the network framing around the path, without flatc / flatbuffers / SEAL -----------------
 * What the reference pins is the FlatBuffers framing of its messages; these functions read and write it:
 *   ReceiverOperationHeader                   common/apsu/network/rop_header.fbs ; receiver_operation.cpp:27-87
 *   ReceiverOperation{QueryRequest}           common/apsu/network/rop.fbs        ; receiver_operation.cpp:180-350
 *   ReceiverOperationResponse{QueryResponse}  common/apsu/network/rop_response.fbs
 *   ResultPackage                             common/apsu/network/result_package.fbs ; result_package.cpp:29-150
 * All buffers are size-prefixed (FinishSizePrefixed), exactly what ZMQChannel / StreamChannel carry.  The byte vectors
 * inside (Ciphertext.data, QueryRequest.relin_keys) are SEAL's own serialisation (seal_object.h:161-219); the reference
 * does not pin that format, so they are passed through as opaque byte ranges.  Parsers verify every offset, length and
 * alignment before use (like flatbuffers::Verifier) and fail with the reference's messages; returned pointers point INTO
 * the caller's buffer.  Builders return a malloc'ed buffer to release with apsu_he_wire_buffer_free.
 * ReceiverOperationType: 0 unknown, 1 parms, 2 oprf, 3 query, 4 response. */
typedef struct apsu_he_wire_query apsu_he_wire_query;
int apsu_he_wire_buffer_free(uint8_t *p);
int apsu_he_wire_build_header(uint32_t version, uint32_t type, uint8_t **out, size_t *out_size);
int apsu_he_wire_parse_header(const uint8_t *buf, size_t size, uint32_t *version, uint32_t *type);
/* ReceiverOperationQuery::save (receiver_operation.cpp:180-247): part i holds cts_per_part[i] ciphertext blobs, taken in
 * order from ct_data / ct_sizes; relin_keys NULL = field absent. */
int apsu_he_wire_build_query_request(uint8_t compression_type, const uint8_t *relin_keys, size_t relin_keys_size, uint32_t n_parts,
                                     const uint32_t *exponents, const uint32_t *cts_per_part, const uint8_t *const *ct_data,
                                     const size_t *ct_sizes, uint8_t **out, size_t *out_size);
/* ReceiverOperationQuery::load (receiver_operation.cpp:249-350): "unexpected operation type", "unsupported compression
 * mode", "invalid query data" (duplicate exponent) as in the reference */
int apsu_he_wire_parse_query_request(const uint8_t *buf, size_t size, apsu_he_wire_query **out);
int apsu_he_wire_query_free(apsu_he_wire_query *q);
int apsu_he_wire_query_info(const apsu_he_wire_query *q, uint8_t *compression_type, int *has_relin_keys, const uint8_t **relin_keys,
                            size_t *relin_keys_size, uint32_t *n_parts);
int apsu_he_wire_query_part(const apsu_he_wire_query *q, uint32_t part, uint32_t *exponent, uint32_t *n_cts);
int apsu_he_wire_query_ct(const apsu_he_wire_query *q, uint32_t part, uint32_t ct, const uint8_t **data, size_t *size);
int apsu_he_wire_build_query_response(uint32_t package_count, uint32_t alpha_max_cache_count, uint8_t **out, size_t *out_size);
int apsu_he_wire_parse_query_response(const uint8_t *buf, size_t size, uint32_t *package_count, uint32_t *alpha_max_cache_count);
/* ResultPackage::save / load (result_package.cpp:29-150) */
int apsu_he_wire_build_result_package(uint32_t bundle_idx, uint32_t cache_idx, const uint8_t *psu_result, size_t psu_result_size,
                                      uint32_t label_byte_count, uint32_t nonce_byte_count, uint32_t n_labels,
                                      const uint8_t *const *label_data, const size_t *label_sizes, uint8_t **out, size_t *out_size);
int apsu_he_wire_parse_result_package(const uint8_t *buf, size_t size, uint32_t *bundle_idx, uint32_t *cache_idx,
                                      const uint8_t **psu_result, size_t *psu_result_size, uint32_t *label_byte_count,
                                      uint32_t *nonce_byte_count, uint32_t *n_labels);
int apsu_he_wire_result_label(const uint8_t *buf, size_t size, uint32_t index, const uint8_t **data, size_t *data_size);
/* every label of a package with ONE parse (capacity 0: only *n_labels) */
int apsu_he_wire_result_labels(const uint8_t *buf, size_t size, uint32_t capacity, const uint8_t **data, size_t *sizes, uint32_t *n_labels);

/* ---- SEAL's own object serialisation: what is INSIDE Ciphertext.data / QueryRequest.relin_keys (host only, no GPU) ----
 * **UNPINNED**: restated from memory of upstream SEAL >= 3.6 / 4.x (apsu_amd/csrc/seal_codec.h lists every field); nothing in
 * the reference or this image can confirm it, the tests hold it against an independent Python model only (zlib and the BLAKE2b
 * core are pinned by python's zlib / hashlib).  Covers what the reference really puts on the wire
 * (sender/apsu/plaintext_powers.cpp:41-46, sender_osn.cpp:223-227,488, receiver/apsu/query.cpp:44-80, seal_object.h:161-219):
 *   - SEALHeader + body, compr_mode none, zlib or zstd (zstd through the system's libzstd.so.1, loaded at run time; refused with a
 *     clear message where that library is absent);
 *   - seeded ciphertexts (Serializable<Ciphertext> of encrypt_symmetric): c1 is expanded from the stored seed with SEAL's
 *     Blake2xb generator and util::sample_poly_uniform;
 *   - KSwitchKeys / RelinKeys (seeded or not) into the [decomp][2][K][n] array apsu_he_relin_upload takes;
 *   - parms_id = BLAKE2b-256 over {scheme, n, coeff moduli, plain modulus} for every level of the chain.
 * apsu_he_seal_ctx holds a parameter set's modulus chain; chain_idx as everywhere (K - 1 or -1 = the key level). */
typedef struct apsu_he_seal_ctx apsu_he_seal_ctx;
#define APSU_HE_SEAL_COMPR_NONE 0
#define APSU_HE_SEAL_COMPR_ZLIB 1
int apsu_he_seal_ctx_create(const char *psu_params_json, apsu_he_seal_ctx **out);
int apsu_he_seal_ctx_create_raw(uint64_t poly_modulus_degree, const uint64_t *coeff_modulus, int k, uint64_t plain_modulus,
                                apsu_he_seal_ctx **out);
int apsu_he_seal_ctx_free(apsu_he_seal_ctx *c);
int apsu_he_seal_parms_id(const apsu_he_seal_ctx *c, int chain_idx, uint64_t out[4]);
/* util::sample_poly_uniform under the Blake2xb generator seeded with seed[8]: out[L][n] at that level */
int apsu_he_seal_sample_poly_uniform(const apsu_he_seal_ctx *c, int chain_idx, const uint64_t seed[8], uint64_t *out);
/* Ciphertext::load: data receives size * L * n words ([poly][limb][coeff]), c1 expanded when the object was seeded.
 * c may be NULL for unseeded objects (chain_idx is then -1).  *consumed = bytes of the object. */
int apsu_he_seal_ct_load(const apsu_he_seal_ctx *c, const uint8_t *buf, size_t size, uint64_t parms_id[4], int *chain_idx, int *is_ntt_form,
                         uint64_t *ct_size, uint64_t *poly_modulus_degree, uint64_t *coeff_modulus_size, int *was_seeded, uint64_t *data,
                         size_t data_capacity_words, size_t *consumed);
/* The same WITHOUT expanding a seed: a seeded object yields c0 (L * n words), *was_seeded = 1 and seed[8] -- feed the seed to
 * apsu_he_seed_expand, which writes c1 into device memory; an unseeded object yields all its words. */
int apsu_he_seal_ct_load_unexpanded(const apsu_he_seal_ctx *c, const uint8_t *buf, size_t size, int *chain_idx, int *is_ntt_form,
                                    uint64_t *ct_size, uint64_t *coeff_modulus_size, int *was_seeded, uint64_t seed[8], uint64_t *data,
                                    size_t data_capacity_words, size_t *consumed);
/* Ciphertext::save at that level; seed != NULL writes the seeded form (c1 is NOT written: the caller guarantees it equals
 * apsu_he_seal_sample_poly_uniform(seed)) */
int apsu_he_seal_ct_save(const apsu_he_seal_ctx *c, int chain_idx, int is_ntt_form, uint64_t ct_size, const uint64_t *data, const uint64_t *seed,
                         int compr_mode, int version_major, int version_minor, uint8_t **out, size_t *out_size);
/* Plaintext::load / ::save (parms_id zero = coefficient form: *chain_idx = -1; else the level it was transformed at).  data NULL: only
 * the dimensions.  These are the objects a BinBundle's cache holds per coefficient (bin_bundle.cpp:421-428, compr none or zstd). */
int apsu_he_seal_pt_load(const apsu_he_seal_ctx *c, const uint8_t *buf, size_t size, int *chain_idx, uint64_t *coeff_count, uint64_t *data,
                         size_t data_capacity_words, size_t *consumed);
int apsu_he_seal_pt_save(const apsu_he_seal_ctx *c, int chain_idx, const uint64_t *data, uint64_t coeff_count, int compr_mode, int version_major,
                         int version_minor, uint8_t **out, size_t *out_size);
/* apsu_he_db_upload_bundle from the reference's own representation: blobs[d] / blob_sizes[d] = batched_coeffs[d].data() / .size() of
 * BatchedPlaintextPolyn (receiver/apsu/bin_bundle.h:52-134), i.e. SEAL-serialised Plaintexts; form and level are read from each
 * object's parms_id and checked against the rule of bin_bundle.cpp:385-389,418-420.  No SEAL on the host.  (ABI 4) */
int apsu_he_db_upload_bundle_serialized(apsu_he_ctx *ctx, const apsu_he_seal_ctx *seal_ctx, uint32_t bundle_idx, uint32_t cache_idx, uint32_t n_coeffs,
                                        const uint8_t *const *blobs, const size_t *blob_sizes, apsu_he_bundle **out);
/* The rest of the framing around the path (ABI 4).  apsu_he_wire_peek_type: the union tag of a ReceiverOperation (1 ParmsRequest,
 * 2 OPRFRequest, 3 QueryRequest, 4 plainResponse; rop.fbs) or of a ReceiverOperationResponse (1 ParmsResponse, 2 OPRFResponse,
 * 3 QueryResponse; rop_response.fbs).  The parameter exchange: ParmsRequest {} / ParmsResponse { data } with data = PSUParams::save's
 * bytes (psu_params.fbs + SEAL's EncryptionParameters object, psu_params.cpp:182-290): apsu_he_wire_psu_params_save turns the JSON
 * apsu_he_create takes into those bytes, _load turns them back into that JSON (explicit coefficient primes must be
 * CoeffModulus::Create's for their bit sizes, which is what every JSON-created parameter set has).  plainResponse { bundle_idx,
 * psu_result:[uint64], cache_idx }: the querier's decrypted results on their way back to the DB side (receiver_operation.cpp).
 * The EncryptionParameters object is restated from memory of upstream SEAL like the rest of apsu_he_seal_*: UNPINNED. */
int apsu_he_wire_peek_type(const uint8_t *buf, size_t size, int is_response, int *type);
int apsu_he_wire_psu_params_save(const char *psu_params_json, uint8_t **out, size_t *out_size);
int apsu_he_wire_psu_params_load(const uint8_t *buf, size_t size, uint8_t **psu_params_json, size_t *json_size);   /* not NUL-terminated */
int apsu_he_wire_build_parms_request(uint8_t **out, size_t *out_size);
int apsu_he_wire_build_parms_response(const uint8_t *psu_params, size_t psu_params_size, uint8_t **out, size_t *out_size);
int apsu_he_wire_parse_parms_response(const uint8_t *buf, size_t size, const uint8_t **psu_params, size_t *psu_params_size);   /* into buf */
int apsu_he_wire_build_plain_response(uint32_t bundle_idx, uint32_t cache_idx, const uint64_t *psu_result, size_t count, uint8_t **out, size_t *out_size);
int apsu_he_wire_parse_plain_response(const uint8_t *buf, size_t size, uint32_t *bundle_idx, uint32_t *cache_idx, uint64_t *psu_result, size_t capacity,
                                      size_t *count);
/* The header ReceiverDB::save writes in front of the BinBundles (receiver/apsu/receiver_db.fbs, receiver_db.cpp:1182-1232): the
 * parameters as JSON (release with apsu_he_wire_buffer_free; NULL: skip), item count, BinBundle count, flags; *consumed = offset of
 * the first BinBundle, to be walked with apsu_he_wire_bin_bundle_info / apsu_he_db_upload_saved_bundle below. */
int apsu_he_wire_receiver_db_header(const uint8_t *buf, size_t size, uint8_t **psu_params_json, size_t *json_size, uint64_t *item_count,
                                    uint32_t *bin_bundle_count, int *compressed, int *stripped, uint32_t *label_byte_count, size_t *consumed);
/* One BinBundle as the reference persists it: the size-prefixed FlatBuffers buffer BinBundle::save appends to a saved ReceiverDB
 * (receiver/apsu/bin_bundle.fbs; bin_bundle.cpp:1085-1168; ReceiverDB::save receiver_db.cpp:1182-1260 writes them one after the
 * other behind its own header).  apsu_he_wire_bin_bundle_info: its dimensions and *consumed = where the next one starts.
 * apsu_he_db_upload_saved_bundle: onto the device -- from the saved cache when there is one (SEAL Plaintext objects, read by the
 * codec above: seal_ctx required), else rebuilt from the item bins on the GPU (apsu_he_db_build_bundle); the checks of
 * BinBundle::load (bin_bundle.cpp:1170-1230: field modulus, number of bins, bin sizes) -> APSU_HE_RUNTIME_ERROR "failed to load
 * BinBundle".  Labels and interpolation polynomials are not read (the unlabeled protocol).  (ABI 4) */
int apsu_he_wire_bin_bundle_info(const uint8_t *buf, size_t size, uint32_t *bundle_idx, uint64_t *mod, int *stripped, uint32_t *n_bins,
                                 uint32_t *largest_bin, uint32_t *cache_coeffs, size_t *consumed);
int apsu_he_db_upload_saved_bundle(apsu_he_ctx *ctx, const apsu_he_seal_ctx *seal_ctx, const uint8_t *buf, size_t size, uint32_t cache_idx,
                                   apsu_he_bundle **out, size_t *consumed);
/* RelinKeys::load -> ksk[K-1][2][K][n] (ksk NULL: only *words); ::save (seeds[K-1][8] or NULL) */
int apsu_he_seal_relin_keys_load(const apsu_he_seal_ctx *c, const uint8_t *buf, size_t size, uint64_t *ksk, size_t capacity_words, size_t *words,
                                 size_t *consumed);
int apsu_he_seal_relin_keys_save(const apsu_he_seal_ctx *c, const uint64_t *ksk, const uint64_t *seeds, int compr_mode, int version_major,
                                 int version_minor, uint8_t **out, size_t *out_size);
/* The seed expansion ON THE DEVICE (needs a GPU context): c1 of `count` seeded objects at chain_idx (-1 / K - 1 = key level)
 * = util::sample_poly_uniform under SEAL's Blake2xb generator seeded with seeds[i*8 .. i*8+7], written to the device buffers
 * dst_device[i] ([L][n] words).  Same values as apsu_he_seal_sample_poly_uniform (the host form costs ~0.2 ms per 384 KiB
 * ciphertext and core; 24 query ciphertexts expand in one launch pair here).  Synchronous. */
int apsu_he_seed_expand(apsu_he_ctx *ctx, int chain_idx, int count, const uint64_t *seeds, uint64_t *const *dst_device);
/* Receiver::RunQuery from the wire, without SEAL on the host (receiver_osn.cpp:160-364, query.cpp:44-80, result_package.cpp:29-76):
 * `request` = the size-prefixed ReceiverOperation{QueryRequest} buffer as ZMQChannel::receive_operation hands it over.  Its
 * RelinKeys are decoded and uploaded, every query ciphertext is decoded (c0 through page-locked memory, a seeded c1 expanded on
 * the device), ComputePowers runs for the bundle indices of the given BinBundles, every BinBundle is evaluated with its mask,
 * and packages[i] / package_sizes[i] receive one size-prefixed ResultPackage per BinBundle (result ciphertext as a SEAL object at
 * the last level under result_compr_mode; release each with apsu_he_wire_buffer_free).  Errors as the reference raises them:
 * exponents that do not match the parameters' query_powers, a wrong ciphertext count per exponent, missing RelinKeys -> 
 * APSU_HE_INVALID_ARGUMENT.  UNPINNED as far as SEAL's object format goes (apsu_he_seal_*). */
int apsu_he_run_query_request(apsu_he_ctx *ctx, const apsu_he_seal_ctx *seal_ctx, const uint8_t *request, size_t request_size,
                              const apsu_he_bundle *const *bundles, int count, const uint64_t *const *masks, int masks_on_device,
                              int result_compr_mode, uint8_t **packages, size_t *package_sizes);
/* The same for the multi-device handle: the request is decoded once onto the handle's first device (seeded c1
 * expanded there), its RelinKeys go to every device, the other devices fetch the ciphertexts of their bundle indices over xGMI, and
 * EVERY BinBundle registered in the handle gets its ResultPackage: packages[id] / package_sizes[id], capacity >= the handle's
 * BinBundle count.  masks[id]: host memory.  (ABI 4) */
int apsu_he_multi_run_query_request(apsu_he_multi *m, const apsu_he_seal_ctx *seal_ctx, const uint8_t *request, size_t request_size,
                                    const uint64_t *const *masks, int result_compr_mode, uint8_t **packages, size_t *package_sizes, int capacity);
/* round-2 entry points, kept: one unseeded ciphertext without a context (zlib bodies are inflated on load) */
int apsu_he_wire_seal_ct_save(const uint64_t parms_id[4], int is_ntt_form, uint64_t ct_size, uint64_t poly_modulus_degree,
                              uint64_t coeff_modulus_size, uint64_t correction_factor, double scale, const uint64_t *data,
                              int version_major, int version_minor, uint8_t **out, size_t *out_size);
int apsu_he_wire_seal_ct_load(const uint8_t *buf, size_t size, uint64_t parms_id[4], int *is_ntt_form, uint64_t *ct_size,
                              uint64_t *poly_modulus_degree, uint64_t *coeff_modulus_size, uint64_t *correction_factor, double *scale,
                              uint64_t *data, size_t data_capacity_words, int *version_major, int *version_minor);

/* ---- measurement hooks (replace the reference's STOPWATCH timers, receiver_osn.cpp:167,403,504) ----
 * Per-kernel-class device time from HIP events recorded on the engine's stream around each launch.
 * Classes (index): 0 ntt_fwd, 1 ntt_inv, 2 dyadic_mac, 3 behz_ext, 4 behz_tensor, 5 behz_finish,
 * 6 keyswitch, 7 modswitch, 8 other, 9 ntt_fused (inverse transforms whose load forms the BEHZ tensor product or the key
 * switch's inner product: k_intt_tensor, k_intt_ks).  units: limb polynomials for 0/1/9, plaintext limb-terms for 2.
 * mode 2 times classes 0, 1 and 9 only. */
#define APSU_HE_PROFILE_CLASSES 10
int apsu_he_profile_enable(apsu_he_ctx *ctx, int mode);   /* 0 off, 1 every class, 2 NTT launches only */
int apsu_he_profile_read(apsu_he_ctx *ctx, double *ms, uint64_t *launches, uint64_t *units, int capacity, int reset);

/* Phase timers under the reference's own STOPWATCH names (receiver_osn.cpp:167,403,504; report format cli/common_utils.cpp:54-76:
 * instances, average, minimum, maximum): 0 "Receiver::RunQuery" (start of apsu_he_compute_powers .. end of the last
 * apsu_he_eval_bundles before the next apsu_he_compute_powers), 1 "Receiver::ComputePowers", 2 "Receiver::ProcessBinBundleCache"
 * (one span per apsu_he_eval_bundles call, i.e. over ALL its BinBundles: they are evaluated as one batch).  Device time from
 * HIP events on the context's streams; enable costs two to five event records per call.  reset != 0 clears all three. */
#define APSU_HE_PHASES 3
const char *apsu_he_phase_name(int phase);
int apsu_he_phase_enable(apsu_he_ctx *ctx, int on);
int apsu_he_phase_read(apsu_he_ctx *ctx, int phase, uint64_t *count, double *avg_ms, double *min_ms, double *max_ms, int reset);

/* Host-side events inside the engine that cost a query time without being a kernel (diagnosis of launch-bound shards):
 * 0 host waits, 1 job-table uploads (cache misses), 2 job-table hits, 3 workspace-arena growths, 4 powers-buffer
 * allocations, 5 wraps of the pinned staging area, 6 job-table re-allocations, 7 (ABI 6) ComputePowers calls that were pipelined
 * with the evaluation in front (apsu_he_set_query_overlap).  Monotonic since apsu_he_create. */
#define APSU_HE_DEBUG_COUNTERS 8
int apsu_he_debug_counters(apsu_he_ctx *ctx, uint64_t *out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* APSU_HE_H */
