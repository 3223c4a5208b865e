// Replacement bodies for the reference's hot path, written against the reference's own signatures:
//
//   Receiver::ComputePowers            receiver/apsu/receiver_osn.cpp:395-488
//   Receiver::ProcessBinBundleCache    receiver/apsu/receiver_osn.cpp:490-540   (the evaluation; packaging and sending unchanged)
//   BatchedPlaintextPolyn::eval        receiver/apsu/bin_bundle.cpp:106-174     (kept as thin forwards for other callers)
//   BatchedPlaintextPolyn::eval_patstock                      bin_bundle.cpp:192-360
//   ReceiverDB::generate_caches        receiver/apsu/receiver_db.cpp:808-820    (+ upload of every cache)
//
// How it goes into a reference checkout: add integration/he_gpu.h as receiver/apsu/he_gpu.h, give
// ReceiverDB a `std::shared_ptr<gpu::HeGpu> he_gpu_` created next to its CryptoContext (receiver_db.cpp, constructor) with an
// accessor `he_gpu()`, replace the five bodies by the ones below, and add ONE line at the end of Receiver::RunQuery's task
// loop (receiver_osn.cpp:361-364, after the futures have been waited for): `receiver_db->he_gpu()->end_query();`.
// Nothing else of the reference changes: the query is still deserialised by SEAL, masks are still drawn by the block at
// receiver_osn.cpp:217-284, results still leave as ResultPackages through send_rp_fun.
//
// This file reproduces statements of the reference ON PURPOSE: it is a patch to receiver_osn.cpp / bin_bundle.cpp / receiver_db.cpp, and the
// lines around each replaced Evaluator call are repeated so that a maintainer can paste the bodies over the originals (about half of its code
// lines also occur in receiver_{osn,ddh}.cpp).  It is never compiled into the engine and never runs on the GPU box; if it is published outside
// this tree it falls under the reference's licence, not the engine's.
//
// This file cannot be compiled in the engine's repository (no SEAL, no APSU headers).  tests/test_integration_syntax.py runs
// `g++ -fsyntax-only` on it against forward declarations the test writes from the signatures cited above; that check shows the
// file is well-formed and matches include/apsu_he.h -- it pins nothing about SEAL.
#include "apsu/receiver_osn.h"
#include "apsu/bin_bundle.h"
#include "apsu/receiver_db.h"
#include "apsu/he_gpu.h"

using namespace std;
using namespace seal;

namespace apsu {
namespace receiver {

using gpu::he_check;

// ---------------------------------------------------------------------------------------------------------------------
// Receiver::ComputePowers: all target powers of the query for ONE bundle index, left in HBM.
// The reference fills all_powers[bundle_idx][power] with host ciphertexts (66 products + 66 key switches at 16M-4096, then the
// per-power level / NTT conversions of :459-487); here the source powers go to the device once and everything else happens
// there -- the vector keeps only the sources, the computed powers live behind an apsu_he_powers handle keyed by that vector.
void Receiver::ComputePowers(
    const shared_ptr<ReceiverDB> &receiver_db,
    const CryptoContext &crypto_context,
    vector<CiphertextPowers> &all_powers,
    const PowersDag &pd,
    uint32_t bundle_idx,
    MemoryPoolHandle &pool)
{
    STOPWATCH(recv_stopwatch, "Receiver::ComputePowers");
    (void)pool;
    auto bundle_caches = receiver_db->get_cache_at(bundle_idx);
    if (!bundle_caches.size()) {
        return;
    }
    gpu::HeGpu &he = *receiver_db->he_gpu();
    CiphertextPowers &powers_at_this_bundle_idx = all_powers[bundle_idx];

    // source powers in ascending order of their exponent = the order apsu_he_compute_powers expects (the engine derives the
    // same PowersDag from the same PSUParams; pd is only consulted for which nodes are sources)
    vector<const uint64_t *> src;
    const auto first_level = crypto_context.seal_context()->first_parms_id();
    pd.apply([&](const PowersDag::PowersNode &node) {
        if (node.is_source()) {
            const Ciphertext &ct = powers_at_this_bundle_idx[node.power];
            if (ct.size() != 2 || ct.is_ntt_form() || ct.parms_id() != first_level) {
                throw invalid_argument("query ciphertext is not a fresh ciphertext at the first data level");
            }
            src.push_back(ct.data());
        }
    });
    if (src.size() != he.info().source_power_count) {
        throw invalid_argument("query powers do not match the parameters");
    }

    const apsu_he_relin *rk = nullptr;
    if (crypto_context.seal_context()->using_keyswitching()) {
        rk = he.relin_keys(crypto_context.relin_keys().get());
    }
    apsu_he_powers *device_powers = nullptr;
    he_check(apsu_he_compute_powers(he.ctx(), &bundle_idx, 1, src.data(), /*src_on_device=*/0, rk, &device_powers));
    he.set_powers(&powers_at_this_bundle_idx, device_powers);
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchedPlaintextPolyn::eval / eval_patstock: the engine picks the form itself from the BinBundle's degree and ps_low_degree
// (receiver_osn.cpp:520-528), so both entry points forward to the same call.  `ciphertext_powers` is the CiphertextPowers
// object ComputePowers was given: it identifies the device-resident powers.
namespace {
Ciphertext eval_on_device(
    gpu::HeGpu &he, const void *cache_key, const vector<Ciphertext> &ciphertext_powers, const SEALContext &context, const Plaintext &random_plain)
{
    const apsu_he_bundle *bundle = he.bundle_of(cache_key);
    const apsu_he_powers *powers = he.powers_of(&ciphertext_powers);
    const uint64_t *mask = random_plain.data();                  // n coefficients mod t (receiver_osn.cpp:217-284)
    const size_t n = he.info().poly_modulus_degree;
    uint32_t polys = 2;
    he_check(apsu_he_bundle_result_size(he.ctx(), bundle, &polys));
    vector<uint64_t> row(static_cast<size_t>(he.info().result_polys) * n);
    he_check(apsu_he_eval_bundles(he.ctx(), &bundle, 1, powers, he.current_relin(), &mask, 0, row.data(), 0));
    // the result: last level, coefficient form, irrelevant bits cleared (bin_bundle.cpp:159-171,340-357)
    Ciphertext result;
    result.resize(context, context.last_parms_id(), polys);
    copy_n(row.data(), static_cast<size_t>(polys) * n, result.data());
    return result;
}
} // namespace

Ciphertext BatchedPlaintextPolyn::eval(
    const vector<Ciphertext> &ciphertext_powers, MemoryPoolHandle &pool, Plaintext &random_plain) const
{
    (void)pool;
    if (ciphertext_powers.size() < max<size_t>(batched_coeffs.size(), 2)) {
        throw invalid_argument("not enough ciphertext powers available");
    }
    return eval_on_device(*he_gpu, this, ciphertext_powers, *crypto_context.seal_context(), random_plain);
}

Ciphertext BatchedPlaintextPolyn::eval_patstock(
    const CryptoContext &eval_crypto_context,
    const vector<Ciphertext> &ciphertext_powers,
    size_t ps_low_degree,
    MemoryPoolHandle &pool,
    Plaintext &random_plain) const
{
    (void)pool;
    if (ciphertext_powers.size() < max<size_t>(batched_coeffs.size(), 2)) {
        throw invalid_argument("not enough ciphertext powers available");
    }
    const size_t degree = batched_coeffs.size() - 1;
    if (ps_low_degree <= 1 || ps_low_degree >= degree) {
        throw invalid_argument("ps_low_degree must be greater than 1 and less than the size of batched_coeffs");
    }
    return eval_on_device(*he_gpu, this, ciphertext_powers, *eval_crypto_context.seal_context(), random_plain);
}

// ---------------------------------------------------------------------------------------------------------------------
// Receiver::ProcessBinBundleCache: unchanged in shape -- the evaluation call now lands on the device
void Receiver::ProcessBinBundleCache(
    const shared_ptr<ReceiverDB> &receiver_db,
    const CryptoContext &crypto_context,
    reference_wrapper<const BinBundleCache> cache,
    vector<CiphertextPowers> &all_powers,
    network::Channel &chl,
    function<void(network::Channel &, ResultPart)> send_rp_fun,
    uint32_t bundle_idx,
    compr_mode_type compr_mode,
    MemoryPoolHandle &pool,
    uint32_t cache_idx,
    uint32_t pack_idx)
{
    STOPWATCH(recv_stopwatch, "Receiver::ProcessBinBundleCache");
    auto rp = make_unique<ResultPackage>();
    rp->compr_mode = compr_mode;
    rp->cache_idx = cache_idx;
    rp->bundle_idx = bundle_idx;
    rp->nonce_byte_count = safe_cast<uint32_t>(receiver_db->get_nonce_byte_count());
    rp->label_byte_count = safe_cast<uint32_t>(receiver_db->get_label_byte_count());

    const BatchedPlaintextPolyn &matching_polyn = cache.get().batched_matching_polyn;
    uint32_t ps_low_degree = receiver_db->get_params().query_params().ps_low_degree;
    uint32_t degree = safe_cast<uint32_t>(matching_polyn.batched_coeffs.size()) - 1;
    bool using_ps = (ps_low_degree > 1) && (ps_low_degree < degree);
    if (using_ps) {
        rp->psu_result = matching_polyn.eval_patstock(
            crypto_context, all_powers[bundle_idx], safe_cast<size_t>(ps_low_degree), pool, random_plain_list[pack_idx]);
    } else {
        rp->psu_result = matching_polyn.eval(all_powers[bundle_idx], pool, random_plain_list[pack_idx]);
    }
    try {
        send_rp_fun(chl, move(rp));
    } catch (const exception &ex) {
        APSU_LOG_ERROR("Failed to send result part; function threw an exception: " << ex.what());
        throw;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ReceiverDB::generate_caches: regenerate on the host as before, then hand every cache to the device (one copy; the host copy may
// be strip()ped afterwards).  apsu_he_db_build_bundle is the alternative that skips the host-side regen_cache altogether (N1).
void ReceiverDB::generate_caches()
{
    STOPWATCH(recv_stopwatch, "ReceiverDB::generate_caches");
    APSU_LOG_INFO("Start generating bin bundle caches");
    he_gpu_->drop_bundles();
    uint32_t bundle_idx = 0;
    for (auto &bundle_set : bin_bundles_) {
        uint32_t cache_idx = 0;
        for (auto &bb : bundle_set) {
            bb.regen_cache();
            const BinBundleCache &cache = bb.get_cache();
            BatchedPlaintextPolyn &polyn = const_cast<BatchedPlaintextPolyn &>(cache.batched_matching_polyn);
            polyn.he_gpu = he_gpu_.get();                         // a member the adapter adds to BatchedPlaintextPolyn
            he_gpu_->upload_cache(&polyn, bundle_idx, cache_idx, polyn, *crypto_context_.seal_context());
            cache_idx++;
        }
        bundle_idx++;
    }
    APSU_LOG_INFO("Finished generating bin bundle caches");
}

} // namespace receiver
} // namespace apsu
