// The hot section of Receiver::RunQuery in its BATCHED form -- the call pattern bench.py measures (`adapter_calls.batched_ms`):
//
//   receiver/apsu/receiver_osn.cpp:320-328   for every bundle index: ComputePowers(...)            -> ONE apsu_he_compute_powers
//   receiver/apsu/receiver_osn.cpp:334-359   one thread-pool task per BinBundle: evaluate + send   -> ONE apsu_he_eval_bundles,
//                                                                                                      then one task per BinBundle
//                                                                                                      that only packages + sends
//   receiver/apsu/receiver_osn.cpp:361-364   wait for the tasks                                    -> unchanged
//
// integration/receiver_hot_path.cpp keeps the reference's call structure (ComputePowers per bundle index, one evaluation per
// ProcessBinBundleCache task): the smallest patch, but every one of its 4 + 28 calls (16M-4096) crosses PCIe on its own, waits for
// the device on its own and fills the chip with one BinBundle's worth of work.  This file is the other end: the sources of ALL
// bundle indices go to the device in one call, ALL BinBundle caches are evaluated by one call into one host buffer (57 kernel
// launches for the whole query instead of ~30 per BinBundle), and the reference's per-BinBundle fan-out keeps only what is
// per-BinBundle by nature -- wrapping a result into its ResultPackage and sending it (receiver_osn.cpp:507-539; arrival order is
// irrelevant, the querier places each package by its (bundle_idx, cache_idx), sender_osn.cpp:698-705).
// bench.py times both patterns on the same inputs from host memory (`adapter_calls`); INTEGRATION.md section 4 has the numbers.
//
// How it goes into a reference checkout: everything of integration/receiver_hot_path.cpp's instructions (he_gpu.h, the HeGpu member
// of ReceiverDB, generate_caches), then declare `EvaluateQueryOnDevice` next to ComputePowers in receiver_osn.h and replace
// receiver_osn.cpp:320-364 -- from "Compute query powers for the bundle indexes" down to and including the loop that waits for
// the futures -- by the single call
//
//     EvaluateQueryOnDevice(receiver_db, crypto_context, all_powers, pd, chl, send_rp_fun, query.compr_mode(), tpm);
//
// ComputePowers / ProcessBinBundleCache / eval / eval_patstock are then no longer called from RunQuery (their replacement bodies
// in receiver_hot_path.cpp stay valid for other callers).
//
// Like the other adapter files this one cannot be compiled here (no SEAL, no APSU headers): tests/test_integration_syntax.py runs
// `g++ -fsyntax-only` on it against forward declarations written from the cited signatures -- well-formedness and the match with
// include/apsu_he.h, nothing about SEAL.
#include "apsu/receiver_osn.h"
#include "apsu/bin_bundle.h"
#include "apsu/receiver_db.h"
#include "apsu/he_gpu.h"

#include <future>

using namespace std;
using namespace seal;

namespace apsu {
namespace receiver {

using gpu::he_check;

void Receiver::EvaluateQueryOnDevice(
    const shared_ptr<ReceiverDB> &receiver_db,
    const CryptoContext &crypto_context,
    vector<CiphertextPowers> &all_powers,
    const PowersDag &pd,
    network::Channel &chl,
    function<void(network::Channel &, ResultPart)> send_rp_fun,
    compr_mode_type compr_mode,
    ThreadPoolMgr &tpm)
{
    gpu::HeGpu &he = *receiver_db->he_gpu();
    const SEALContext &context = *crypto_context.seal_context();
    const uint32_t bundle_idx_count = safe_cast<uint32_t>(all_powers.size());
    const size_t n = he.info().poly_modulus_degree;
    const size_t row_words = static_cast<size_t>(he.info().result_polys) * n;

    // ---- receiver_osn.cpp:320-328: the source powers of every bundle index that has BinBundles, in ONE call.
    // Source order per bundle index = ascending exponent, the order of the PowersDag's source nodes (as in ComputePowers).
    vector<uint32_t> indices;
    vector<const uint64_t *> src;
    const auto first_level = context.first_parms_id();
    for (uint32_t bundle_idx = 0; bundle_idx < bundle_idx_count; bundle_idx++) {
        if (!receiver_db->get_cache_at(bundle_idx).size()) {
            continue;                                            // ComputePowers returns early for it (receiver_osn.cpp:406-409)
        }
        indices.push_back(bundle_idx);
        const CiphertextPowers &powers = all_powers[bundle_idx];
        size_t found = 0;
        pd.apply([&](const PowersDag::PowersNode &node) {
            if (node.is_source()) {
                const Ciphertext &ct = powers[node.power];
                if (ct.size() != 2 || ct.is_ntt_form() || ct.parms_id() != first_level) {
                    throw invalid_argument("query ciphertext is not a fresh ciphertext at the first data level");
                }
                src.push_back(ct.data());
                found++;
            }
        });
        if (found != he.info().source_power_count) {
            throw invalid_argument("query powers do not match the parameters");
        }
    }
    if (indices.empty()) {
        return;
    }
    const apsu_he_relin *rk = nullptr;
    if (context.using_keyswitching()) {
        rk = he.relin_keys(crypto_context.relin_keys().get());
    }
    struct PowersGuard {                                         // the device powers live until the evaluation has returned
        apsu_he_powers *p = nullptr;
        ~PowersGuard() { if (p) apsu_he_powers_free(p); }
    } device_powers;
    {
        STOPWATCH(recv_stopwatch, "Receiver::ComputePowers");
        he_check(apsu_he_compute_powers(
            he.ctx(), indices.data(), static_cast<int>(indices.size()), src.data(), /*src_on_device=*/0, rk, &device_powers.p));
    }

    // ---- receiver_osn.cpp:334-359, the evaluations: every BinBundle cache of every bundle index in ONE call.
    // pack_idx as the reference computes it (it indexes random_plain_list, the masks drawn at receiver_osn.cpp:217-284).
    struct Unit { uint32_t bundle_idx, cache_idx; };
    vector<Unit> units;
    vector<const apsu_he_bundle *> bundles;
    vector<const uint64_t *> masks;
    for (uint32_t bundle_idx = 0; bundle_idx < bundle_idx_count; bundle_idx++) {
        auto bundle_caches = receiver_db->get_cache_at(bundle_idx);
        uint32_t cache_idx = 0;
        for (auto &cache : bundle_caches) {
            pack_cnt++;
            const size_t pack_idx = bundle_idx + static_cast<size_t>(cache_idx) * bundle_idx_count;
            units.push_back(Unit{ bundle_idx, cache_idx });
            bundles.push_back(he.bundle_of(&cache.get().batched_matching_polyn));
            masks.push_back(random_plain_list[pack_idx].data());           // n coefficients mod t
            cache_idx++;
        }
    }
    vector<uint64_t> rows(units.size() * row_words);
    {
        STOPWATCH(recv_stopwatch, "Receiver::ProcessBinBundleCache");
        he_check(apsu_he_eval_bundles(
            he.ctx(), bundles.data(), static_cast<int>(bundles.size()), device_powers.p, rk, masks.data(), /*masks_on_device=*/0,
            rows.data(), /*out_on_device=*/0));
    }

    // ---- the per-BinBundle fan-out, unchanged in what it sends (receiver_osn.cpp:507-539): one ResultPackage per BinBundle,
    // serialised and sent from a pool thread (the channel serialises the sends, zmq_channel.cpp:560)
    const uint32_t nonce_byte_count = safe_cast<uint32_t>(receiver_db->get_nonce_byte_count());
    const uint32_t label_byte_count = safe_cast<uint32_t>(receiver_db->get_label_byte_count());
    vector<future<void>> futures;
    for (size_t i = 0; i < units.size(); i++) {
        futures.push_back(tpm.thread_pool().enqueue([&, i]() {
            auto rp = make_unique<ResultPackage>();
            rp->compr_mode = compr_mode;
            rp->cache_idx = units[i].cache_idx;
            rp->bundle_idx = units[i].bundle_idx;
            rp->nonce_byte_count = nonce_byte_count;
            rp->label_byte_count = label_byte_count;
            // the result: last level, coefficient form, irrelevant bits cleared (bin_bundle.cpp:159-171,340-357)
            uint32_t polys = 2;
            he_check(apsu_he_bundle_result_size(he.ctx(), bundles[i], &polys));
            rp->psu_result.resize(context, context.last_parms_id(), polys);
            copy_n(rows.data() + i * row_words, static_cast<size_t>(polys) * n, rp->psu_result.data());
            try {
                send_rp_fun(chl, move(rp));
            } catch (const exception &ex) {
                APSU_LOG_ERROR("Failed to send result part; function threw an exception: " << ex.what());
                throw;
            }
        }));
    }
    for (auto &f : futures) {
        f.get();
    }
    he.end_query();                                              // the query's relinearisation keys leave HBM
}

} // namespace receiver
} // namespace apsu
