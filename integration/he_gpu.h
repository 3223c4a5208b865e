// Adapter between APSU's DB-side types and the C ABI of the MI355X engine (include/apsu_he.h).
//
// This header is what a maintainer of the reference adds as receiver/apsu/he_gpu.h.  It is written against the reference's own
// signatures (CryptoContext common/apsu/crypto_context.h:28-125, BinBundleCache receiver/apsu/bin_bundle.h:137-171,
// CiphertextPowers receiver/apsu/receiver_osn.h:41) and against Microsoft SEAL's public API -- NEITHER of which exists in the
// engine's repository or image.  The repository only checks that the file is well-formed C++ against forward declarations
// that tests/test_integration_syntax.py generates (g++ -fsyntax-only); that check pins nothing about SEAL.
//
// Ownership: one HeGpu per ReceiverDB (the lifetime of its CryptoContext).  BinBundle images live in HBM from
// ReceiverDB::generate_caches on; per query the relinearisation keys and one apsu_he_powers handle per bundle index live from
// Receiver::ComputePowers until HeGpu::end_query.  Thread safety: the C ABI serialises calls per context; the maps below are
// guarded for the reference's task fan-out (receiver_osn.cpp:334-359 calls ProcessBinBundleCache from pool threads).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "apsu_he.h"
#include "seal/seal.h"

namespace apsu {
namespace receiver {
namespace gpu {

// SEAL-style exceptions for the dispatcher (receiver_dispatcher_osn.cpp:193-195 logs what() and carries on)
inline void he_check(int rc)
{
    if (rc == APSU_HE_OK) return;
    if (rc == APSU_HE_INVALID_ARGUMENT) throw std::invalid_argument(apsu_he_last_error());
    if (rc == APSU_HE_LOGIC_ERROR) throw std::logic_error(apsu_he_last_error());
    throw std::runtime_error(apsu_he_last_error());
}

// the engine names levels by SEAL's chain_index (util/utils.cpp:179-189 gives parms_id -> level in the reference)
inline int chain_idx_of(const seal::SEALContext &context, const seal::parms_id_type &id)
{
    auto data = context.get_context_data(id);
    if (!data) throw std::invalid_argument("parms_id is not in the modulus chain");
    return static_cast<int>(data->chain_index());
}

class HeGpu {
public:
    // psu_params_json: PSUParams::to_string() (common/apsu/psu_params.h:172), the JSON the CLI loads
    explicit HeGpu(const std::string &psu_params_json, int device = 0)
    {
        if (apsu_he_abi_version() != APSU_HE_ABI_VERSION) throw std::runtime_error("libapsu_he_gpu.so: ABI version mismatch");
        he_check(apsu_he_create(psu_params_json.c_str(), device, &ctx_));
        he_check(apsu_he_get_info(ctx_, &info_));
    }
    HeGpu(const HeGpu &) = delete;
    HeGpu &operator=(const HeGpu &) = delete;
    ~HeGpu()
    {
        end_query();
        for (auto &kv : bundles_) apsu_he_bundle_free(kv.second);
        apsu_he_destroy(ctx_);
    }

    apsu_he_ctx *ctx() const { return ctx_; }
    const apsu_he_info &info() const { return info_; }

    // ---- DB side: one image per BinBundleCache, keyed by the cache object (stable: caches live inside their BinBundle)
    // polyn.batched_coeffs[d] is a serialised seal::Plaintext (bin_bundle.cpp:421-428); form and level as the ctor chose them
    template <class BatchedPlaintextPolynT>
    apsu_he_bundle *upload_cache(const void *cache_key, std::uint32_t bundle_idx, std::uint32_t cache_idx, const BatchedPlaintextPolynT &polyn,
                                 const seal::SEALContext &context)
    {
        const auto &coeffs = polyn.batched_coeffs;
        std::vector<seal::Plaintext> pts(coeffs.size());
        std::vector<const std::uint64_t *> ptrs(coeffs.size());
        std::vector<std::uint8_t> is_ntt(coeffs.size());
        for (std::size_t d = 0; d < coeffs.size(); d++) {
            pts[d].unsafe_load(context, reinterpret_cast<const seal::seal_byte *>(coeffs[d].data()), coeffs[d].size());
            ptrs[d] = pts[d].data();
            is_ntt[d] = pts[d].is_ntt_form() ? 1 : 0;
        }
        apsu_he_bundle *b = nullptr;
        he_check(apsu_he_db_upload_bundle(ctx_, bundle_idx, cache_idx, static_cast<std::uint32_t>(ptrs.size()), ptrs.data(), is_ntt.data(), &b));
        std::lock_guard<std::mutex> g(mu_);
        auto it = bundles_.find(cache_key);
        if (it != bundles_.end()) { apsu_he_bundle_free(it->second); it->second = b; }
        else bundles_.emplace(cache_key, b);
        return b;
    }
    apsu_he_bundle *bundle_of(const void *cache_key) const
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = bundles_.find(cache_key);
        if (it == bundles_.end()) throw std::logic_error("BinBundleCache has no device image: generate_caches has not run");
        return it->second;
    }
    void drop_bundles()
    {
        std::lock_guard<std::mutex> g(mu_);
        for (auto &kv : bundles_) apsu_he_bundle_free(kv.second);
        bundles_.clear();
    }

    // ---- per query
    // RelinKeys of the query (crypto_context.h:45-49): KSwitchKeys::data()[0] = the keys of c2, one PublicKey per decomposition
    // prime, each a size-2 NTT-form ciphertext over all key-level primes = exactly the layout apsu_he_relin_upload takes
    const apsu_he_relin *relin_keys(const seal::RelinKeys *keys)
    {
        std::lock_guard<std::mutex> g(mu_);
        if (!keys) return nullptr;
        if (rk_ && rk_for_ == keys) return rk_;
        if (rk_) { apsu_he_relin_free(rk_); rk_ = nullptr; }
        const auto &row = keys->data()[0];
        std::vector<std::uint64_t> ksk;
        for (const auto &pk : row) {
            const seal::Ciphertext &ct = pk.data();
            const std::size_t words = ct.size() * ct.coeff_modulus_size() * ct.poly_modulus_degree();
            ksk.insert(ksk.end(), ct.data(), ct.data() + words);
        }
        he_check(apsu_he_relin_upload(ctx_, ksk.data(), &rk_));
        rk_for_ = keys;
        return rk_;
    }
    const apsu_he_relin *current_relin() const { std::lock_guard<std::mutex> g(mu_); return rk_; }
    // device powers of one bundle index, keyed by the caller's CiphertextPowers object (all_powers[bundle_idx])
    void set_powers(const void *powers_key, apsu_he_powers *p)
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = powers_.find(powers_key);
        if (it != powers_.end()) { apsu_he_powers_free(it->second); it->second = p; }
        else powers_.emplace(powers_key, p);
    }
    const apsu_he_powers *powers_of(const void *powers_key) const
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = powers_.find(powers_key);
        if (it == powers_.end()) throw std::invalid_argument("not enough ciphertext powers available");     // bin_bundle.cpp:116-118
        return it->second;
    }
    // Receiver::RunQuery calls this when the last ProcessBinBundleCache task has finished (receiver_osn.cpp:361-364)
    void end_query()
    {
        std::lock_guard<std::mutex> g(mu_);
        for (auto &kv : powers_) apsu_he_powers_free(kv.second);
        powers_.clear();
        if (rk_) { apsu_he_relin_free(rk_); rk_ = nullptr; rk_for_ = nullptr; }
    }

private:
    apsu_he_ctx *ctx_ = nullptr;
    apsu_he_info info_{};
    mutable std::mutex mu_;
    std::map<const void *, apsu_he_bundle *> bundles_;
    std::map<const void *, apsu_he_powers *> powers_;
    apsu_he_relin *rk_ = nullptr;
    const seal::RelinKeys *rk_for_ = nullptr;
};

} // namespace gpu
} // namespace receiver
} // namespace apsu
