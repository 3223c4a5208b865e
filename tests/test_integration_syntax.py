"""CPU tier: the adapter a maintainer of the reference would add (integration/he_gpu.h, integration/receiver_hot_path.cpp) is
well-formed C++ and matches include/apsu_he.h.

The adapter is written against Microsoft SEAL's public API and against APSU's own headers; neither exists in this repository or
image.  This test writes minimal FORWARD DECLARATIONS of exactly the members the adapter touches -- restated from the signatures
the adapter's comments cite (crypto_context.h:28-125, bin_bundle.h:52-171, receiver_osn.h:41,220-250, receiver_db.h, powers.h:42-160,
result_package.h) and from SEAL's public class interfaces as recalled -- and runs `g++ -fsyntax-only`.  It checks syntax, overload
resolution against include/apsu_he.h and nothing else: IT PINS NOTHING ABOUT SEAL OR ABOUT THE REFERENCE (a wrong recollection in a
stub would go unnoticed here and fail at the maintainer's first real build)."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SEAL_H = r'''
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>
namespace seal {
using seal_byte = std::byte;
using parms_id_type = std::array<std::uint64_t, 4>;
enum class compr_mode_type : std::uint8_t { none = 0, zlib = 1, zstd = 2 };
class MemoryPoolHandle {};
class SEALContext {
public:
    class ContextData { public: std::size_t chain_index() const; };
    std::shared_ptr<const ContextData> get_context_data(parms_id_type id) const;
    const parms_id_type &first_parms_id() const;
    const parms_id_type &last_parms_id() const;
    bool using_keyswitching() const;
};
class Plaintext {
public:
    std::uint64_t *data();
    const std::uint64_t *data() const;
    bool is_ntt_form() const;
    void unsafe_load(const SEALContext &context, const seal_byte *in, std::size_t size);
};
class Ciphertext {
public:
    std::uint64_t *data();
    const std::uint64_t *data() const;
    std::size_t size() const;
    std::size_t coeff_modulus_size() const;
    std::size_t poly_modulus_degree() const;
    bool is_ntt_form() const;
    const parms_id_type &parms_id() const;
    void resize(const SEALContext &context, parms_id_type parms_id, std::size_t size);
};
class PublicKey { public: const Ciphertext &data() const; };
class KSwitchKeys { public: const std::vector<std::vector<PublicKey>> &data() const; };
class RelinKeys : public KSwitchKeys {};
class Evaluator;
}
'''

APSU_COMMON = r'''
#pragma once
#include <algorithm>
#include <functional>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>
#include "seal/seal.h"
#define STOPWATCH(sw, name) do {} while (0)
struct ApsuLogSink { template <class T> ApsuLogSink &operator<<(const T &) { return *this; } };
#define APSU_LOG_INFO(msg) do { ApsuLogSink s_; s_ << msg; } while (0)
#define APSU_LOG_ERROR(msg) do { ApsuLogSink s_; s_ << msg; } while (0)
namespace apsu {
namespace util { template <class T, class S> T safe_cast(S v) { return static_cast<T>(v); } }
using util::safe_cast;
struct PSUParams {
    struct QueryParams { std::uint32_t ps_low_degree; std::set<std::uint32_t> query_powers; };
    const QueryParams &query_params() const;
    std::string to_string() const;
};
class CryptoContext {
public:
    std::shared_ptr<seal::SEALContext> seal_context() const;
    std::shared_ptr<seal::RelinKeys> relin_keys() const;
    std::shared_ptr<seal::Evaluator> evaluator() const;
};
class PowersDag {
public:
    struct PowersNode { std::uint32_t power, depth; std::pair<std::uint32_t, std::uint32_t> parents; bool is_source() const; };
    std::set<std::uint32_t> target_powers() const;
    template <class Func> void apply(Func &&func) const { for (const PowersNode &nd : nodes_) func(nd); }
    template <class Func> void parallel_apply(Func &&func) const { apply(func); }
private:
    std::vector<PowersNode> nodes_;
};
namespace network { class Channel {}; }
namespace receiver {
struct ResultPackage {
    seal::compr_mode_type compr_mode;
    std::uint32_t cache_idx, bundle_idx, nonce_byte_count, label_byte_count;
    seal::Ciphertext psu_result;
};
using ResultPart = std::unique_ptr<ResultPackage>;
}
}
'''

BIN_BUNDLE_H = r'''
#pragma once
#include "apsu/common_stub.h"
namespace apsu { namespace receiver {
namespace gpu { class HeGpu; }
struct BatchedPlaintextPolyn {
    std::vector<std::vector<unsigned char>> batched_coeffs;
    CryptoContext crypto_context;
    gpu::HeGpu *he_gpu = nullptr;                 // added by the adapter
    seal::Ciphertext eval(const std::vector<seal::Ciphertext> &ciphertext_powers, seal::MemoryPoolHandle &pool, seal::Plaintext &random_plain) const;
    seal::Ciphertext eval_patstock(const CryptoContext &eval_crypto_context, const std::vector<seal::Ciphertext> &ciphertext_powers,
                                   std::size_t ps_low_degree, seal::MemoryPoolHandle &pool, seal::Plaintext &random_plain) const;
};
struct BinBundleCache { BatchedPlaintextPolyn batched_matching_polyn; };
class BinBundle { public: void regen_cache(); const BinBundleCache &get_cache() const; };
}}
'''

RECEIVER_DB_H = r'''
#pragma once
#include "apsu/bin_bundle.h"
namespace apsu { namespace receiver {
class ReceiverDB {
public:
    void generate_caches();
    std::vector<std::reference_wrapper<const BinBundleCache>> get_cache_at(std::uint32_t bundle_idx);
    const PSUParams &get_params() const;
    std::size_t get_nonce_byte_count() const;
    std::size_t get_label_byte_count() const;
    const std::shared_ptr<gpu::HeGpu> &he_gpu() const { return he_gpu_; }      // added by the adapter
private:
    std::vector<std::vector<BinBundle>> bin_bundles_;
    CryptoContext crypto_context_;
    std::shared_ptr<gpu::HeGpu> he_gpu_;                                       // added by the adapter
};
}}
'''

RECEIVER_OSN_H = r'''
#pragma once
#include "apsu/receiver_db.h"
namespace apsu { namespace receiver {
using CiphertextPowers = std::vector<seal::Ciphertext>;
extern std::vector<seal::Plaintext> random_plain_list;
class Receiver {
public:
    static void ComputePowers(const std::shared_ptr<ReceiverDB> &receiver_db, const CryptoContext &crypto_context,
                              std::vector<CiphertextPowers> &all_powers, const PowersDag &pd, std::uint32_t bundle_idx,
                              seal::MemoryPoolHandle &pool);
    static void ProcessBinBundleCache(const std::shared_ptr<ReceiverDB> &receiver_db, const CryptoContext &crypto_context,
                                      std::reference_wrapper<const BinBundleCache> cache, std::vector<CiphertextPowers> &all_powers,
                                      network::Channel &chl, std::function<void(network::Channel &, ResultPart)> send_rp_fun,
                                      std::uint32_t bundle_idx, seal::compr_mode_type compr_mode, seal::MemoryPoolHandle &pool,
                                      std::uint32_t cache_idx, std::uint32_t pack_idx);
};
}}
'''


def test_adapter_is_well_formed_against_forward_declarations(tmp_path):
    inc = tmp_path / "inc"
    (inc / "seal").mkdir(parents=True)
    (inc / "apsu").mkdir()
    (inc / "seal" / "seal.h").write_text(SEAL_H)
    (inc / "apsu" / "common_stub.h").write_text(APSU_COMMON)
    (inc / "apsu" / "bin_bundle.h").write_text(BIN_BUNDLE_H)
    (inc / "apsu" / "receiver_db.h").write_text(RECEIVER_DB_H)
    (inc / "apsu" / "receiver_osn.h").write_text(RECEIVER_OSN_H)
    shutil.copy(os.path.join(ROOT, "integration", "he_gpu.h"), inc / "apsu" / "he_gpu.h")
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", str(inc), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "integration", "receiver_hot_path.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
