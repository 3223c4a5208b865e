"""CPU tier: the adapter a maintainer of the reference would add (integration/he_gpu.h, integration/receiver_hot_path.cpp,
integration/receiver_run_query.cpp) is well-formed C++ and matches include/apsu_he.h.

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
#include <iostream>
#include <memory>
#include <vector>
#define SEAL_VERSION_MAJOR 4
#define SEAL_VERSION_MINOR 1
#define SEAL_VERSION_PATCH 1
#define SEAL_USE_ZLIB
#define SEAL_USE_ZSTD
namespace seal {
using seal_byte = std::byte;
using parms_id_type = std::array<std::uint64_t, 4>;
enum class compr_mode_type : std::uint8_t { none = 0, zlib = 1, zstd = 2 };
enum class scheme_type : std::uint8_t { none = 0, bfv = 1, ckks = 2 };
enum class sec_level_type : int { none = 0, tc128 = 128 };
class MemoryPoolHandle {};
class Modulus { public: std::uint64_t value() const; int bit_count() const; };
struct CoeffModulus { static std::vector<Modulus> Create(std::size_t poly_modulus_degree, std::vector<int> bit_sizes); };
struct PlainModulus { static Modulus Batching(std::size_t poly_modulus_degree, int bit_size); };
class EncryptionParameters {
public:
    EncryptionParameters(scheme_type scheme);
    void set_poly_modulus_degree(std::size_t n);
    void set_coeff_modulus(const std::vector<Modulus> &q);
    void set_plain_modulus(const Modulus &t);
    std::size_t poly_modulus_degree() const;
    const std::vector<Modulus> &coeff_modulus() const;
    const Modulus &plain_modulus() const;
};
namespace util {
class NTTTables { public: std::uint64_t get_root() const; };
int get_power_of_two(std::uint64_t value);
}
class SEALContext {
public:
    class ContextData {
    public:
        std::size_t chain_index() const;
        const EncryptionParameters &parms() const;
        const parms_id_type &parms_id() const;
        std::shared_ptr<const ContextData> next_context_data() const;
        const util::NTTTables *small_ntt_tables() const;
    };
    SEALContext(const EncryptionParameters &parms, bool expand_mod_chain = true, sec_level_type sec_level = sec_level_type::tc128);
    std::shared_ptr<const ContextData> get_context_data(parms_id_type id) const;
    std::shared_ptr<const ContextData> key_context_data() const;
    std::shared_ptr<const ContextData> first_context_data() const;
    std::shared_ptr<const ContextData> last_context_data() const;
    const parms_id_type &key_parms_id() const;
    const parms_id_type &first_parms_id() const;
    const parms_id_type &last_parms_id() const;
    bool using_keyswitching() const;
};
class Plaintext {
public:
    Plaintext();
    explicit Plaintext(std::size_t coeff_count);
    std::uint64_t *data();
    const std::uint64_t *data() const;
    std::uint64_t &operator[](std::size_t i);
    std::size_t coeff_count() const;
    void set_zero();
    bool is_ntt_form() const;
    parms_id_type &parms_id();
    void unsafe_load(const SEALContext &context, const seal_byte *in, std::size_t size);
    std::streamoff save(std::ostream &stream, compr_mode_type compr_mode) const;
    std::streamoff load(const SEALContext &context, std::istream &stream);
};
class Ciphertext {
public:
    std::uint64_t *data();
    const std::uint64_t *data() const;
    std::uint64_t *data(std::size_t poly_index);
    const std::uint64_t *data(std::size_t poly_index) const;
    std::size_t size() const;
    std::size_t coeff_modulus_size() const;
    std::size_t poly_modulus_degree() const;
    bool &is_ntt_form();
    bool is_ntt_form() const;
    const parms_id_type &parms_id() const;
    void resize(const SEALContext &context, parms_id_type parms_id, std::size_t size);
    std::streamoff save(std::ostream &stream, compr_mode_type compr_mode) const;
    std::streamoff load(const SEALContext &context, std::istream &stream);
};
template <class T> class Serializable { public: std::streamoff save(std::ostream &stream, compr_mode_type compr_mode) const; };
class PublicKey { public: Ciphertext &data(); const Ciphertext &data() const; };
class KSwitchKeys {
public:
    std::vector<std::vector<PublicKey>> &data();
    const std::vector<std::vector<PublicKey>> &data() const;
    parms_id_type &parms_id();
    std::streamoff save(std::ostream &stream, compr_mode_type compr_mode) const;
    std::streamoff load(const SEALContext &context, std::istream &stream);
};
class RelinKeys : public KSwitchKeys {};
class SecretKey { public: Plaintext &data(); const Plaintext &data() const; };
class KeyGenerator {
public:
    KeyGenerator(const SEALContext &context);
    const SecretKey &secret_key() const;
    void create_relin_keys(RelinKeys &destination);
    Serializable<RelinKeys> create_relin_keys();
};
class Encryptor {
public:
    Encryptor(const SEALContext &context, const SecretKey &secret_key);
    void encrypt_symmetric(const Plaintext &plain, Ciphertext &destination) const;
    Serializable<Ciphertext> encrypt_symmetric(const Plaintext &plain) const;
};
class Decryptor {
public:
    Decryptor(const SEALContext &context, const SecretKey &secret_key);
    void decrypt(const Ciphertext &encrypted, Plaintext &destination);
    int invariant_noise_budget(const Ciphertext &encrypted);
};
class BatchEncoder {
public:
    BatchEncoder(const SEALContext &context);
    void encode(const std::vector<std::uint64_t> &values, Plaintext &destination) const;
    void decode(const Plaintext &plain, std::vector<std::uint64_t> &destination) const;
};
class Evaluator {
public:
    Evaluator(const SEALContext &context);
    void transform_to_ntt(const Ciphertext &encrypted, Ciphertext &destination) const;
    void transform_to_ntt(const Plaintext &plain, parms_id_type parms_id, Plaintext &destination) const;
    void transform_to_ntt_inplace(Ciphertext &encrypted) const;
    void transform_to_ntt_inplace(Plaintext &plain, parms_id_type parms_id) const;
    void transform_from_ntt_inplace(Ciphertext &encrypted_ntt) const;
    void multiply_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination) const;
    void add(const Ciphertext &a, const Ciphertext &b, Ciphertext &destination) const;
    void add_inplace(Ciphertext &a, const Ciphertext &b) const;
    void add_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination) const;
    void add_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const;
    void mod_switch_to_next(const Ciphertext &encrypted, Ciphertext &destination) const;
    void mod_switch_to_next_inplace(Ciphertext &encrypted) const;
    void mod_switch_to_inplace(Ciphertext &encrypted, parms_id_type parms_id) const;
    void multiply(const Ciphertext &a, const Ciphertext &b, Ciphertext &destination) const;
    void multiply_inplace(Ciphertext &a, const Ciphertext &b) const;
    void square(const Ciphertext &encrypted, Ciphertext &destination) const;
    void relinearize(const Ciphertext &encrypted, const RelinKeys &relin_keys, Ciphertext &destination) const;
    void relinearize_inplace(Ciphertext &encrypted, const RelinKeys &relin_keys) const;
};
}
'''

APSU_COMMON = r'''
#pragma once
#include <algorithm>
#include <functional>
#include <future>
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
namespace util {
class ThreadPool {                                           // common/apsu/util/thread_pool.h:63
public:
    template <class F> auto enqueue(F &&f) -> std::future<decltype(f())> { return std::async(std::launch::deferred, std::forward<F>(f)); }
};
}
class ThreadPoolMgr {                                        // common/apsu/thread_pool_mgr.h:17-45
public:
    util::ThreadPool &thread_pool() const;
};
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
class Receiver {                                             // receiver/apsu/receiver_osn.h:148-250: non-static members
public:
    Receiver();
private:
    void ComputePowers(const std::shared_ptr<ReceiverDB> &receiver_db, const CryptoContext &crypto_context,
                       std::vector<CiphertextPowers> &all_powers, const PowersDag &pd, std::uint32_t bundle_idx,
                       seal::MemoryPoolHandle &pool);
    void ProcessBinBundleCache(const std::shared_ptr<ReceiverDB> &receiver_db, const CryptoContext &crypto_context,
                               std::reference_wrapper<const BinBundleCache> cache, std::vector<CiphertextPowers> &all_powers,
                               network::Channel &chl, std::function<void(network::Channel &, ResultPart)> send_rp_fun,
                               std::uint32_t bundle_idx, seal::compr_mode_type compr_mode, seal::MemoryPoolHandle &pool,
                               std::uint32_t cache_idx, std::uint32_t pack_idx);
    // added by the adapter (integration/receiver_run_query.cpp)
    void EvaluateQueryOnDevice(const std::shared_ptr<ReceiverDB> &receiver_db, const CryptoContext &crypto_context,
                               std::vector<CiphertextPowers> &all_powers, const PowersDag &pd, network::Channel &chl,
                               std::function<void(network::Channel &, ResultPart)> send_rp_fun, seal::compr_mode_type compr_mode,
                               ThreadPoolMgr &tpm);
    std::uint32_t pack_cnt;                                  // receiver_osn.h:237
    std::vector<seal::Plaintext> random_plain_list;          // receiver_osn.h:243
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
    # the batched form of RunQuery's hot section (what bench.py's adapter_calls.batched_ms measures)
    cmd[-1] = os.path.join(ROOT, "integration", "receiver_run_query.cpp")
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_seal_fixture_generator_is_well_formed_against_forward_declarations(tmp_path):
    """integration/seal_fixtures.cpp (the SEAL cross-check kit) against the same hand-written declarations: syntax and overloads only"""
    inc = tmp_path / "inc"
    (inc / "seal").mkdir(parents=True)
    (inc / "seal" / "seal.h").write_text(SEAL_H)
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", str(inc), os.path.join(ROOT, "integration", "seal_fixtures.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
