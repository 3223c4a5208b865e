// Host-side parameter objects for the query-evaluation engine.
//
// PSUParams mirrors apsu::PSUParams (common/apsu/psu_params.h:112-155): same JSON schema
// (common/apsu/psu_params.cpp:290-374), same validation rules and derived sizes (:95-180),
// same exception type (std::invalid_argument).  HeParams derives every number-theoretic
// constant the hot path needs: the SEAL-compatible primes, the modulus
// chain, NTT tables, plaintext scaling constants, BEHZ base-conversion tables and the
// key-switching constants.  Nothing here touches the GPU.
#pragma once
#include <cstdint>
#include <set>
#include <string>
#include <vector>

namespace apsu_he {

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef uint32_t u32;

struct PSUParams {
    struct TableParams { uint32_t hash_func_count = 0, table_size = 0, max_items_per_bin = 0; } table_params;
    struct ItemParams { uint32_t felts_per_item = 0; } item_params;
    struct QueryParams { uint32_t ps_low_degree = 0; std::set<uint32_t> query_powers; } query_params;
    struct SEALParams {
        size_t poly_modulus_degree = 0;
        std::vector<int> coeff_modulus_bits;
        u64 plain_modulus = 0;          // literal, or derived from plain_modulus_bits
        int plain_modulus_bits = 0;
    } seal_params;

    // derived (psu_params.cpp:150-180)
    uint32_t item_bit_count_per_felt = 0, item_bit_count = 0;
    uint32_t items_per_bundle = 0, bins_per_bundle = 0, bundle_idx_count = 0;

    // throws std::invalid_argument / std::runtime_error like PSUParams::Load
    static PSUParams Load(const std::string &json_text);
    void initialize(u64 resolved_plain_modulus);
};

// utils.cpp:146-177
std::set<uint32_t> create_powers_set(uint32_t ps_low_degree, uint32_t target_degree);

// -------------------------------------------------------------------- number theory
struct ModulusInfo {
    u64 value = 0;
    u64 ratio[2] = { 0, 0 };            // floor(2^128 / value)
    int bits = 0;
    explicit ModulusInfo(u64 v = 0);
    u64 reduce(u128 x) const { return (u64)(x % value); }
    u64 mul(u64 a, u64 b) const { return (u64)(((u128)a * b) % value); }
    u64 pow(u64 a, u64 e) const;
    u64 inv(u64 a) const;               // generic (extended Euclid); throws if not invertible
    u64 shoup(u64 w) const { return (u64)(((u128)w << 64) / value); }
};

bool is_prime_u64(u64 v);
std::vector<u64> get_primes(u64 factor, int bit_size, size_t count);            // App. B1
std::vector<u64> coeff_modulus_create(size_t n, const std::vector<int> &bits);  // CoeffModulus::Create
u64 plain_modulus_batching(size_t n, int bits);                                 // PlainModulus::Batching
u64 minimal_primitive_root(u64 degree, const ModulusInfo &m);                   // App. B3

struct NttTablesHost {                  // one modulus
    ModulusInfo mod;
    u64 psi = 0;
    std::vector<u64> fwd, fwd_q;        // fwd[k] = psi^brv(k) and Shoup quotient
    std::vector<u64> inv, inv_q;        // inv[k] = psi^-brv(k)
    std::vector<u64> dit, dit_q;        // dit[g + j] = psi^(-j*n/g)   (decimation-in-time cyclic inverse)
    std::vector<u64> scale, scale_q;    // scale[j] = n^-1 * psi^-j
    u64 ninv = 0, ninv_q = 0;
};

constexpr int MAXL = 8;                 // limbs of q at a data level (<= K-1)
constexpr int MAXB = MAXL + 2;          // |Bsk| <= L + 2

struct LevelConstants {                 // one data level, chain_idx = L - 1
    int L = 0, nB = 0;
    std::vector<u64> q;                 // q_0 .. q_{L-1}
    std::vector<u64> B;                 // auxiliary base, |B| = nB
    u64 m_sk = 0, gamma = 0;
    // plaintext scaling (B5, B7)
    std::vector<u64> coeff_div_plain;   // floor(Q/t) mod q_j
    u64 q_mod_t = 0, upper_half_threshold = 0;
    std::vector<u64> upper_half_incr;   // q_j - t
    // drop-last-limb (B8)
    std::vector<u64> inv_q_last;        // q_{L-1}^-1 mod q_j, j < L-1
    // BEHZ (B9)
    std::vector<u64> inv_punct_q;                     // (Q/q_j)^-1 mod q_j
    std::vector<std::vector<u64>> q_to_bsk;           // [i in Bsk][j] = (Q/q_j) mod Bsk_i   (Bsk = B.., m_sk)
    std::vector<u64> q_to_mtilde;                     // (Q/q_j) mod 2^32
    u64 neg_inv_q_mod_mtilde = 0;
    std::vector<u64> prod_q_mod_bsk, inv_prod_q_mod_bsk, inv_mtilde_mod_bsk;
    std::vector<u64> inv_punct_B;                     // (B/b_i)^-1 mod b_i
    std::vector<std::vector<u64>> B_to_q;             // [j][i] = (B/b_i) mod q_j
    std::vector<u64> B_to_msk;                        // (B/b_i) mod m_sk
    u64 inv_prod_B_mod_msk = 0;
    std::vector<u64> prod_B_mod_q;
};

struct HeParams {
    size_t n = 0;
    int logn = 0;
    int K = 0;                          // limbs at key level
    int first_chain_idx = 0;            // K-2 (K>1) else 0
    bool using_keyswitching = false;
    u64 t = 0;
    std::vector<u64> key_q;             // q_0 .. q_{K-1}; q_{K-1} = special prime when K>1
    std::vector<u64> aux_primes;        // get_primes(2n, 61, .): [0]=m_sk, [1]=gamma, [2..]=B
    // modulus ids: 0..K-1 = key_q ; K + i = aux_primes[i]
    std::vector<NttTablesHost> ntt;     // indexed by modulus id
    std::vector<LevelConstants> level;  // indexed by chain_idx, 0..first_chain_idx
    std::vector<u64> inv_p_mod_q;       // special prime^-1 mod q_j  (B10)
    int irrelevant_bit_count = 0;       // bin_bundle.cpp:67-97
    // BatchEncoder (App. B4), only when t = 1 (mod 2n): modulus id plain_id() in `ntt`, slot -> coefficient map
    bool batching = false;
    std::vector<uint32_t> slot_map;     // matrix_reps_index_map
    int plain_id() const { return K + (int)aux_primes.size(); }

    static HeParams Create(size_t n, const std::vector<u64> &coeff_modulus, u64 plain_modulus);
    static HeParams FromPSUParams(const PSUParams &p);
    int clamp_chain_idx(int chain_idx) const { return chain_idx > first_chain_idx ? first_chain_idx : chain_idx; }
    int aux_id(int i) const { return K + i; }             // modulus id of aux_primes[i]
    int bsk_id(int level_nB, int i) const { return i < level_nB ? K + 2 + i : K + 0; }   // Bsk_i -> modulus id
};

} // namespace apsu_he
