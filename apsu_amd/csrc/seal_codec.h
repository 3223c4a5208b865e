// N3: Microsoft SEAL's own object serialisation — what travels INSIDE the framed messages of wire.h
// (Ciphertext.data, QueryRequest.relin_keys) — so that a DB-side process can take a real QueryRequest without SEAL on the host.
//
// **UNPINNED.**  Nothing in /root/reference pins this format (SEAL is an external dependency, no vectors, no tests), and
// SEAL is not in this image: everything here restates upstream SEAL (>= 3.6 / 4.x; native/src/seal/{serialization,ciphertext,
// kswitchkeys,publickey,dynarray,randomgen,encryptionparams}.{h,cpp}, util/{rlwe,hash,ztools}.cpp) from memory, and is
// checked only against an independent Python model of the same description (tests/test_seal_codec.py; zlib and the BLAKE2b
// core ARE pinned, by python's zlib / hashlib).  What the reference does with these objects:
//   querier: Encryptor::encrypt_symmetric -> Serializable<Ciphertext> (c1 replaced by the seed of the PRNG that sampled it)
//            sender/apsu/plaintext_powers.cpp:41-46 ; KeyGenerator::create_relin_keys -> Serializable<RelinKeys> sender_osn.cpp:223-227
//            both saved with compr_mode_default (zstd if SEAL was built with it, else zlib) common/apsu/seal_object.h:183-196
//            -- all three modes are read and written here
//   DB side: SEALObject::load / extract -> Ciphertext::load expands the seed  receiver/apsu/query.cpp:44-80
//
// Layout restated (all integers little-endian):
//   SEALHeader (16 B): magic 0xA15E u16 | header_size 0x10 u8 | version major u8 | minor u8 | compr_mode u8 (0 none, 1 zlib,
//                      2 zstd) | reserved u16 | size u64 (header + body as stored)
//   body as stored   : the member bytes; with zlib ONE deflate stream (zlib format) of them, with zstd ONE zstd frame
//   Ciphertext       : parms_id 4 x u64 | is_ntt_form u8 | size u64 | poly_modulus_degree u64 | coeff_modulus_size u64 |
//                      correction_factor u64 (version 4.x) | scale f64 | DynArray  [| UniformRandomGeneratorInfo  if seeded]
//                      seeded: the DynArray holds c0 only (size/2 of the words); c1 = sample_poly_uniform(PRNG(seed))
//   DynArray<u64>    : own SEALHeader (compr none) | count u64 | words
//   UniformRandomGeneratorInfo : own SEALHeader | prng_type u8 (1 blake2xb, 2 shake256) | seed 64 B
//                      shake256 (keccak.h): buffer k = SHAKE256(seed || k as u64) squeezed to 4096 bytes; such objects are expanded on the
//                      host whatever the caller asked for (the device kernels know Blake2xb only) and then count as unseeded
//   PublicKey        : exactly its Ciphertext's object (PublicKey::save forwards to pk_.save: ONE SEALHeader, no envelope of its own)
//   KSwitchKeys      : parms_id 4 x u64 | dim1 u64 | for each: dim2 u64 | dim2 x PublicKey (= Ciphertext) object
//                      RelinKeys: dim1 = 1, dim2 = decomposition count = K - 1, every key ciphertext size 2 over all K primes, NTT form
//   parms_id         : BLAKE2b-256 of the u64 words {scheme (BFV = 1), poly_modulus_degree, coeff moduli..., plain_modulus}
//   sample_poly_uniform (util/rlwe.cpp): fill L*n words from the generator, then per limb replace every word >= the largest
//                      multiple of q below 2^64 - 1 by fresh 64-bit draws taken from the SAME stream, and reduce mod q
//   generator        : Blake2xb (blake2x.h): 4096-byte buffers = blake2xb(4096, counter u64, key = seed), bytes handed out in order
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace apsu_he {
namespace sealio {

enum : uint8_t { COMPR_NONE = 0, COMPR_ZLIB = 1, COMPR_ZSTD = 2 };

// encryption parameters of one level of the modulus chain: parms_id identifies it inside serialised objects
struct Level { uint64_t parms_id[4]; std::vector<uint64_t> q; };
void compute_parms_id(uint64_t out[4], uint64_t scheme, uint64_t poly_modulus_degree, const uint64_t *coeff_modulus, size_t count,
                      uint64_t plain_modulus);
// key level first (all K primes), then every data level down to one prime (SEALContext's chain)
std::vector<Level> modulus_chain(uint64_t poly_modulus_degree, const std::vector<uint64_t> &key_moduli, uint64_t plain_modulus);

// util::sample_poly_uniform under one of SEAL's generators seeded with `seed`: dst[L][n].  prng_type 1 = Blake2xb (SEAL's default),
// 2 = Shake256 (a SEAL built with SEAL_DEFAULT_PRNG=Shake256; host only -- such objects are always expanded by the codec itself)
enum : uint8_t { PRNG_BLAKE2XB = 1, PRNG_SHAKE256 = 2 };
void sample_poly_uniform(const uint64_t seed[8], const uint64_t *q, size_t L, size_t n, uint64_t *dst, uint8_t prng_type = PRNG_BLAKE2XB);

struct Ciphertext {
    uint64_t parms_id[4] = { 0, 0, 0, 0 };
    uint8_t is_ntt_form = 0;
    uint64_t size = 0, poly_modulus_degree = 0, coeff_modulus_size = 0, correction_factor = 1;
    double scale = 1.0;
    bool seeded = false;                 // load: the object carried a seed (expanded into data); save: write the seeded form
    uint8_t prng_type = PRNG_BLAKE2XB;   // generator of a seeded object
    uint64_t seed[8] = { 0 };
    uint8_t version_major = 4, version_minor = 0;
    std::vector<uint64_t> data;          // [size][coeff_modulus_size][poly_modulus_degree], expanded
};

// One serialised object of `size` bytes at buf (its header says how many are used: *consumed).  zlib and zstd bodies are inflated
// (zstd through the system's libzstd.so.1, loaded at run time; refused with a clear message where that is absent).
// Throws std::runtime_error on malformed input.
// expand = false: a seeded object's c1 is left zero and only ct.seed is filled (the caller expands it, e.g. on the device with
// apsu_he_seed_expand); the chain is then not consulted.
Ciphertext load_ciphertext(const uint8_t *buf, size_t size, const std::vector<Level> &chain, size_t *consumed = nullptr, bool expand = true);
// compr: COMPR_NONE, COMPR_ZLIB or COMPR_ZSTD.  ct.seeded: c1 is NOT written, the seed is (the caller guarantees c1 = sample(seed)).
std::vector<uint8_t> save_ciphertext(const Ciphertext &ct, uint8_t compr);

// seal::Plaintext (plaintext.cpp save_members [SEAL-recall]): parms_id 4 x u64 (all zero = coefficient form, else the level the
// plaintext was transformed to NTT form at) | coeff_count u64 | scale f64 | DynArray.  What the reference keeps per coefficient of
// a BinBundle's batched polynomial (bin_bundle.cpp:421-428: Plaintext::save, compr_mode none or zstd) and re-reads with
// unsafe_load on every use (bin_bundle.cpp:143-146).
struct Plaintext {
    uint64_t parms_id[4] = { 0, 0, 0, 0 };
    uint64_t coeff_count = 0;
    double scale = 1.0;
    uint8_t version_major = 4, version_minor = 0;
    std::vector<uint64_t> data;          // coeff_count words ([L][n] in NTT form)
    bool is_ntt_form() const { return (parms_id[0] | parms_id[1] | parms_id[2] | parms_id[3]) != 0; }
};
Plaintext load_plaintext(const uint8_t *buf, size_t size, size_t *consumed = nullptr);
std::vector<uint8_t> save_plaintext(const Plaintext &pt, uint8_t compr);

// seal::EncryptionParameters (encryptionparams.cpp save_members [SEAL-recall]): scheme u8 (bfv = 1) | poly_modulus_degree u64 |
// coeff_modulus_size u64 | that many Modulus objects | plain_modulus Modulus object; a Modulus object = its own SEALHeader + value u64.
// What PSUParams::save embeds (psu_params.cpp:203-209, compr_mode none) in the parameter exchange and in a saved ReceiverDB.
struct EncryptionParameters {
    uint8_t scheme = 1;
    uint64_t poly_modulus_degree = 0;
    std::vector<uint64_t> coeff_modulus;
    uint64_t plain_modulus = 0;
    uint8_t version_major = 4, version_minor = 0;
};
EncryptionParameters load_encryption_parameters(const uint8_t *buf, size_t size, size_t *consumed = nullptr);
std::vector<uint8_t> save_encryption_parameters(const EncryptionParameters &p, uint8_t compr);

struct KSwitchKeys {
    uint64_t parms_id[4] = { 0, 0, 0, 0 };
    std::vector<std::vector<Ciphertext>> keys;          // [dim1][dim2]
    uint8_t version_major = 4, version_minor = 0;
};
// expand = false: seeded keys keep c1 zero and their seed (Ciphertext::seeded / seed), for a caller that expands them on the device
KSwitchKeys load_kswitch_keys(const uint8_t *buf, size_t size, const std::vector<Level> &chain, size_t *consumed = nullptr, bool expand = true);
std::vector<uint8_t> save_kswitch_keys(const KSwitchKeys &k, uint8_t compr);
// RelinKeys -> the [decomp][2][K][n] array apsu_he_relin_upload takes (keys[0][*], every one size 2 at the key level, NTT form)
std::vector<uint64_t> relin_keys_layout(const KSwitchKeys &k, size_t K, size_t n);

} // namespace sealio
} // namespace apsu_he
