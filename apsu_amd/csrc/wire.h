// N3: the reference's network framing around the query-evaluation path, without flatc / flatbuffers.
//
// What the reference pins — and what is implemented here, reader and writer — is the FlatBuffers framing of
//   ReceiverOperationHeader          common/apsu/network/rop_header.fbs, receiver_operation.cpp:27-87
//   ReceiverOperation{QueryRequest}  common/apsu/network/rop.fbs,        receiver_operation.cpp:180-350
//   ReceiverOperationResponse{QueryResponse}  rop_response.fbs,          receiver_operation_response.cpp
//   ResultPackage                    common/apsu/network/result_package.fbs, result_package.cpp:29-150
// (size-prefixed buffers, FinishSizePrefixed / VerifySizePrefixed...Buffer).  The byte vectors inside
// (Ciphertext.data, QueryRequest.relin_keys) are SEAL's own serialisation (seal_object.h:161-219 -> Ciphertext::save);
// nothing in /root/reference pins that format, so they are OPAQUE byte ranges here.  seal_envelope_* restates the
// uncompressed (compr_mode::none) envelope from memory of upstream SEAL and is marked UNPINNED.
//
// Host code only (no HIP).  Parsing verifies every offset, length and alignment before use, like flatbuffers::Verifier,
// and fails with the reference's messages ("failed to load ...: invalid buffer").
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace apsu_he {
namespace wire {

struct Span { const uint8_t *p = nullptr; size_t n = 0; };

// rop_header.fbs: enum ReceiverOperationType : uint32 { rop_unknown = 0, rop_parms, rop_oprf, rop_query, rop_response }
struct Header { uint32_t version = 0, type = 0; };
std::vector<uint8_t> build_header(const Header &h);
Header parse_header(const uint8_t *buf, size_t size);

// rop.fbs: QueryRequest { compression_type:ubyte; relin_keys:[ubyte]; query:[QueryRequestPart] (required) }
//          QueryRequestPart { exponent:uint32; cts:[Ciphertext] (required) }   Ciphertext { data:[ubyte] (required) }
struct QueryPart { uint32_t exponent = 0; std::vector<Span> cts; };
struct QueryRequest {
    uint8_t compression_type = 0;
    bool has_relin_keys = false;
    Span relin_keys;
    std::vector<QueryPart> parts;
};
std::vector<uint8_t> build_query_request(const QueryRequest &q);
// throws std::runtime_error: invalid buffer / unexpected operation type / invalid query data (duplicate exponent)
QueryRequest parse_query_request(const uint8_t *buf, size_t size);

// rop_response.fbs: QueryResponse { package_count:uint32; alpha_max_cache_count:uint32 }
struct QueryResponse { uint32_t package_count = 0, alpha_max_cache_count = 0; };
std::vector<uint8_t> build_query_response(const QueryResponse &r);
QueryResponse parse_query_response(const uint8_t *buf, size_t size);

// result_package.fbs
struct ResultPackage {
    uint32_t bundle_idx = 0, cache_idx = 0;
    Span psu_result;
    uint32_t label_byte_count = 0, nonce_byte_count = 0;
    std::vector<Span> label_result;
};
std::vector<uint8_t> build_result_package(const ResultPackage &r);
ResultPackage parse_result_package(const uint8_t *buf, size_t size);

// ---- UNPINNED: SEAL's uncompressed object envelope for a Ciphertext (restated from memory) ----
// 16-byte SEALHeader {magic 0xA15E, header_size 0x10, version major/minor, compr_mode, reserved, size u64} followed by
// parms_id (4 x u64, opaque here: SEAL derives it by hashing the encryption parameters), is_ntt_form (1 byte),
// size, poly_modulus_degree, coeff_modulus_size (u64 each), correction_factor (u64, SEAL 4.x), scale (double), then the
// coefficient DynArray as its own object: header + u64 element count + raw little-endian words.
struct SealCt {
    uint64_t parms_id[4] = { 0, 0, 0, 0 };
    uint8_t is_ntt_form = 0;
    uint64_t size = 0, poly_modulus_degree = 0, coeff_modulus_size = 0, correction_factor = 1;
    double scale = 1.0;
    const uint64_t *data = nullptr;      // size * coeff_modulus_size * poly_modulus_degree words
};
std::vector<uint8_t> seal_envelope_save(const SealCt &ct, uint8_t version_major, uint8_t version_minor);
// returns the header fields and a pointer INTO buf for the words; throws std::runtime_error on malformed input,
// on a compressed object (compr_mode != none) and on a seeded ciphertext (not expanded here)
SealCt seal_envelope_load(const uint8_t *buf, size_t size, uint8_t *version_major, uint8_t *version_minor);

} // namespace wire
} // namespace apsu_he
