// N3: the reference's network framing around the query-evaluation path, without flatc / flatbuffers.
//
// What the reference pins — and what is implemented here, reader and writer — is the FlatBuffers framing of
//   ReceiverOperationHeader          common/apsu/network/rop_header.fbs, receiver_operation.cpp:27-87
//   ReceiverOperation{QueryRequest}  common/apsu/network/rop.fbs,        receiver_operation.cpp:180-350
//   ReceiverOperationResponse{QueryResponse}  rop_response.fbs,          receiver_operation_response.cpp
//   ResultPackage                    common/apsu/network/result_package.fbs, result_package.cpp:29-150
// (size-prefixed buffers, FinishSizePrefixed / VerifySizePrefixed...Buffer).  The byte vectors inside
// (Ciphertext.data, QueryRequest.relin_keys) are SEAL's own serialisation (seal_object.h:161-219 -> Ciphertext::save);
// nothing in /root/reference pins that format, so they are OPAQUE byte ranges here; seal_codec.h restates SEAL's
// object serialisation (seeded ciphertexts, zlib bodies, KSwitchKeys, parms_id) from memory of upstream SEAL and is marked UNPINNED.
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

// The other members of the two unions (rop.fbs Request: 1 ParmsRequest, 2 OPRFRequest, 3 QueryRequest, 4 plainResponse;
// rop_response.fbs Response: 1 ParmsResponse, 2 OPRFResponse, 3 QueryResponse).  The parameter exchange and the querier's
// decrypted results on their way back frame the path on either side; OPRF messages are opaque here (byte vector only).
uint8_t peek_request_type(const uint8_t *buf, size_t size);          // the union tag of a ReceiverOperation
uint8_t peek_response_type(const uint8_t *buf, size_t size);         // ... of a ReceiverOperationResponse
std::vector<uint8_t> build_parms_request();                           // ParmsRequest {}
std::vector<uint8_t> build_parms_response(Span psu_params);           // ParmsResponse { data:[ubyte] } = PSUParams::save bytes
Span parse_parms_response(const uint8_t *buf, size_t size);           // empty span when data is absent
// plainResponse { bundle_idx:uint32; psu_result:[uint64] (required); cache_idx:uint32 }
struct PlainResponse { uint32_t bundle_idx = 0, cache_idx = 0; std::vector<uint64_t> psu_result; };
std::vector<uint8_t> build_plain_response(const PlainResponse &p);
PlainResponse parse_plain_response(const uint8_t *buf, size_t size);

// common/apsu/psu_params.fbs -- PSUParams::save / Load (psu_params.cpp:182-290; size-prefixed):
//   PSUParams { version:uint32; item_params:ItemParams; table_params:TableParams; query_params:QueryParams; seal_params:SEALParams (required) }
//   struct ItemParams { felts_per_item:uint32 }   struct TableParams { table_size, max_items_per_bin, hash_func_count : uint32 }
//   QueryParams { ps_low_degree:uint32; query_powers:[uint32] }   SEALParams { data:[ubyte] (required) } = EncryptionParameters::save bytes
struct PsuParamsWire {
    uint32_t version = 1;                                             // apsu_serialization_version (version.cpp:12)
    uint32_t felts_per_item = 0, table_size = 0, max_items_per_bin = 0, hash_func_count = 0, ps_low_degree = 0;
    std::vector<uint32_t> query_powers;
    Span seal_params;
};
std::vector<uint8_t> build_psu_params(const PsuParamsWire &p);
// throws "failed to load parameters: invalid buffer" / "... incompatible serialization version"
PsuParamsWire parse_psu_params(const uint8_t *buf, size_t size);

// receiver/apsu/receiver_db.fbs -- the header ReceiverDB::save writes in front of the BinBundles (receiver_db.cpp:1182-1232):
//   ReceiverDB { params:[ubyte] (required); info:ReceiverDBInfo; oprf_key:[ubyte] (required); hashed_items:[HashedItem] (required);
//                bin_bundle_count:uint32 }   struct ReceiverDBInfo { label_byte_count, nonce_byte_count : uint32; item_count:uint64;
//                compressed, stripped : bool }   struct HashedItem { low_word, high_word : uint64 }
struct ReceiverDbHeader {
    Span params, oprf_key;
    uint32_t label_byte_count = 0, nonce_byte_count = 0, bin_bundle_count = 0;
    uint64_t item_count = 0, hashed_item_count = 0;
    bool compressed = false, stripped = false;
    size_t consumed = 0;                                              // where the first BinBundle starts
};
ReceiverDbHeader parse_receiver_db_header(const uint8_t *buf, size_t size);

// receiver/apsu/bin_bundle.fbs -- what ReceiverDB::save appends per BinBundle (BinBundle::save, bin_bundle.cpp:1085-1168; size-prefixed):
//   BinBundle { bundle_idx:uint32; mod:uint64; item_bins:FEltMatrix (required); label_bins:[FEltMatrix]; cache:BinBundleCache; stripped:bool }
//   FEltMatrix { rows:[FEltArray] (required) }   FEltArray { felts:[uint64] (required) }
//   BinBundleCache { felt_matching_polyns:FEltMatrix (required); batched_matching_polyn:BatchedPlaintextPolyn (required); ... }
//   BatchedPlaintextPolyn { coeffs:[Plaintext] (required) }   Plaintext { data:[ubyte] (required) }   (SEAL-serialised seal::Plaintext)
// The hot path needs the cache's batched matching polynomial (-> apsu_he_db_upload_bundle_serialized) or, when the cache was not
// saved, the item bins (-> apsu_he_db_build_bundle rebuilds it on the GPU).  Labels and interpolation polynomials are not read
// (APSU is the unlabeled protocol on this path).
struct SavedBinBundle {
    uint32_t bundle_idx = 0;
    uint64_t mod = 0;
    bool stripped = false, has_cache = false;
    std::vector<std::vector<uint64_t>> item_bins;        // [bin] -> field elements
    std::vector<Span> batched_coeffs;                    // cache.batched_matching_polyn.coeffs[d].data
    size_t consumed = 0;                                 // bytes of the buffer this BinBundle occupies (prefix included)
};
// throws std::runtime_error("failed to load BinBundle: invalid buffer")
SavedBinBundle parse_bin_bundle(const uint8_t *buf, size_t size);

} // namespace wire
} // namespace apsu_he
