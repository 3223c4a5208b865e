// N2: on-disk image of the whole HBM-resident database -- the counterpart of ReceiverDB::save / Load
// (receiver/apsu/receiver_db.cpp:1182-1429: PSUParams + every BinBundle's flatbuffers / SEAL-serialised cache) for a DB that
// lives on GPUs as raw limb arrays.  One file = one parameter set: a header with the parameter fingerprint, a table of the
// BinBundles (bundle index, cache index, degree, offset, size) and the engine-native BinBundle images (Engine::save_bundle: 256-byte
// header + raw arrays + checksum) at 4096-byte aligned offsets.  The file is opened with mmap: a process that holds a shard of
// the DB (one device of a node) touches only the pages of its own BinBundles, and every array goes to the device with one copy
// straight from the mapping.  75 GiB at 256M-4096 need not be rebuilt (or even read whole) per process.
//
//   DbFileHeader (256 B, little-endian) | DbFileEntry[count] | padding | image 0 | padding | image 1 | ...
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace apsu_he {

class Engine;
struct Bundle;

struct DbFileEntry {
    uint32_t bundle_idx, cache_idx, degree, reserved;
    uint64_t offset, bytes;                  // of the BinBundle image inside the file
};

// Writes the file: every BinBundle is downloaded from its device once (Engine::save_bundle) and appended.  bundles[i] may live on
// different engines of the same parameters (engines[i]): the shards of a multi-device DB go to one file.
void db_file_save(const std::string &path, Engine *const *engines, const Bundle *const *bundles, size_t count);

class DbFile {
public:
    explicit DbFile(const std::string &path);            // mmap, read-only; validates the header and the table against the file size
    ~DbFile();
    DbFile(const DbFile &) = delete;
    DbFile &operator=(const DbFile &) = delete;
    size_t count() const { return entries_.size(); }
    const DbFileEntry &entry(size_t i) const { return entries_.at(i); }
    const unsigned char *image(size_t i) const { return base_ + entries_.at(i).offset; }
    size_t file_bytes() const { return bytes_; }
    // the parameters the file was written for must be the engine's (n, t, coefficient primes, ps_low_degree, max_items_per_bin)
    void check_parameters(const Engine &e) const;
    // BinBundle i onto the engine's device (Engine::load_bundle on the mapped image: checksum, shape and parameter checks)
    std::unique_ptr<Bundle> load(Engine &e, size_t i) const;

private:
    int fd_ = -1;
    const unsigned char *base_ = nullptr;
    size_t bytes_ = 0;
    std::vector<DbFileEntry> entries_;
};

} // namespace apsu_he
