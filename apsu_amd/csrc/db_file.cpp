// N2: on-disk image of the whole database (db_file.h).  Host-side file handling only; the BinBundle images themselves are written
// and read by Engine::save_bundle / load_bundle.
#include "db_file.h"

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "engine.h"

namespace apsu_he {

namespace {
struct DbFileHeader {                        // little-endian, 256 bytes
    char magic[8];                           // "APSUHED1"
    uint64_t header_bytes, table_offset, count, total_bytes;
    uint64_t n, t, K, q[8];
    uint32_t ps_low_degree, max_items_per_bin, table_size, felts_per_item;
    uint64_t table_checksum;                 // FNV-1a over the table
    unsigned char pad[256 - 8 - 32 - 88 - 16 - 8];
};
static_assert(sizeof(DbFileHeader) == 256, "database file header layout");
static_assert(sizeof(DbFileEntry) == 32, "database file table entry layout");
constexpr uint64_t ALIGN = 4096;

uint64_t fnv1a64(const unsigned char *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
uint64_t align_up(uint64_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

void fill_fingerprint(DbFileHeader &hd, const Engine &e)
{
    const HeParams &hp = e.he();
    const PSUParams *psu = e.psu();
    if (!psu) throw std::logic_error("context was created without PSUParams");
    hd.n = hp.n; hd.t = hp.t; hd.K = (uint64_t)hp.K;
    for (int j = 0; j < hp.K && j < 8; j++) hd.q[j] = hp.key_q[j];
    hd.ps_low_degree = psu->query_params.ps_low_degree;
    hd.max_items_per_bin = psu->table_params.max_items_per_bin;
    hd.table_size = psu->table_params.table_size;
    hd.felts_per_item = psu->item_params.felts_per_item;
}

struct File {
    std::FILE *f = nullptr;
    ~File() { if (f) std::fclose(f); }
};
void write_all(std::FILE *f, const void *p, size_t bytes, const std::string &path)
{
    if (bytes && std::fwrite(p, 1, bytes, f) != bytes) throw std::runtime_error("cannot write " + path + ": " + std::strerror(errno));
}
void pad_to(std::FILE *f, uint64_t &pos, uint64_t target, const std::string &path)
{
    static const unsigned char zeros[ALIGN] = { 0 };
    while (pos < target) {
        const size_t chunk = (size_t)std::min<uint64_t>(target - pos, ALIGN);
        write_all(f, zeros, chunk, path);
        pos += chunk;
    }
}
} // namespace

void db_file_save(const std::string &path, Engine *const *engines, const Bundle *const *bundles, size_t count)
{
    if (count && (!engines || !bundles)) throw std::invalid_argument("null argument");
    if (!count) throw std::invalid_argument("an empty database cannot be saved (the file carries the parameters of its BinBundles' context)");
    DbFileHeader hd;
    std::memset(&hd, 0, sizeof(hd));
    std::memcpy(hd.magic, "APSUHED1", 8);
    fill_fingerprint(hd, *engines[0]);
    hd.header_bytes = sizeof(hd);
    hd.table_offset = sizeof(hd);
    hd.count = count;
    std::vector<DbFileEntry> table(count);
    uint64_t pos = align_up(sizeof(hd) + count * sizeof(DbFileEntry));
    for (size_t i = 0; i < count; i++) {
        if (!engines[i] || !bundles[i]) throw std::invalid_argument("null bundle");
        DbFileHeader other;
        std::memset(&other, 0, sizeof(other));
        fill_fingerprint(other, *engines[i]);
        if (std::memcmp(&other.n, &hd.n, offsetof(DbFileHeader, table_checksum) - offsetof(DbFileHeader, n)) != 0)
            throw std::invalid_argument("the BinBundles belong to contexts with different parameters");
        table[i] = DbFileEntry{ bundles[i]->bundle_idx, bundles[i]->cache_idx, bundles[i]->degree, 0, pos, engines[i]->bundle_image_size(*bundles[i]) };
        pos = align_up(pos + table[i].bytes);
    }
    hd.total_bytes = pos;
    hd.table_checksum = fnv1a64(reinterpret_cast<const unsigned char *>(table.data()), count * sizeof(DbFileEntry));
    const std::string tmp = path + ".tmp";
    {
        File out;
        out.f = std::fopen(tmp.c_str(), "wb");
        if (!out.f) throw std::runtime_error("cannot create " + tmp + ": " + std::strerror(errno));
        uint64_t at = 0;
        write_all(out.f, &hd, sizeof(hd), tmp); at += sizeof(hd);
        write_all(out.f, table.data(), count * sizeof(DbFileEntry), tmp); at += count * sizeof(DbFileEntry);
        std::vector<unsigned char> buf;
        for (size_t i = 0; i < count; i++) {
            pad_to(out.f, at, table[i].offset, tmp);
            buf.resize(table[i].bytes);
            const size_t wrote = engines[i]->save_bundle(*bundles[i], buf.data(), buf.size());
            if (wrote != table[i].bytes) throw std::logic_error("BinBundle image size changed while saving");
            write_all(out.f, buf.data(), wrote, tmp); at += wrote;
        }
        pad_to(out.f, at, hd.total_bytes, tmp);
        if (std::fflush(out.f) != 0) throw std::runtime_error("cannot write " + tmp + ": " + std::strerror(errno));
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) {             // readers never see a half-written database
        const std::string why = std::strerror(errno);
        std::remove(tmp.c_str());
        throw std::runtime_error("cannot move " + tmp + " to " + path + ": " + why);
    }
}

DbFile::DbFile(const std::string &path)
{
    fd_ = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd_ < 0) throw std::runtime_error("cannot open " + path + ": " + std::strerror(errno));
    struct stat st;
    if (::fstat(fd_, &st) != 0 || st.st_size < (off_t)sizeof(DbFileHeader)) {
        ::close(fd_); fd_ = -1;
        throw std::invalid_argument(path + " is not a database file (too short)");
    }
    bytes_ = (size_t)st.st_size;
    void *m = ::mmap(nullptr, bytes_, PROT_READ, MAP_SHARED, fd_, 0);
    if (m == MAP_FAILED) {
        const std::string why = std::strerror(errno);
        ::close(fd_); fd_ = -1;
        throw std::runtime_error("cannot map " + path + ": " + why);
    }
    base_ = static_cast<const unsigned char *>(m);
    try {
        DbFileHeader hd;
        std::memcpy(&hd, base_, sizeof(hd));
        if (std::memcmp(hd.magic, "APSUHED1", 8) != 0 || hd.header_bytes != sizeof(hd)) throw std::invalid_argument(path + " is not a database file");
        if (hd.total_bytes != bytes_) throw std::invalid_argument(path + ": database file size mismatch (truncated?)");
        if (hd.table_offset != sizeof(hd) || hd.count > (bytes_ - sizeof(hd)) / sizeof(DbFileEntry)) throw std::invalid_argument(path + ": bad BinBundle table");
        entries_.resize((size_t)hd.count);
        std::memcpy(entries_.data(), base_ + hd.table_offset, entries_.size() * sizeof(DbFileEntry));
        if (fnv1a64(reinterpret_cast<const unsigned char *>(entries_.data()), entries_.size() * sizeof(DbFileEntry)) != hd.table_checksum)
            throw std::invalid_argument(path + ": BinBundle table is corrupt (checksum)");
        const uint64_t first = sizeof(hd) + entries_.size() * sizeof(DbFileEntry);
        for (const DbFileEntry &e : entries_)
            if (e.offset < first || e.offset % ALIGN || e.bytes > bytes_ || e.offset > bytes_ - e.bytes)
                throw std::invalid_argument(path + ": BinBundle table points outside the file");
    } catch (...) {
        ::munmap(const_cast<unsigned char *>(base_), bytes_);
        ::close(fd_);
        base_ = nullptr; fd_ = -1;
        throw;
    }
}

DbFile::~DbFile()
{
    if (base_) ::munmap(const_cast<unsigned char *>(base_), bytes_);
    if (fd_ >= 0) ::close(fd_);
}

void DbFile::check_parameters(const Engine &e) const
{
    DbFileHeader hd, mine;
    std::memcpy(&hd, base_, sizeof(hd));
    std::memset(&mine, 0, sizeof(mine));
    fill_fingerprint(mine, e);
    if (std::memcmp(&mine.n, &hd.n, offsetof(DbFileHeader, table_checksum) - offsetof(DbFileHeader, n)) != 0)
        throw std::invalid_argument("the database file was built for different parameters");
}

std::unique_ptr<Bundle> DbFile::load(Engine &e, size_t i) const
{
    const DbFileEntry &en = entries_.at(i);
    std::unique_ptr<Bundle> b = e.load_bundle(base_ + en.offset, (size_t)en.bytes);
    if (b->bundle_idx != en.bundle_idx || b->cache_idx != en.cache_idx || b->degree != en.degree)
        throw std::invalid_argument("database file: table entry and BinBundle image disagree");
    return b;
}

} // namespace apsu_he
