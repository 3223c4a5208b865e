// See wire.h.  A hand-written FlatBuffers reader / writer for the four message shapes around the query-evaluation path
// (the public FlatBuffers binary layout: little-endian scalars, vtables, forward 32-bit offsets, length-prefixed
// vectors, unions as a type byte + a table offset).  No generated code, no flatbuffers library.
#include "wire.h"

#include <cstring>
#include <set>
#include <stdexcept>
#include <string>

namespace apsu_he {
namespace wire {

namespace {

// ------------------------------------------------------------------------------------------------ writer
// Tables are written front to back: [vtable][table][children...], so every child offset points forward (the format
// only requires that) and a vtable sits in front of its table (positive soffset), like in flatc's output.
struct Writer {
    std::vector<uint8_t> b;
    size_t pos() const { return b.size(); }
    void align(size_t a) { while (b.size() % a) b.push_back(0); }
    void u8(uint8_t v) { b.push_back(v); }
    void u16(uint16_t v) { b.push_back((uint8_t)v); b.push_back((uint8_t)(v >> 8)); }
    void u32(uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void u64(uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void bytes(const uint8_t *p, size_t n) { if (n) b.insert(b.end(), p, p + n); }
    void patch32(size_t at, uint32_t v) { for (int i = 0; i < 4; i++) b[at + i] = (uint8_t)(v >> (8 * i)); }
    // forward offset stored at `at`, pointing to the current position
    void link(size_t at) { patch32(at, (uint32_t)(pos() - at)); }
};

// One table under construction: fields are declared in id order with their inline size (0 = absent).
struct TableWriter {
    Writer &w;
    std::vector<uint16_t> sizes;          // inline size per field id (0 absent)
    std::vector<size_t> where;            // absolute position of each present field
    size_t table_pos = 0;
    explicit TableWriter(Writer &w_) : w(w_) {}
    // sizes_in: inline byte size of every field id (1 u8, 4 u32 / offset, 4 k: a struct of k u32), 0 = not stored
    void begin(const std::vector<uint16_t> &sizes_in)
    {
        sizes = sizes_in;
        // trailing absent fields are trimmed from the vtable (flatc does the same)
        size_t nf = sizes.size();
        while (nf && !sizes[nf - 1]) nf--;
        // inline layout: soffset, then 4-byte fields, then 1-byte fields (largest first keeps everything aligned)
        std::vector<uint16_t> off(sizes.size(), 0);
        uint16_t cur = 4;
        for (size_t i = 0; i < sizes.size(); i++) if (sizes[i] >= 4) { off[i] = cur; cur = (uint16_t)(cur + sizes[i]); }   // u32, offsets, structs of u32
        for (size_t i = 0; i < sizes.size(); i++) if (sizes[i] == 1) { off[i] = cur; cur += 1; }
        const uint16_t tbl_size = cur;
        const uint16_t vt_size = (uint16_t)(4 + 2 * nf);
        w.align(4);
        if (vt_size % 4) w.u16(0);                                   // so that the table behind the vtable is 4-aligned
        const size_t vt_pos = w.pos();
        w.u16(vt_size);
        w.u16(tbl_size);
        for (size_t i = 0; i < nf; i++) w.u16(off[i]);
        table_pos = w.pos();
        w.u32((uint32_t)(table_pos - vt_pos));                        // soffset: vtable = table - soffset
        where.assign(sizes.size(), 0);
        for (uint16_t o = 4; o < tbl_size; o++) w.u8(0);
        for (size_t i = 0; i < sizes.size(); i++) if (sizes[i]) where[i] = table_pos + off[i];
        w.align(4);
    }
    void set_u32(int id, uint32_t v) { w.patch32(where[id], v); }
    void set_u8(int id, uint8_t v) { w.b[where[id]] = v; }
    void link(int id) { w.align(4); w.link(where[id]); }              // the child starts here
};

void write_byte_vector(Writer &w, Span s)
{
    w.align(4);
    w.u32((uint32_t)s.n);
    w.bytes(s.p, s.n);
    w.align(4);
}

// vector of Ciphertext tables at the current position
void write_ciphertext_vector(Writer &w, const std::vector<Span> &cts)
{
    w.align(4);
    w.u32((uint32_t)cts.size());
    const size_t slots = w.pos();
    for (size_t i = 0; i < cts.size(); i++) w.u32(0);
    for (size_t i = 0; i < cts.size(); i++) {
        TableWriter t(w);
        t.begin({ 4 });
        // the element offset points at the TABLE, which sits behind its vtable
        w.patch32(slots + 4 * i, (uint32_t)(t.table_pos - (slots + 4 * i)));
        t.link(0);
        write_byte_vector(w, cts[i]);
    }
}

std::vector<uint8_t> finish_size_prefixed(Writer &body_with_root_slot)
{
    // layout: [u32 size][u32 root offset][...]; the caller reserved the first 8 bytes
    Writer &w = body_with_root_slot;
    w.align(4);
    w.patch32(0, (uint32_t)(w.pos() - 4));
    return std::move(w.b);
}

// ------------------------------------------------------------------------------------------------ reader
struct Reader {
    const uint8_t *b;
    size_t n;
    const char *what;
    // Work budget (flatbuffers::Verifier has max_tables for the same reason): offsets may legally point at shared children,
    // so a small buffer can describe a quadratic number of (part, ciphertext) visits.  Every table and every vector element
    // visited costs one unit; a buffer without sharing needs at most n / 4 of them (each owns four bytes).
    mutable size_t visits = 0;
    [[noreturn]] void bad() const { throw std::runtime_error(std::string("failed to load ") + what + ": invalid buffer"); }
    void visit(size_t units = 1) const { visits += units; if (visits > n / 4 + 1024) bad(); }
    void need(size_t off, size_t len) const { if (off > n || len > n - off) bad(); }
    uint8_t u8(size_t off) const { need(off, 1); return b[off]; }
    uint16_t u16(size_t off) const { need(off, 2); if (off % 2) bad(); return (uint16_t)(b[off] | (b[off + 1] << 8)); }
    uint32_t u32(size_t off) const
    {
        need(off, 4);
        if (off % 4) bad();
        return (uint32_t)b[off] | ((uint32_t)b[off + 1] << 8) | ((uint32_t)b[off + 2] << 16) | ((uint32_t)b[off + 3] << 24);
    }
    struct Table { size_t pos = 0, vt = 0; uint16_t vt_size = 0, tbl_size = 0; };
    // uoffset stored at `at` -> absolute target (forward, in bounds)
    size_t follow(size_t at) const
    {
        const uint32_t o = u32(at);
        if (o == 0) bad();
        need(at, o);
        return at + o;
    }
    Table table(size_t pos) const
    {
        visit();
        Table t;
        t.pos = pos;
        const int32_t so = (int32_t)u32(pos);
        const int64_t vt = (int64_t)pos - so;
        if (vt < 0 || (uint64_t)vt > n) bad();
        t.vt = (size_t)vt;
        t.vt_size = u16(t.vt);
        t.tbl_size = u16(t.vt + 2);
        if (t.vt_size < 4 || t.vt_size % 2) bad();
        need(t.vt, t.vt_size);
        if (t.tbl_size < 4) bad();
        need(t.pos, t.tbl_size);
        return t;
    }
    // absolute position of field `id` with `size` inline bytes, 0 when absent
    size_t field(const Table &t, int id, size_t size) const
    {
        const size_t slot = 4 + 2 * (size_t)id;
        if (slot + 2 > t.vt_size) return 0;
        const uint16_t off = u16(t.vt + slot);
        if (!off) return 0;
        if ((size_t)off + size > t.tbl_size) bad();
        if (size > 1 && (t.pos + off) % size) bad();
        return t.pos + off;
    }
    uint32_t get_u32(const Table &t, int id, uint32_t dflt) const { const size_t p = field(t, id, 4); return p ? u32(p) : dflt; }
    // inline struct: `size` bytes at an `align`-aligned position
    size_t field_struct(const Table &t, int id, size_t size, size_t align) const
    {
        const size_t slot = 4 + 2 * (size_t)id;
        if (slot + 2 > t.vt_size) return 0;
        const uint16_t off = u16(t.vt + slot);
        if (!off) return 0;
        if ((size_t)off + size > t.tbl_size) bad();
        if ((t.pos + off) % align) bad();
        need(t.pos + off, size);
        return t.pos + off;
    }
    std::vector<uint32_t> u32_vector(size_t pos) const
    {
        const uint32_t len = u32(pos);
        if ((uint64_t)len * 4 > n) bad();
        need(pos + 4, (size_t)len * 4);
        visit(len / 8 + 1);
        std::vector<uint32_t> out(len);
        for (uint32_t i = 0; i < len; i++) out[i] = u32(pos + 4 + 4 * (size_t)i);
        return out;
    }
    uint64_t u64(size_t off) const { need(off, 8); if (off % 8) bad(); uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | b[off + i]; return v; }
    uint64_t get_u64(const Table &t, int id, uint64_t dflt) const { const size_t p = field(t, id, 8); return p ? u64(p) : dflt; }
    // [uint64]: length, then the elements at an 8-byte aligned position
    std::vector<uint64_t> u64_vector(size_t pos) const
    {
        const uint32_t len = u32(pos);
        if ((uint64_t)len * 8 > n) bad();
        need(pos + 4, (size_t)len * 8);
        if (len && (pos + 4) % 8) bad();
        visit(len / 4 + 1);
        std::vector<uint64_t> out(len);
        for (uint32_t i = 0; i < len; i++) out[i] = u64(pos + 4 + 8 * (size_t)i);
        return out;
    }
    uint8_t get_u8(const Table &t, int id, uint8_t dflt) const { const size_t p = field(t, id, 1); return p ? u8(p) : dflt; }
    // offset field -> absolute position of the child, 0 when absent; required fields must be present
    size_t child(const Table &t, int id, bool required) const
    {
        const size_t p = field(t, id, 4);
        if (!p) { if (required) bad(); return 0; }
        return follow(p);
    }
    Span byte_vector(size_t pos) const
    {
        const uint32_t len = u32(pos);
        need(pos + 4, len);
        return Span{ b + pos + 4, len };
    }
    // vector of offsets to tables -> absolute positions
    std::vector<size_t> table_vector(size_t pos) const
    {
        const uint32_t len = u32(pos);
        if ((uint64_t)len * 4 > n) bad();
        need(pos + 4, (size_t)len * 4);
        visit(len);
        std::vector<size_t> out(len);
        for (uint32_t i = 0; i < len; i++) out[i] = follow(pos + 4 + 4 * (size_t)i);
        return out;
    }
    // size-prefixed buffer -> root table position
    size_t root() const
    {
        if (n < 8) bad();
        if (u32(0) != n - 4) bad();
        return follow(4);
    }
    Span ciphertext(size_t pos) const
    {
        const Table t = table(pos);
        return byte_vector(child(t, 0, true));
    }
};

} // namespace

// ================================================================================================ header
std::vector<uint8_t> build_header(const Header &h)
{
    Writer w;
    w.u32(0); w.u32(0);
    TableWriter t(w);
    t.begin({ (uint16_t)(h.version ? 4 : 0), (uint16_t)(h.type ? 4 : 0) });
    w.patch32(4, (uint32_t)(t.table_pos - 4));
    if (h.version) t.set_u32(0, h.version);
    if (h.type) t.set_u32(1, h.type);
    return finish_size_prefixed(w);
}

Header parse_header(const uint8_t *buf, size_t size)
{
    Reader r{ buf, size, "ReceiverOperationHeader" };
    const Reader::Table t = r.table(r.root());
    Header h;
    h.version = r.get_u32(t, 0, 0);
    h.type = r.get_u32(t, 1, 0);           // raw value: the reference's verifier does not range-check the enum either
    return h;                              // (receiver_operation.cpp:67-86); an unknown type is the dispatcher's rop_invalid
}

// ================================================================================================ query request
std::vector<uint8_t> build_query_request(const QueryRequest &q)
{
    Writer w;
    w.u32(0); w.u32(0);
    TableWriter rop(w);
    rop.begin({ 1, 4 });                                               // request_type, request
    w.patch32(4, (uint32_t)(rop.table_pos - 4));
    rop.set_u8(0, 3);                                                  // Request_QueryRequest
    // QueryRequest
    TableWriter qr(w);
    w.align(4);
    {
        // the union value points at the QueryRequest table (behind its vtable)
        qr.begin({ (uint16_t)(q.compression_type ? 1 : 0), (uint16_t)(q.has_relin_keys ? 4 : 0), 4 });
        w.patch32(rop.where[1], (uint32_t)(qr.table_pos - rop.where[1]));
    }
    if (q.compression_type) qr.set_u8(0, q.compression_type);
    if (q.has_relin_keys) { qr.link(1); write_byte_vector(w, q.relin_keys); }
    // query: vector of QueryRequestPart
    qr.link(2);
    w.u32((uint32_t)q.parts.size());
    const size_t slots = w.pos();
    for (size_t i = 0; i < q.parts.size(); i++) w.u32(0);
    for (size_t i = 0; i < q.parts.size(); i++) {
        TableWriter pt(w);
        pt.begin({ (uint16_t)(q.parts[i].exponent ? 4 : 0), 4 });
        w.patch32(slots + 4 * i, (uint32_t)(pt.table_pos - (slots + 4 * i)));
        if (q.parts[i].exponent) pt.set_u32(0, q.parts[i].exponent);
        pt.link(1);
        write_ciphertext_vector(w, q.parts[i].cts);
    }
    return finish_size_prefixed(w);
}

QueryRequest parse_query_request(const uint8_t *buf, size_t size)
{
    Reader r{ buf, size, "ReceiverOperation" };
    const Reader::Table rop = r.table(r.root());
    const uint8_t type = r.get_u8(rop, 0, 0);
    const size_t req_pos = r.child(rop, 1, true);
    if (type != 3) throw std::runtime_error("unexpected operation type");              // receiver_operation.cpp:273-275
    const Reader::Table qr = r.table(req_pos);
    QueryRequest q;
    q.compression_type = r.get_u8(qr, 0, 0);
    if (q.compression_type > 2) throw std::runtime_error("unsupported compression mode");   // :280-282 (none, zlib, zstd)
    if (const size_t rk = r.child(qr, 1, false)) { q.has_relin_keys = true; q.relin_keys = r.byte_vector(rk); }
    std::set<uint32_t> seen;
    for (size_t ppos : r.table_vector(r.child(qr, 2, true))) {
        const Reader::Table pt = r.table(ppos);
        QueryPart part;
        part.exponent = r.get_u32(pt, 0, 0);
        if (!seen.insert(part.exponent).second) throw std::runtime_error("invalid query data");   // :315-317
        for (size_t cpos : r.table_vector(r.child(pt, 1, true))) part.cts.push_back(r.ciphertext(cpos));
        q.parts.push_back(std::move(part));
    }
    return q;
}

// ================================================================================================ query response
std::vector<uint8_t> build_query_response(const QueryResponse &q)
{
    Writer w;
    w.u32(0); w.u32(0);
    TableWriter rr(w);
    rr.begin({ 1, 4 });
    w.patch32(4, (uint32_t)(rr.table_pos - 4));
    rr.set_u8(0, 3);                                                   // Response_QueryResponse
    TableWriter t(w);
    t.begin({ (uint16_t)(q.package_count ? 4 : 0), (uint16_t)(q.alpha_max_cache_count ? 4 : 0) });
    w.patch32(rr.where[1], (uint32_t)(t.table_pos - rr.where[1]));
    if (q.package_count) t.set_u32(0, q.package_count);
    if (q.alpha_max_cache_count) t.set_u32(1, q.alpha_max_cache_count);
    return finish_size_prefixed(w);
}

QueryResponse parse_query_response(const uint8_t *buf, size_t size)
{
    Reader r{ buf, size, "ReceiverOperationResponse" };
    const Reader::Table rr = r.table(r.root());
    const uint8_t type = r.get_u8(rr, 0, 0);
    const size_t pos = r.child(rr, 1, true);
    if (type != 3) throw std::runtime_error("unexpected operation type");
    const Reader::Table t = r.table(pos);
    QueryResponse q;
    q.package_count = r.get_u32(t, 0, 0);
    q.alpha_max_cache_count = r.get_u32(t, 1, 0);
    return q;
}

// ================================================================================================ result package
std::vector<uint8_t> build_result_package(const ResultPackage &p)
{
    Writer w;
    w.u32(0); w.u32(0);
    TableWriter t(w);
    // label_result is always written by the reference (an empty vector when there are no labels, result_package.cpp:45-60)
    t.begin({ (uint16_t)(p.bundle_idx ? 4 : 0), (uint16_t)(p.cache_idx ? 4 : 0), 4, (uint16_t)(p.label_byte_count ? 4 : 0),
              (uint16_t)(p.nonce_byte_count ? 4 : 0), 4 });
    w.patch32(4, (uint32_t)(t.table_pos - 4));
    if (p.bundle_idx) t.set_u32(0, p.bundle_idx);
    if (p.cache_idx) t.set_u32(1, p.cache_idx);
    if (p.label_byte_count) t.set_u32(3, p.label_byte_count);
    if (p.nonce_byte_count) t.set_u32(4, p.nonce_byte_count);
    {
        w.align(4);
        TableWriter ct(w);
        ct.begin({ 4 });
        w.patch32(t.where[2], (uint32_t)(ct.table_pos - t.where[2]));
        ct.link(0);
        write_byte_vector(w, p.psu_result);
    }
    t.link(5);
    write_ciphertext_vector(w, p.label_result);
    return finish_size_prefixed(w);
}

ResultPackage parse_result_package(const uint8_t *buf, size_t size)
{
    Reader r{ buf, size, "ResultPackage" };
    const Reader::Table t = r.table(r.root());
    ResultPackage p;
    p.bundle_idx = r.get_u32(t, 0, 0);
    p.cache_idx = r.get_u32(t, 1, 0);
    p.psu_result = r.ciphertext(r.child(t, 2, true));
    p.label_byte_count = r.get_u32(t, 3, 0);
    p.nonce_byte_count = r.get_u32(t, 4, 0);
    if (const size_t lv = r.child(t, 5, false))
        for (size_t cpos : r.table_vector(lv)) p.label_result.push_back(r.ciphertext(cpos));
    return p;
}

// ================================================================================================ the unions' other members
uint8_t peek_request_type(const uint8_t *buf, size_t size)
{
    Reader r{ buf, size, "ReceiverOperation" };
    const Reader::Table rop = r.table(r.root());
    (void)r.child(rop, 1, true);
    return r.get_u8(rop, 0, 0);
}
uint8_t peek_response_type(const uint8_t *buf, size_t size)
{
    Reader r{ buf, size, "ReceiverOperationResponse" };
    const Reader::Table rr = r.table(r.root());
    (void)r.child(rr, 1, true);
    return r.get_u8(rr, 0, 0);
}

// root { type:u8, value:offset } + the union member's table; returns the member's TableWriter ready for its fields
static void begin_union(Writer &w, uint8_t tag, TableWriter &member, const std::vector<uint16_t> &member_sizes)
{
    w.u32(0); w.u32(0);
    TableWriter root(w);
    root.begin({ 1, 4 });
    w.patch32(4, (uint32_t)(root.table_pos - 4));
    root.set_u8(0, tag);
    w.align(4);
    member.begin(member_sizes);
    w.patch32(root.where[1], (uint32_t)(member.table_pos - root.where[1]));
}

std::vector<uint8_t> build_parms_request()
{
    Writer w;
    TableWriter m(w);
    begin_union(w, 1, m, {});
    return finish_size_prefixed(w);
}

std::vector<uint8_t> build_parms_response(Span psu_params)
{
    Writer w;
    TableWriter m(w);
    begin_union(w, 1, m, { 4 });
    m.link(0);
    write_byte_vector(w, psu_params);
    return finish_size_prefixed(w);
}

Span parse_parms_response(const uint8_t *buf, size_t size)
{
    Reader r{ buf, size, "ReceiverOperationResponse" };
    const Reader::Table rr = r.table(r.root());
    const uint8_t type = r.get_u8(rr, 0, 0);
    const size_t pos = r.child(rr, 1, true);
    if (type != 1) throw std::runtime_error("unexpected operation type");
    const Reader::Table t = r.table(pos);
    if (const size_t d = r.child(t, 0, false)) return r.byte_vector(d);
    return Span{};
}

std::vector<uint8_t> build_plain_response(const PlainResponse &p)
{
    Writer w;
    TableWriter m(w);
    begin_union(w, 4, m, { (uint16_t)(p.bundle_idx ? 4 : 0), 4, (uint16_t)(p.cache_idx ? 4 : 0) });
    if (p.bundle_idx) m.set_u32(0, p.bundle_idx);
    if (p.cache_idx) m.set_u32(2, p.cache_idx);
    // [uint64]: the length word directly in front of the 8-aligned elements
    w.align(4);
    if ((w.pos() + 4) % 8) w.u32(0);
    m.w.link(m.where[1]);
    w.u32((uint32_t)p.psu_result.size());
    for (uint64_t v : p.psu_result) w.u64(v);
    return finish_size_prefixed(w);
}

PlainResponse parse_plain_response(const uint8_t *buf, size_t size)
{
    Reader r{ buf, size, "ReceiverOperation" };
    const Reader::Table rop = r.table(r.root());
    const uint8_t type = r.get_u8(rop, 0, 0);
    const size_t pos = r.child(rop, 1, true);
    if (type != 4) throw std::runtime_error("unexpected operation type");
    const Reader::Table t = r.table(pos);
    PlainResponse p;
    p.bundle_idx = r.get_u32(t, 0, 0);
    p.psu_result = r.u64_vector(r.child(t, 1, true));
    p.cache_idx = r.get_u32(t, 2, 0);
    return p;
}

// ================================================================================================ PSUParams (psu_params.fbs)
std::vector<uint8_t> build_psu_params(const PsuParamsWire &p)
{
    Writer w;
    w.u32(0); w.u32(0);
    TableWriter t(w);
    t.begin({ (uint16_t)(p.version ? 4 : 0), 4, 12, 4, 4 });
    w.patch32(4, (uint32_t)(t.table_pos - 4));
    if (p.version) t.set_u32(0, p.version);
    w.patch32(t.where[1], p.felts_per_item);
    w.patch32(t.where[2], p.table_size);
    w.patch32(t.where[2] + 4, p.max_items_per_bin);
    w.patch32(t.where[2] + 8, p.hash_func_count);
    {   // QueryParams
        TableWriter q(w);
        w.align(4);
        q.begin({ (uint16_t)(p.ps_low_degree ? 4 : 0), 4 });
        w.patch32(t.where[3], (uint32_t)(q.table_pos - t.where[3]));
        if (p.ps_low_degree) q.set_u32(0, p.ps_low_degree);
        q.link(1);
        w.u32((uint32_t)p.query_powers.size());
        for (uint32_t v : p.query_powers) w.u32(v);
    }
    {   // SEALParams
        TableWriter sp(w);
        w.align(4);
        sp.begin({ 4 });
        w.patch32(t.where[4], (uint32_t)(sp.table_pos - t.where[4]));
        sp.link(0);
        write_byte_vector(w, p.seal_params);
    }
    return finish_size_prefixed(w);
}

PsuParamsWire parse_psu_params(const uint8_t *buf, size_t size)
{
    Reader r{ buf, size, "parameters" };
    const Reader::Table t = r.table(r.root());
    PsuParamsWire p;
    p.version = r.get_u32(t, 0, 0);
    if (p.version != 1) throw std::runtime_error("failed to load parameters: incompatible serialization version");   // psu_params.cpp:240-248
    const size_t ip = r.field_struct(t, 1, 4, 4), tp = r.field_struct(t, 2, 12, 4);
    if (!ip || !tp) r.bad();                                          // the reference dereferences both unconditionally
    p.felts_per_item = r.u32(ip);
    p.table_size = r.u32(tp); p.max_items_per_bin = r.u32(tp + 4); p.hash_func_count = r.u32(tp + 8);
    const size_t qp = r.child(t, 3, false);
    if (!qp) r.bad();
    const Reader::Table q = r.table(qp);
    p.ps_low_degree = r.get_u32(q, 0, 0);
    if (const size_t v = r.child(q, 1, false)) p.query_powers = r.u32_vector(v);
    const Reader::Table sp = r.table(r.child(t, 4, true));
    p.seal_params = r.byte_vector(r.child(sp, 0, true));
    return p;
}

// ================================================================================================ a saved ReceiverDB's header
ReceiverDbHeader parse_receiver_db_header(const uint8_t *buf, size_t size)
{
    if (!buf || size < 8) throw std::runtime_error("failed to load ReceiverDB");
    const uint64_t body = (uint64_t)buf[0] | ((uint64_t)buf[1] << 8) | ((uint64_t)buf[2] << 16) | ((uint64_t)buf[3] << 24);
    if (body + 4 > size) throw std::runtime_error("failed to load ReceiverDB");
    Reader r{ buf, (size_t)body + 4, "ReceiverDB" };
    const Reader::Table t = r.table(r.root());
    ReceiverDbHeader h;
    h.consumed = (size_t)body + 4;
    h.params = r.byte_vector(r.child(t, 0, true));
    if (const size_t ip = r.field_struct(t, 1, 24, 8)) {
        h.label_byte_count = r.u32(ip); h.nonce_byte_count = r.u32(ip + 4);
        h.item_count = r.u64(ip + 8);
        h.compressed = r.u8(ip + 16) != 0; h.stripped = r.u8(ip + 17) != 0;
    } else {
        r.bad();                                                      // ReceiverDB::Load dereferences info unconditionally
    }
    h.oprf_key = r.byte_vector(r.child(t, 2, true));
    {
        const size_t hv = r.child(t, 3, true);
        const uint32_t len = r.u32(hv);
        if ((uint64_t)len * 16 > r.n) r.bad();
        r.need(hv + 4, (size_t)len * 16);
        if (len && (hv + 4) % 8) r.bad();
        h.hashed_item_count = len;
    }
    h.bin_bundle_count = r.get_u32(t, 4, 0);
    return h;
}

// ================================================================================================ a saved BinBundle
SavedBinBundle parse_bin_bundle(const uint8_t *buf, size_t size)
{
    // ReceiverDB::save writes the BinBundles one after the other: this one ends where its size prefix says
    if (!buf || size < 8) throw std::runtime_error("failed to load BinBundle: invalid buffer");
    const uint64_t body = (uint64_t)buf[0] | ((uint64_t)buf[1] << 8) | ((uint64_t)buf[2] << 16) | ((uint64_t)buf[3] << 24);
    if (body + 4 > size) throw std::runtime_error("failed to load BinBundle: invalid buffer");
    Reader r{ buf, (size_t)body + 4, "BinBundle" };
    const Reader::Table t = r.table(r.root());
    SavedBinBundle out;
    out.consumed = (size_t)body + 4;
    out.bundle_idx = r.get_u32(t, 0, 0);
    out.mod = r.get_u64(t, 1, 0);
    out.stripped = r.get_u8(t, 5, 0) != 0;
    {
        const Reader::Table m = r.table(r.child(t, 2, true));                       // item_bins
        for (size_t row : r.table_vector(r.child(m, 0, true))) {
            const Reader::Table a = r.table(row);
            out.item_bins.push_back(r.u64_vector(r.child(a, 0, true)));
        }
    }
    if (const size_t cpos = r.child(t, 4, false)) {
        const Reader::Table c = r.table(cpos);
        (void)r.child(c, 0, true);                                                  // felt_matching_polyns: required, not needed here
        const Reader::Table bp = r.table(r.child(c, 1, true));                      // batched_matching_polyn
        for (size_t ppos : r.table_vector(r.child(bp, 0, true))) out.batched_coeffs.push_back(r.ciphertext(ppos));   // Plaintext { data } has Ciphertext's shape
        out.has_cache = true;
    }
    return out;
}

} // namespace wire
} // namespace apsu_he
