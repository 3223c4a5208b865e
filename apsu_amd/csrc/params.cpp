// See params.h.  Reference behaviour mirrored: common/apsu/psu_params.cpp:95-180 (initialize),
// :290-374 (Load from JSON), common/apsu/util/utils.cpp:146-177 (create_powers_set).
// SEAL-defined constants follow SURVEY.md App. B ([SEAL-recall]: SEAL is not in the image).
#include "params.h"

#include <algorithm>
#include <cctype>
#include <map>
#include <memory>
#include <stdexcept>

namespace apsu_he {

// ======================================================================== tiny JSON reader
namespace {

struct JValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    bool is_integer = false, negative = false;
    unsigned long long u = 0;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;

    const JValue *find(const std::string &key) const
    {
        for (auto &kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

class JParser {
public:
    explicit JParser(const std::string &s) : s_(s) {}
    JValue parse()
    {
        JValue v = value();
        ws();
        if (pos_ != s_.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string &s_;
    size_t pos_ = 0;
    [[noreturn]] void fail(const char *what) const
    {
        throw std::runtime_error(std::string("JSON parse error at offset ") + std::to_string(pos_) + ": " + what);
    }
    void ws() { while (pos_ < s_.size() && std::isspace((unsigned char)s_[pos_])) pos_++; }
    char peek() { ws(); if (pos_ >= s_.size()) fail("unexpected end"); return s_[pos_]; }
    void expect(char c) { if (peek() != c) fail("unexpected character"); pos_++; }
    JValue value()
    {
        char c = peek();
        JValue v;
        if (c == '{') {
            v.kind = JValue::Object; pos_++;
            if (peek() == '}') { pos_++; return v; }
            for (;;) {
                JValue k = string_value();
                expect(':');
                v.obj.emplace_back(k.str, value());
                if (peek() == ',') { pos_++; continue; }
                expect('}');
                return v;
            }
        }
        if (c == '[') {
            v.kind = JValue::Array; pos_++;
            if (peek() == ']') { pos_++; return v; }
            for (;;) {
                v.arr.push_back(value());
                if (peek() == ',') { pos_++; continue; }
                expect(']');
                return v;
            }
        }
        if (c == '"') return string_value();
        if (s_.compare(pos_, 4, "true") == 0) { pos_ += 4; v.kind = JValue::Bool; v.b = true; return v; }
        if (s_.compare(pos_, 5, "false") == 0) { pos_ += 5; v.kind = JValue::Bool; return v; }
        if (s_.compare(pos_, 4, "null") == 0) { pos_ += 4; return v; }
        return number();
    }
    JValue string_value()
    {
        expect('"');
        JValue v; v.kind = JValue::String;
        while (pos_ < s_.size() && s_[pos_] != '"') {
            if (s_[pos_] == '\\') {
                pos_++;
                if (pos_ >= s_.size()) fail("bad escape");
                char e = s_[pos_];
                v.str.push_back(e == 'n' ? '\n' : e == 't' ? '\t' : e);
            } else v.str.push_back(s_[pos_]);
            pos_++;
        }
        if (pos_ >= s_.size()) fail("unterminated string");
        pos_++;
        return v;
    }
    JValue number()
    {
        size_t start = pos_;
        JValue v; v.kind = JValue::Number; v.is_integer = true;
        if (s_[pos_] == '-') { v.negative = true; pos_++; }
        if (pos_ >= s_.size() || !std::isdigit((unsigned char)s_[pos_])) fail("invalid value");
        while (pos_ < s_.size() && (std::isdigit((unsigned char)s_[pos_]) || s_[pos_] == '.' || s_[pos_] == 'e' ||
                                    s_[pos_] == 'E' || s_[pos_] == '+' || s_[pos_] == '-')) {
            if (!std::isdigit((unsigned char)s_[pos_])) v.is_integer = false;
            pos_++;
        }
        std::string tok = s_.substr(start, pos_ - start);
        v.num = std::stod(tok);
        if (v.is_integer) v.u = std::stoull(v.negative ? tok.substr(1) : tok);
        return v;
    }
};

// helpers with the reference's error behaviour (psu_params.cpp json_value_* / get_non_null_json_value)
const JValue &non_null(const JValue &parent, const std::string &name)
{
    const JValue *v = parent.find(name);
    if (!v || v->kind == JValue::Null) throw std::runtime_error("no valid entry for " + name + " found");
    return *v;
}
unsigned long long as_u64(const JValue &v, const std::string &name)
{
    if (v.kind != JValue::Number || !v.is_integer || v.negative)
        throw std::runtime_error(name + " should be an unsigned integer");
    return v.u;
}
uint32_t as_u32(const JValue &v, const std::string &name)
{
    unsigned long long x = as_u64(v, name);
    if (x > 0xFFFFFFFFull) throw std::runtime_error(name + " should be an unsigned int32");
    return (uint32_t)x;
}
int as_int(const JValue &v, const std::string &name)
{
    if (v.kind != JValue::Number || !v.is_integer || v.u > 0x7FFFFFFFull) throw std::runtime_error(name + " should be an int");
    return v.negative ? -(int)v.u : (int)v.u;
}

} // namespace

PSUParams PSUParams::Load(const std::string &json_text)
{
    JValue root = JParser(json_text).parse();
    if (root.kind != JValue::Object) throw std::runtime_error("JSON root must be an object");
    PSUParams p;

    const JValue &tp = non_null(root, "table_params");
    p.table_params.hash_func_count = as_u32(non_null(tp, "hash_func_count"), "hash_func_count");
    p.table_params.table_size = as_u32(non_null(tp, "table_size"), "table_size");
    p.table_params.max_items_per_bin = as_u32(non_null(tp, "max_items_per_bin"), "max_items_per_bin");

    const JValue &ip = non_null(root, "item_params");
    p.item_params.felts_per_item = as_u32(non_null(ip, "felts_per_item"), "felts_per_item");

    const JValue &qp = non_null(root, "query_params");
    p.query_params.ps_low_degree = as_u32(non_null(qp, "ps_low_degree"), "ps_low_degree");
    const JValue &powers = non_null(qp, "query_powers");
    p.query_params.query_powers.insert(1);                       // "Should always contain 1" (:326)
    for (auto &v : powers.arr) p.query_params.query_powers.insert(as_u32(v, "query_powers"));

    const JValue &sp = non_null(root, "seal_params");
    const JValue &bits = non_null(sp, "coeff_modulus_bits");
    p.seal_params.poly_modulus_degree = (size_t)as_u64(non_null(sp, "poly_modulus_degree"), "poly_modulus_degree");
    bool has_pm = sp.find("plain_modulus") != nullptr, has_pb = sp.find("plain_modulus_bits") != nullptr;
    if (has_pm && has_pb) throw std::runtime_error("only one of plain_modulus and plain_modulus_bits must be specified");
    if (!has_pm && !has_pb) throw std::runtime_error("neither plain_modulus nor plain_modulus_bits was specified");
    size_t n = p.seal_params.poly_modulus_degree;
    if (n < 2 || (n & (n - 1)) || n > 32768) throw std::invalid_argument("poly_modulus_degree is invalid");
    u64 t;
    if (has_pm) t = as_u64(*sp.find("plain_modulus"), "plain_modulus");
    else {
        p.seal_params.plain_modulus_bits = as_int(*sp.find("plain_modulus_bits"), "plain_modulus_bits");
        t = plain_modulus_batching(n, p.seal_params.plain_modulus_bits);
    }
    for (auto &v : bits.arr) p.seal_params.coeff_modulus_bits.push_back(as_int(v, "coeff_modulus_bits"));
    p.initialize(t);
    return p;
}

void PSUParams::initialize(u64 t)
{
    using std::invalid_argument;
    seal_params.plain_modulus = t;
    if (!table_params.table_size) throw invalid_argument("table_size cannot be zero");
    if (!table_params.max_items_per_bin) throw invalid_argument("max_items_per_bin cannot be zero");
    if (table_params.hash_func_count < 1 || table_params.hash_func_count > 8)
        throw invalid_argument("hash_func_count is too large or too small");
    if (item_params.felts_per_item < 2 || item_params.felts_per_item > 32)
        throw invalid_argument("felts_per_item is too large or too small");
    if (query_params.ps_low_degree > table_params.max_items_per_bin)
        throw invalid_argument("ps_low_degree cannot be larger than max_items_per_bin");
    if (query_params.query_powers.count(0) || !query_params.query_powers.count(1))
        throw invalid_argument("query_powers cannot contain 0 and must contain 1");
    if (query_params.query_powers.size() > table_params.max_items_per_bin)
        throw invalid_argument("query_powers cannot be larger than max_items_per_bin");
    for (uint32_t p : query_params.query_powers) {
        if (p > table_params.max_items_per_bin)
            throw invalid_argument("query_powers cannot contain values larger than max_items_per_bin");
        uint32_t h = query_params.ps_low_degree + 1;
        if (p > query_params.ps_low_degree && (p % h) != 0)
            throw invalid_argument("query_powers cannot contain values larger than ps_low_degree that are not "
                                   "multiples ps_low_degree + 1");
    }
    // SEALContext validity, restricted to what the hot path relies on (tc128 bit budget is a
    // security policy of the caller and is not re-checked here).
    size_t n = seal_params.poly_modulus_degree;
    if (seal_params.coeff_modulus_bits.empty() || seal_params.coeff_modulus_bits.size() > (size_t)MAXL)
        throw invalid_argument("Microsoft SEAL parameters are invalid: coeff_modulus size");
    for (int b : seal_params.coeff_modulus_bits)
        if (b < 2 || b > 60) throw invalid_argument("Microsoft SEAL parameters are invalid: coeff_modulus bit size");
    if (t < 2 || !is_prime_u64(t) || (t - 1) % (2 * n) != 0)
        throw invalid_argument("Microsoft SEAL parameters do not support batching; plain_modulus must be a prime "
                               "congruent to 1 modulo 2*poly_modulus_degree");

    int t_bits = 64 - __builtin_clzll(t);
    item_bit_count_per_felt = (uint32_t)(t_bits - 1);
    item_bit_count = item_bit_count_per_felt * item_params.felts_per_item;
    if (item_bit_count < 80 || item_bit_count > 128)
        throw invalid_argument("parameters result in too large or too small item_bit_count");
    items_per_bundle = (uint32_t)n / item_params.felts_per_item;
    if (!items_per_bundle) throw invalid_argument("poly_modulus_degree is too small");
    bins_per_bundle = items_per_bundle * item_params.felts_per_item;
    if (table_params.table_size % items_per_bundle)
        throw invalid_argument("table_size must be a multiple of floor(poly_modulus_degree / felts_per_item)");
    bundle_idx_count = table_params.table_size / items_per_bundle;
}

std::set<uint32_t> create_powers_set(uint32_t ps_low_degree, uint32_t target_degree)
{
    if (ps_low_degree > target_degree) throw std::invalid_argument("ps_low_degree cannot be bigger than target_degree");
    if (!target_degree) throw std::invalid_argument("target_degree cannot be zero");
    std::set<uint32_t> result;
    if (ps_low_degree) {
        for (uint32_t p = 1; p <= ps_low_degree; p++) result.insert(p);
        uint32_t first = ps_low_degree + 1;
        for (uint32_t p = first; p <= (target_degree / first) * first; p += first) result.insert(p);
    } else {
        for (uint32_t p = 1; p <= target_degree; p++) result.insert(p);
    }
    return result;
}

// ======================================================================== number theory
ModulusInfo::ModulusInfo(u64 v) : value(v)
{
    if (!v) return;
    bits = 64 - __builtin_clzll(v);
    // floor(2^128 / v) without a 129-bit type: 2^128 = v*k + r  with  (2^128 - 1) = v*k' + r'
    u128 all = ~(u128)0;
    u128 k = all / v;
    if (all % v == (u128)(v - 1)) k += 1;
    ratio[0] = (u64)k;
    ratio[1] = (u64)(k >> 64);
}

u64 ModulusInfo::pow(u64 a, u64 e) const
{
    u64 r = 1 % value;
    a %= value;
    for (; e; e >>= 1) {
        if (e & 1) r = mul(r, a);
        a = mul(a, a);
    }
    return r;
}

u64 ModulusInfo::inv(u64 a) const
{
    __int128 r0 = value, r1 = a % value, s0 = 0, s1 = 1;
    while (r1) {
        __int128 qq = r0 / r1, tmp = r0 - qq * r1;
        r0 = r1; r1 = tmp;
        tmp = s0 - qq * s1; s0 = s1; s1 = tmp;
    }
    if (r0 != 1) throw std::invalid_argument("value is not invertible");
    if (s0 < 0) s0 += value;
    return (u64)s0;
}

bool is_prime_u64(u64 v)
{
    static const u64 bases[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };   // deterministic for 64-bit
    if (v < 2) return false;
    for (u64 b : bases) {
        if (v == b) return true;
        if (v % b == 0) return false;
    }
    ModulusInfo m(v);
    u64 d = v - 1;
    int r = 0;
    while (!(d & 1)) { d >>= 1; r++; }
    for (u64 b : bases) {
        u64 x = m.pow(b, d);
        if (x == 1 || x == v - 1) continue;
        bool composite = true;
        for (int i = 1; i < r && composite; i++) {
            x = m.mul(x, x);
            if (x == v - 1) composite = false;
        }
        if (composite) return false;
    }
    return true;
}

std::vector<u64> get_primes(u64 factor, int bit_size, size_t count)
{
    if (bit_size < 2 || bit_size > 62) throw std::invalid_argument("bit_size is invalid");
    std::vector<u64> out;
    u64 lower = (u64)1 << (bit_size - 1);
    u64 v = (((u64)1 << bit_size) - 1) / factor * factor + 1;
    while (out.size() < count && v > lower) {
        if (is_prime_u64(v)) out.push_back(v);
        v -= factor;
    }
    if (out.size() != count) throw std::logic_error("failed to find enough qualifying primes");
    return out;
}

std::vector<u64> coeff_modulus_create(size_t n, const std::vector<int> &bits)
{
    std::map<int, size_t> count;
    for (int b : bits) count[b]++;
    std::map<int, std::vector<u64>> lists;
    for (auto &kv : count) lists[kv.first] = get_primes(2 * (u64)n, kv.first, kv.second);
    std::vector<u64> out;
    for (int b : bits) {                       // each entry takes the smallest remaining prime of its size
        out.push_back(lists[b].back());
        lists[b].pop_back();
    }
    return out;
}

u64 plain_modulus_batching(size_t n, int bits) { return coeff_modulus_create(n, { bits })[0]; }

u64 minimal_primitive_root(u64 degree, const ModulusInfo &m)
{
    if ((m.value - 1) % degree) throw std::invalid_argument("modulus does not support this root of unity");
    u64 root = 0;
    for (u64 g = 2; g < m.value && !root; g++) {
        u64 r = m.pow(g, (m.value - 1) / degree);
        if (m.pow(r, degree / 2) == m.value - 1) root = r;
    }
    u64 step = m.mul(root, root), cur = root, best = root;
    for (u64 i = 0; i < degree / 2; i++) {     // all odd powers = all primitive roots
        best = std::min(best, cur);
        cur = m.mul(cur, step);
    }
    return best;
}

namespace {

u32 bit_reverse(u32 x, int bits)
{
    u32 r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

NttTablesHost make_ntt_tables(u64 q, size_t n, int logn)
{
    NttTablesHost t;
    t.mod = ModulusInfo(q);
    t.psi = minimal_primitive_root(2 * (u64)n, t.mod);
    u64 ipsi = t.mod.inv(t.psi);
    t.fwd.resize(n); t.fwd_q.resize(n); t.inv.resize(n); t.inv_q.resize(n);
    u64 p = 1, ip = 1;
    for (size_t i = 0; i < n; i++) {
        u32 k = bit_reverse((u32)i, logn);
        t.fwd[k] = p;  t.fwd_q[k] = t.mod.shoup(p);
        t.inv[k] = ip; t.inv_q[k] = t.mod.shoup(ip);
        p = t.mod.mul(p, t.psi);
        ip = t.mod.mul(ip, ipsi);
    }
    t.ninv = t.mod.inv((u64)n % q);
    t.ninv_q = t.mod.shoup(t.ninv);
    // tables of the decimation-in-time inverse (used for narrow moduli): powers of psi^-1 by index
    std::vector<u64> ipow(n);
    ipow[0] = 1;
    for (size_t i = 1; i < n; i++) ipow[i] = t.mod.mul(ipow[i - 1], ipsi);
    t.dit.assign(n, 1); t.dit_q.assign(n, t.mod.shoup(1));
    t.scale.resize(n); t.scale_q.resize(n);
    for (size_t g = 1; g < n; g <<= 1)
        for (size_t j = 0; j < g; j++) {
            u64 w = ipow[(j * (n / g)) % n];                 // psi^(-j*n/g), exponent < n
            t.dit[g + j] = w;
            t.dit_q[g + j] = t.mod.shoup(w);
        }
    for (size_t j = 0; j < n; j++) {
        u64 s = t.mod.mul(t.ninv, ipow[j]);
        t.scale[j] = s;
        t.scale_q[j] = t.mod.shoup(s);
    }
    return t;
}

// product of base (optionally skipping one index) modulo m
u64 prod_mod(const std::vector<u64> &base, int skip, u64 m)
{
    u128 r = 1 % m;
    for (size_t i = 0; i < base.size(); i++)
        if ((int)i != skip) r = r * (base[i] % m) % m;
    return (u64)r;
}

// multi-precision product / division by a word (enough for <= 8 limbs of 60 bits + slack)
struct Big {
    std::vector<u64> w;
    explicit Big(u64 v = 0) : w(12, 0) { w[0] = v; }
    void mul(u64 m)
    {
        u64 carry = 0;
        for (auto &x : w) { u128 p = (u128)x * m + carry; x = (u64)p; carry = (u64)(p >> 64); }
    }
    u64 divmod(u64 d)
    {
        u128 rem = 0;
        for (size_t i = w.size(); i-- > 0;) { u128 cur = (rem << 64) | w[i]; w[i] = (u64)(cur / d); rem = cur % d; }
        return (u64)rem;
    }
    u64 mod(u64 d) const { Big c = *this; return c.divmod(d); }
    int bits() const
    {
        for (size_t i = w.size(); i-- > 0;) if (w[i]) return (int)(64 * i + 64 - __builtin_clzll(w[i]));
        return 0;
    }
};

LevelConstants make_level(const HeParams &hp, int L)
{
    LevelConstants lv;
    lv.L = L;
    lv.q.assign(hp.key_q.begin(), hp.key_q.begin() + L);
    const u64 t = hp.t;
    Big Q(1);
    for (u64 qj : lv.q) Q.mul(qj);
    int q_bits = Q.bits();
    Big Qt = Q;
    lv.q_mod_t = Qt.divmod(t);                                   // Qt = floor(Q / t)
    lv.upper_half_threshold = (t + 1) >> 1;
    for (u64 qj : lv.q) {
        if (qj <= t) throw std::invalid_argument("coeff_modulus primes must exceed plain_modulus (fast plain lift)");
        lv.coeff_div_plain.push_back(Qt.mod(qj));
        lv.upper_half_incr.push_back(qj - t);
    }
    for (int j = 0; j + 1 < L; j++) lv.inv_q_last.push_back(ModulusInfo(lv.q[j]).inv(lv.q[L - 1] % lv.q[j]));

    int t_bits = 64 - __builtin_clzll(t);
    lv.nB = L + ((32 + t_bits + q_bits >= 61 * L + 61) ? 1 : 0);
    if ((size_t)lv.nB + 2 > hp.aux_primes.size()) throw std::logic_error("auxiliary prime list too short");
    lv.m_sk = hp.aux_primes[0];
    lv.gamma = hp.aux_primes[1];
    lv.B.assign(hp.aux_primes.begin() + 2, hp.aux_primes.begin() + 2 + lv.nB);
    std::vector<u64> bsk = lv.B;
    bsk.push_back(lv.m_sk);
    const u64 mt = (u64)1 << 32;

    for (int j = 0; j < L; j++) {
        ModulusInfo mj(lv.q[j]);
        lv.inv_punct_q.push_back(mj.inv(prod_mod(lv.q, j, lv.q[j])));
        lv.q_to_mtilde.push_back(prod_mod(lv.q, j, mt));
    }
    for (u64 m : bsk) {
        std::vector<u64> row;
        for (int j = 0; j < L; j++) row.push_back(prod_mod(lv.q, j, m));
        lv.q_to_bsk.push_back(row);
        ModulusInfo mi(m);
        u64 pq = prod_mod(lv.q, -1, m);
        lv.prod_q_mod_bsk.push_back(pq);
        lv.inv_prod_q_mod_bsk.push_back(mi.inv(pq));
        lv.inv_mtilde_mod_bsk.push_back(mi.inv(mt % m));
    }
    {
        ModulusInfo mm(mt);
        u64 inv = mm.inv(prod_mod(lv.q, -1, mt));
        lv.neg_inv_q_mod_mtilde = inv ? mt - inv : 0;
    }
    for (int i = 0; i < lv.nB; i++) {
        lv.inv_punct_B.push_back(ModulusInfo(lv.B[i]).inv(prod_mod(lv.B, i, lv.B[i])));
        lv.B_to_msk.push_back(prod_mod(lv.B, i, lv.m_sk));
    }
    for (int j = 0; j < L; j++) {
        std::vector<u64> row;
        for (int i = 0; i < lv.nB; i++) row.push_back(prod_mod(lv.B, i, lv.q[j]));
        lv.B_to_q.push_back(row);
        lv.prod_B_mod_q.push_back(prod_mod(lv.B, -1, lv.q[j]));
    }
    lv.inv_prod_B_mod_msk = ModulusInfo(lv.m_sk).inv(prod_mod(lv.B, -1, lv.m_sk));
    return lv;
}

} // namespace

HeParams HeParams::Create(size_t n, const std::vector<u64> &coeff_modulus, u64 plain_modulus)
{
    HeParams hp;
    if (n < 2 || (n & (n - 1))) throw std::invalid_argument("poly_modulus_degree must be a power of two");
    if (coeff_modulus.empty() || coeff_modulus.size() > (size_t)MAXL) throw std::invalid_argument("coeff_modulus size");
    hp.n = n;
    while (((size_t)1 << hp.logn) < n) hp.logn++;
    hp.K = (int)coeff_modulus.size();
    hp.using_keyswitching = hp.K > 1;
    hp.first_chain_idx = hp.K > 1 ? hp.K - 2 : 0;
    hp.t = plain_modulus;
    hp.key_q = coeff_modulus;
    for (u64 q : hp.key_q)
        if (q >> 61 || (q - 1) % (2 * n) || !is_prime_u64(q)) throw std::invalid_argument("coeff_modulus prime is not NTT-friendly");
    int maxL = hp.first_chain_idx + 1;
    hp.aux_primes = get_primes(2 * (u64)n, 61, (size_t)maxL + 3);
    for (u64 q : hp.key_q) hp.ntt.push_back(make_ntt_tables(q, n, hp.logn));
    for (u64 q : hp.aux_primes) hp.ntt.push_back(make_ntt_tables(q, n, hp.logn));
    if (plain_modulus > 1 && (plain_modulus - 1) % (2 * (u64)n) == 0 && is_prime_u64(plain_modulus)) {
        // BatchEncoder tables [SEAL-recall batchencoder.cpp]: NTT mod t and matrix_reps_index_map (generator 3)
        hp.batching = true;
        hp.ntt.push_back(make_ntt_tables(plain_modulus, n, hp.logn));
        hp.slot_map.resize(n);
        const u64 m = 2 * (u64)n;
        u64 pos = 1;
        const size_t row = n >> 1;
        for (size_t i = 0; i < row; i++) {
            hp.slot_map[i] = bit_reverse((u32)((pos - 1) >> 1), hp.logn);
            hp.slot_map[row | i] = bit_reverse((u32)((m - pos - 1) >> 1), hp.logn);
            pos = (pos * 3) & (m - 1);
        }
    }
    for (int c = 0; c <= hp.first_chain_idx; c++) hp.level.push_back(make_level(hp, c + 1));
    if (hp.K > 1) {
        u64 p = hp.key_q[hp.K - 1];
        for (int j = 0; j < hp.K - 1; j++) hp.inv_p_mod_q.push_back(ModulusInfo(hp.key_q[j]).inv(p % hp.key_q[j]));
    }
    int t_bits = 64 - __builtin_clzll(plain_modulus);
    int q0_bits = 64 - __builtin_clzll(hp.key_q[0]);
    int irr = q0_bits - (t_bits + (hp.logn + 1) - 1);
    hp.irrelevant_bit_count = irr > 0 ? irr : 0;
    return hp;
}

HeParams HeParams::FromPSUParams(const PSUParams &p)
{
    return Create(p.seal_params.poly_modulus_degree,
                  coeff_modulus_create(p.seal_params.poly_modulus_degree, p.seal_params.coeff_modulus_bits),
                  p.seal_params.plain_modulus);
}

} // namespace apsu_he
