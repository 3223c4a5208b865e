// SEAL cross-check kit: emits, from REAL Microsoft SEAL, the fixtures that turn this repository's "parity unpinned" into
// "pinned" (VERDICT r03 item 3).  Written against SEAL's public API only (seal/seal.h, 3.7 ... 4.1).
//
// THIS PROGRAM HAS NEVER BEEN COMPILED OR RUN BY THE ENGINE'S BUILD: SEAL is not in its image and cannot be fetched.
// tests/test_integration_syntax.py only checks that it is well-formed C++ against hand-written forward declarations
// (which pin nothing about SEAL).  Anyone who has SEAL:
//
//     g++ -std=c++17 -O2 integration/seal_fixtures.cpp -I<seal include dir> -lseal-4.1 -o seal_fixtures
//     ./seal_fixtures tests/golden              # writes tests/golden/seal_*.json
//     python -m pytest tests/test_seal_fixtures.py            # CPU: oracle + object codec;   -m gpu: the HIP path
//
// What it writes (hex strings for every 64-bit word; arrays are [poly][limb][coeff] = Ciphertext::data() order):
//   seal_ops_<tag>.json      the schema of tests/golden/ops_n64.json: per level of the modulus chain random operands and SEAL's
//                            results for transform_to_ntt, multiply_plain (NTT x NTT, coefficient form, one-coefficient
//                            monomial), add, add_plain, mod_switch_to_next, multiply, square, relinearize; primes, psi, parms_ids
//   seal_path_<tag>.json     the schema of tests/golden/path_n64.json: a query encrypted by SEAL, Receiver::ComputePowers and
//                            BatchedPlaintextPolyn::eval / eval_patstock restated call by call in the reference's order
//                            (receiver/apsu/receiver_osn.cpp:412-487, receiver/apsu/bin_bundle.cpp:106-174,192-360), every
//                            target power and every BinBundle result
//   seal_objects_<tag>.json  serialised objects for the N3 codec: parms_id per level; seeded symmetric ciphertexts, RelinKeys
//                            (seeded and expanded), plaintexts (coefficient and NTT form) under compr none / zlib / zstd,
//                            each with the words SEAL's own load gives back
// Tags: n64 (the toy chain of the existing golden files: coefficient bits 40,40,40,36, t 17 bits; needs sec_level_type::none)
// and 16M (16M-4096.json: n = 8192, bits 56,56,56,50, t = 22 bits; ops and objects only, the path uses a reduced DAG).
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <map>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "seal/seal.h"

using namespace seal;
using std::size_t;
using std::string;
using std::uint32_t;
using std::uint64_t;
using std::vector;

namespace {

// ---------------------------------------------------------------------------------------------- JSON by hand
struct Json {
    std::ostringstream o;
    bool first = true;
    void sep() { if (!first) o << ","; first = false; }
    void key(const string &k) { sep(); o << "\"" << k << "\":"; }
    static string hex(uint64_t v) { char b[20]; std::snprintf(b, sizeof b, "\"%llx\"", (unsigned long long)v); return b; }
    void num(const string &k, long long v) { key(k); o << v; }
    void str(const string &k, const string &v) { key(k); o << "\"" << v << "\""; }
    void hexv(const string &k, uint64_t v) { key(k); o << hex(v); }
    void raw(const string &k, const string &v) { key(k); o << v; }
};
string hex_list(const uint64_t *p, size_t n)
{
    string s = "[";
    for (size_t i = 0; i < n; i++) { if (i) s += ","; s += Json::hex(p[i]); }
    return s + "]";
}
// [polys][limbs][n]
string ct_json(const Ciphertext &ct)
{
    const size_t n = ct.poly_modulus_degree(), L = ct.coeff_modulus_size();
    string s = "[";
    for (size_t p = 0; p < ct.size(); p++) {
        if (p) s += ",";
        s += "[";
        for (size_t j = 0; j < L; j++) { if (j) s += ","; s += hex_list(ct.data(p) + j * n, n); }
        s += "]";
    }
    return s + "]";
}
string pt_json(const Plaintext &pt) { return hex_list(pt.data(), pt.coeff_count()); }
// NTT-form plaintext as [limbs][n]
string pt_ntt_json(const Plaintext &pt, size_t n)
{
    string s = "[";
    for (size_t j = 0; j * n < pt.coeff_count(); j++) { if (j) s += ","; s += hex_list(pt.data() + j * n, n); }
    return s + "]";
}
string bytes_hex(const string &b)
{
    static const char *d = "0123456789abcdef";
    string s = "\"";
    for (unsigned char c : b) { s += d[c >> 4]; s += d[c & 15]; }
    return s + "\"";
}
string ints_json(const vector<long long> &v)
{
    string s = "[";
    for (size_t i = 0; i < v.size(); i++) { if (i) s += ","; s += std::to_string(v[i]); }
    return s + "]";
}
void write_file(const string &path, const string &body)
{
    std::ofstream f(path, std::ios::binary);
    f << "{" << body << "}";
    std::printf("wrote %s (%zu bytes)\n", path.c_str(), body.size() + 2);
}

// ---------------------------------------------------------------------------------------------- the chain
struct Chain {
    SEALContext context;
    vector<parms_id_type> level;                 // level[c] = parms_id of chain_index c (data levels), key level last
    size_t n, K;
    int first;                                   // chain_index of the first data level
    explicit Chain(const EncryptionParameters &parms) : context(parms, true, sec_level_type::none)
    {
        n = parms.poly_modulus_degree();
        K = parms.coeff_modulus().size();
        auto cd = context.first_context_data();
        first = static_cast<int>(cd->chain_index());
        level.resize(static_cast<size_t>(first) + 1);
        for (; cd; cd = cd->next_context_data()) level[cd->chain_index()] = cd->parms_id();
    }
    vector<uint64_t> primes(int chain_idx) const
    {
        vector<uint64_t> q;
        for (auto &m : context.get_context_data(level[static_cast<size_t>(chain_idx)])->parms().coeff_modulus()) q.push_back(m.value());
        return q;
    }
};

EncryptionParameters make_parms(size_t n, const vector<int> &bits, int plain_bits)
{
    EncryptionParameters parms(scheme_type::bfv);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::Create(n, bits));
    parms.set_plain_modulus(PlainModulus::Batching(n, plain_bits));
    return parms;
}

// a ciphertext of `size` polynomials at a level, every word uniform below its prime (the evaluator never looks at more than
// the metadata; the operands need not be encryptions)
Ciphertext random_ct(const Chain &C, int chain_idx, size_t size, std::mt19937_64 &rng, bool ntt_form = false)
{
    Ciphertext ct;
    ct.resize(C.context, C.level[static_cast<size_t>(chain_idx)], size);
    ct.is_ntt_form() = ntt_form;
    const vector<uint64_t> q = C.primes(chain_idx);
    for (size_t p = 0; p < size; p++)
        for (size_t j = 0; j < q.size(); j++)
            for (size_t k = 0; k < C.n; k++) ct.data(p)[j * C.n + k] = rng() % q[j];
    return ct;
}
Plaintext random_pt(const Chain &C, uint64_t t, std::mt19937_64 &rng)
{
    Plaintext pt(C.n);
    for (size_t k = 0; k < C.n; k++) pt[k] = rng() % t;
    return pt;
}
// RelinKeys holding given words: [decomp K-1][2][K][n], NTT form, key level
RelinKeys random_relin_keys(const Chain &C, std::mt19937_64 &rng)
{
    RelinKeys rk;
    const parms_id_type key_id = C.context.key_parms_id();
    rk.parms_id() = key_id;
    rk.data().resize(1);
    rk.data()[0].resize(C.K - 1);
    vector<uint64_t> q;
    for (auto &m : C.context.key_context_data()->parms().coeff_modulus()) q.push_back(m.value());
    for (size_t d = 0; d + 1 < C.K; d++) {
        Ciphertext &ct = rk.data()[0][d].data();
        ct.resize(C.context, key_id, 2);
        ct.is_ntt_form() = true;
        for (size_t p = 0; p < 2; p++)
            for (size_t j = 0; j < C.K; j++)
                for (size_t k = 0; k < C.n; k++) ct.data(p)[j * C.n + k] = rng() % q[j];
    }
    return rk;
}
string rk_json(const RelinKeys &rk)
{
    string s = "[";
    bool f = true;
    for (auto &pk : rk.data()[0]) { if (!f) s += ","; f = false; s += ct_json(pk.data()); }
    return s + "]";
}

// ---------------------------------------------------------------------------------------------- seal_ops_<tag>.json
void emit_ops(const string &dir, const string &tag, size_t n, const vector<int> &bits, int plain_bits)
{
    Chain C(make_parms(n, bits, plain_bits));
    Evaluator ev(C.context);
    std::mt19937_64 rng(0x41505355);
    const uint64_t t = C.context.first_context_data()->parms().plain_modulus().value();
    Json J;
    J.num("n", (long long)n);
    { vector<uint64_t> q; for (auto &m : C.context.key_context_data()->parms().coeff_modulus()) q.push_back(m.value()); J.raw("coeff_modulus", hex_list(q.data(), q.size())); }
    J.hexv("plain_modulus", t);
    { vector<long long> b(bits.begin(), bits.end()); J.raw("coeff_bits", ints_json(b)); }
    J.num("plain_bits", plain_bits);
    { vector<uint64_t> psi; for (size_t j = 0; j < C.K; j++) psi.push_back(C.context.key_context_data()->small_ntt_tables()[j].get_root()); J.raw("psi", hex_list(psi.data(), psi.size())); }
    J.str("seal_version", std::to_string(SEAL_VERSION_MAJOR) + "." + std::to_string(SEAL_VERSION_MINOR) + "." + std::to_string(SEAL_VERSION_PATCH));
    RelinKeys rk = random_relin_keys(C, rng);
    J.raw("rk", rk_json(rk));
    string levels = "[";
    for (int lvl = C.first; lvl >= 0; lvl--) {
        Json c;
        c.num("chain_idx", lvl);
        Ciphertext ct = random_ct(C, lvl, 2, rng), ct2 = random_ct(C, lvl, 2, rng), ct3 = random_ct(C, lvl, 3, rng);
        Plaintext pt = random_pt(C, t, rng), mono(C.n);
        mono.set_zero();
        mono[5] = t - 3;
        c.raw("ct", ct_json(ct)); c.raw("ct2", ct_json(ct2)); c.raw("ct3", ct_json(ct3)); c.raw("pt", pt_json(pt)); c.raw("mono", pt_json(mono));
        Ciphertext ntt; ev.transform_to_ntt(ct, ntt);
        c.raw("ntt", ct_json(ntt));
        Plaintext ptn; ev.transform_to_ntt(pt, C.level[static_cast<size_t>(lvl)], ptn);
        c.raw("pt_ntt", pt_ntt_json(ptn, C.n));
        Ciphertext r;
        ev.multiply_plain(ntt, ptn, r); c.raw("multiply_plain_ntt", ct_json(r));
        ev.multiply_plain(ct, pt, r); c.raw("multiply_plain", ct_json(r));
        ev.multiply_plain(ct, mono, r); c.raw("multiply_plain_mono", ct_json(r));
        ev.add(ct, ct2, r); c.raw("add", ct_json(r));
        ev.add_plain(ct, pt, r); c.raw("add_plain", ct_json(r));
        if (lvl > 0) { ev.mod_switch_to_next(ct, r); c.raw("mod_switch", ct_json(r)); }
        ev.multiply(ct, ct2, r); c.raw("multiply", ct_json(r));
        ev.square(ct, r); c.raw("square", ct_json(r));
        if (C.context.using_keyswitching()) { ev.relinearize(ct3, rk, r); c.raw("relinearize", ct_json(r)); }
        levels += (lvl == C.first ? "{" : ",{") + c.o.str() + "}";
    }
    J.raw("levels", levels + "]");
    write_file(dir + "/seal_ops_" + tag + ".json", J.o.str());
}

// ---------------------------------------------------------------------------------------------- the reference's call order
// PowersDag::configure (common/apsu/powers.cpp:22-107): parents of every target chosen to minimise depth, first minimal pair in
// ascending order of the smaller parent.  nodes: power -> (depth, parent1, parent2); sources have parents (0, 0).
struct Node { uint32_t depth, p1, p2; };
std::map<uint32_t, Node> powers_dag(const std::set<uint32_t> &sources, const std::set<uint32_t> &targets)
{
    std::map<uint32_t, Node> nodes;
    for (uint32_t s : sources) nodes[s] = Node{ 0, 0, 0 };
    for (uint32_t curr : targets) {
        if (nodes.count(curr)) continue;
        uint32_t best = ~0u, b1 = 0, b2 = 0;
        for (uint32_t s1 : targets) {
            if (s1 > curr / 2 || s1 >= curr) break;
            const uint32_t s2 = curr - s1;
            if (!nodes.count(s1) || !nodes.count(s2)) continue;
            const uint32_t d = std::max(nodes[s1].depth, nodes[s2].depth) + 1;
            if (d < best) { best = d; b1 = s1; b2 = s2; }
        }
        nodes[curr] = Node{ best, b1, b2 };
    }
    return nodes;
}
// create_powers_set (common/apsu/util/utils.cpp:146-177)
std::set<uint32_t> create_powers_set(uint32_t ps_low_degree, uint32_t target_degree)
{
    std::set<uint32_t> s;
    if (ps_low_degree <= 1 || ps_low_degree >= target_degree) { for (uint32_t i = 1; i <= target_degree; i++) s.insert(i); return s; }
    for (uint32_t i = 1; i <= ps_low_degree; i++) s.insert(i);
    for (uint32_t i = ps_low_degree + 1; i <= target_degree; i += ps_low_degree + 1) s.insert(i);
    return s;
}
// get_parms_id_for_chain_idx (util/utils.cpp:179-189): walk down from the first data level, stop at chain_idx or at the last level
parms_id_type parms_id_for_chain_idx(const SEALContext &context, size_t chain_idx)
{
    auto cd = context.first_context_data();
    while (cd->chain_index() > chain_idx && cd->next_context_data()) cd = cd->next_context_data();
    return cd->parms_id();
}
// try_clear_irrelevant_bits (receiver/apsu/bin_bundle.cpp:67-97): one prime left -> clear the low bits no decryption can see
void clear_irrelevant_bits(const EncryptionParameters &parms, Ciphertext &ct)
{
    if (parms.coeff_modulus().size() != 1) return;
    // bits kept = bit_count(t) + significant bits of n (log2 n + 1) - 1
    int n_bits = 0;
    for (size_t v = parms.poly_modulus_degree(); v; v >>= 1) n_bits++;
    const int irrelevant = parms.coeff_modulus()[0].bit_count() - (parms.plain_modulus().bit_count() + n_bits - 1);
    if (irrelevant <= 0) return;
    const uint64_t mask = ~((uint64_t(1) << irrelevant) - 1);
    for (size_t p = 0; p < ct.size(); p++)
        for (size_t k = 0; k < ct.poly_modulus_degree(); k++) ct.data(p)[k] &= mask;
}

// BatchedPlaintextPolyn::eval (bin_bundle.cpp:106-174)
Ciphertext ref_eval(const Chain &C, Evaluator &ev, const vector<Ciphertext> &powers, const vector<Plaintext> &coeffs, Plaintext mask)
{
    const parms_id_type encode_id = powers[1].parms_id();
    Ciphertext result, temp;
    result.resize(C.context, encode_id, 2);
    result.is_ntt_form() = true;
    for (size_t p = 0; p < 2; p++) for (size_t k = 0; k < result.coeff_modulus_size() * C.n; k++) result.data(p)[k] = 0;
    for (size_t deg = 1; deg < coeffs.size(); deg++) {
        ev.multiply_plain(powers[deg], coeffs[deg], temp);
        ev.add_inplace(result, temp);
    }
    ev.transform_from_ntt_inplace(result);
    ev.add_plain_inplace(result, coeffs[0]);
    ev.add_plain_inplace(result, mask);
    while (result.parms_id() != C.context.last_parms_id()) ev.mod_switch_to_next_inplace(result);
    clear_irrelevant_bits(C.context.last_context_data()->parms(), result);
    return result;
}
// BatchedPlaintextPolyn::eval_patstock (bin_bundle.cpp:192-360)
Ciphertext ref_eval_patstock(const Chain &C, Evaluator &ev, const RelinKeys &rk, const vector<Ciphertext> &powers, const vector<Plaintext> &coeffs,
                             size_t ps_low_degree, Plaintext mask)
{
    const size_t degree = coeffs.size() - 1;
    const size_t ps_high_degree = ps_low_degree + 1;
    const size_t ps_high_degree_powers = degree / ps_high_degree;
    const bool relinearize = C.context.using_keyswitching();
    const parms_id_type high_id = powers[ps_high_degree].parms_id();
    Ciphertext result, temp, temp_in;
    result.resize(C.context, high_id, 3);
    result.is_ntt_form() = false;
    for (size_t p = 0; p < 3; p++) for (size_t k = 0; k < result.coeff_modulus_size() * C.n; k++) result.data(p)[k] = 0;
    auto zero_ntt = [&](Ciphertext &c, const parms_id_type &id) {
        c.resize(C.context, id, 2);
        c.is_ntt_form() = true;
        for (size_t p = 0; p < 2; p++) for (size_t k = 0; k < c.coeff_modulus_size() * C.n; k++) c.data(p)[k] = 0;
    };
    for (size_t i = 1; i < ps_high_degree_powers; i++) {                                    // :232-275
        zero_ntt(temp_in, powers[1].parms_id());
        for (size_t j = 1; j < ps_high_degree; j++) {
            ev.multiply_plain(powers[j], coeffs[i * ps_high_degree + j], temp);
            ev.add_inplace(temp_in, temp);
        }
        ev.transform_from_ntt_inplace(temp_in);
        ev.mod_switch_to_inplace(temp_in, high_id);
        ev.multiply_inplace(temp_in, powers[i * ps_high_degree]);
        ev.add_inplace(result, temp_in);
    }
    if (degree % ps_high_degree > 0 && ps_high_degree_powers > 0) {                           // :277-305
        zero_ntt(temp_in, powers[1].parms_id());
        for (size_t j = 1; j <= degree % ps_high_degree; j++) {
            ev.multiply_plain(powers[j], coeffs[ps_high_degree_powers * ps_high_degree + j], temp);
            ev.add_inplace(temp_in, temp);
        }
        ev.transform_from_ntt_inplace(temp_in);
        ev.mod_switch_to_inplace(temp_in, high_id);
        ev.multiply_inplace(temp_in, powers[ps_high_degree_powers * ps_high_degree]);
        ev.add_inplace(result, temp_in);
    }
    if (relinearize) ev.relinearize_inplace(result, rk);                                      // :308-310
    for (size_t j = 1; j < ps_high_degree; j++) {                                             // :313-325
        ev.multiply_plain(powers[j], coeffs[j], temp);
        ev.transform_from_ntt_inplace(temp);
        ev.mod_switch_to_inplace(temp, high_id);
        ev.add_inplace(result, temp);
    }
    for (size_t i = 1; i < ps_high_degree_powers + 1; i++) {                                  // :327-338
        ev.multiply_plain(powers[i * ps_high_degree], coeffs[i * ps_high_degree], temp);
        ev.mod_switch_to_inplace(temp, high_id);
        ev.add_inplace(result, temp);
    }
    ev.add_plain_inplace(result, coeffs[0]);                                                  // :340-346
    ev.add_plain_inplace(result, mask);
    while (result.parms_id() != C.context.last_parms_id()) ev.mod_switch_to_next_inplace(result);
    clear_irrelevant_bits(C.context.last_context_data()->parms(), result);
    return result;
}

// ---------------------------------------------------------------------------------------------- seal_path_<tag>.json
void emit_path(const string &dir, const string &tag, size_t n, const vector<int> &bits, int plain_bits, uint32_t ps_low, uint32_t max_items,
               const vector<uint32_t> &query_powers, const vector<uint32_t> &degrees)
{
    Chain C(make_parms(n, bits, plain_bits));
    Evaluator ev(C.context);
    KeyGenerator keygen(C.context);
    const SecretKey sk = keygen.secret_key();
    RelinKeys rk;
    if (C.context.using_keyswitching()) keygen.create_relin_keys(rk);
    Encryptor enc(C.context, sk);
    Decryptor dec(C.context, sk);
    BatchEncoder be(C.context);
    std::mt19937_64 rng(0x41505356);
    const uint64_t t = C.context.first_context_data()->parms().plain_modulus().value();
    auto mulmod = [&](uint64_t a, uint64_t b) { return static_cast<uint64_t>((unsigned __int128)a * b % t); };
    auto powmod = [&](uint64_t b, uint64_t e) { uint64_t r = 1; for (; e; e >>= 1, b = mulmod(b, b)) if (e & 1) r = mulmod(r, b); return r; };

    const std::set<uint32_t> targets = create_powers_set(ps_low, max_items);
    const std::set<uint32_t> sources(query_powers.begin(), query_powers.end());
    const std::map<uint32_t, Node> dag = powers_dag(sources, targets);
    uint32_t depth = 0;
    for (auto &kv : dag) depth = std::max(depth, kv.second.depth);

    vector<uint64_t> x(n);
    for (auto &v : x) v = rng() % t;
    vector<Ciphertext> powers(static_cast<size_t>(max_items) + 1);
    Json J;
    J.num("n", (long long)n);
    { vector<long long> b(bits.begin(), bits.end()); J.raw("coeff_bits", ints_json(b)); }
    { vector<uint64_t> q; for (auto &m : C.context.key_context_data()->parms().coeff_modulus()) q.push_back(m.value()); J.raw("coeff_modulus", hex_list(q.data(), q.size())); }
    J.hexv("plain_modulus", t);
    J.str("seal_version", std::to_string(SEAL_VERSION_MAJOR) + "." + std::to_string(SEAL_VERSION_MINOR) + "." + std::to_string(SEAL_VERSION_PATCH));
    J.num("plain_bits", plain_bits); J.num("ps_low_degree", ps_low); J.num("max_items_per_bin", max_items);
    { vector<long long> v(query_powers.begin(), query_powers.end()); J.raw("query_powers", ints_json(v)); }
    { vector<long long> v(targets.begin(), targets.end()); J.raw("targets", ints_json(v)); }
    J.num("dag_depth", depth);
    { string s = "["; bool f = true; for (auto &kv : dag) { if (!f) s += ","; f = false; s += ints_json({ (long long)kv.first, (long long)kv.second.depth, (long long)kv.second.p1, (long long)kv.second.p2 }); } J.raw("dag_nodes", s + "]"); }
    // the secret key as SEAL holds it: NTT form over all key-level primes, [K][n]
    J.raw("secret_ntt", pt_ntt_json(sk.data(), n));
    J.raw("x", hex_list(x.data(), n));
    if (C.context.using_keyswitching()) J.raw("rk", rk_json(rk));
    string src_json = "{";
    for (uint32_t e : query_powers) {                                                         // the querier: plaintext_powers.cpp:41-46
        vector<uint64_t> slots(n);
        for (size_t k = 0; k < n; k++) slots[k] = powmod(x[k], e);
        Plaintext pt;
        be.encode(slots, pt);
        enc.encrypt_symmetric(pt, powers[e]);
        src_json += (src_json.size() > 1 ? ",\"" : "\"") + std::to_string(e) + "\":" + ct_json(powers[e]);
    }
    J.raw("sources", src_json + "}");
    // Receiver::ComputePowers (receiver_osn.cpp:412-487), nodes in ascending depth
    for (uint32_t d = 1; d <= depth; d++)
        for (auto &kv : dag) {
            if (kv.second.depth != d) continue;
            Ciphertext prod;
            if (kv.second.p1 == kv.second.p2) ev.square(powers[kv.second.p1], prod);
            else ev.multiply(powers[kv.second.p1], powers[kv.second.p2], prod);
            if (C.context.using_keyswitching()) ev.relinearize_inplace(prod, rk);
            powers[kv.first] = prod;
        }
    const parms_id_type high_id = parms_id_for_chain_idx(C.context, 1), low_id = parms_id_for_chain_idx(C.context, 2);
    string pw_json = "{";
    for (uint32_t p : targets) {
        if (!ps_low) { ev.mod_switch_to_inplace(powers[p], high_id); ev.transform_to_ntt_inplace(powers[p]); }
        else if (p <= ps_low) { ev.mod_switch_to_inplace(powers[p], low_id); ev.transform_to_ntt_inplace(powers[p]); }
        else ev.mod_switch_to_inplace(powers[p], high_id);
        pw_json += (pw_json.size() > 1 ? ",\"" : "\"") + std::to_string(p) + "\":" + ct_json(powers[p]);
    }
    J.raw("powers", pw_json + "}");
    // BinBundles: BatchedPlaintextPolyn's constructor rule (bin_bundle.cpp:366-430): with Paterson-Stockmeyer the coefficients whose
    // degree is a multiple of ps_low_degree + 1 stay in coefficient form, all others go to NTT form at the LOW powers' level
    string bundles = "[";
    for (size_t bi = 0; bi < degrees.size(); bi++) {
        const uint32_t degree = degrees[bi];
        vector<vector<uint64_t>> A(degree + 1, vector<uint64_t>(n));
        for (auto &row : A) for (auto &v : row) v = rng() % t;
        for (auto &v : A[degree]) v = 1;
        vector<Plaintext> coeffs(degree + 1);
        vector<long long> flags;
        string cj = "[";
        const bool using_ps = ps_low > 1 && ps_low < degree;
        for (uint32_t dgr = 0; dgr <= degree; dgr++) {
            be.encode(A[dgr], coeffs[dgr]);
            const bool is_ntt = using_ps ? (dgr % (ps_low + 1) != 0) : (dgr != 0);
            if (is_ntt) ev.transform_to_ntt_inplace(coeffs[dgr], powers[1].parms_id());
            flags.push_back(is_ntt ? 1 : 0);
            cj += (dgr ? "," : "") + (is_ntt ? pt_ntt_json(coeffs[dgr], n) : pt_json(coeffs[dgr]));
        }
        vector<uint64_t> mask_vals(n);
        for (auto &v : mask_vals) v = rng() % t;
        Plaintext mask;
        be.encode(mask_vals, mask);
        Ciphertext res = using_ps ? ref_eval_patstock(C, ev, rk, powers, coeffs, ps_low, mask) : ref_eval(C, ev, powers, coeffs, mask);
        Plaintext out;
        dec.decrypt(res, out);
        vector<uint64_t> slots;
        be.decode(out, slots);
        vector<uint64_t> expect(n);
        for (size_t k = 0; k < n; k++) {
            uint64_t acc = 0;
            for (uint32_t dgr = degree + 1; dgr-- > 0;) acc = (mulmod(acc, x[k]) + A[dgr][k]) % t;
            expect[k] = (acc + mask_vals[k]) % t;
        }
        if (slots != expect) std::printf("WARNING: decrypt(eval) != P(x) + mask for BinBundle of degree %u (noise budget %d)\n", degree, dec.invariant_noise_budget(res));
        Json b;
        b.num("degree", degree);
        b.raw("coeffs", cj + "]");
        { string f = "["; for (size_t i = 0; i < flags.size(); i++) f += (i ? "," : "") + string(flags[i] ? "true" : "false"); b.raw("is_ntt", f + "]"); }
        b.raw("mask", pt_json(mask));
        b.raw("expected_slots", hex_list(expect.data(), n));
        b.raw("result", ct_json(res));
        b.num("noise_budget", dec.invariant_noise_budget(res));
        bundles += (bi ? ",{" : "{") + b.o.str() + "}";
    }
    J.raw("bundles", bundles + "]");
    write_file(dir + "/seal_path_" + tag + ".json", J.o.str());
}

// ---------------------------------------------------------------------------------------------- seal_objects_<tag>.json
template <class T> string saved(const T &obj, compr_mode_type mode)
{
    std::stringstream ss(std::ios::in | std::ios::out | std::ios::binary);
    obj.save(ss, mode);
    return ss.str();
}
void emit_objects(const string &dir, const string &tag, size_t n, const vector<int> &bits, int plain_bits)
{
    Chain C(make_parms(n, bits, plain_bits));
    Evaluator ev(C.context);
    KeyGenerator keygen(C.context);
    const SecretKey sk = keygen.secret_key();
    Encryptor enc(C.context, sk);
    BatchEncoder be(C.context);
    std::mt19937_64 rng(0x41505357);
    const uint64_t t = C.context.first_context_data()->parms().plain_modulus().value();
    Json J;
    J.num("n", (long long)n);
    { vector<uint64_t> q; for (auto &m : C.context.key_context_data()->parms().coeff_modulus()) q.push_back(m.value()); J.raw("coeff_modulus", hex_list(q.data(), q.size())); }
    J.hexv("plain_modulus", t);
    J.str("seal_version", std::to_string(SEAL_VERSION_MAJOR) + "." + std::to_string(SEAL_VERSION_MINOR) + "." + std::to_string(SEAL_VERSION_PATCH));
    {   // parms_id: key level, then every data level by chain_index
        const parms_id_type kid = C.context.key_parms_id();
        J.raw("key_parms_id", hex_list(kid.data(), 4));
        string s = "[";
        for (size_t c = 0; c < C.level.size(); c++) s += (c ? "," : "") + hex_list(C.level[c].data(), 4);
        J.raw("parms_id_by_chain_idx", s + "]");
    }
    vector<compr_mode_type> modes = { compr_mode_type::none };
#ifdef SEAL_USE_ZLIB
    modes.push_back(compr_mode_type::zlib);
#endif
#ifdef SEAL_USE_ZSTD
    modes.push_back(compr_mode_type::zstd);
#endif
    string cts = "[", rks = "[", pts = "[";
    for (compr_mode_type mode : modes) {
        // the querier's objects (sender/apsu/plaintext_powers.cpp:41-46, sender_osn.cpp:223-227): Serializable<> = seeded
        vector<uint64_t> slots(n);
        for (auto &v : slots) v = rng() % t;
        Plaintext pt;
        be.encode(slots, pt);
        const string blob = saved(enc.encrypt_symmetric(pt), mode);
        Ciphertext back;
        { std::stringstream ss(blob, std::ios::in | std::ios::binary); back.load(C.context, ss); }
        Json c;
        c.num("compr", (long long)static_cast<int>(mode)); c.num("chain_idx", C.first); c.raw("seeded", "true");
        c.raw("blob", bytes_hex(blob)); c.raw("data", ct_json(back));
        cts += (cts.size() > 1 ? ",{" : "{") + c.o.str() + "}";
        // the same ciphertext saved expanded
        const string blob2 = saved(back, mode);
        Json c2;
        c2.num("compr", (long long)static_cast<int>(mode)); c2.num("chain_idx", C.first); c2.raw("seeded", "false");
        c2.raw("blob", bytes_hex(blob2)); c2.raw("data", ct_json(back));
        cts += ",{" + c2.o.str() + "}";
        if (C.context.using_keyswitching()) {
            const string kb = saved(keygen.create_relin_keys(), mode);
            RelinKeys kback;
            { std::stringstream ss(kb, std::ios::in | std::ios::binary); kback.load(C.context, ss); }
            Json k;
            k.num("compr", (long long)static_cast<int>(mode)); k.raw("seeded", "true"); k.raw("blob", bytes_hex(kb)); k.raw("data", rk_json(kback));
            rks += (rks.size() > 1 ? ",{" : "{") + k.o.str() + "}";
            const string kb2 = saved(kback, mode);
            Json k2;
            k2.num("compr", (long long)static_cast<int>(mode)); k2.raw("seeded", "false"); k2.raw("blob", bytes_hex(kb2)); k2.raw("data", rk_json(kback));
            rks += ",{" + k2.o.str() + "}";
        }
        // plaintexts as BatchedPlaintextPolyn stores them (bin_bundle.cpp:421-428): coefficient form and NTT form at the low level
        Json p;
        p.num("compr", (long long)static_cast<int>(mode)); p.num("chain_idx", -1); p.raw("blob", bytes_hex(saved(pt, mode))); p.raw("data", pt_json(pt));
        pts += (pts.size() > 1 ? ",{" : "{") + p.o.str() + "}";
        const int low = C.first < 2 ? C.first : 2;
        Plaintext ptn;
        ev.transform_to_ntt(pt, C.level[static_cast<size_t>(low)], ptn);
        Json p2;
        p2.num("compr", (long long)static_cast<int>(mode)); p2.num("chain_idx", low); p2.raw("blob", bytes_hex(saved(ptn, mode))); p2.raw("data", pt_ntt_json(ptn, n));
        pts += ",{" + p2.o.str() + "}";
    }
    J.raw("ciphertexts", cts + "]");
    J.raw("relin_keys", rks + "]");
    J.raw("plaintexts", pts + "]");
    write_file(dir + "/seal_objects_" + tag + ".json", J.o.str());
}

} // namespace

int main(int argc, char **argv)
{
    const string dir = argc > 1 ? argv[1] : ".";
    // the toy chain of tests/golden/*.json
    emit_ops(dir, "n64", 64, { 40, 40, 40, 36 }, 17);
    emit_path(dir, "n64", 64, { 40, 40, 40, 36 }, 17, 3, 11, { 1, 4 }, { 10, 8, 3, 11 });
    emit_objects(dir, "n64", 64, { 40, 40, 40, 36 }, 17);
    // no Paterson-Stockmeyer, one coefficient prime (100K-1's shape: nothing is relinearised, every power is sent)
    emit_path(dir, "n64_single", 64, { 48 }, 17, 0, 6, { 1, 2, 3, 4, 5, 6 }, { 6, 1 });
    // 16M-4096.json's encryption parameters (parameters/16M-4096.json): primes, parms_ids, one set of operations per level, objects
    emit_ops(dir, "16M", 8192, { 56, 56, 56, 50 }, 22);
    emit_objects(dir, "16M", 8192, { 56, 56, 56, 50 }, 22);
    // ... and its path on a reduced workload (same chain, same ps_low_degree rule, a two-level PowersDag)
    emit_path(dir, "16M_small", 8192, { 56, 56, 56, 50 }, 22, 3, 11, { 1, 4 }, { 11, 7 });
    return 0;
}
