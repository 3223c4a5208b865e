// CPU tier, sanitizer build (AddressSanitizer + UndefinedBehaviorSanitizer) of the product's HOST code that parses untrusted or
// structured input: the N3 framing reader (wire.cpp), the PSUParams JSON reader + constant derivation (params.cpp), the
// PowersDag (powers_dag.cpp) and the partition rule (sharding.cpp).  Built and run by tests/test_host_sanitizers.py; any
// out-of-bounds read in the verifier, signed overflow or misaligned access aborts the run.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../apsu_amd/csrc/params.h"
#include "../../apsu_amd/csrc/powers_dag.h"
#include "../../apsu_amd/csrc/sharding.h"
#include "../../apsu_amd/csrc/wire.h"
#include "../../apsu_amd/csrc/seal_codec.h"

using namespace apsu_he;

static uint64_t rng_state = 0x9e3779b97f4a7c15ULL;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}

template <class F> static int fuzz(const std::vector<uint8_t> &good, F &&parse, int rounds)
{
    int rejected = 0;
    for (int r = 0; r < rounds; r++) {
        // heap copy of EXACT size: a read past the end is an ASan error
        const size_t cut = (r % 7 == 0) ? rnd() % (good.size() + 1) : good.size();
        uint8_t *m = static_cast<uint8_t *>(std::malloc(cut ? cut : 1));
        std::memcpy(m, good.data(), cut);
        const int flips = 1 + (int)(rnd() % 4);
        for (int f = 0; f < flips && cut; f++) m[rnd() % cut] = (uint8_t)rnd();
        try { parse(m, cut); } catch (const std::runtime_error &) { rejected++; }
        std::free(m);
    }
    return rejected;
}

int main(int argc, char **argv)
{
    // ---- wire framing
    std::vector<uint8_t> blob(3000);
    for (auto &b : blob) b = (uint8_t)rnd();
    wire::QueryRequest q;
    q.compression_type = 2; q.has_relin_keys = true; q.relin_keys = wire::Span{ blob.data(), 1000 };
    for (uint32_t e : { 1u, 3u, 11u, 18u, 45u, 225u }) {
        wire::QueryPart p; p.exponent = e;
        for (int b = 0; b < 4; b++) p.cts.push_back(wire::Span{ blob.data() + 17 * b, (size_t)(40 + e % 50) });
        q.parts.push_back(p);
    }
    const std::vector<uint8_t> qbuf = wire::build_query_request(q);
    const wire::QueryRequest back = wire::parse_query_request(qbuf.data(), qbuf.size());
    if (back.parts.size() != 6 || back.parts[5].exponent != 225 || back.relin_keys.n != 1000) return 10;
    wire::ResultPackage rp;
    rp.bundle_idx = 2; rp.cache_idx = 5; rp.psu_result = wire::Span{ blob.data(), 2048 }; rp.label_byte_count = 16; rp.nonce_byte_count = 8;
    rp.label_result = { wire::Span{ blob.data(), 10 }, wire::Span{ blob.data() + 5, 0 } };
    const std::vector<uint8_t> rbuf = wire::build_result_package(rp);
    if (wire::parse_result_package(rbuf.data(), rbuf.size()).label_result.size() != 2) return 11;
    const std::vector<uint8_t> hbuf = wire::build_header(wire::Header{ 7, 3 }), sbuf = wire::build_query_response(wire::QueryResponse{ 28, 7 });
    int rej = 0;
    rej += fuzz(qbuf, [](const uint8_t *p, size_t n) { (void)wire::parse_query_request(p, n); }, 40000);
    rej += fuzz(rbuf, [](const uint8_t *p, size_t n) { (void)wire::parse_result_package(p, n); }, 40000);
    rej += fuzz(hbuf, [](const uint8_t *p, size_t n) { (void)wire::parse_header(p, n); }, 5000);
    rej += fuzz(sbuf, [](const uint8_t *p, size_t n) { (void)wire::parse_query_response(p, n); }, 5000);
    {   // a small valid buffer whose parts share ONE ciphertext vector: quadratic visits, refused by the reader's work budget
        wire::QueryRequest sq;
        std::vector<wire::Span> many(2000, wire::Span{ blob.data(), 8 });
        for (uint32_t e = 1; e <= 4; e++) { wire::QueryPart p; p.exponent = e; p.cts = many; sq.parts.push_back(p); }
        std::vector<uint8_t> sb = wire::build_query_request(sq);
        (void)wire::parse_query_request(sb.data(), sb.size());         // unshared: fine
        // re-point every part's cts offset at part 0's vector is what an attacker would do; here the cheaper equivalent:
        // the budget must hold for ANY buffer, so the fuzz loop below covers offsets that alias by chance
        rej += fuzz(sb, [](const uint8_t *p, size_t n) { (void)wire::parse_query_request(p, n); }, 300);
    }
    // ---- SEAL object codec (seal_codec.cpp): plain, seeded and zlib ciphertexts, RelinKeys
    {
        const std::vector<uint64_t> kq = { 0xffffffffc001ULL, 0xffffee001ULL, 0x1ffc001ULL };
        const std::vector<sealio::Level> chain = sealio::modulus_chain(64, kq, 65537);
        sealio::Ciphertext ct;
        std::memcpy(ct.parms_id, chain[1].parms_id, 32);
        ct.size = 2; ct.coeff_modulus_size = 2; ct.poly_modulus_degree = 64;
        ct.data.resize(2 * 2 * 64);
        for (size_t i = 0; i < ct.data.size(); i++) ct.data[i] = rnd() % kq[(i / 64) % 2];
        for (uint8_t compr : { sealio::COMPR_NONE, sealio::COMPR_ZLIB, sealio::COMPR_ZSTD }) {
            const std::vector<uint8_t> eb = sealio::save_ciphertext(ct, compr);
            if (sealio::load_ciphertext(eb.data(), eb.size(), chain).data != ct.data) return 12;
            rej += fuzz(eb, [&](const uint8_t *p, size_t n) { (void)sealio::load_ciphertext(p, n, chain); }, 8000);
            sealio::Ciphertext sc = ct;
            sc.seeded = true;
            for (int i = 0; i < 8; i++) sc.seed[i] = rnd();
            sealio::sample_poly_uniform(sc.seed, chain[1].q.data(), 2, 64, sc.data.data() + 128);
            const std::vector<uint8_t> sbuf2 = sealio::save_ciphertext(sc, compr);
            if (sealio::load_ciphertext(sbuf2.data(), sbuf2.size(), chain).data != sc.data) return 13;
            rej += fuzz(sbuf2, [&](const uint8_t *p, size_t n) { (void)sealio::load_ciphertext(p, n, chain); }, 8000);
        }
        {   // a seeded ciphertext under SEAL's Shake256 generator: expanded by the host codec in both load modes
            sealio::Ciphertext sh = ct;
            sh.seeded = true; sh.prng_type = sealio::PRNG_SHAKE256;
            for (int i = 0; i < 8; i++) sh.seed[i] = rnd();
            sealio::sample_poly_uniform(sh.seed, chain[1].q.data(), 2, 64, sh.data.data() + 128, sealio::PRNG_SHAKE256);
            const std::vector<uint8_t> hb = sealio::save_ciphertext(sh, sealio::COMPR_ZLIB);
            if (sealio::load_ciphertext(hb.data(), hb.size(), chain).data != sh.data) return 21;
            const sealio::Ciphertext un = sealio::load_ciphertext(hb.data(), hb.size(), chain, nullptr, false);
            if (un.seeded || un.data != sh.data) return 22;
            rej += fuzz(hb, [&](const uint8_t *p, size_t n) { (void)sealio::load_ciphertext(p, n, chain); }, 2000);
        }
        {   // forged dimensions: (a) a tiny object that claims 64 x 64 x 2^20 words must allocate nothing; (b) a level's parms_id with
            // another coeff_modulus_size is refused with and without seed expansion
            sealio::Ciphertext big = ct;
            big.size = 64; big.coeff_modulus_size = 64; big.poly_modulus_degree = 1u << 20;
            std::vector<uint8_t> eb = sealio::save_ciphertext(ct, sealio::COMPR_NONE);
            auto put64 = [&](size_t at, uint64_t v) { std::memcpy(eb.data() + at, &v, 8); };
            put64(16 + 33, 64); put64(16 + 41, (uint64_t)1 << 20); put64(16 + 49, 64);
            bool refused = false;
            try { (void)sealio::load_ciphertext(eb.data(), eb.size(), chain); } catch (const std::exception &) { refused = true; }
            if (!refused) return 19;
            sealio::Ciphertext one = ct;
            one.coeff_modulus_size = 1; one.data.resize(2 * 64);
            const std::vector<uint8_t> ob = sealio::save_ciphertext(one, sealio::COMPR_NONE);
            for (bool expand : { true, false }) {
                refused = false;
                try { (void)sealio::load_ciphertext(ob.data(), ob.size(), chain, nullptr, expand); } catch (const std::exception &) { refused = true; }
                if (!refused) return 20;
            }
            rej += 3;
        }
        sealio::KSwitchKeys kk;
        std::memcpy(kk.parms_id, chain[0].parms_id, 32);
        kk.keys.resize(1);
        for (int j = 0; j < 2; j++) {
            sealio::Ciphertext k = ct;
            std::memcpy(k.parms_id, chain[0].parms_id, 32);
            k.is_ntt_form = 1; k.coeff_modulus_size = 3; k.data.assign(2 * 3 * 64, 0);
            for (size_t i = 0; i < k.data.size(); i++) k.data[i] = rnd() % kq[(i / 64) % 3];
            kk.keys[0].push_back(k);
        }
        const std::vector<uint8_t> kb = sealio::save_kswitch_keys(kk, sealio::COMPR_ZLIB);
        if (sealio::relin_keys_layout(sealio::load_kswitch_keys(kb.data(), kb.size(), chain), 3, 64).size() != 2 * 2 * 3 * 64) return 14;
        rej += fuzz(kb, [&](const uint8_t *p, size_t n) { (void)sealio::load_kswitch_keys(p, n, chain); }, 8000);
    }
    // ---- parameter exchange, plainResponse, PSUParams in binary form (with SEAL's EncryptionParameters object inside)
    {
        sealio::EncryptionParameters ep;
        ep.poly_modulus_degree = 64; ep.coeff_modulus = { 1099511590913ull, 1099511592577ull, 68719403009ull }; ep.plain_modulus = 65537;
        const std::vector<uint8_t> eb = sealio::save_encryption_parameters(ep, sealio::COMPR_NONE);
        if (sealio::load_encryption_parameters(eb.data(), eb.size()).coeff_modulus != ep.coeff_modulus) return 16;
        wire::PsuParamsWire pw;
        pw.felts_per_item = 5; pw.table_size = 24; pw.max_items_per_bin = 11; pw.hash_func_count = 3; pw.ps_low_degree = 3; pw.query_powers = { 1, 4 };
        pw.seal_params = wire::Span{ eb.data(), eb.size() };
        const std::vector<uint8_t> pb = wire::build_psu_params(pw);
        const wire::PsuParamsWire pr = wire::parse_psu_params(pb.data(), pb.size());
        if (pr.query_powers != pw.query_powers || pr.table_size != 24 || pr.seal_params.n != eb.size()) return 17;
        wire::PlainResponse pl; pl.bundle_idx = 3; pl.cache_idx = 0; pl.psu_result = { 1, 2, 3, ~0ull };
        const std::vector<uint8_t> plb = wire::build_plain_response(pl), rq = wire::build_parms_request(),
                                   rs = wire::build_parms_response(wire::Span{ pb.data(), pb.size() });
        if (wire::parse_plain_response(plb.data(), plb.size()).psu_result != pl.psu_result || wire::peek_request_type(rq.data(), rq.size()) != 1 ||
            wire::parse_parms_response(rs.data(), rs.size()).n != pb.size()) return 18;
        rej += fuzz(eb, [](const uint8_t *p, size_t n) { (void)sealio::load_encryption_parameters(p, n); }, 20000);
        rej += fuzz(pb, [](const uint8_t *p, size_t n) { (void)wire::parse_psu_params(p, n); }, 20000);
        rej += fuzz(plb, [](const uint8_t *p, size_t n) { (void)wire::parse_plain_response(p, n); }, 10000);
        rej += fuzz(rs, [](const uint8_t *p, size_t n) { (void)wire::parse_parms_response(p, n); (void)wire::peek_response_type(p, n); }, 10000);
    }
    // ---- a saved ReceiverDB's header (receiver_db.fbs; seeds from the test's FlatBuffers model, like the BinBundles below)
    for (int i = 1; i < argc; i++) {
        const std::string name = argv[i];
        if (name.size() < 9 || name.substr(name.size() - 9) != ".dbheader") continue;
        std::ifstream f(name, std::ios::binary);
        std::vector<uint8_t> seed((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (wire::parse_receiver_db_header(seed.data(), seed.size()).bin_bundle_count != 2) return 19;
        rej += fuzz(seed, [](const uint8_t *p, size_t n) { (void)wire::parse_receiver_db_header(p, n); }, 20000);
    }
    // ---- saved BinBundles (bin_bundle.fbs; the library has no writer for them): seeds written by the test's FlatBuffers model
    for (int i = 1; i < argc; i++) {
        const std::string name = argv[i];
        if (name.size() < 10 || name.substr(name.size() - 10) != ".binbundle") continue;
        std::ifstream f(name, std::ios::binary);
        std::vector<uint8_t> seed((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const wire::SavedBinBundle sb = wire::parse_bin_bundle(seed.data(), seed.size());
        if (sb.consumed != seed.size() || sb.item_bins.empty()) return 15;
        rej += fuzz(seed, [](const uint8_t *p, size_t n) { (void)wire::parse_bin_bundle(p, n); }, 30000);
    }
    std::printf("wire: %d malformed buffers rejected, none crashed\n", rej);

    // ---- PSUParams JSON + derived constants + PowersDag for the parameter files given on the command line
    for (int i = 1; i < argc; i++) {
        if (std::string(argv[i]).find(".json") == std::string::npos) continue;
        std::ifstream f(argv[i]);
        std::stringstream ss; ss << f.rdbuf();
        PSUParams p = PSUParams::Load(ss.str());
        HeParams hp = HeParams::FromPSUParams(p);
        PowersDag dag;
        if (!dag.configure(p.query_params.query_powers, create_powers_set(p.query_params.ps_low_degree, p.table_params.max_items_per_bin))) return 20;
        std::printf("%s: n=%zu K=%d t=%llu depth=%u\n", argv[i], hp.n, hp.K, (unsigned long long)hp.t, dag.depth());
        // truncated / garbled JSON must throw, not crash
        const std::string js = ss.str();
        for (int r = 0; r < 300; r++) {
            std::string m = js.substr(0, rnd() % (js.size() + 1));
            if (!m.empty() && r % 2) m[rnd() % m.size()] = (char)rnd();
            try { (void)PSUParams::Load(m); } catch (const std::exception &) {}
        }
    }

    // ---- partition rule
    for (int r = 0; r < 2000; r++) {
        const uint32_t nb = 1 + rnd() % 10;
        std::vector<ShardUnit> u(rnd() % 80);
        for (auto &x : u) x = ShardUnit{ (uint32_t)(rnd() % nb), (uint32_t)(rnd() % 40), (uint32_t)(rnd() % 9000) };
        const int world = 1 + (int)(rnd() % 8);
        const std::vector<int> s = partition_units(u, nb, world);
        for (int v : s) if (v < 0 || v >= world) return 30;
    }
    std::printf("ok\n");
    return 0;
}
