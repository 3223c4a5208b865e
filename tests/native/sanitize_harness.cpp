// CPU tier, sanitizer build (AddressSanitizer + UndefinedBehaviorSanitizer) of the product's HOST code that parses untrusted or
// structured input: the N3 framing reader (wire.cpp), the PSUParams JSON reader + constant derivation (params.cpp), the
// PowersDag (powers_dag.cpp) and the partition rule (sharding.cpp).  Built and run by tests/test_host_sanitizers.py; any
// out-of-bounds read in the verifier, signed overflow or misaligned access aborts the run.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../apsu_amd/csrc/params.h"
#include "../../apsu_amd/csrc/powers_dag.h"
#include "../../apsu_amd/csrc/sharding.h"
#include "../../apsu_amd/csrc/wire.h"

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
    std::vector<uint64_t> words(2 * 3 * 64);
    for (auto &w : words) w = rnd();
    wire::SealCt ct; ct.size = 2; ct.coeff_modulus_size = 3; ct.poly_modulus_degree = 64; ct.data = words.data();
    const std::vector<uint8_t> ebuf = wire::seal_envelope_save(ct, 4, 1);
    rej += fuzz(ebuf, [](const uint8_t *p, size_t n) { uint8_t a, b; (void)wire::seal_envelope_load(p, n, &a, &b); }, 20000);
    std::printf("wire: %d malformed buffers rejected, none crashed\n", rej);

    // ---- PSUParams JSON + derived constants + PowersDag for the parameter files given on the command line
    for (int i = 1; i < argc; i++) {
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
