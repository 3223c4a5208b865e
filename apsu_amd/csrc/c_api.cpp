// extern "C" surface of the engine (include/apsu_he.h).  Exceptions are translated to status
// codes the way the reference's callers expect to see SEAL's exceptions.
#include "../../include/apsu_he.h"

#include <algorithm>
#include <thread>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "db_file.h"
#include "engine.h"
#include "multi.h"
#include "wire.h"
#include "seal_codec.h"

using namespace apsu_he;

#include <condition_variable>
#include <mutex>
#include <unordered_set>

struct apsu_he_powers;
// live_powers: the apsu_he_powers handles whose buffers return to this context's pool when they are freed.
// apsu_he_destroy orphans them (their device buffers are released there), so freeing a handle after its context is
// safe — the Python binding's garbage collector does exactly that.
struct apsu_he_ctx { std::unique_ptr<Engine> eng; std::unordered_set<apsu_he_powers *> live_powers; int recycling = 0; };
struct apsu_he_relin { std::unique_ptr<RelinKeys> rk; };
struct apsu_he_bundle { std::unique_ptr<Bundle> b; };
struct apsu_he_powers { std::unique_ptr<Powers> p; apsu_he_ctx *ctx = nullptr; };
struct apsu_he_multi { std::unique_ptr<MultiEngine> m; };
static std::mutex g_registry_mu;      // guards every ctx::live_powers / ctx::recycling and powers::ctx; never held while an Engine lock is taken
static std::condition_variable g_registry_cv;

static thread_local std::string g_last_error;

// The SEAL objects of a query are independent: decoded (inflate) and encoded (deflate, the expensive direction) on a few host threads,
// the way the reference builds its ResultPackages inside its thread-pool tasks (receiver_osn.cpp:334-364,507-539).
template <class F> static void parallel_for(size_t count, F &&fn)
{
    const size_t hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t T = std::min(count, std::min<size_t>(16, hw));
    if (T <= 1) { for (size_t i = 0; i < count; i++) fn(i); return; }
    std::atomic<size_t> next{ 0 };
    std::exception_ptr err;
    std::mutex mu;
    auto body = [&] {
        for (size_t i; (i = next.fetch_add(1)) < count;) {
            try { fn(i); }
            catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); }
        }
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < T; k++) th.emplace_back(body);
    body();
    for (auto &t : th) t.join();
    if (err) std::rethrow_exception(err);
}

template <class F> static int guarded(F &&fn)
{
    try {
        fn();
        return APSU_HE_OK;
    } catch (const std::invalid_argument &e) { g_last_error = e.what(); return APSU_HE_INVALID_ARGUMENT;
    } catch (const std::logic_error &e) { g_last_error = e.what(); return APSU_HE_LOGIC_ERROR;
    } catch (const std::bad_alloc &e) { g_last_error = e.what(); return APSU_HE_OUT_OF_MEMORY;
    } catch (const HipError &e) {
        g_last_error = e.what();
        if (g_last_error.find("no HIP device") != std::string::npos) return APSU_HE_NO_DEVICE;
        if (g_last_error.find("out of memory") != std::string::npos) return APSU_HE_OUT_OF_MEMORY;
        return APSU_HE_RUNTIME_ERROR;
    } catch (const std::exception &e) { g_last_error = e.what(); return APSU_HE_RUNTIME_ERROR;
    } catch (...) { g_last_error = "unknown error"; return APSU_HE_RUNTIME_ERROR; }
}

#define REQUIRE(cond, msg) do { if (!(cond)) throw std::invalid_argument(msg); } while (0)

extern "C" {

const char *apsu_he_last_error(void) { return g_last_error.c_str(); }
int apsu_he_abi_version(void) { return APSU_HE_ABI_VERSION; }

int apsu_he_create(const char *json, int device, apsu_he_ctx **out)
{
    return guarded([&] {
        REQUIRE(json && out, "null argument");
        PSUParams p = PSUParams::Load(json);
        HeParams hp = HeParams::FromPSUParams(p);
        auto c = new apsu_he_ctx;
        try { c->eng = std::make_unique<Engine>(hp, &p, device); } catch (...) { delete c; throw; }
        *out = c;
    });
}

int apsu_he_create_raw(uint64_t n, const uint64_t *coeff_modulus, int k, uint64_t plain_modulus, int device, apsu_he_ctx **out)
{
    return guarded([&] {
        REQUIRE(coeff_modulus && out && k > 0, "null argument");
        HeParams hp = HeParams::Create((size_t)n, std::vector<u64>(coeff_modulus, coeff_modulus + k), plain_modulus);
        auto c = new apsu_he_ctx;
        try { c->eng = std::make_unique<Engine>(hp, nullptr, device); } catch (...) { delete c; throw; }
        *out = c;
    });
}

int apsu_he_destroy(apsu_he_ctx *ctx)
{
    return guarded([&] {
        if (!ctx) return;
        {
            std::unique_lock<std::mutex> g(g_registry_mu);
            g_registry_cv.wait(g, [&] { return ctx->recycling == 0; });       // a powers_free in flight finishes its recycle first
            for (apsu_he_powers *p : ctx->live_powers) { p->ctx = nullptr; p->p.reset(); }
            ctx->live_powers.clear();
        }
        delete ctx;
    });
}

int apsu_he_get_info(const apsu_he_ctx *ctx, apsu_he_info *out)
{
    return guarded([&] {
        REQUIRE(ctx && out, "null argument");
        const HeParams &hp = ctx->eng->he();
        std::memset(out, 0, sizeof(*out));
        out->poly_modulus_degree = hp.n;
        out->plain_modulus = hp.t;
        out->coeff_modulus_size = hp.K;
        out->first_chain_idx = hp.first_chain_idx;
        out->using_keyswitching = hp.using_keyswitching;
        out->irrelevant_bit_count = hp.irrelevant_bit_count;
        for (int j = 0; j < hp.K && j < 8; j++) out->coeff_modulus[j] = hp.key_q[j];
        if (const PSUParams *p = ctx->eng->psu()) {
            out->ps_low_degree = p->query_params.ps_low_degree;
            out->max_items_per_bin = p->table_params.max_items_per_bin;
            out->bundle_idx_count = p->bundle_idx_count;
            out->items_per_bundle = p->items_per_bundle;
            out->source_power_count = ctx->eng->dag().source_count();
            out->target_power_count = (uint32_t)ctx->eng->dag().target_powers().size();
            out->powers_dag_depth = ctx->eng->dag().depth();
        }
        out->result_polys = ctx->eng->result_polys();
    });
}

int apsu_he_get_powers_dag(const apsu_he_ctx *ctx, apsu_he_dag_node *nodes, int capacity, int *n_nodes)
{
    return guarded([&] {
        REQUIRE(ctx && n_nodes, "null argument");
        const auto &m = ctx->eng->dag().nodes();
        *n_nodes = (int)m.size();
        int i = 0;
        for (auto &kv : m) {
            if (nodes && i < capacity)
                nodes[i] = apsu_he_dag_node{ kv.second.power, kv.second.depth, kv.second.parents.first, kv.second.parents.second };
            i++;
        }
    });
}

// ---- tier 1
int apsu_he_transform_to_ntt(apsu_he_ctx *c, uint64_t *ct, int polys, int ci)
{ return guarded([&] { REQUIRE(c && ct && polys > 0, "null argument"); c->eng->transform_to_ntt(ct, polys, ci); }); }
int apsu_he_transform_from_ntt(apsu_he_ctx *c, uint64_t *ct, int polys, int ci)
{ return guarded([&] { REQUIRE(c && ct && polys > 0, "null argument"); c->eng->transform_from_ntt(ct, polys, ci); }); }
int apsu_he_transform_plain_to_ntt(apsu_he_ctx *c, const uint64_t *pt, size_t cnt, uint64_t *out, int ci)
{ return guarded([&] { REQUIRE(c && pt && out, "null argument"); c->eng->transform_plain_to_ntt(pt, cnt, out, ci); }); }
int apsu_he_multiply_plain_ntt(apsu_he_ctx *c, const uint64_t *ct, const uint64_t *pt, uint64_t *out, int polys, int ci)
{ return guarded([&] { REQUIRE(c && ct && pt && out && polys > 0, "null argument"); c->eng->multiply_plain_ntt(ct, pt, out, polys, ci); }); }
int apsu_he_multiply_plain(apsu_he_ctx *c, const uint64_t *ct, const uint64_t *pt, size_t cnt, uint64_t *out, int polys, int ci)
{ return guarded([&] { REQUIRE(c && ct && pt && out && polys > 0, "null argument"); c->eng->multiply_plain(ct, pt, cnt, out, polys, ci); }); }
int apsu_he_add(apsu_he_ctx *c, uint64_t *acc, const uint64_t *x, int polys, int ci)
{ return guarded([&] { REQUIRE(c && acc && x && polys > 0, "null argument"); c->eng->add(acc, x, polys, ci); }); }
int apsu_he_add_plain(apsu_he_ctx *c, uint64_t *ct, const uint64_t *pt, size_t cnt, int ci)
{ return guarded([&] { REQUIRE(c && ct && pt, "null argument"); c->eng->add_plain(ct, pt, cnt, ci); }); }
int apsu_he_multiply(apsu_he_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out3, int ci)
{ return guarded([&] { REQUIRE(c && a && b && out3, "null argument"); c->eng->multiply(a, b, out3, ci); }); }
int apsu_he_square(apsu_he_ctx *c, const uint64_t *a, uint64_t *out3, int ci)
{ return guarded([&] { REQUIRE(c && a && out3, "null argument"); c->eng->multiply(a, a, out3, ci); }); }
int apsu_he_multiply_sized(apsu_he_ctx *c, const uint64_t *a, int size_a, const uint64_t *b, int size_b, uint64_t *out, int ci)
{ return guarded([&] { REQUIRE(c && a && b && out, "null argument"); c->eng->multiply_sized(a, size_a, b, size_b, out, ci); }); }
int apsu_he_relinearize(apsu_he_ctx *c, uint64_t *ct3, const apsu_he_relin *rk, int ci)
{ return guarded([&] { REQUIRE(c && ct3 && rk, "null argument"); c->eng->relinearize(ct3, *rk->rk, ci); }); }
int apsu_he_mod_switch_to_next(apsu_he_ctx *c, uint64_t *ct, int polys, int ci)
{ return guarded([&] { REQUIRE(c && ct && polys > 0, "null argument"); c->eng->mod_switch_to_next(ct, polys, ci); }); }
int apsu_he_clear_irrelevant_bits(apsu_he_ctx *c, uint64_t *ct, int polys)
{ return guarded([&] { REQUIRE(c && ct && polys > 0, "null argument"); c->eng->clear_irrelevant_bits(ct, polys); }); }

// ---- tier 2
int apsu_he_relin_upload(apsu_he_ctx *c, const uint64_t *ksk, apsu_he_relin **out)
{
    return guarded([&] {
        REQUIRE(c && ksk && out, "null argument");
        auto r = new apsu_he_relin;
        try { r->rk = c->eng->upload_relin_keys(ksk); } catch (...) { delete r; throw; }
        *out = r;
    });
}
int apsu_he_relin_free(apsu_he_relin *rk) { return guarded([&] { delete rk; }); }

int apsu_he_db_upload_bundle(apsu_he_ctx *c, uint32_t bundle_idx, uint32_t cache_idx, uint32_t n_coeffs,
                             const uint64_t *const *coeff_ptrs, const uint8_t *is_ntt, apsu_he_bundle **out)
{
    return guarded([&] {
        REQUIRE(c && coeff_ptrs && is_ntt && out, "null argument");
        auto b = new apsu_he_bundle;
        try { b->b = c->eng->upload_bundle(bundle_idx, cache_idx, n_coeffs, coeff_ptrs, is_ntt); } catch (...) { delete b; throw; }
        *out = b;
    });
}
int apsu_he_db_random_bundle(apsu_he_ctx *c, uint32_t bundle_idx, uint32_t cache_idx, uint32_t degree, uint64_t seed,
                             apsu_he_bundle **out)
{
    return guarded([&] {
        REQUIRE(c && out, "null argument");
        auto b = new apsu_he_bundle;
        try { b->b = c->eng->random_bundle(bundle_idx, cache_idx, degree, seed); } catch (...) { delete b; throw; }
        *out = b;
    });
}
int apsu_he_db_build_bundle(apsu_he_ctx *c, uint32_t bundle_idx, uint32_t cache_idx, const uint64_t *roots, const uint32_t *counts,
                            uint32_t bins, uint32_t stride, apsu_he_bundle **out)
{
    return guarded([&] {
        REQUIRE(c && counts && out && (roots || !bins), "null argument");
        auto b = new apsu_he_bundle;
        try { b->b = c->eng->build_bundle(bundle_idx, cache_idx, roots, counts, bins, stride); } catch (...) { delete b; throw; }
        *out = b;
    });
}
int apsu_he_algebraize_items(apsu_he_ctx *c, const uint8_t *items, size_t count, int items_on_device, uint64_t *felts, int felts_on_device)
{
    return guarded([&] {
        REQUIRE(c && (!count || (items && felts)), "null argument");
        c->eng->algebraize_items(items, count, items_on_device != 0, felts, felts_on_device != 0);
    });
}
int apsu_he_bundle_download(apsu_he_ctx *c, const apsu_he_bundle *b, uint32_t degree, uint64_t *out, size_t capacity_words,
                            size_t *words, int *kind)
{
    return guarded([&] {
        REQUIRE(c && b && out, "null argument");
        size_t w = c->eng->download_coeff(*b->b, degree, out, capacity_words, kind);
        if (words) *words = w;
    });
}
int apsu_he_bundle_image_size(apsu_he_ctx *c, const apsu_he_bundle *b, uint64_t *bytes)
{ return guarded([&] { REQUIRE(c && b && bytes, "null argument"); *bytes = c->eng->bundle_image_size(*b->b); }); }
int apsu_he_bundle_save(apsu_he_ctx *c, const apsu_he_bundle *b, uint8_t *buf, uint64_t capacity, uint64_t *written)
{
    return guarded([&] {
        REQUIRE(c && b && buf, "null argument");
        size_t w = c->eng->save_bundle(*b->b, buf, (size_t)capacity);
        if (written) *written = w;
    });
}
int apsu_he_bundle_load(apsu_he_ctx *c, const uint8_t *buf, uint64_t size, apsu_he_bundle **out)
{
    return guarded([&] {
        REQUIRE(c && buf && out, "null argument");
        auto b = new apsu_he_bundle;
        try { b->b = c->eng->load_bundle(buf, (size_t)size); } catch (...) { delete b; throw; }
        *out = b;
    });
}
int apsu_he_set_two_stream(apsu_he_ctx *c, int mode)
{ return guarded([&] { REQUIRE(c, "null argument"); c->eng->set_two_stream(mode); }); }
int apsu_he_set_async_results(apsu_he_ctx *c, int on)
{ return guarded([&] { REQUIRE(c, "null argument"); c->eng->set_async_results(on != 0); }); }
int apsu_he_set_query_overlap(apsu_he_ctx *c, int on)
{ return guarded([&] { REQUIRE(c, "null argument"); if (on < 0 || on > 3) throw std::invalid_argument("apsu_he_set_query_overlap: mode must be 0 .. 3"); c->eng->set_query_overlap(on); }); }
int apsu_he_set_tier1_on_device(apsu_he_ctx *c, int on)
{ return guarded([&] { REQUIRE(c, "null argument"); c->eng->set_tier1_on_device(on != 0); }); }
int apsu_he_sync(apsu_he_ctx *c)
{ return guarded([&] { REQUIRE(c, "null argument"); c->eng->wait(); }); }
int apsu_he_stream(apsu_he_ctx *c, void **hip_stream)
{ return guarded([&] { REQUIRE(c && hip_stream, "null argument"); *hip_stream = (void *)c->eng->stream(); }); }
int apsu_he_mask_generate(apsu_he_ctx *c, uint64_t seed, uint32_t count, uint64_t *masks_dev, uint64_t *values, uint64_t *blocks)
{ return guarded([&] { REQUIRE(c && (masks_dev || !count), "null argument"); c->eng->mask_generate(seed, count, masks_dev, values, blocks); }); }
int apsu_he_mask_generate_blake2xb(apsu_he_ctx *c, const uint64_t *seed, uint64_t first_value, uint32_t count, uint64_t *masks_dev,
                                   uint64_t *values, uint64_t *blocks)
{
    return guarded([&] {
        REQUIRE(c && seed && (masks_dev || !count), "null argument");
        c->eng->mask_generate_blake2xb(seed, first_value, count, masks_dev, values, blocks);
    });
}
int apsu_he_decrypt_decode(apsu_he_ctx *c, const uint64_t *sk_ntt, const uint64_t *cts, int cts_on_device, uint32_t count,
                           uint64_t *values, uint64_t *blocks)
{ return guarded([&] { REQUIRE(c && sk_ntt && (cts || !count), "null argument"); c->eng->decrypt_decode(sk_ntt, cts, cts_on_device != 0, count, values, blocks); }); }
int apsu_he_bundle_degree(const apsu_he_bundle *b, uint32_t *degree)
{ return guarded([&] { REQUIRE(b && degree, "null argument"); *degree = b->b->degree; }); }
int apsu_he_bundle_result_size(const apsu_he_ctx *c, const apsu_he_bundle *b, uint32_t *polys)
{ return guarded([&] { REQUIRE(c && b && polys, "null argument"); *polys = c->eng->result_size(*b->b); }); }
int apsu_he_power_size(const apsu_he_ctx *c, uint32_t power, uint32_t *polys)
{ return guarded([&] { REQUIRE(c && polys, "null argument"); *polys = c->eng->power_size(power); }); }
int apsu_he_bundle_free(apsu_he_bundle *b) { return guarded([&] { delete b; }); }
int apsu_he_bundle_bytes(const apsu_he_bundle *b, uint64_t *db_bytes)
{ return guarded([&] { REQUIRE(b && db_bytes, "null argument"); *db_bytes = b->b->db_bytes(); }); }

int apsu_he_compute_powers(apsu_he_ctx *c, const uint32_t *bundle_indices, int nb, const uint64_t *const *src, int on_device,
                           const apsu_he_relin *rk, apsu_he_powers **out)
{
    return guarded([&] {
        REQUIRE(c && bundle_indices && src && out, "null argument");
        auto p = new apsu_he_powers;
        try { p->p = c->eng->compute_powers(bundle_indices, nb, src, on_device != 0, rk ? rk->rk.get() : nullptr); }
        catch (...) { delete p; throw; }
        {
            std::lock_guard<std::mutex> g(g_registry_mu);
            p->ctx = c;
            c->live_powers.insert(p);
        }
        *out = p;
    });
}
/* buffers go back to the pool of the context if it is still alive; after apsu_he_destroy only the handle is left */
int apsu_he_powers_free(apsu_he_powers *p)
{
    return guarded([&] {
        if (!p) return;
        // the registry lock only covers the bookkeeping: recycling takes the Engine's own lock, which a long synchronous call
        // on that context may hold -- it must not stall powers_free / compute_powers / destroy of OTHER contexts.  A context
        // being destroyed concurrently is kept alive for the recycle by its `recycling` count (apsu_he_destroy waits for 0).
        apsu_he_ctx *owner = nullptr;
        {
            std::lock_guard<std::mutex> g(g_registry_mu);
            if (p->ctx) {
                owner = p->ctx;
                owner->live_powers.erase(p);
                owner->recycling++;
                p->ctx = nullptr;
            }
        }
        if (owner) {
            try { owner->eng->recycle_powers(std::move(p->p)); } catch (...) { }
            std::lock_guard<std::mutex> g(g_registry_mu);
            owner->recycling--;
            g_registry_cv.notify_all();
        }
        delete p;
    });
}

int apsu_he_powers_download(apsu_he_ctx *c, const apsu_he_powers *p, uint32_t bundle_idx, uint32_t power, uint64_t *out,
                            size_t capacity_words, int *chain_idx, int *is_ntt)
{
    return guarded([&] {
        REQUIRE(c && p && out, "null argument");
        REQUIRE(p->p, "the powers object outlived its context");
        c->eng->download_power(*p->p, bundle_idx, power, out, capacity_words, chain_idx, is_ntt);
    });
}

int apsu_he_eval_bundles(apsu_he_ctx *c, const apsu_he_bundle *const *bundles, int count, const apsu_he_powers *powers,
                         const apsu_he_relin *rk, const uint64_t *const *masks, int masks_on_device, uint64_t *out, int out_on_device)
{
    return guarded([&] {
        REQUIRE(c && bundles && powers && masks && out && count >= 0, "null argument");
        REQUIRE(powers->p, "the powers object outlived its context");
        std::vector<const Bundle *> bs(count);
        for (int i = 0; i < count; i++) { REQUIRE(bundles[i], "null bundle"); bs[i] = bundles[i]->b.get(); }
        c->eng->eval_bundles(bs.data(), count, *powers->p, rk ? rk->rk.get() : nullptr, masks, masks_on_device != 0, out,
                             out_on_device != 0);
    });
}

// ---- N2: the whole database in one file (db_file.h)
struct apsu_he_db_file { std::unique_ptr<DbFile> f; };
int apsu_he_db_file_save(apsu_he_ctx *c, const char *path, const apsu_he_bundle *const *bundles, int count)
{
    return guarded([&] {
        REQUIRE(c && path && bundles && count > 0, "null argument");
        std::vector<Engine *> engs(count, c->eng.get());
        std::vector<const Bundle *> bs(count);
        for (int i = 0; i < count; i++) { REQUIRE(bundles[i], "null bundle"); bs[i] = bundles[i]->b.get(); }
        db_file_save(path, engs.data(), bs.data(), (size_t)count);
    });
}
int apsu_he_db_file_open(const char *path, apsu_he_db_file **out)
{
    return guarded([&] {
        REQUIRE(path && out, "null argument");
        auto h = std::make_unique<apsu_he_db_file>();
        h->f = std::make_unique<DbFile>(path);
        *out = h.release();
    });
}
int apsu_he_db_file_close(apsu_he_db_file *f) { return guarded([&] { delete f; }); }
int apsu_he_db_file_count(const apsu_he_db_file *f, int *count, uint64_t *file_bytes)
{
    return guarded([&] {
        REQUIRE(f && count, "null argument");
        *count = (int)f->f->count();
        if (file_bytes) *file_bytes = f->f->file_bytes();
    });
}
int apsu_he_db_file_entry(const apsu_he_db_file *f, int i, uint32_t *bundle_idx, uint32_t *cache_idx, uint32_t *degree, uint64_t *image_bytes)
{
    return guarded([&] {
        REQUIRE(f && i >= 0 && (size_t)i < f->f->count(), "entry out of range");
        const DbFileEntry &e = f->f->entry((size_t)i);
        if (bundle_idx) *bundle_idx = e.bundle_idx;
        if (cache_idx) *cache_idx = e.cache_idx;
        if (degree) *degree = e.degree;
        if (image_bytes) *image_bytes = e.bytes;
    });
}
int apsu_he_db_file_load(apsu_he_ctx *c, const apsu_he_db_file *f, int i, apsu_he_bundle **out)
{
    return guarded([&] {
        REQUIRE(c && f && out && i >= 0 && (size_t)i < f->f->count(), "bad argument");
        f->f->check_parameters(*c->eng);
        auto b = new apsu_he_bundle;
        try { b->b = f->f->load(*c->eng, (size_t)i); } catch (...) { delete b; throw; }
        *out = b;
    });
}

// ---- several GPUs behind one handle (multi.h)
int apsu_he_partition_bundles(uint32_t bundle_idx_count, int n_devices, const uint32_t *bundle_idx, const uint32_t *cache_idx,
                              const uint32_t *degree, int count, int *device_slot)
{
    return guarded([&] {
        REQUIRE(count >= 0 && (count == 0 || (bundle_idx && cache_idx && degree && device_slot)), "null argument");
        std::vector<ShardUnit> u(count);
        for (int i = 0; i < count; i++) u[i] = ShardUnit{ bundle_idx[i], cache_idx[i], degree[i] };
        auto r = partition_units(u, bundle_idx_count, n_devices);
        for (int i = 0; i < count; i++) device_slot[i] = r[i];
    });
}
int apsu_he_partition_bundles_ex(uint32_t bundle_idx_count, int n_devices, const uint32_t *bundle_idx, const uint32_t *cache_idx,
                                 const uint32_t *degree, int count, uint64_t compute_powers_cost, int *device_slot)
{
    return guarded([&] {
        REQUIRE(count >= 0 && (count == 0 || (bundle_idx && cache_idx && degree && device_slot)), "null argument");
        std::vector<ShardUnit> u(count);
        for (int i = 0; i < count; i++) u[i] = ShardUnit{ bundle_idx[i], cache_idx[i], degree[i] };
        auto r = partition_units(u, bundle_idx_count, n_devices, compute_powers_cost);
        for (int i = 0; i < count; i++) device_slot[i] = r[i];
    });
}
// ComputePowers for ONE bundle index in the partition rule's cost unit (degree + 64 per BinBundle): about 110 per ciphertext
// product of the PowersDag on MI355X (16M-4096: 66 products = 0.3-0.47 ms against 0.082 ms per BinBundle of degree 1303;
// 256M-4096: 311 products = 2.0 ms against 0.213 ms per BinBundle of degree 3999; profiles/r02_rank_cost*.txt)
int apsu_he_compute_powers_cost(const apsu_he_ctx *ctx, uint64_t *cost)
{
    return guarded([&] {
        REQUIRE(ctx && cost, "null argument");
        REQUIRE(ctx->eng->psu(), "context was created without PSUParams");
        const PowersDag &dag = ctx->eng->dag();
        *cost = 110u * (uint64_t)(dag.target_powers().size() - dag.source_count());
    });
}

int apsu_he_multi_create(const char *json, const int *devices, int n_devices, apsu_he_multi **out)
{
    return guarded([&] {
        REQUIRE(json && devices && out && n_devices > 0, "null argument");
        PSUParams p = PSUParams::Load(json);
        HeParams hp = HeParams::FromPSUParams(p);
        auto m = new apsu_he_multi;
        try { m->m = std::make_unique<MultiEngine>(hp, p, std::vector<int>(devices, devices + n_devices)); } catch (...) { delete m; throw; }
        *out = m;
    });
}
int apsu_he_multi_destroy(apsu_he_multi *m) { return guarded([&] { delete m; }); }
int apsu_he_multi_device_count(const apsu_he_multi *m, int *n_devices)
{ return guarded([&] { REQUIRE(m && n_devices, "null argument"); *n_devices = m->m->device_count(); }); }
int apsu_he_multi_result_polys(const apsu_he_multi *m, uint32_t *polys)
{ return guarded([&] { REQUIRE(m && polys, "null argument"); *polys = m->m->engine(0).result_polys(); }); }
int apsu_he_multi_relin_upload(apsu_he_multi *m, const uint64_t *ksk)
{ return guarded([&] { REQUIRE(m && ksk, "null argument"); m->m->upload_relin_keys(ksk); }); }
int apsu_he_multi_db_upload_bundle(apsu_he_multi *m, int device_slot, uint32_t bundle_idx, uint32_t cache_idx, uint32_t n_coeffs,
                                   const uint64_t *const *coeff_ptrs, const uint8_t *is_ntt, int *bundle_id)
{
    return guarded([&] {
        REQUIRE(m && coeff_ptrs && is_ntt && bundle_id, "null argument");
        REQUIRE(device_slot >= 0 && device_slot < m->m->device_count(), "device slot out of range");
        *bundle_id = m->m->upload_bundle(device_slot, bundle_idx, cache_idx, n_coeffs, coeff_ptrs, is_ntt);
    });
}
int apsu_he_multi_db_random_bundle(apsu_he_multi *m, int device_slot, uint32_t bundle_idx, uint32_t cache_idx, uint32_t degree,
                                   uint64_t seed, int *bundle_id)
{
    return guarded([&] {
        REQUIRE(m && bundle_id, "null argument");
        REQUIRE(device_slot >= 0 && device_slot < m->m->device_count(), "device slot out of range");
        *bundle_id = m->m->random_bundle(device_slot, bundle_idx, cache_idx, degree, seed);
    });
}
int apsu_he_multi_db_load_file(apsu_he_multi *m, const apsu_he_db_file *f, int *n_loaded)
{
    return guarded([&] {
        REQUIRE(m && f, "null argument");
        const int k = m->m->load_file(*f->f);
        if (n_loaded) *n_loaded = k;
    });
}
int apsu_he_multi_db_save_file(apsu_he_multi *m, const char *path)
{ return guarded([&] { REQUIRE(m && path, "null argument"); m->m->save_file(path); }); }
int apsu_he_multi_db_clear(apsu_he_multi *m) { return guarded([&] { REQUIRE(m, "null argument"); m->m->clear_bundles(); }); }
int apsu_he_eval_all(apsu_he_multi *m, const uint64_t *const *src_cts, const uint64_t *const *masks, uint64_t *out_cts, int out_device_slot)
{
    return guarded([&] {
        REQUIRE(m && src_cts && masks && out_cts, "null argument");
        m->m->eval_all(src_cts, masks, out_cts, out_device_slot);
    });
}
int apsu_he_eval_all_ex(apsu_he_multi *m, const uint64_t *const *src_cts, const uint64_t *const *masks, uint64_t *out_cts, int out_device_slot,
                        unsigned flags, int in_device_slot)
{
    return guarded([&] {
        REQUIRE(m && src_cts && masks && out_cts, "null argument");
        REQUIRE((flags & ~0x3fu) == 0, "unknown flag");
        m->m->eval_all(src_cts, masks, out_cts, out_device_slot, flags, in_device_slot);
    });
}
const char *apsu_he_multi_last_gather(const apsu_he_multi *m) { return m ? m->m->last_gather() : ""; }
int apsu_he_multi_phase_enable(apsu_he_multi *m, int on)
{ return guarded([&] { REQUIRE(m, "null argument"); m->m->phase_enable(on != 0); }); }
int apsu_he_multi_phase_read(apsu_he_multi *m, uint64_t *count, double *avg_ms, double *min_ms, double *max_ms, int reset)
{
    return guarded([&] {
        REQUIRE(m, "null argument");
        Engine::PhaseSummary all[Engine::PH_COUNT];
        m->m->phase_read(all, reset != 0);
        for (int i = 0; i < Engine::PH_COUNT; i++) {
            if (count) count[i] = all[i].count;
            if (avg_ms) avg_ms[i] = all[i].count ? all[i].sum_ms / (double)all[i].count : 0.0;
            if (min_ms) min_ms[i] = all[i].min_ms;
            if (max_ms) max_ms[i] = all[i].max_ms;
        }
    });
}
int apsu_he_host_alloc(size_t bytes, void **out)
{
    return guarded([&] {
        REQUIRE(out && bytes, "null argument");
        void *p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); throw std::bad_alloc(); }
        *out = p;
    });
}
int apsu_he_host_free(void *p) { return guarded([&] { if (p && hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); throw std::invalid_argument("not an apsu_he_host_alloc pointer"); } }); }

// ---- N3: network framing (wire.h); host only
struct apsu_he_wire_query { wire::QueryRequest q; };
static int wire_out(std::vector<uint8_t> &&v, uint8_t **out, size_t *out_size)
{
    uint8_t *p = static_cast<uint8_t *>(std::malloc(v.size() ? v.size() : 1));
    if (!p) throw std::bad_alloc();
    if (!v.empty()) std::memcpy(p, v.data(), v.size());
    *out = p; *out_size = v.size();
    return 0;
}
int apsu_he_wire_buffer_free(uint8_t *p) { std::free(p); return APSU_HE_OK; }
int apsu_he_wire_build_header(uint32_t version, uint32_t type, uint8_t **out, size_t *out_size)
{ return guarded([&] { REQUIRE(out && out_size, "null argument"); wire_out(wire::build_header(wire::Header{ version, type }), out, out_size); }); }
int apsu_he_wire_parse_header(const uint8_t *buf, size_t size, uint32_t *version, uint32_t *type)
{
    return guarded([&] {
        REQUIRE(buf && version && type, "null argument");
        const wire::Header h = wire::parse_header(buf, size);
        *version = h.version; *type = h.type;
    });
}
int apsu_he_wire_build_query_request(uint8_t compression_type, const uint8_t *relin_keys, size_t relin_keys_size, uint32_t n_parts,
                                     const uint32_t *exponents, const uint32_t *cts_per_part, const uint8_t *const *ct_data,
                                     const size_t *ct_sizes, uint8_t **out, size_t *out_size)
{
    return guarded([&] {
        REQUIRE(out && out_size && (n_parts == 0 || (exponents && cts_per_part)), "null argument");
        wire::QueryRequest q;
        q.compression_type = compression_type;
        q.has_relin_keys = relin_keys != nullptr;
        q.relin_keys = wire::Span{ relin_keys, relin_keys_size };
        size_t k = 0;
        for (uint32_t i = 0; i < n_parts; i++) {
            wire::QueryPart part;
            part.exponent = exponents[i];
            for (uint32_t j = 0; j < cts_per_part[i]; j++, k++) {
                REQUIRE(ct_data && ct_sizes && (ct_data[k] || !ct_sizes[k]), "null ciphertext");
                part.cts.push_back(wire::Span{ ct_data[k], ct_sizes[k] });
            }
            q.parts.push_back(std::move(part));
        }
        wire_out(wire::build_query_request(q), out, out_size);
    });
}
int apsu_he_wire_parse_query_request(const uint8_t *buf, size_t size, apsu_he_wire_query **out)
{
    return guarded([&] {
        REQUIRE(buf && out, "null argument");
        auto m = new apsu_he_wire_query;
        try { m->q = wire::parse_query_request(buf, size); } catch (...) { delete m; throw; }
        *out = m;
    });
}
int apsu_he_wire_query_free(apsu_he_wire_query *m) { return guarded([&] { delete m; }); }
int apsu_he_wire_query_info(const apsu_he_wire_query *m, uint8_t *compression_type, int *has_relin_keys, const uint8_t **relin_keys,
                            size_t *relin_keys_size, uint32_t *n_parts)
{
    return guarded([&] {
        REQUIRE(m, "null argument");
        if (compression_type) *compression_type = m->q.compression_type;
        if (has_relin_keys) *has_relin_keys = m->q.has_relin_keys ? 1 : 0;
        if (relin_keys) *relin_keys = m->q.relin_keys.p;
        if (relin_keys_size) *relin_keys_size = m->q.relin_keys.n;
        if (n_parts) *n_parts = (uint32_t)m->q.parts.size();
    });
}
int apsu_he_wire_query_part(const apsu_he_wire_query *m, uint32_t part, uint32_t *exponent, uint32_t *n_cts)
{
    return guarded([&] {
        REQUIRE(m && part < m->q.parts.size(), "part out of range");
        if (exponent) *exponent = m->q.parts[part].exponent;
        if (n_cts) *n_cts = (uint32_t)m->q.parts[part].cts.size();
    });
}
int apsu_he_wire_query_ct(const apsu_he_wire_query *m, uint32_t part, uint32_t ct, const uint8_t **data, size_t *size)
{
    return guarded([&] {
        REQUIRE(m && data && size && part < m->q.parts.size() && ct < m->q.parts[part].cts.size(), "index out of range");
        *data = m->q.parts[part].cts[ct].p; *size = m->q.parts[part].cts[ct].n;
    });
}
int apsu_he_wire_build_query_response(uint32_t package_count, uint32_t alpha_max_cache_count, uint8_t **out, size_t *out_size)
{ return guarded([&] { REQUIRE(out && out_size, "null argument"); wire_out(wire::build_query_response(wire::QueryResponse{ package_count, alpha_max_cache_count }), out, out_size); }); }
int apsu_he_wire_parse_query_response(const uint8_t *buf, size_t size, uint32_t *package_count, uint32_t *alpha_max_cache_count)
{
    return guarded([&] {
        REQUIRE(buf && package_count && alpha_max_cache_count, "null argument");
        const wire::QueryResponse r = wire::parse_query_response(buf, size);
        *package_count = r.package_count; *alpha_max_cache_count = r.alpha_max_cache_count;
    });
}
int apsu_he_wire_build_result_package(uint32_t bundle_idx, uint32_t cache_idx, const uint8_t *psu_result, size_t psu_result_size,
                                      uint32_t label_byte_count, uint32_t nonce_byte_count, uint32_t n_labels,
                                      const uint8_t *const *label_data, const size_t *label_sizes, uint8_t **out, size_t *out_size)
{
    return guarded([&] {
        REQUIRE(out && out_size && (psu_result || !psu_result_size), "null argument");
        wire::ResultPackage p;
        p.bundle_idx = bundle_idx; p.cache_idx = cache_idx; p.psu_result = wire::Span{ psu_result, psu_result_size };
        p.label_byte_count = label_byte_count; p.nonce_byte_count = nonce_byte_count;
        for (uint32_t i = 0; i < n_labels; i++) {
            REQUIRE(label_data && label_sizes, "null label");
            p.label_result.push_back(wire::Span{ label_data[i], label_sizes[i] });
        }
        wire_out(wire::build_result_package(p), out, out_size);
    });
}
int apsu_he_wire_parse_result_package(const uint8_t *buf, size_t size, uint32_t *bundle_idx, uint32_t *cache_idx, const uint8_t **psu_result,
                                      size_t *psu_result_size, uint32_t *label_byte_count, uint32_t *nonce_byte_count, uint32_t *n_labels)
{
    return guarded([&] {
        REQUIRE(buf, "null argument");
        const wire::ResultPackage p = wire::parse_result_package(buf, size);
        if (bundle_idx) *bundle_idx = p.bundle_idx;
        if (cache_idx) *cache_idx = p.cache_idx;
        if (psu_result) *psu_result = p.psu_result.p;
        if (psu_result_size) *psu_result_size = p.psu_result.n;
        if (label_byte_count) *label_byte_count = p.label_byte_count;
        if (nonce_byte_count) *nonce_byte_count = p.nonce_byte_count;
        if (n_labels) *n_labels = (uint32_t)p.label_result.size();
    });
}
int apsu_he_wire_result_label(const uint8_t *buf, size_t size, uint32_t index, const uint8_t **data, size_t *data_size)
{
    return guarded([&] {
        REQUIRE(buf && data && data_size, "null argument");
        const wire::ResultPackage p = wire::parse_result_package(buf, size);
        REQUIRE(index < p.label_result.size(), "label index out of range");
        *data = p.label_result[index].p; *data_size = p.label_result[index].n;
    });
}
// every label of a package with ONE parse: data / sizes hold n_labels entries (as reported by apsu_he_wire_parse_result_package)
int apsu_he_wire_result_labels(const uint8_t *buf, size_t size, uint32_t capacity, const uint8_t **data, size_t *sizes, uint32_t *n_labels)
{
    return guarded([&] {
        REQUIRE(buf && (capacity == 0 || (data && sizes)), "null argument");
        const wire::ResultPackage p = wire::parse_result_package(buf, size);
        if (n_labels) *n_labels = (uint32_t)p.label_result.size();
        REQUIRE(capacity == 0 || capacity >= p.label_result.size(), "label arrays too small");
        for (size_t i = 0; i < p.label_result.size() && i < capacity; i++) { data[i] = p.label_result[i].p; sizes[i] = p.label_result[i].n; }
    });
}
int apsu_he_seed_expand(apsu_he_ctx *c, int chain_idx, int count, const uint64_t *seeds, uint64_t *const *dst_device)
{ return guarded([&] { REQUIRE(c, "null argument"); c->eng->seed_expand(chain_idx, count, seeds, dst_device); }); }

// ---- N3: SEAL's object serialisation (seal_codec.h; UNPINNED)
struct apsu_he_seal_ctx { std::vector<sealio::Level> chain; size_t n = 0, K = 0; u64 t = 0; };
static const sealio::Level &seal_level(const apsu_he_seal_ctx *c, int chain_idx)
{
    // chain[0] = key level (all K primes), chain[1 + i] = the level with K - 1 - i primes
    const int K = (int)c->K;
    if (chain_idx < 0 || chain_idx == K - 1) return c->chain[0];
    REQUIRE(chain_idx < K - 1, "chain_idx out of range");
    return c->chain[(size_t)(K - 1 - chain_idx)];
}
static int seal_chain_idx(const apsu_he_seal_ctx *c, const uint64_t id[4])
{
    for (size_t i = 0; i < c->chain.size(); i++)
        if (!std::memcmp(c->chain[i].parms_id, id, 32)) return (int)c->chain[i].q.size() - 1;
    return -1;
}
int apsu_he_seal_ctx_create_raw(uint64_t n, const uint64_t *coeff_modulus, int k, uint64_t plain_modulus, apsu_he_seal_ctx **out)
{
    return guarded([&] {
        REQUIRE(coeff_modulus && out && k > 0 && k <= 64 && n >= 2 && n <= (1u << 20), "bad argument");
        auto c = new apsu_he_seal_ctx;
        c->n = (size_t)n; c->K = (size_t)k; c->t = plain_modulus;
        c->chain = sealio::modulus_chain(n, std::vector<uint64_t>(coeff_modulus, coeff_modulus + k), plain_modulus);
        *out = c;
    });
}
int apsu_he_seal_ctx_create(const char *json, apsu_he_seal_ctx **out)
{
    return guarded([&] {
        REQUIRE(json && out, "null argument");
        PSUParams p = PSUParams::Load(json);
        HeParams hp = HeParams::FromPSUParams(p);
        auto c = new apsu_he_seal_ctx;
        c->n = hp.n; c->K = (size_t)hp.K; c->t = hp.t;
        c->chain = sealio::modulus_chain(hp.n, std::vector<uint64_t>(hp.key_q.begin(), hp.key_q.begin() + hp.K), hp.t);
        *out = c;
    });
}
int apsu_he_seal_ctx_free(apsu_he_seal_ctx *c) { return guarded([&] { delete c; }); }
int apsu_he_seal_parms_id(const apsu_he_seal_ctx *c, int chain_idx, uint64_t out[4])
{ return guarded([&] { REQUIRE(c && out, "null argument"); std::memcpy(out, seal_level(c, chain_idx).parms_id, 32); }); }
int apsu_he_seal_sample_poly_uniform(const apsu_he_seal_ctx *c, int chain_idx, const uint64_t seed[8], uint64_t *out)
{
    return guarded([&] {
        REQUIRE(c && seed && out, "null argument");
        const sealio::Level &l = seal_level(c, chain_idx);
        sealio::sample_poly_uniform(seed, l.q.data(), l.q.size(), c->n, out);
    });
}
int apsu_he_seal_ct_load(const apsu_he_seal_ctx *c, const uint8_t *buf, size_t size, uint64_t parms_id[4], int *chain_idx, int *is_ntt_form,
                         uint64_t *ct_size, uint64_t *poly_modulus_degree, uint64_t *coeff_modulus_size, int *was_seeded, uint64_t *data,
                         size_t data_capacity_words, size_t *consumed)
{
    return guarded([&] {
        REQUIRE(buf, "null argument");
        static const std::vector<sealio::Level> none;
        const sealio::Ciphertext ct = sealio::load_ciphertext(buf, size, c ? c->chain : none, consumed);
        if (parms_id) std::memcpy(parms_id, ct.parms_id, 32);
        if (chain_idx) *chain_idx = c ? seal_chain_idx(c, ct.parms_id) : -1;
        if (is_ntt_form) *is_ntt_form = ct.is_ntt_form;
        if (ct_size) *ct_size = ct.size;
        if (poly_modulus_degree) *poly_modulus_degree = ct.poly_modulus_degree;
        if (coeff_modulus_size) *coeff_modulus_size = ct.coeff_modulus_size;
        if (was_seeded) *was_seeded = ct.seeded ? 1 : 0;
        if (data) {
            REQUIRE(data_capacity_words >= ct.data.size(), "output buffer too small");
            std::memcpy(data, ct.data.data(), ct.data.size() * sizeof(uint64_t));
        }
    });
}
int apsu_he_seal_ct_load_unexpanded(const apsu_he_seal_ctx *c, const uint8_t *buf, size_t size, int *chain_idx, int *is_ntt_form, uint64_t *ct_size,
                                    uint64_t *coeff_modulus_size, int *was_seeded, uint64_t seed[8], uint64_t *data, size_t data_capacity_words,
                                    size_t *consumed)
{
    return guarded([&] {
        REQUIRE(buf, "null argument");
        static const std::vector<sealio::Level> none;
        const sealio::Ciphertext ct = sealio::load_ciphertext(buf, size, c ? c->chain : none, consumed, false);
        if (chain_idx) *chain_idx = c ? seal_chain_idx(c, ct.parms_id) : -1;
        if (is_ntt_form) *is_ntt_form = ct.is_ntt_form;
        if (ct_size) *ct_size = ct.size;
        if (coeff_modulus_size) *coeff_modulus_size = ct.coeff_modulus_size;
        if (was_seeded) *was_seeded = ct.seeded ? 1 : 0;
        if (seed && ct.seeded) std::memcpy(seed, ct.seed, 64);
        if (data) {
            const size_t words = ct.seeded ? ct.data.size() / 2 : ct.data.size();       // seeded: c0 only
            REQUIRE(data_capacity_words >= words, "output buffer too small");
            std::memcpy(data, ct.data.data(), words * sizeof(uint64_t));
        }
    });
}
int apsu_he_seal_ct_save(const apsu_he_seal_ctx *c, int chain_idx, int is_ntt_form, uint64_t ct_size, const uint64_t *data, const uint64_t *seed,
                         int compr_mode, int version_major, int version_minor, uint8_t **out, size_t *out_size)
{
    return guarded([&] {
        REQUIRE(c && data && out && out_size && ct_size >= 1 && ct_size <= 64, "bad argument");
        const sealio::Level &l = seal_level(c, chain_idx);
        sealio::Ciphertext ct;
        std::memcpy(ct.parms_id, l.parms_id, 32);
        ct.is_ntt_form = is_ntt_form ? 1 : 0; ct.size = ct_size; ct.poly_modulus_degree = c->n; ct.coeff_modulus_size = l.q.size();
        ct.version_major = (uint8_t)version_major; ct.version_minor = (uint8_t)version_minor;
        ct.data.assign(data, data + ct_size * l.q.size() * c->n);
        if (seed) { ct.seeded = true; std::memcpy(ct.seed, seed, 64); }
        wire_out(sealio::save_ciphertext(ct, (uint8_t)compr_mode), out, out_size);
    });
}
int apsu_he_seal_pt_load(const apsu_he_seal_ctx *c, const uint8_t *buf, size_t size, int *chain_idx, uint64_t *coeff_count, uint64_t *data,
                         size_t data_capacity_words, size_t *consumed)
{
    return guarded([&] {
        REQUIRE(c && buf, "null argument");
        size_t used = 0;
        const sealio::Plaintext pt = sealio::load_plaintext(buf, size, &used);
        int ci = -1;                                                      // coefficient form
        if (pt.is_ntt_form()) {
            ci = seal_chain_idx(c, pt.parms_id);
            if (ci < 0) throw std::invalid_argument("parms_id of the plaintext is not in this context's modulus chain");
        }
        if (chain_idx) *chain_idx = ci;
        if (coeff_count) *coeff_count = pt.coeff_count;
        if (consumed) *consumed = used;
        if (data) {
            if (data_capacity_words < pt.data.size()) throw std::invalid_argument("output buffer too small");
            std::memcpy(data, pt.data.data(), pt.data.size() * sizeof(uint64_t));
        }
    });
}
int apsu_he_seal_pt_save(const apsu_he_seal_ctx *c, int chain_idx, const uint64_t *data, uint64_t coeff_count, int compr_mode, int version_major,
                         int version_minor, uint8_t **out, size_t *out_size)
{
    return guarded([&] {
        REQUIRE(c && data && out && out_size, "null argument");
        sealio::Plaintext pt;
        if (chain_idx >= 0) {
            const sealio::Level &l = seal_level(c, chain_idx);
            std::memcpy(pt.parms_id, l.parms_id, 32);
            REQUIRE(coeff_count == l.q.size() * c->n, "an NTT-form plaintext has one polynomial per coefficient prime of its level");
        }
        pt.coeff_count = coeff_count;
        pt.version_major = (uint8_t)version_major; pt.version_minor = (uint8_t)version_minor;
        pt.data.assign(data, data + coeff_count);
        wire_out(sealio::save_plaintext(pt, (uint8_t)compr_mode), out, out_size);
    });
}
// BinBundleCache::batched_matching_polyn as the reference holds it -- batched_coeffs[d] = a SEAL-serialised Plaintext
// (bin_bundle.cpp:421-428, compr_mode none or zstd) -- straight into apsu_he_db_upload_bundle, without SEAL on the host
static std::unique_ptr<Bundle> upload_serialized(apsu_he_ctx *c, const apsu_he_seal_ctx *sc, uint32_t bundle_idx, uint32_t cache_idx,
                                                 const std::vector<wire::Span> &blobs)
{
    const HeParams &hp = c->eng->he();
    REQUIRE(sc->n == hp.n && sc->K == (size_t)hp.K && sc->t == hp.t, "the SEAL context belongs to other parameters");
    const size_t n_coeffs = blobs.size();
    REQUIRE(n_coeffs > 0, "a BinBundle has at least one coefficient");
    std::vector<std::vector<uint64_t>> keep(n_coeffs);
    std::vector<const uint64_t *> ptrs(n_coeffs);
    std::vector<unsigned char> is_ntt(n_coeffs);
    int ntt_level = -1;
    for (size_t d = 0; d < n_coeffs; d++) {
        REQUIRE(blobs[d].p, "null plaintext");
        sealio::Plaintext pt = sealio::load_plaintext(blobs[d].p, blobs[d].n);
        is_ntt[d] = pt.is_ntt_form() ? 1 : 0;
        if (pt.is_ntt_form()) {
            const int ci = seal_chain_idx(sc, pt.parms_id);
            if (ci < 0 || ci > hp.first_chain_idx) throw std::invalid_argument("plaintext parms_id is not a data level of these parameters");
            if (ntt_level >= 0 && ci != ntt_level) throw std::invalid_argument("the NTT-form plaintexts of a BinBundle share one level (bin_bundle.cpp:385-389)");
            ntt_level = ci;
            if (pt.coeff_count != (uint64_t)(ci + 1) * hp.n) throw std::invalid_argument("NTT-form plaintext has the wrong coefficient count");
        } else {
            if (pt.coeff_count > hp.n) throw std::invalid_argument("coefficient-form plaintext is longer than the ring");
            pt.data.resize(hp.n, 0);                                 // BatchEncoder::encode writes n coefficients; shorter ones are zero-extended
        }
        keep[d] = std::move(pt.data);
        ptrs[d] = keep[d].data();
    }
    std::unique_ptr<Bundle> b = c->eng->upload_bundle(bundle_idx, cache_idx, (uint32_t)n_coeffs, ptrs.data(), is_ntt.data());
    if (ntt_level >= 0 && b->pt_level != ntt_level)
        throw std::invalid_argument("the NTT-form plaintexts are not at the level the BinBundle rule prescribes (bin_bundle.cpp:385-389)");
    return b;
}
int apsu_he_db_upload_bundle_serialized(apsu_he_ctx *c, const apsu_he_seal_ctx *sc, uint32_t bundle_idx, uint32_t cache_idx, uint32_t n_coeffs,
                                        const uint8_t *const *blobs, const size_t *blob_sizes, apsu_he_bundle **out)
{
    return guarded([&] {
        REQUIRE(c && sc && blobs && blob_sizes && out && n_coeffs > 0, "null argument");
        std::vector<wire::Span> spans(n_coeffs);
        for (uint32_t d = 0; d < n_coeffs; d++) spans[d] = wire::Span{ blobs[d], blob_sizes[d] };
        auto b = new apsu_he_bundle;
        try { b->b = upload_serialized(c, sc, bundle_idx, cache_idx, spans); } catch (...) { delete b; throw; }
        *out = b;
    });
}
// ---- PSUParams in their binary form (psu_params.fbs + SEAL's EncryptionParameters object): the parameter exchange and a saved ReceiverDB
namespace {
// -> the JSON form PSUParams::Load(json) / apsu_he_create take (psu_params.cpp:290-374).  The JSON names the coefficient primes by
// their bit sizes (CoeffModulus::Create picks the primes), so the binary form's explicit primes must be exactly those.
std::string psu_params_to_json(const wire::PsuParamsWire &w)
{
    const sealio::EncryptionParameters e = sealio::load_encryption_parameters(w.seal_params.p, w.seal_params.n);
    if (e.scheme != 1) throw std::runtime_error("failed to load parameters: invalid scheme type");      // psu_params.cpp:283-285
    std::vector<int> bits;
    for (uint64_t q : e.coeff_modulus) { int b = 0; while (b < 64 && (q >> b)) b++; bits.push_back(b); }
    if (e.coeff_modulus.empty() || e.poly_modulus_degree < 2 || (e.poly_modulus_degree & (e.poly_modulus_degree - 1)))
        throw std::runtime_error("failed to load parameters: invalid encryption parameters");
    for (int b : bits) if (b < 2 || b > 60) throw std::runtime_error("failed to load parameters: invalid coefficient modulus");
    if (coeff_modulus_create((size_t)e.poly_modulus_degree, bits) != e.coeff_modulus)
        throw std::runtime_error("failed to load parameters: the coefficient primes are not CoeffModulus::Create's for their bit sizes "
                                 "(the JSON form cannot name them)");
    std::string j = "{\"table_params\":{\"hash_func_count\":" + std::to_string(w.hash_func_count) + ",\"table_size\":" + std::to_string(w.table_size) +
                    ",\"max_items_per_bin\":" + std::to_string(w.max_items_per_bin) + "},\"item_params\":{\"felts_per_item\":" +
                    std::to_string(w.felts_per_item) + "},\"query_params\":{\"ps_low_degree\":" + std::to_string(w.ps_low_degree) + ",\"query_powers\":[";
    for (size_t i = 0; i < w.query_powers.size(); i++) j += (i ? "," : "") + std::to_string(w.query_powers[i]);
    j += "]},\"seal_params\":{\"plain_modulus\":" + std::to_string(e.plain_modulus) + ",\"poly_modulus_degree\":" + std::to_string(e.poly_modulus_degree) +
         ",\"coeff_modulus_bits\":[";
    for (size_t i = 0; i < bits.size(); i++) j += (i ? "," : "") + std::to_string(bits[i]);
    j += "]}}";
    (void)PSUParams::Load(j);                                            // the reference's own validation of what was just read
    return j;
}
} // namespace
int apsu_he_wire_psu_params_save(const char *psu_params_json, uint8_t **out, size_t *out_size)
{
    return guarded([&] {
        REQUIRE(psu_params_json && out && out_size, "null argument");
        const PSUParams p = PSUParams::Load(psu_params_json);
        const HeParams hp = HeParams::FromPSUParams(p);
        sealio::EncryptionParameters e;
        e.poly_modulus_degree = hp.n; e.coeff_modulus.assign(hp.key_q.begin(), hp.key_q.begin() + hp.K); e.plain_modulus = hp.t;
        const std::vector<uint8_t> sp = sealio::save_encryption_parameters(e, sealio::COMPR_NONE);
        wire::PsuParamsWire w;
        w.felts_per_item = p.item_params.felts_per_item;
        w.table_size = p.table_params.table_size; w.max_items_per_bin = p.table_params.max_items_per_bin; w.hash_func_count = p.table_params.hash_func_count;
        w.ps_low_degree = p.query_params.ps_low_degree;
        w.query_powers.assign(p.query_params.query_powers.begin(), p.query_params.query_powers.end());
        w.seal_params = wire::Span{ sp.data(), sp.size() };
        wire_out(wire::build_psu_params(w), out, out_size);
    });
}
int apsu_he_wire_psu_params_load(const uint8_t *buf, size_t size, uint8_t **json_out_buf, size_t *json_size)
{
    return guarded([&] {
        REQUIRE(buf && json_out_buf && json_size, "null argument");
        const std::string j = psu_params_to_json(wire::parse_psu_params(buf, size));
        wire_out(std::vector<uint8_t>(j.begin(), j.end()), json_out_buf, json_size);
    });
}
int apsu_he_wire_peek_type(const uint8_t *buf, size_t size, int is_response, int *type)
{
    return guarded([&] {
        REQUIRE(buf && type, "null argument");
        *type = is_response ? wire::peek_response_type(buf, size) : wire::peek_request_type(buf, size);
    });
}
int apsu_he_wire_build_parms_request(uint8_t **out, size_t *out_size)
{ return guarded([&] { REQUIRE(out && out_size, "null argument"); wire_out(wire::build_parms_request(), out, out_size); }); }
int apsu_he_wire_build_parms_response(const uint8_t *psu_params, size_t psu_params_size, uint8_t **out, size_t *out_size)
{
    return guarded([&] {
        REQUIRE(out && out_size && (psu_params || !psu_params_size), "null argument");
        wire_out(wire::build_parms_response(wire::Span{ psu_params, psu_params_size }), out, out_size);
    });
}
int apsu_he_wire_parse_parms_response(const uint8_t *buf, size_t size, const uint8_t **psu_params, size_t *psu_params_size)
{
    return guarded([&] {
        REQUIRE(buf && psu_params && psu_params_size, "null argument");
        const wire::Span s = wire::parse_parms_response(buf, size);
        *psu_params = s.p; *psu_params_size = s.n;
    });
}
int apsu_he_wire_build_plain_response(uint32_t bundle_idx, uint32_t cache_idx, const uint64_t *psu_result, size_t count, uint8_t **out, size_t *out_size)
{
    return guarded([&] {
        REQUIRE(out && out_size && (psu_result || !count), "null argument");
        wire::PlainResponse p;
        p.bundle_idx = bundle_idx; p.cache_idx = cache_idx;
        p.psu_result.assign(psu_result, psu_result + count);
        wire_out(wire::build_plain_response(p), out, out_size);
    });
}
int apsu_he_wire_parse_plain_response(const uint8_t *buf, size_t size, uint32_t *bundle_idx, uint32_t *cache_idx, uint64_t *psu_result, size_t capacity,
                                      size_t *count)
{
    return guarded([&] {
        REQUIRE(buf && count, "null argument");
        const wire::PlainResponse p = wire::parse_plain_response(buf, size);
        if (bundle_idx) *bundle_idx = p.bundle_idx;
        if (cache_idx) *cache_idx = p.cache_idx;
        *count = p.psu_result.size();
        if (psu_result) {
            REQUIRE(capacity >= p.psu_result.size(), "output buffer too small");
            std::memcpy(psu_result, p.psu_result.data(), p.psu_result.size() * sizeof(uint64_t));
        }
    });
}
// The header of a database the reference saved (receiver_db.fbs): its PSUParams as JSON, counts and flags; *consumed = first BinBundle
int apsu_he_wire_receiver_db_header(const uint8_t *buf, size_t size, uint8_t **psu_params_json, size_t *json_size, uint64_t *item_count,
                                    uint32_t *bin_bundle_count, int *compressed, int *stripped, uint32_t *label_byte_count, size_t *consumed)
{
    return guarded([&] {
        REQUIRE(buf, "null argument");
        const wire::ReceiverDbHeader h = wire::parse_receiver_db_header(buf, size);
        if (psu_params_json) {
            REQUIRE(json_size, "null argument");
            const std::string j = psu_params_to_json(wire::parse_psu_params(h.params.p, h.params.n));
            wire_out(std::vector<uint8_t>(j.begin(), j.end()), psu_params_json, json_size);
        }
        if (item_count) *item_count = h.item_count;
        if (bin_bundle_count) *bin_bundle_count = h.bin_bundle_count;
        if (compressed) *compressed = h.compressed ? 1 : 0;
        if (stripped) *stripped = h.stripped ? 1 : 0;
        if (label_byte_count) *label_byte_count = h.label_byte_count;
        if (consumed) *consumed = h.consumed;
    });
}
// One BinBundle as ReceiverDB::save wrote it (bin_bundle.fbs; BinBundle::save bin_bundle.cpp:1085-1168): dimensions only
int apsu_he_wire_bin_bundle_info(const uint8_t *buf, size_t size, uint32_t *bundle_idx, uint64_t *mod, int *stripped, uint32_t *n_bins,
                                 uint32_t *largest_bin, uint32_t *cache_coeffs, size_t *consumed)
{
    return guarded([&] {
        REQUIRE(buf, "null argument");
        const wire::SavedBinBundle sb = wire::parse_bin_bundle(buf, size);
        if (bundle_idx) *bundle_idx = sb.bundle_idx;
        if (mod) *mod = sb.mod;
        if (stripped) *stripped = sb.stripped ? 1 : 0;
        if (n_bins) *n_bins = (uint32_t)sb.item_bins.size();
        if (largest_bin) { size_t m = 0; for (const auto &b : sb.item_bins) m = std::max(m, b.size()); *largest_bin = (uint32_t)m; }
        if (cache_coeffs) *cache_coeffs = sb.has_cache ? (uint32_t)sb.batched_coeffs.size() : 0;
        if (consumed) *consumed = sb.consumed;
    });
}
// ... and onto the device: from its saved cache when there is one (no SEAL, no flatbuffers on the host), else rebuilt from the item
// bins on the GPU (N1).  The checks are BinBundle::load's (bin_bundle.cpp:1170-1230): field modulus, number of bins, bin sizes.
int apsu_he_db_upload_saved_bundle(apsu_he_ctx *c, const apsu_he_seal_ctx *sc, const uint8_t *buf, size_t size, uint32_t cache_idx,
                                   apsu_he_bundle **out, size_t *consumed)
{
    return guarded([&] {
        REQUIRE(c && buf && out, "null argument");
        const PSUParams *psu = c->eng->psu();
        REQUIRE(psu, "context was created without PSUParams");
        const wire::SavedBinBundle sb = wire::parse_bin_bundle(buf, size);
        auto fail = [](const char *why) { throw std::runtime_error(std::string("failed to load BinBundle: ") + why); };
        if (sb.mod != c->eng->he().t) fail("the field modulus differs from the plain_modulus of these parameters");
        if (sb.bundle_idx >= psu->bundle_idx_count) fail("bundle index out of range");
        if (!sb.stripped) {
            if (sb.item_bins.size() != psu->bins_per_bundle) fail("wrong number of item bins");
            for (const auto &bin : sb.item_bins) if (bin.size() > psu->table_params.max_items_per_bin) fail("an item bin exceeds max_items_per_bin");
        }
        auto b = new apsu_he_bundle;
        try {
            if (sb.has_cache) {
                REQUIRE(sc, "the saved cache holds SEAL objects: a SEAL context is needed");
                b->b = upload_serialized(c, sc, sb.bundle_idx, cache_idx, sb.batched_coeffs);
            } else {
                if (sb.stripped) fail("a stripped BinBundle without its cache cannot be evaluated");
                size_t stride = 1;
                for (const auto &bin : sb.item_bins) stride = std::max(stride, bin.size());
                std::vector<uint64_t> roots(sb.item_bins.size() * stride, 0);
                std::vector<uint32_t> counts(sb.item_bins.size());
                for (size_t i = 0; i < sb.item_bins.size(); i++) {
                    counts[i] = (uint32_t)sb.item_bins[i].size();
                    std::copy(sb.item_bins[i].begin(), sb.item_bins[i].end(), roots.begin() + i * stride);
                }
                b->b = c->eng->build_bundle(sb.bundle_idx, cache_idx, roots.data(), counts.data(), (uint32_t)counts.size(), (uint32_t)stride);
            }
        } catch (...) { delete b; throw; }
        *out = b;
        if (consumed) *consumed = sb.consumed;
    });
}
int apsu_he_seal_relin_keys_load(const apsu_he_seal_ctx *c, const uint8_t *buf, size_t size, uint64_t *ksk, size_t capacity_words, size_t *words,
                                 size_t *consumed)
{
    return guarded([&] {
        REQUIRE(c && buf, "null argument");
        const sealio::KSwitchKeys k = sealio::load_kswitch_keys(buf, size, c->chain, consumed);
        if (std::memcmp(k.parms_id, c->chain[0].parms_id, 32)) throw std::invalid_argument("RelinKeys were generated for other encryption parameters");
        const std::vector<uint64_t> flat = sealio::relin_keys_layout(k, c->K, c->n);
        if (words) *words = flat.size();
        if (ksk) {
            REQUIRE(capacity_words >= flat.size(), "output buffer too small");
            std::memcpy(ksk, flat.data(), flat.size() * sizeof(uint64_t));
        }
    });
}
int apsu_he_seal_relin_keys_save(const apsu_he_seal_ctx *c, const uint64_t *ksk, const uint64_t *seeds, int compr_mode, int version_major,
                                 int version_minor, uint8_t **out, size_t *out_size)
{
    return guarded([&] {
        REQUIRE(c && ksk && out && out_size && c->K >= 2, "bad argument");
        sealio::KSwitchKeys k;
        std::memcpy(k.parms_id, c->chain[0].parms_id, 32);
        k.version_major = (uint8_t)version_major; k.version_minor = (uint8_t)version_minor;
        k.keys.resize(1);
        const size_t per = 2 * c->K * c->n;
        for (size_t j = 0; j + 1 < c->K; j++) {
            sealio::Ciphertext ct;
            std::memcpy(ct.parms_id, c->chain[0].parms_id, 32);
            ct.is_ntt_form = 1; ct.size = 2; ct.poly_modulus_degree = c->n; ct.coeff_modulus_size = c->K;
            ct.version_major = k.version_major; ct.version_minor = k.version_minor;
            ct.data.assign(ksk + j * per, ksk + (j + 1) * per);
            if (seeds) { ct.seeded = true; std::memcpy(ct.seed, seeds + 8 * j, 64); }
            k.keys[0].push_back(std::move(ct));
        }
        wire_out(sealio::save_kswitch_keys(k, (uint8_t)compr_mode), out, out_size);
    });
}
// ---- Receiver::RunQuery from the wire (receiver_osn.cpp:160-364 + query.cpp:44-80 + result_package.cpp:29-76), without SEAL
namespace {
// The query of one request, decoded onto engine E's device: relinearisation keys in apsu_he_relin_upload's layout (empty without
// key switching) and the source ciphertexts of the bundle indices `idx`, in (index, ascending exponent) order.
// seal::is_data_valid_for: every word of limb j of every polynomial is below q_j
static void check_residues(const u64 *data, size_t polys, const u64 *q, size_t limbs, size_t n, const char *what)
{
    for (size_t p = 0; p < polys; p++)
        for (size_t j = 0; j < limbs; j++) {
            const u64 *row = data + (p * limbs + j) * n, qj = q[j];
            u64 bad = 0;
            for (size_t k = 0; k < n; k++) bad |= (u64)(row[k] >= qj);
            if (bad) throw std::invalid_argument(what);
        }
}
struct DecodedQuery {
    std::vector<uint64_t> relin_flat;
    // seeded RelinKeys left for the device to expand (keys_on_device): c1 of key d sits at word key_c1_at[i] of relin_flat / of the uploaded keys
    std::vector<uint64_t> key_seeds;
    std::vector<size_t> key_c1_at;
    std::vector<const u64 *> src;            // device pointers into the engine's staging buffer: [idx][source][2][first_L][n]
};
// dev / host: the engine's staging (Engine::wire_stage, at least idx.size() * query_powers.size() ciphertexts), caller holds E.wire_mutex()
void decode_query(Engine &E, const apsu_he_seal_ctx *sc, const uint8_t *request, size_t request_size, const std::vector<uint32_t> &idx, DecodedQuery &out,
                  bool keys_on_device, u64 *dev, u64 *host)
{
    const PSUParams *psu = E.psu();
    REQUIRE(psu, "context was created without PSUParams");
    const HeParams &hp = E.he();
    REQUIRE(sc->n == hp.n && sc->K == (size_t)hp.K && sc->t == hp.t, "the SEAL context belongs to other parameters");
    const size_t n = hp.n;
    const int first = hp.first_chain_idx;
    const size_t Lf = (size_t)first + 1, ct_words = 2 * Lf * n;
    // Query::Query (receiver/apsu/query.cpp:44-80): relin keys, then one ciphertext per (exponent, bundle index)
    const wire::QueryRequest q = wire::parse_query_request(request, request_size);
    const auto &want = psu->query_params.query_powers;
    if (q.parts.size() != want.size()) throw std::invalid_argument("query powers do not match the parameters (query.cpp:63-68)");
    std::vector<const wire::QueryPart *> parts;                    // ascending exponent = the PowersDag's source order
    for (uint32_t e : want) {
        const wire::QueryPart *hit = nullptr;
        for (const auto &p : q.parts) if (p.exponent == e) hit = &p;
        if (!hit) throw std::invalid_argument("query powers do not match the parameters (query.cpp:63-68)");
        if (hit->cts.size() != psu->bundle_idx_count) throw std::invalid_argument("one ciphertext per bundle index expected (query.cpp:69-74)");
        parts.push_back(hit);
    }
    if (hp.using_keyswitching && !q.has_relin_keys) throw std::invalid_argument("the query carries no relinearization keys");
    // every object of the request is decoded on its own (host threads): the relinearisation keys (task 0) and one task per ciphertext
    const size_t n_cts = idx.size() * parts.size();
    struct OnDevice {                                              // the copies below are queued on E's stream: its device must be the current one
        int prev = -1;
        explicit OnDevice(int d) { int cur = -1; if (hipGetDevice(&cur) == hipSuccess && cur != d) { prev = cur; (void)hipSetDevice(d); } }
        ~OnDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
    } on_device(E.device());
    std::vector<unsigned char> seeded(n_cts, 0);
    std::vector<uint64_t> seed_of(n_cts * 8, 0);
    parallel_for(n_cts + 1, [&](size_t task) {
        if (task == 0) {
            if (!hp.using_keyswitching) return;
            // (expanding the seeded halves of the keys on the host is ~0.5 ms per key on this call's critical path; the single-device caller
            //  leaves it to apsu_he_seed_expand's kernels on the uploaded keys)
            const sealio::KSwitchKeys kk = sealio::load_kswitch_keys(q.relin_keys.p, q.relin_keys.n, sc->chain, nullptr, !keys_on_device);
            if (std::memcmp(kk.parms_id, sc->chain[0].parms_id, 32)) throw std::invalid_argument("RelinKeys were generated for other encryption parameters");
            out.relin_flat = sealio::relin_keys_layout(kk, sc->K, n);
            // (is_valid_for on the keys as well; a seeded entry's c1 is still zero here and sampled below [0, q) by construction)
            for (size_t d = 0; d + 1 < sc->K; d++)
                check_residues(out.relin_flat.data() + d * 2 * sc->K * n, 2, hp.key_q.data(), sc->K, n, "RelinKeys hold a coefficient outside [0, q)");
            if (keys_on_device && !kk.keys.empty())
                for (size_t d = 0; d < kk.keys[0].size(); d++)
                    if (kk.keys[0][d].seeded) {
                        if (std::memcmp(kk.keys[0][d].parms_id, sc->chain[0].parms_id, 32)) throw std::invalid_argument("a seeded RelinKeys entry is not at the key level");
                        out.key_seeds.insert(out.key_seeds.end(), kk.keys[0][d].seed, kk.keys[0][d].seed + 8);
                        out.key_c1_at.push_back((d * 2 + 1) * sc->K * n);
                    }
            return;
        }
        const size_t k = task - 1, b = k / parts.size(), s2 = k % parts.size();
        const wire::Span blob = parts[s2]->cts[idx[b]];
        const sealio::Ciphertext ct = sealio::load_ciphertext(blob.p, blob.n, sc->chain, nullptr, false);
        // (SEALObject::extract -> is_valid_for in the reference, seal_object.h:161-219: dimensions must be the level's, not the peer's claim)
        if (std::memcmp(ct.parms_id, seal_level(sc, first).parms_id, 32) || ct.size != 2 || ct.is_ntt_form || ct.poly_modulus_degree != n ||
            ct.coeff_modulus_size != Lf || ct.data.size() != ct_words)
            throw std::invalid_argument("query ciphertext is not a fresh size-2 ciphertext at the first data level");
        // ... and every coefficient a canonical residue of its limb's prime (seal::is_data_valid_for, part of is_valid_for): the engine's
        // lazy transforms take source limbs as they are (Engine::run_dag, ntt_gather_nored_ok), a word >= q_j would overflow their range
        check_residues(ct.data.data(), ct.seeded ? 1 : 2, hp.key_q.data(), Lf, n, "query ciphertext holds a coefficient outside [0, q)");
        std::memcpy(host + k * ct_words, ct.data.data(), (ct.seeded ? ct_words / 2 : ct_words) * sizeof(u64));
        seeded[k] = ct.seeded ? 1 : 0;
        if (ct.seeded) std::memcpy(&seed_of[k * 8], ct.seed, 64);
    });
    if (idx.empty()) return;
    // ciphertexts: c0 (and c1 when the object is not seeded) through page-locked memory, seeded c1 expanded on the device
    std::vector<uint64_t> seeds;
    std::vector<u64 *> c1;
    out.src.resize(n_cts);
    hipStream_t st = E.stream();
    for (size_t k = 0; k < n_cts; k++) {
        u64 *d = dev + k * ct_words;
        const size_t words = seeded[k] ? ct_words / 2 : ct_words;
        if (hipMemcpyAsync(d, host + k * ct_words, words * sizeof(u64), hipMemcpyHostToDevice, st) != hipSuccess) throw HipError("upload of a query ciphertext failed");
        if (seeded[k]) { seeds.insert(seeds.end(), seed_of.begin() + k * 8, seed_of.begin() + k * 8 + 8); c1.push_back(d + ct_words / 2); }
        out.src[k] = d;
    }
    if (!c1.empty()) E.seed_expand(first, (int)c1.size(), seeds.data(), c1.data());
    E.wait();                                                       // the page-locked staging may be refilled by the next query
}

// ResultPackage of one BinBundle (receiver_osn.cpp:507-539): its result ciphertext saved at the last level
void result_package(const apsu_he_seal_ctx *sc, size_t n, uint32_t bundle_idx, uint32_t cache_idx, const u64 *row, uint32_t polys, int compr_mode,
                    uint8_t **package, size_t *package_size)
{
    sealio::Ciphertext rc;
    std::memcpy(rc.parms_id, seal_level(sc, 0).parms_id, 32);
    rc.size = polys; rc.poly_modulus_degree = n; rc.coeff_modulus_size = 1;
    rc.data.assign(row, row + (size_t)polys * n);
    const std::vector<uint8_t> body = sealio::save_ciphertext(rc, (uint8_t)compr_mode);
    wire::ResultPackage rp;
    rp.bundle_idx = bundle_idx; rp.cache_idx = cache_idx;
    rp.psu_result = wire::Span{ body.data(), body.size() };
    wire_out(wire::build_result_package(rp), package, package_size);
}
} // namespace

int apsu_he_run_query_request(apsu_he_ctx *c, const apsu_he_seal_ctx *sc, const uint8_t *request, size_t request_size,
                              const apsu_he_bundle *const *bundles, int count, const uint64_t *const *masks, int masks_on_device,
                              int result_compr_mode, uint8_t **packages, size_t *package_sizes)
{
    return guarded([&] {
        REQUIRE(c && sc && request && count >= 0 && (count == 0 || (bundles && masks && packages && package_sizes)), "null argument");
        Engine &E = *c->eng;
        std::lock_guard<std::mutex> one_query(E.wire_mutex());             // the staging buffers serve one query at a time
        // the bundle indices the given BinBundles need, ascending
        std::vector<uint32_t> idx;
        for (int i = 0; i < count; i++) { REQUIRE(bundles[i], "null bundle"); idx.push_back(bundles[i]->b->bundle_idx); }
        std::sort(idx.begin(), idx.end());
        idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
        const PSUParams *psu = E.psu();
        REQUIRE(psu, "context was created without PSUParams");
        // one staging area for the whole query, kept across calls: [source ciphertexts | masks | results], device and page-locked host
        const size_t n = E.he().n, R = E.result_polys();                // 2 with key switching; longer results without
        const size_t ct_words = (size_t)2 * (E.he().first_chain_idx + 1) * n;
        const size_t src_words = idx.size() * psu->query_params.query_powers.size() * ct_words;
        const size_t mask_words = masks_on_device ? 0 : (size_t)count * n, res_words = (size_t)count * R * n;
        u64 *dev = nullptr, *host = nullptr;
        E.wire_stage((src_words + mask_words + res_words) * sizeof(u64) + 64, &dev, &host);
        DecodedQuery dq;
        decode_query(E, sc, request, request_size, idx, dq, true, dev, host);
        if (idx.empty()) return;
        struct OnDevice {
            int prev = -1;
            explicit OnDevice(int d) { int cur = -1; if (hipGetDevice(&cur) == hipSuccess && cur != d) { prev = cur; (void)hipSetDevice(d); } }
            ~OnDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
        } on_device(E.device());
        std::unique_ptr<RelinKeys> rk;
        if (!dq.relin_flat.empty()) {
            rk = E.upload_relin_keys(dq.relin_flat.data());
            if (!dq.key_c1_at.empty()) {                                 // c1 of the seeded keys: sampled by the device, in place
                std::vector<u64 *> dst;
                for (size_t at : dq.key_c1_at) dst.push_back(rk->data.u() + at);
                E.seed_expand(-1, (int)dst.size(), dq.key_seeds.data(), dst.data());
            }
        }
        std::unique_ptr<Powers> pw = E.compute_powers(idx.data(), (int)idx.size(), dq.src.data(), true, rk.get());
        std::vector<const Bundle *> bs(count);
        for (int i = 0; i < count; i++) bs[i] = bundles[i]->b.get();
        // masks through the page-locked staging in one copy (28 pageable 64 KiB copies cost more than the evaluation's launch overhead)
        std::vector<const u64 *> mptr(count);
        if (masks_on_device) {
            for (int i = 0; i < count; i++) mptr[i] = masks[i];
        } else {
            for (int i = 0; i < count; i++) { REQUIRE(masks[i], "null mask"); std::memcpy(host + src_words + (size_t)i * n, masks[i], n * sizeof(u64)); mptr[i] = dev + src_words + (size_t)i * n; }
            if (hipMemcpyAsync(dev + src_words, host + src_words, mask_words * sizeof(u64), hipMemcpyHostToDevice, E.stream()) != hipSuccess)
                throw HipError("upload of the masks failed");
        }
        u64 *res_dev = dev + src_words + mask_words, *res_host = host + src_words + mask_words;
        E.eval_bundles(bs.data(), count, *pw, rk.get(), mptr.data(), true, res_dev, true);
        if (hipMemcpyAsync(res_host, res_dev, res_words * sizeof(u64), hipMemcpyDeviceToHost, E.stream()) != hipSuccess) throw HipError("download of the results failed");
        E.wait();
        E.recycle_powers(std::move(pw));                                // the next query takes the buffers from the pool (no hipMalloc / hipFree per query)
        for (int i = 0; i < count; i++) { packages[i] = nullptr; package_sizes[i] = 0; }
        try {
            parallel_for((size_t)count, [&](size_t i) {
                result_package(sc, n, bs[i]->bundle_idx, bs[i]->cache_idx, res_host + i * R * n, E.result_size(*bs[i]), result_compr_mode, &packages[i],
                               &package_sizes[i]);
            });
        } catch (...) { for (int i = 0; i < count; i++) { std::free(packages[i]); packages[i] = nullptr; } throw; }
    });
}

// The same for the multi-device handle: the query is decoded once onto the first device (seeded c1 expanded there), the relinearisation
// keys go to every device, the other devices fetch the ciphertexts of their bundle indices over xGMI (IO_SRC_ON_DEVICE), and every
// BinBundle registered in the handle gets its ResultPackage, in id order.  masks: one per bundle id, host memory.
int apsu_he_multi_run_query_request(apsu_he_multi *m, const apsu_he_seal_ctx *sc, const uint8_t *request, size_t request_size,
                                    const uint64_t *const *masks, int result_compr_mode, uint8_t **packages, size_t *package_sizes, int capacity)
{
    return guarded([&] {
        REQUIRE(m && sc && request, "null argument");
        MultiEngine &M = *m->m;
        const int count = M.bundle_count();
        REQUIRE(count == 0 || (masks && packages && package_sizes && capacity >= count), "one mask and one package slot per registered BinBundle are needed");
        Engine &E = M.engine(0);
        std::lock_guard<std::mutex> one_query(E.wire_mutex());
        std::vector<uint32_t> idx(M.psu().bundle_idx_count);            // eval_all takes the sources of every bundle index
        for (uint32_t b = 0; b < idx.size(); b++) idx[b] = b;
        const size_t ct_words = (size_t)2 * (E.he().first_chain_idx + 1) * E.he().n;
        u64 *dev = nullptr, *host = nullptr;
        E.wire_stage(idx.size() * M.psu().query_params.query_powers.size() * ct_words * sizeof(u64) + 64, &dev, &host);
        DecodedQuery dq;
        decode_query(E, sc, request, request_size, idx, dq, true, dev, host);
        if (!count) return;
        if (!dq.relin_flat.empty())                                       // every device samples the seeded halves of its copy itself
            M.upload_relin_keys_seeded(dq.relin_flat.data(), dq.key_seeds.data(), dq.key_c1_at.data(), (int)dq.key_c1_at.size());
        const size_t n = E.he().n, R = E.result_polys();
        std::vector<u64> out((size_t)count * R * n);
        M.eval_all(dq.src.data(), masks, out.data(), -1, MultiEngine::IO_SRC_ON_DEVICE, 0);
        for (int i = 0; i < count; i++) { packages[i] = nullptr; package_sizes[i] = 0; }
        try {
            parallel_for((size_t)count, [&](size_t i) {
                const Bundle &b = M.bundle((int)i);
                result_package(sc, n, b.bundle_idx, b.cache_idx, out.data() + i * R * n, E.result_size(b), result_compr_mode, &packages[i], &package_sizes[i]);
            });
        } catch (...) { for (int i = 0; i < count; i++) { std::free(packages[i]); packages[i] = nullptr; } throw; }
    });
}

// the round-2 entry points (no context: unseeded objects only; zlib bodies are inflated)
int apsu_he_wire_seal_ct_save(const uint64_t parms_id[4], int is_ntt_form, uint64_t ct_size, uint64_t poly_modulus_degree,
                              uint64_t coeff_modulus_size, uint64_t correction_factor, double scale, const uint64_t *data,
                              int version_major, int version_minor, uint8_t **out, size_t *out_size)
{
    return guarded([&] {
        REQUIRE(parms_id && data && out && out_size, "null argument");
        REQUIRE(ct_size <= 64 && coeff_modulus_size <= 64 && poly_modulus_degree <= (1u << 20), "implausible dimensions");
        sealio::Ciphertext ct;
        std::memcpy(ct.parms_id, parms_id, 32);
        ct.is_ntt_form = is_ntt_form ? 1 : 0; ct.size = ct_size; ct.poly_modulus_degree = poly_modulus_degree;
        ct.coeff_modulus_size = coeff_modulus_size; ct.correction_factor = correction_factor; ct.scale = scale;
        ct.version_major = (uint8_t)version_major; ct.version_minor = (uint8_t)version_minor;
        ct.data.assign(data, data + ct_size * coeff_modulus_size * poly_modulus_degree);
        wire_out(sealio::save_ciphertext(ct, sealio::COMPR_NONE), out, out_size);
    });
}
int apsu_he_wire_seal_ct_load(const uint8_t *buf, size_t size, uint64_t parms_id[4], int *is_ntt_form, uint64_t *ct_size,
                              uint64_t *poly_modulus_degree, uint64_t *coeff_modulus_size, uint64_t *correction_factor, double *scale,
                              uint64_t *data, size_t data_capacity_words, int *version_major, int *version_minor)
{
    return guarded([&] {
        REQUIRE(buf, "null argument");
        const sealio::Ciphertext ct = sealio::load_ciphertext(buf, size, {});
        if (parms_id) std::memcpy(parms_id, ct.parms_id, 32);
        if (is_ntt_form) *is_ntt_form = ct.is_ntt_form;
        if (ct_size) *ct_size = ct.size;
        if (poly_modulus_degree) *poly_modulus_degree = ct.poly_modulus_degree;
        if (coeff_modulus_size) *coeff_modulus_size = ct.coeff_modulus_size;
        if (correction_factor) *correction_factor = ct.correction_factor;
        if (scale) *scale = ct.scale;
        if (version_major) *version_major = ct.version_major;
        if (version_minor) *version_minor = ct.version_minor;
        if (data) {
            REQUIRE(data_capacity_words >= ct.data.size(), "output buffer too small");
            std::memcpy(data, ct.data.data(), ct.data.size() * sizeof(uint64_t));
        }
    });
}

const char *apsu_he_phase_name(int phase) { return Engine::phase_name(phase); }
int apsu_he_phase_enable(apsu_he_ctx *c, int on)
{ return guarded([&] { REQUIRE(c, "null argument"); c->eng->phase_enable(on != 0); }); }
int apsu_he_phase_read(apsu_he_ctx *c, int phase, uint64_t *count, double *avg_ms, double *min_ms, double *max_ms, int reset)
{
    return guarded([&] {
        REQUIRE(c && phase >= 0 && phase < Engine::PH_COUNT, "unknown phase");
        Engine::PhaseSummary all[Engine::PH_COUNT];
        c->eng->phase_read(all, false);
        const Engine::PhaseSummary &p = all[phase];
        if (count) *count = p.count;
        if (avg_ms) *avg_ms = p.count ? p.sum_ms / (double)p.count : 0.0;
        if (min_ms) *min_ms = p.min_ms;
        if (max_ms) *max_ms = p.max_ms;
        if (reset) c->eng->phase_read(nullptr, true);
    });
}

int apsu_he_debug_counters(apsu_he_ctx *c, uint64_t *out, int capacity)
{ return guarded([&] { REQUIRE(c && out && capacity >= 0, "null argument"); c->eng->counters_read(out, capacity); }); }

int apsu_he_profile_enable(apsu_he_ctx *c, int on)
{ return guarded([&] { REQUIRE(c, "null argument"); c->eng->profile_enable(on); }); }

int apsu_he_profile_read(apsu_he_ctx *c, double *ms, uint64_t *launches, uint64_t *units, int capacity, int reset)
{
    return guarded([&] {
        REQUIRE(c, "null argument");
        Engine::ProfStats st;
        c->eng->profile_read(&st, reset != 0);
        for (int i = 0; i < capacity && i < Engine::P_COUNT; i++) {
            if (ms) ms[i] = st.ms[i];
            if (launches) launches[i] = st.launches[i];
            if (units) units[i] = st.units[i];
        }
    });
}

} // extern "C"
