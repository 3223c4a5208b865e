// See multi.h.  Replaces, for a node with several GPUs, the loop of Receiver::RunQuery that runs ComputePowers per
// bundle index and enqueues one ProcessBinBundleCache task per BinBundle (receiver/apsu/receiver_osn.cpp:320-364).
#include "multi.h"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <stdexcept>
#include <string>

namespace apsu_he {

void throw_hip(hipError_t e, const char *file, int line);
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__); } while (0)

void MultiEngine::worker(Dev *d)
{
    // the worker's current device is the Dev's for its whole life (HIP's current device is per thread)
    (void)hipSetDevice(d->device);
    std::unique_lock<std::mutex> lk(d->mu);
    for (;;) {
        d->cv.wait(lk, [&] { return d->has_job || d->quit; });
        if (d->quit) return;
        std::function<void()> job = std::move(d->job);
        d->has_job = false;
        lk.unlock();
        std::exception_ptr err;
        try { job(); } catch (...) { err = std::current_exception(); }
        lk.lock();
        d->error = err;
        d->done = true;
        d->cv.notify_all();
    }
}

MultiEngine::MultiEngine(const HeParams &hp, const PSUParams &psu, const std::vector<int> &devices) : hp_(hp), psu_(psu)
{
    if (devices.empty()) throw std::invalid_argument("no devices given");
    for (int dev : devices) {
        auto d = std::make_unique<Dev>();
        d->device = dev;
        d->eng = std::make_unique<Engine>(hp_, &psu_, dev);       // validates the device index
        devs_.push_back(std::move(d));
    }
    // results are gathered with peer copies when the caller wants them on one device
    for (size_t i = 0; i < devs_.size(); i++)
        for (size_t j = 0; j < devs_.size(); j++) {
            if (devs_[i]->device == devs_[j]->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devs_[i]->device, devs_[j]->device) == hipSuccess && can) {
                int prev = 0;
                (void)hipGetDevice(&prev);
                if (hipSetDevice(devs_[i]->device) == hipSuccess) {
                    hipError_t e = hipDeviceEnablePeerAccess(devs_[j]->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
                (void)hipSetDevice(prev);
            }
        }
    for (auto &d : devs_) d->th = std::thread(worker, d.get());
}

MultiEngine::~MultiEngine()
{
    rccl_.reset();                                               // communicators go before the engines' streams
    for (auto &d : devs_) {
        {
            std::lock_guard<std::mutex> g(d->mu);
            d->quit = true;
        }
        d->cv.notify_all();
        if (d->th.joinable()) d->th.join();
    }
    for (auto &d : devs_) {
        int prev = 0;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(d->device);
        if (d->host_out) (void)hipHostFree(d->host_out);
        if (d->host_in) (void)hipHostFree(d->host_in);
        d->in.release();
        d->gath.release();
        d->bundles.clear();
        d->rk.reset();
        d->out.release();
        d->eng.reset();
        (void)hipSetDevice(prev);
    }
}

// fn runs once per device on that device's worker thread; the first exception (by device order) is rethrown
void MultiEngine::run_all(const std::function<void(Dev &)> &fn)
{
    for (auto &d : devs_) {
        std::lock_guard<std::mutex> g(d->mu);
        Dev *dp = d.get();
        d->job = [&fn, dp] { fn(*dp); };
        d->has_job = true;
        d->done = false;
        d->error = nullptr;
        d->cv.notify_all();
    }
    std::exception_ptr first;
    for (auto &d : devs_) {
        std::unique_lock<std::mutex> lk(d->mu);
        d->cv.wait(lk, [&] { return d->done; });
        if (d->error && !first) first = d->error;
    }
    if (first) std::rethrow_exception(first);
}

void MultiEngine::upload_relin_keys(const u64 *ksk)
{
    std::lock_guard<std::mutex> g(mu_);
    run_all([&](Dev &d) { d.rk = d.eng->upload_relin_keys(ksk); });
}

void MultiEngine::upload_relin_keys_seeded(const u64 *ksk, const u64 *seeds, const size_t *c1_at, int n_seeded)
{
    std::lock_guard<std::mutex> g(mu_);
    run_all([&](Dev &d) {
        d.rk = d.eng->upload_relin_keys(ksk);
        if (n_seeded <= 0) return;
        std::vector<u64 *> dst(n_seeded);
        for (int i = 0; i < n_seeded; i++) dst[i] = d.rk->data.u() + c1_at[i];
        d.eng->seed_expand(-1, n_seeded, seeds, dst.data());
    });
}

int MultiEngine::upload_bundle(int slot, uint32_t bundle_idx, uint32_t cache_idx, uint32_t n_coeffs, const u64 *const *coeff_ptrs,
                               const unsigned char *is_ntt)
{
    std::lock_guard<std::mutex> g(mu_);
    Dev &d = *devs_.at(slot);
    d.bundles.push_back(d.eng->upload_bundle(bundle_idx, cache_idx, n_coeffs, coeff_ptrs, is_ntt));
    const int id = (int)where_.size();
    d.ids.push_back(id);
    where_.push_back({ slot, (int)d.bundles.size() - 1 });
    return id;
}

int MultiEngine::random_bundle(int slot, uint32_t bundle_idx, uint32_t cache_idx, uint32_t degree, u64 seed)
{
    std::lock_guard<std::mutex> g(mu_);
    Dev &d = *devs_.at(slot);
    d.bundles.push_back(d.eng->random_bundle(bundle_idx, cache_idx, degree, seed));
    const int id = (int)where_.size();
    d.ids.push_back(id);
    where_.push_back({ slot, (int)d.bundles.size() - 1 });
    return id;
}

int MultiEngine::load_file(const DbFile &f)
{
    std::lock_guard<std::mutex> g(mu_);
    f.check_parameters(*devs_[0]->eng);
    std::vector<ShardUnit> units(f.count());
    for (size_t i = 0; i < f.count(); i++) units[i] = ShardUnit{ f.entry(i).bundle_idx, f.entry(i).cache_idx, f.entry(i).degree };
    const PowersDag &dag = devs_[0]->eng->dag();
    const uint64_t cp_cost = 110u * (uint64_t)(dag.target_powers().size() - dag.source_count());   // apsu_he_compute_powers_cost
    const std::vector<int> slot = partition_units(units, psu_.bundle_idx_count, (int)devs_.size(), cp_cost);
    std::vector<std::vector<size_t>> mine(devs_.size());
    for (size_t i = 0; i < units.size(); i++) mine[slot[i]].push_back(i);
    const int base = (int)where_.size();
    std::vector<std::vector<std::unique_ptr<Bundle>>> loaded(devs_.size());
    run_all([&](Dev &d) {
        size_t me = 0;
        for (size_t k = 0; k < devs_.size(); k++) if (devs_[k].get() == &d) me = k;
        for (size_t i : mine[me]) loaded[me].push_back(f.load(*d.eng, i));
    });
    // all shards are on their devices: register in table order (a failed load above leaves the handle unchanged)
    std::vector<size_t> next(devs_.size(), 0);
    for (size_t i = 0; i < units.size(); i++) {
        Dev &d = *devs_[slot[i]];
        d.bundles.push_back(std::move(loaded[slot[i]][next[slot[i]]++]));
        d.ids.push_back(base + (int)i);
        where_.push_back({ slot[i], (int)d.bundles.size() - 1 });
    }
    return (int)units.size();
}

void MultiEngine::save_file(const std::string &path)
{
    std::lock_guard<std::mutex> g(mu_);
    std::vector<Engine *> engs;
    std::vector<const Bundle *> bs;
    for (const auto &w : where_) { engs.push_back(devs_[w.first]->eng.get()); bs.push_back(devs_[w.first]->bundles[w.second].get()); }
    db_file_save(path, engs.data(), bs.data(), bs.size());
}

void MultiEngine::clear_bundles()
{
    std::lock_guard<std::mutex> g(mu_);
    run_all([&](Dev &d) { d.bundles.clear(); d.ids.clear(); });
    where_.clear();
}

// ---- RCCL, loaded at run time: the library has no link-time dependency on it, and a node without it (or a device list
// that repeats a device, which a communicator cannot hold) falls back to peer copies
struct MultiEngine::Rccl {
    void *lib = nullptr;
    typedef int (*InitAll)(void **, int, const int *);
    typedef int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t);
    typedef int (*Destroy)(void *);
    typedef const char *(*ErrStr)(int);
    InitAll init_all = nullptr; AllGather all_gather = nullptr; Destroy destroy = nullptr, abort = nullptr; ErrStr err = nullptr;
    std::vector<void *> comms;
    std::atomic<bool> broken{ false };                           // a collective failed: communicators were aborted, fall back to peer copies
    std::mutex abort_mu;
    // Called from worker threads whose all-gather failed.  Exactly ONE thread aborts: several workers may fail in the same collective,
    // and ncclCommAbort must not run twice on a communicator nor next to a worker that is still reading its entry of `comms` --
    // every worker copies its communicator before the rendezvous (comm_of), the entries are cleared under the lock.
    void abort_all()
    {
        std::lock_guard<std::mutex> g(abort_mu);
        if (broken.exchange(true)) return;
        if (abort) for (void *&c : comms) if (c) { (void)abort(c); c = nullptr; }
    }
    void *comm_of(int slot) { std::lock_guard<std::mutex> g(abort_mu); return broken ? nullptr : comms[(size_t)slot]; }
    ~Rccl() { if (destroy) for (void *c : comms) if (c) (void)destroy(c); if (lib) dlclose(lib); }
};

bool MultiEngine::rccl_ready()
{
    if (rccl_ && rccl_->broken) rccl_.reset();                     // an earlier collective failed (eval_all): peer copies from now on
    if (rccl_tried_) return (bool)rccl_;
    rccl_tried_ = true;
    std::vector<int> devlist;
    for (auto &d : devs_) devlist.push_back(d->device);
    std::vector<int> uniq = devlist;
    std::sort(uniq.begin(), uniq.end());
    if (std::adjacent_find(uniq.begin(), uniq.end()) != uniq.end()) return false;        // a repeated device
    auto r = std::make_unique<Rccl>();
    r->lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!r->lib) r->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!r->lib) return false;
    r->init_all = (Rccl::InitAll)dlsym(r->lib, "ncclCommInitAll");
    r->all_gather = (Rccl::AllGather)dlsym(r->lib, "ncclAllGather");
    r->destroy = (Rccl::Destroy)dlsym(r->lib, "ncclCommDestroy");
    r->err = (Rccl::ErrStr)dlsym(r->lib, "ncclGetErrorString");
    r->abort = (Rccl::Destroy)dlsym(r->lib, "ncclCommAbort");
    if (!r->init_all || !r->all_gather || !r->destroy) return false;
    r->comms.assign(devlist.size(), nullptr);
    if (r->init_all(r->comms.data(), (int)devlist.size(), devlist.data()) != 0) { r->comms.clear(); return false; }
    rccl_ = std::move(r);
    return true;
}

void MultiEngine::phase_enable(bool on)
{
    std::lock_guard<std::mutex> g(mu_);
    phase_on_ = on;
    run_all([&](Dev &d) { d.eng->phase_enable(on); });
}

void MultiEngine::phase_read(Engine::PhaseSummary *out, bool reset)
{
    std::lock_guard<std::mutex> g(mu_);
    Engine::PhaseSummary worst[Engine::PH_COUNT];
    for (auto &d : devs_) {
        Engine::PhaseSummary one[Engine::PH_COUNT];
        d->eng->phase_read(one, reset);
        for (int i = 0; i < Engine::PH_COUNT; i++)
            if (one[i].count && (!worst[i].count || one[i].sum_ms / (double)one[i].count > worst[i].sum_ms / (double)worst[i].count)) worst[i] = one[i];
    }
    worst[Engine::PH_RUN_QUERY] = run_query_;
    if (out) for (int i = 0; i < Engine::PH_COUNT; i++) out[i] = worst[i];
    if (reset) run_query_ = Engine::PhaseSummary{};
}

static void ensure_pinned(void *&p, size_t &have, size_t need)
{
    if (have >= need) return;
    if (p) (void)hipHostFree(p);
    p = nullptr; have = 0;
    // visible to every device (the staging areas are read / written by kernels, not by copy engines)
    HIP_CHECK(hipHostMalloc(&p, need, hipHostMallocPortable | hipHostMallocMapped));
    have = need;
}

// dst[i] <- src[i] (`bytes` each) on a small persistent pool: one core moves ~10 GB/s, a query's ciphertexts are ~10 MB, and
// starting threads per query would cost as much as it saves
namespace {
class StagePool {
public:
    explicit StagePool(int n) { for (int i = 0; i < n; i++) th_.emplace_back([this] { loop(); }); }
    ~StagePool()
    {
        { std::lock_guard<std::mutex> g(mu_); quit_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run(const std::vector<std::pair<void *, const void *>> &jobs, size_t bytes)
    {
        if (jobs.empty()) return;
        std::unique_lock<std::mutex> callers(callers_mu_);       // one batch at a time (several device workers share the pool)
        {
            std::lock_guard<std::mutex> g(mu_);
            jobs_ = &jobs; bytes_ = bytes; next_ = 0; pending_ = jobs.size(); gen_++;
        }
        cv_.notify_all();
        work();                                                  // the caller copies too
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        jobs_ = nullptr;
    }
private:
    void work()
    {
        for (;;) {
            size_t i;
            const std::vector<std::pair<void *, const void *>> *jobs;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (!jobs_ || next_ >= jobs_->size()) return;
                i = next_++; jobs = jobs_;
            }
            std::memcpy((*jobs)[i].first, (*jobs)[i].second, bytes_);
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return quit_ || gen_ != seen; });
                if (quit_) return;
                seen = gen_;
            }
            work();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, callers_mu_;
    std::condition_variable cv_, done_;
    const std::vector<std::pair<void *, const void *>> *jobs_ = nullptr;
    size_t bytes_ = 0, next_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool quit_ = false;
};
StagePool &stage_pool() { static StagePool pool(5); return pool; }
} // namespace

static void parallel_stage(const std::vector<std::pair<void *, const void *>> &jobs, size_t bytes)
{
    if (jobs.size() * bytes < ((size_t)1 << 20)) { for (auto &j : jobs) std::memcpy(j.first, j.second, bytes); return; }
    stage_pool().run(jobs, bytes);
}

void MultiEngine::eval_all(const u64 *const *src_cts, const u64 *const *masks, u64 *out, int out_slot, unsigned flags, int in_slot)
{
    std::lock_guard<std::mutex> g(mu_);
    if (out_slot >= (int)devs_.size()) throw std::invalid_argument("output device slot out of range");
    const bool src_dev = flags & IO_SRC_ON_DEVICE, mask_dev = flags & IO_MASKS_ON_DEVICE;
    if ((src_dev || mask_dev) && (in_slot < 0 || in_slot >= (int)devs_.size())) throw std::invalid_argument("input device slot out of range");
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t n = hp_.n, row = (size_t)devs_[0]->eng->result_polys() * n;     // 2n with key switching (Engine::result_polys)
    const uint32_t ns = devs_[0]->eng->dag().source_count();
    const size_t src_words = (size_t)2 * (hp_.first_chain_idx + 1) * n;
    const int out_device = out_slot >= 0 ? devs_[out_slot]->device : -1;
    const int in_device = (src_dev || mask_dev) ? devs_[in_slot]->device : -1;
    size_t max_rows = 0;
    for (auto &d : devs_) max_rows = std::max(max_rows, d->bundles.size());
    const bool use_rccl = out_device >= 0 && (flags & IO_GATHER_RCCL) && rccl_ready();
    if (out_device >= 0) last_gather_ = use_rccl ? "rccl" : "peer";
    // Nothing is copied by a copy engine on this path.  Page-locked host memory and peer devices' memory are addressable by
    // kernels, so the kernels that consume / produce the query's buffers anyway (the source gather of ComputePowers, the
    // evaluation's epilogue) read and write them IN PLACE over PCIe / xGMI: 80 small copies per query (24 ciphertexts, 28
    // masks, 28 result rows at ~8 us each) become none.  Pageable host buffers go through the device's own page-locked area.
    // With the RCCL gather every worker must reach the collective or none may: a worker that fails before it (ComputePowers,
    // the evaluation, an allocation) would leave the others waiting in the all-gather for ever, with mu_ held.  So the compute part
    // is guarded per worker, the workers meet once, and the collective is entered only if nobody failed.
    struct Rendezvous {
        std::mutex m; std::condition_variable cv; size_t arrived = 0, expect; size_t failed = 0;
        explicit Rendezvous(size_t e) : expect(e) {}
        bool arrive(bool ok) {                                    // true iff every worker arrived without a failure
            std::unique_lock<std::mutex> lk(m);
            if (!ok) failed++;
            if (++arrived == expect) cv.notify_all(); else cv.wait(lk, [&] { return arrived == expect; });
            return failed == 0;
        }
    } meet(devs_.size());
    run_all([&](Dev &d) {
        const int cnt = (int)d.bundles.size();
        Engine &E = *d.eng;
        hipStream_t st = E.stream();
        std::exception_ptr compute_error;
        try {
        auto reachable = [&](int other_device) {                  // may kernels on d.device address memory of other_device?
            if (other_device == d.device) return true;
            int can = 0;
            return hipDeviceCanAccessPeer(&can, d.device, other_device) == hipSuccess && can != 0;
        };
        if (cnt) {
            E.set_async_results(true);                            // nothing of this query is waited for before its last kernel is queued
            // the bundle indices this device holds, ascending; only their powers are computed here
            std::vector<uint32_t> idx;
            for (auto &b : d.bundles) idx.push_back(b->bundle_idx);
            std::sort(idx.begin(), idx.end());
            idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
            const size_t n_src = idx.size() * ns;
            const size_t in_words = n_src * src_words + (size_t)cnt * n;
            const bool src_direct = src_dev ? reachable(in_device) : (flags & IO_SRC_PINNED) != 0;
            const bool mask_direct = mask_dev ? reachable(in_device) : (flags & IO_MASKS_PINNED) != 0;
            const bool stage_src = !src_dev && !src_direct, stage_mask = !mask_dev && !mask_direct;
            if (stage_src || stage_mask) ensure_pinned(d.host_in, d.host_in_bytes, in_words * sizeof(u64));
            if (((src_dev && !src_direct) || (mask_dev && !mask_direct)) && d.in.bytes() < in_words * sizeof(u64)) { E.drain(); d.in.alloc(in_words * sizeof(u64)); }
            u64 *stage = static_cast<u64 *>(d.host_in);
            // 1. the query ciphertexts, in the order ComputePowers takes them
            std::vector<const u64 *> src(n_src);
            std::vector<std::pair<void *, const void *>> stage_jobs;
            for (size_t i = 0; i < idx.size(); i++)
                for (uint32_t s = 0; s < ns; s++) {
                    const size_t k = i * ns + s;
                    const u64 *from = src_cts[(size_t)idx[i] * ns + s];
                    if (src_direct) src[k] = from;
                    else if (src_dev) {                           // a device without peer access to the holder: copy engine
                        HIP_CHECK(hipMemcpyPeerAsync(d.in.u() + k * src_words, d.device, from, in_device, src_words * sizeof(u64), st));
                        src[k] = d.in.u() + k * src_words;
                    } else { stage_jobs.push_back({ stage + k * src_words, from }); src[k] = stage + k * src_words; }
                }
            if (!stage_jobs.empty()) parallel_stage(stage_jobs, src_words * sizeof(u64));
            std::unique_ptr<Powers> pw = E.compute_powers(idx.data(), (int)idx.size(), src.data(), true, d.rk.get());
            // 2. the masks (read once, by the evaluation's last kernel); pageable ones are staged while ComputePowers runs
            std::vector<const Bundle *> bl;
            std::vector<const u64 *> mk(cnt);
            stage_jobs.clear();
            for (int i = 0; i < cnt; i++) {
                bl.push_back(d.bundles[i].get());
                const u64 *from = masks[d.ids[i]];
                if (mask_direct) mk[i] = from;
                else if (mask_dev) {
                    u64 *to = d.in.u() + n_src * src_words + (size_t)i * n;
                    HIP_CHECK(hipMemcpyPeerAsync(to, d.device, from, in_device, n * sizeof(u64), st));
                    mk[i] = to;
                } else { u64 *sm = stage + n_src * src_words + (size_t)i * n; stage_jobs.push_back({ sm, from }); mk[i] = sm; }
            }
            if (!stage_jobs.empty()) parallel_stage(stage_jobs, n * sizeof(u64));
            // 3. results: written in place by the epilogue kernel wherever kernels can reach the destination
            const size_t bytes = (size_t)cnt * row * sizeof(u64);
            std::vector<u64 *> rows(cnt);
            bool scatter_host = false, peer_copy = false;
            if (use_rccl) {
                if (d.out.bytes() < max_rows * row * sizeof(u64)) { E.drain(); d.out.alloc(max_rows * row * sizeof(u64)); }
                for (int i = 0; i < cnt; i++) rows[i] = d.out.u() + (size_t)i * row;
            } else if (out_device < 0 && (flags & IO_OUT_PINNED)) {
                for (int i = 0; i < cnt; i++) rows[i] = out + (size_t)d.ids[i] * row;
            } else if (out_device < 0) {
                ensure_pinned(d.host_out, d.host_out_bytes, bytes);
                for (int i = 0; i < cnt; i++) rows[i] = static_cast<u64 *>(d.host_out) + (size_t)i * row;
                scatter_host = true;
            } else if (reachable(out_device)) {
                for (int i = 0; i < cnt; i++) rows[i] = out + (size_t)d.ids[i] * row;      // xGMI peer writes: the gather is the store
            } else {
                if (d.out.bytes() < bytes) { E.drain(); d.out.alloc(bytes); }
                for (int i = 0; i < cnt; i++) rows[i] = d.out.u() + (size_t)i * row;
                peer_copy = true;
            }
            E.eval_bundles(bl.data(), cnt, *pw, d.rk.get(), mk.data(), true, nullptr, true, rows.data());
            if (peer_copy)
                for (int i = 0; i < cnt; i++)
                    HIP_CHECK(hipMemcpyPeerAsync(out + (size_t)d.ids[i] * row, out_device, rows[i], d.device, row * sizeof(u64), st));
            if (!use_rccl) {
                HIP_CHECK(hipStreamSynchronize(st));
                // the query's work is complete: a source coefficient outside [0, q) is THIS call's error (round 6: it used to surface at a
                // later, valid query's next wait)
                if (E.take_bad_source()) throw std::invalid_argument("a source ciphertext of apsu_he_eval_all holds a coefficient outside [0, q): "
                                                                     "the results computed from it are not valid (seal::is_data_valid_for)");
            }
            if (scatter_host) {
                stage_jobs.clear();
                for (int i = 0; i < cnt; i++) stage_jobs.push_back({ out + (size_t)d.ids[i] * row, rows[i] });
                parallel_stage(stage_jobs, row * sizeof(u64));
            }
            E.recycle_powers(std::move(pw));
        }
        if (use_rccl) {
            // Everything that may wait or throw happens HERE, in front of the rendezvous: between the rendezvous and the collective a
            // worker that leaves strands its peers inside ncclAllGather (round 6).
            const size_t need = devs_.size() * max_rows * row * sizeof(u64);
            if (d.gath.bytes() < need) { E.drain(); d.gath.alloc(need); }
            if (d.out.bytes() < max_rows * row * sizeof(u64)) { E.drain(); d.out.alloc(max_rows * row * sizeof(u64)); }
        }
        } catch (...) { compute_error = std::current_exception(); }
        if (use_rccl) {
            int my_slot = 0;
            for (size_t i = 0; i < devs_.size(); i++) if (devs_[i].get() == &d) my_slot = (int)i;
            void *const my_comm = rccl_->comm_of(my_slot);        // taken before the rendezvous: a failing peer clears the table
            const bool all_ok = meet.arrive(!compute_error && my_comm != nullptr);
            if (compute_error) std::rethrow_exception(compute_error);
            if (!all_ok) { (void)hipStreamSynchronize(st); throw std::runtime_error("another device failed before the RCCL gather; nothing was gathered"); }
            // ONE all-gather of max_rows fixed-size rows per device, then the output device places them by id
            const int slot = my_slot;
            const int rc = rccl_->all_gather(d.out.u(), d.gath.u(), max_rows * row, /* ncclUint64 */ 5, my_comm, st);
            if (rc != 0) {                                        // the peers may already sit in the collective: abort every communicator so they return
                const std::string why = rccl_->err ? rccl_->err(rc) : "?";
                rccl_->abort_all();
                throw std::runtime_error("RCCL all-gather failed: " + why);
            }
            if (d.device == out_device && slot == out_slot) {
                for (size_t s2 = 0; s2 < devs_.size(); s2++)
                    for (size_t i = 0; i < devs_[s2]->ids.size(); i++)
                        HIP_CHECK(hipMemcpyAsync(out + (size_t)devs_[s2]->ids[i] * row, d.gath.u() + (s2 * max_rows + i) * row, row * sizeof(u64),
                                                 hipMemcpyDeviceToDevice, st));
            }
            HIP_CHECK(hipStreamSynchronize(st));
            // every rank has left the all-gather: only now may this rank report its query's invalid source (thrown earlier, it would have
            // left the peers inside the collective)
            if (E.take_bad_source()) throw std::invalid_argument("a source ciphertext of apsu_he_eval_all holds a coefficient outside [0, q): "
                                                                 "the results computed from it are not valid (seal::is_data_valid_for)");
        } else if (compute_error) std::rethrow_exception(compute_error);
    });
    if (phase_on_) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        run_query_.min_ms = run_query_.count ? std::min(run_query_.min_ms, ms) : ms;
        run_query_.max_ms = run_query_.count ? std::max(run_query_.max_ms, ms) : ms;
        run_query_.sum_ms += ms;
        run_query_.count++;
    }
}

} // namespace apsu_he
