// See multi.h.  Replaces, for a node with several GPUs, the loop of Receiver::RunQuery that runs ComputePowers per
// bundle index and enqueues one ProcessBinBundleCache task per BinBundle (receiver/apsu/receiver_osn.cpp:320-364).
#include "multi.h"

#include <cstring>
#include <stdexcept>

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

void MultiEngine::clear_bundles()
{
    std::lock_guard<std::mutex> g(mu_);
    run_all([&](Dev &d) { d.bundles.clear(); d.ids.clear(); });
    where_.clear();
}

void MultiEngine::eval_all(const u64 *const *src_cts, const u64 *const *masks, u64 *out, int out_slot)
{
    std::lock_guard<std::mutex> g(mu_);
    if (out_slot >= (int)devs_.size()) throw std::invalid_argument("output device slot out of range");
    const size_t n = hp_.n, row = 2 * n;
    const uint32_t ns = devs_[0]->eng->dag().source_count();
    const int out_device = out_slot >= 0 ? devs_[out_slot]->device : -1;
    run_all([&](Dev &d) {
        const int cnt = (int)d.bundles.size();
        if (!cnt) return;
        // the bundle indices this device holds, ascending; only their powers are computed here
        std::vector<uint32_t> idx;
        for (auto &b : d.bundles) idx.push_back(b->bundle_idx);
        std::sort(idx.begin(), idx.end());
        idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
        std::vector<const u64 *> src;
        for (uint32_t b : idx) for (uint32_t s = 0; s < ns; s++) src.push_back(src_cts[(size_t)b * ns + s]);
        std::unique_ptr<Powers> pw = d.eng->compute_powers(idx.data(), (int)idx.size(), src.data(), false, d.rk.get());
        std::vector<const Bundle *> bl;
        std::vector<const u64 *> mk;
        for (int i = 0; i < cnt; i++) { bl.push_back(d.bundles[i].get()); mk.push_back(masks[d.ids[i]]); }
        if (out_device < 0) {
            // host destination: one D2H of this device's rows into pinned staging, scattered by bundle id
            const size_t bytes = (size_t)cnt * row * sizeof(u64);
            if (d.host_out_bytes < bytes) {
                if (d.host_out) (void)hipHostFree(d.host_out);
                d.host_out = nullptr; d.host_out_bytes = 0;
                HIP_CHECK(hipHostMalloc(&d.host_out, bytes));
                d.host_out_bytes = bytes;
            }
            u64 *stage = static_cast<u64 *>(d.host_out);
            d.eng->eval_bundles(bl.data(), cnt, *pw, d.rk.get(), mk.data(), false, stage, false);
            for (int i = 0; i < cnt; i++) std::memcpy(out + (size_t)d.ids[i] * row, stage + (size_t)i * row, row * sizeof(u64));
        } else {
            const size_t bytes = (size_t)cnt * row * sizeof(u64);
            if (d.out.bytes() < bytes) d.out.alloc(bytes);
            d.eng->eval_bundles(bl.data(), cnt, *pw, d.rk.get(), mk.data(), false, d.out.u(), true);
            // the gather: fixed-size rows to the output device (xGMI peer copies; same-device rows are plain copies)
            for (int i = 0; i < cnt; i++)
                HIP_CHECK(hipMemcpyPeerAsync(out + (size_t)d.ids[i] * row, out_device, d.out.u() + (size_t)i * row, d.device,
                                             row * sizeof(u64), d.eng->stream()));
            HIP_CHECK(hipStreamSynchronize(d.eng->stream()));
        }
        d.eng->recycle_powers(std::move(pw));
    });
}

} // namespace apsu_he
