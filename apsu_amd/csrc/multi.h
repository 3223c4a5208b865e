// Several GPUs of one node behind one handle: the in-process counterpart of Receiver::RunQuery's fan-out of BinBundle
// tasks (receiver/apsu/receiver_osn.cpp:320-364).  One Engine per device, one persistent host thread per device;
// BinBundles are the sharded unit: devices are assigned to bundle indices first, an index's BinBundles
// are split over its devices by cost ~ degree (longest-processing-time greedy); every device computes the powers of
// its own indices only, and the only data exchange of a query is the final gather of the fixed-size results.
#pragma once
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "db_file.h"
#include "engine.h"
#include "sharding.h"

namespace apsu_he {

class MultiEngine {
public:
    MultiEngine(const HeParams &hp, const PSUParams &psu, const std::vector<int> &devices);
    ~MultiEngine();
    MultiEngine(const MultiEngine &) = delete;
    MultiEngine &operator=(const MultiEngine &) = delete;

    int device_count() const { return (int)devs_.size(); }
    Engine &engine(int slot) { return *devs_.at(slot)->eng; }
    const PSUParams &psu() const { return psu_; }
    const HeParams &he() const { return hp_; }

    void upload_relin_keys(const u64 *ksk);                      // replicated on every device
    // the same with seeded keys: c1 of entry i (at word c1_at[i] of ksk) is sampled from seeds[i * 8 ..] on every device, in place
    void upload_relin_keys_seeded(const u64 *ksk, const u64 *seeds, const size_t *c1_at, int n_seeded);
    // DB placement: the bundle's id is its registration order (= its row in eval_all's output)
    int upload_bundle(int slot, uint32_t bundle_idx, uint32_t cache_idx, uint32_t n_coeffs, const u64 *const *coeff_ptrs,
                      const unsigned char *is_ntt);
    int random_bundle(int slot, uint32_t bundle_idx, uint32_t cache_idx, uint32_t degree, u64 seed);
    // N2: the whole DB from / to one file (db_file.h).  load_file places the file's BinBundles with the partition rule (spill pass
    // included) and every device reads its own shard from the shared mapping, all devices at once; ids = the file's table order,
    // appended to whatever is registered already.  Returns the number of BinBundles loaded.
    int load_file(const DbFile &f);
    void save_file(const std::string &path);                      // every registered BinBundle, in id order
    int bundle_count() const { return (int)where_.size(); }
    int bundle_device(int id) const { return where_.at(id).first; }
    const Bundle &bundle(int id) const { const auto &w = where_.at(id); return *devs_[w.first]->bundles[w.second]; }
    void clear_bundles();

    // One query.  src_cts[b * source_count + s]: ciphertexts of every bundle index (each device reads its own);
    // masks[id]: n words mod t; out: bundle_count * 2n words, row = bundle id — host memory when out_slot < 0, else device
    // memory on devices[out_slot] (gathered over xGMI: peer copies, or one RCCL all-gather with IO_GATHER_RCCL).
    // flags: where the caller's buffers live.  Host buffers are pageable unless flagged page-locked (IO_*_PINNED: DMA goes
    // straight from / to them; pageable ones are staged through the device's own page-locked area, copy and DMA pipelined).
    // IO_SRC_ON_DEVICE / IO_MASKS_ON_DEVICE: the pointers are device pointers on devices[in_slot] (other devices fetch
    // what they need with peer copies).  Everything of a device's share is queued on its streams before the first wait.
    enum : unsigned { IO_SRC_PINNED = 1, IO_MASKS_PINNED = 2, IO_OUT_PINNED = 4, IO_SRC_ON_DEVICE = 8, IO_MASKS_ON_DEVICE = 16,
                      IO_GATHER_RCCL = 32 };
    void eval_all(const u64 *const *src_cts, const u64 *const *masks, u64 *out, int out_slot, unsigned flags = 0, int in_slot = 0);
    // what the last device-side gather used: "peer" or "rccl" ("" before the first one / host destination)
    const char *last_gather() const { return last_gather_; }
    // phase timers (Engine::phase_*): RunQuery = host wall time of eval_all; the other two = the slowest device's
    void phase_enable(bool on);
    void phase_read(Engine::PhaseSummary *out, bool reset);

private:
    struct Dev {
        int device = 0;
        std::unique_ptr<Engine> eng;
        std::unique_ptr<RelinKeys> rk;
        std::vector<std::unique_ptr<Bundle>> bundles;
        std::vector<int> ids;                                     // global id of bundles[i]
        DevBuf out;                                               // [bundles][2n] results of this device
        void *host_out = nullptr;                                 // pinned staging of the same size
        size_t host_out_bytes = 0;
        DevBuf in;                                                // this device's query inputs: sources, then masks
        void *host_in = nullptr;                                  // pinned staging for pageable inputs
        size_t host_in_bytes = 0;
        DevBuf gath;                                              // RCCL all-gather destination [devices][max rows][2n]
        // worker
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<void()> job;
        bool has_job = false, done = false, quit = false;
        std::exception_ptr error;
    };
    void run_all(const std::function<void(Dev &)> &fn);
    static void worker(Dev *d);

    HeParams hp_;
    PSUParams psu_;
    std::vector<std::unique_ptr<Dev>> devs_;
    std::vector<std::pair<int, int>> where_;                      // id -> (slot, local position)
    std::mutex mu_;                                               // one query / placement call at a time
    const char *last_gather_ = "";
    // RCCL (librccl.so, loaded on first use): one communicator per device slot of this handle, null when unavailable
    struct Rccl;
    std::unique_ptr<Rccl> rccl_;
    bool rccl_tried_ = false;
    bool rccl_ready();
    bool phase_on_ = false;
    Engine::PhaseSummary run_query_;
};

} // namespace apsu_he
