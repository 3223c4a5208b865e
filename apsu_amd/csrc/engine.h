// Device-resident query-evaluation engine: the MI355X replacement for the seal::Evaluator calls
// made by apsu::receiver::Receiver::ComputePowers (receiver/apsu/receiver_osn.cpp:395-488) and
// BatchedPlaintextPolyn::eval / eval_patstock (receiver/apsu/bin_bundle.cpp:106-174,192-360).
//
// Tier 1 = one method per Evaluator call (host buffers in SEAL's [poly][limb][coeff] order),
// used for parity testing and incremental adoption.  Tier 2 = the fused, HBM-resident path:
// BinBundle plaintexts are uploaded once, ComputePowers runs level-synchronously over the
// PowersDag for all bundle indices at once, and every BinBundle of a query is evaluated in a
// few batched launches.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "device.h"
#include "params.h"
#include "powers_dag.h"

namespace apsu_he {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };

class DevBuf {                       // RAII device allocation
public:
    DevBuf() = default;
    explicit DevBuf(size_t bytes) { alloc(bytes); }
    ~DevBuf() { release(); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { release(); p_ = o.p_; bytes_ = o.bytes_; o.p_ = nullptr; o.bytes_ = 0; } return *this; }
    void alloc(size_t bytes);
    void release();
    u64 *u() const { return static_cast<u64 *>(p_); }
    void *p() const { return p_; }
    size_t bytes() const { return bytes_; }
private:
    void *p_ = nullptr;
    size_t bytes_ = 0;
};

struct RelinKeys {                   // [decomp K-1][2][K][n], NTT form (SEAL KSwitchKeys data, index 0)
    DevBuf data;
};

// One uploaded BinBundle cache = the batched matching polynomial of BatchedPlaintextPolyn
// (bin_bundle.h:52-134): coefficient d for all bins, stored per the ctor's rule (bin_bundle.cpp:385-420).
struct Bundle {
    uint32_t bundle_idx = 0, cache_idx = 0;
    uint32_t degree = 0;             // batched_coeffs.size() - 1
    bool use_ps = false;             // (ps_low_degree > 1) && (ps_low_degree < degree)  receiver_osn.cpp:520-522
    uint32_t H = 0, r = 0;           // degree / h, degree % h          (bin_bundle.cpp:225-227)
    int pt_level = 0;                // chain index of the NTT-form plaintexts
    size_t ntt_count = 0;            // NTT-form coefficients, packed in ascending degree order
    DevBuf ntt;                      // [ntt_count][pt_level+1][n]
    DevBuf lifted;                   // PS only: NTT(lift(a_{i*h})) at the high level, i = 1..H : [H][Lh][n]
    DevBuf a0;                       // constant coefficient, n words mod t
    // Row format of `ntt` and `lifted` (round 4).  false: dense 64-bit words as above.  true: every plaintext is a slot of
    // ntt_slot_bytes / lifted_slot_bytes bytes whose limb rows are bit-packed (DevLevel::mac_bits: 7 bytes per coefficient of a
    // 56-bit prime, 6.25 of a 50-bit one); both buffers carry 16 spare bytes behind the last slot (k_mac reads 16-byte windows).
    bool packed = false;
    size_t ntt_slot_bytes = 0, lifted_slot_bytes = 0;
    size_t db_bytes() const { return ntt.bytes() + lifted.bytes() + a0.bytes(); }
};

// Output of ComputePowers for a set of bundle indices (CiphertextPowers, receiver_osn.h:41).
struct Powers {
    int nb = 0;                                  // bundle indices held
    std::vector<uint32_t> bundle_indices;
    int low_level = 0, high_level = 0;
    uint32_t n_low = 0, n_high = 0;              // PS: l low powers, H high powers; no PS: all in `low`
    uint32_t polys = 2;                          // polynomials stored per power (the "2" of the layouts below).  More than 2 only without
                                                 // key switching: products are never relinearised, every power is stored zero-padded to
                                                 // the longest one's size rounded up to even (Engine::power_size gives the real sizes)
    DevBuf low;                                  // [idx][power-1][2][Ll][n]   NTT form
    DevBuf high;                                 // [idx][i-1][2][Lh][n]       coefficient form (power i*h)
    DevBuf hext;                                 // [idx][i-1][2][Eh][n]       extended + NTT form of the same
    // high/hext are produced on the engine's second stream when the PowersDag splits (see Engine::compute_powers);
    // consumers on the main stream wait for this event first.  Null / never recorded = already ordered.
    hipEvent_t high_ready = nullptr;
    bool high_async = false;
    bool low_async = false;                      // pipelined queries (Engine::compute_powers): the low powers come from the second stream too
    // recorded on the main stream behind the last evaluation that read these powers: when the buffer comes back from the pool,
    // the second stream may start writing its high half as soon as THAT evaluation is over -- it does not have to wait for whatever
    // else the main stream has queued since (Engine::compute_powers, inputs_ready_)
    mutable hipEvent_t last_use = nullptr;
    mutable bool last_use_set = false;
    // sequence number of the ComputePowers call that filled this buffer: the word a source coefficient outside [0, q) is reported in
    // (Engine::check_sources) carries it, so the report attaches to THIS query's evaluation / download
    uint32_t seq = 0;
    ~Powers() { if (high_ready) (void)hipEventDestroy(high_ready); if (last_use) (void)hipEventDestroy(last_use); }
    Powers() = default;
    Powers(const Powers &) = delete;
    Powers &operator=(const Powers &) = delete;
    int slot_of(uint32_t bundle_idx) const;
};

struct MacStream;
struct PsBatch;

class Engine {
public:
    Engine(const HeParams &hp, const PSUParams *psu, int device);
    ~Engine();

    const HeParams &he() const { return hp_; }
    const PSUParams *psu() const { return has_psu_ ? &psu_ : nullptr; }
    const PowersDag &dag() const { return dag_; }
    hipStream_t stream() const { return st_; }
    void sync();

    // ---------------- tier 1 (host pointers; each call is H2D, kernels, D2H)
    void transform_to_ntt(u64 *ct, int polys, int chain_idx);
    void transform_from_ntt(u64 *ct, int polys, int chain_idx);
    void multiply_plain_ntt(const u64 *ct, const u64 *pt_ntt, u64 *out, int polys, int chain_idx);
    void multiply_plain(const u64 *ct, const u64 *pt, size_t pt_coeffs, u64 *out, int polys, int chain_idx);
    void transform_plain_to_ntt(const u64 *pt, size_t pt_coeffs, u64 *out, int chain_idx);
    void add(u64 *acc, const u64 *x, int polys, int chain_idx);
    void add_plain(u64 *ct, const u64 *pt, size_t pt_coeffs, int chain_idx);
    void multiply(const u64 *a, const u64 *b, u64 *out3, int chain_idx);
    // Evaluator::multiply for operands of any size (nothing relinearised in between): out has size_a + size_b - 1 polynomials
    void multiply_sized(const u64 *a, int size_a, const u64 *b, int size_b, u64 *out, int chain_idx);
    void relinearize(u64 *ct3, const RelinKeys &rk, int chain_idx);
    void mod_switch_to_next(u64 *ct, int polys, int chain_idx);
    void clear_irrelevant_bits(u64 *ct, int polys);

    // ---------------- tier 2 (device resident)
    std::unique_ptr<RelinKeys> upload_relin_keys(const u64 *rk_host);
    std::unique_ptr<Bundle> upload_bundle(uint32_t bundle_idx, uint32_t cache_idx, uint32_t n_coeffs,
                                          const u64 *const *coeff_ptrs, const unsigned char *is_ntt);
    // Synthetic-DB helper for benchmarks: fills a bundle of the given degree with uniformly random
    // plaintext coefficients generated and transformed on the GPU (no host data).
    std::unique_ptr<Bundle> random_bundle(uint32_t bundle_idx, uint32_t cache_idx, uint32_t degree, u64 seed);
    // N1: BinBundle::regen_polyns + regen_plaintexts + BatchedPlaintextPolyn ctor on the GPU
    // (bin_bundle.cpp:366-430,934-1026): roots[bin * stride + r], r < counts[bin], are the field elements of a bin.
    std::unique_ptr<Bundle> build_bundle(uint32_t bundle_idx, uint32_t cache_idx, const u64 *roots, const uint32_t *counts,
                                         uint32_t bins, uint32_t stride);
    // N1, one step earlier: algebraize_item (common/apsu/util/db_encoding.cpp:209-256,360-366) for `count` hashed items of 16 bytes:
    // out[count][felts_per_item], felt j = bits [j*b, (j+1)*b) of the item's first item_bit_count bits, b = bit_count(t) - 1
    void algebraize_items(const unsigned char *items, size_t count, bool items_on_device, u64 *out, bool out_on_device);
    // N2: engine-native image of a BinBundle cache (raw limb arrays), replacing the flatbuffers +
    // SEAL-serialised blobs of ReceiverDB::save/Load (receiver_db.cpp:1182-1429) for the GPU-resident DB.
    size_t bundle_image_size(const Bundle &b) const;
    size_t save_bundle(const Bundle &b, unsigned char *buf, size_t capacity);
    std::unique_ptr<Bundle> load_bundle(const unsigned char *buf, size_t size);
    // N4: mask generation + block packing (receiver_osn.cpp:53-73,217-284) and the querier's
    // decrypt + decode + packing (result_package.cpp:175-213 ; sender_osn.cpp:675-700)
    void mask_generate(u64 seed, uint32_t count, u64 *masks_dev, u64 *values_host, u64 *blocks_host);
    // the same with the reference's generator: SEAL's Blake2xb PRNG under `seed`, starting at its first_value-th 32-bit output
    void mask_generate_blake2xb(const u64 seed[8], u64 first_value, uint32_t count, u64 *masks_dev, u64 *values_host, u64 *blocks_host);
    void decrypt_decode(const u64 *sk_ntt_host, const u64 *cts, bool on_device, uint32_t count, u64 *values_host, u64 *blocks_host);
    // N3 on the device: c1 of `count` seeded objects at chain_idx (-1 / K - 1: the key level) = util::sample_poly_uniform under
    // SEAL's Blake2xb generator seeded with seeds[i][8], written to the DEVICE buffers dst[i] ([L][n] words each)
    void seed_expand(int chain_idx, int count, const u64 *seeds, u64 *const *dst);
    // test hook: stored form of coefficient d.  kind: 0 = raw mod t (d = 0), 1 = NTT form at pt_level,
    // 2 = pre-lifted + NTT at the high level (coefficient-form a_{i*h}); returns words written
    size_t download_coeff(const Bundle &b, uint32_t d, u64 *out, size_t capacity, int *kind);
    // src[b * n_sources + s]: source power s (ascending power order) of bundle index bundle_indices[b],
    // size-2 coefficient-form ct at the first data level.  on_device: pointers are device pointers.
    std::unique_ptr<Powers> compute_powers(const uint32_t *bundle_indices, int nb, const u64 *const *src, bool on_device,
                                           const RelinKeys *rk);
    // masks[i]: n words mod t (host, or device if on_device).  out: count * 2n words (host or device).
    // out_rows (optional, implies device-accessible destinations): result i goes to out_rows[i] (2n words) instead of
    // out + i * 2n -- rows of another device's buffer (peer access) or of page-locked host memory are written in place.
    void eval_bundles(const Bundle *const *bundles, int count, const Powers &pw, const RelinKeys *rk,
                      const u64 *const *masks, bool masks_on_device, u64 *out, bool out_on_device, u64 *const *out_rows = nullptr);

    // Parameter sets with one coefficient prime have no key switching: the reference then never relinearises
    // (receiver_osn.cpp:416,430-432 ; bin_bundle.cpp:308-310), so powers and results have more than two polynomials.
    // power_size: polynomials of a target power after ComputePowers; result_size: of one BinBundle's result;
    // result_polys: the largest result_size over all degrees = polynomials per row of eval_bundles' output (2 with key switching).
    static constexpr uint32_t CT_SIZE_MAX = 16;          // SEAL_CIPHERTEXT_SIZE_MAX: Ciphertext::resize throws beyond it
    uint32_t power_size(uint32_t power) const;
    uint32_t result_size(const Bundle &b) const;
    uint32_t result_size_for(uint32_t degree) const;
    uint32_t result_polys() const { return result_polys_; }

    // introspection for tests
    size_t workspace_bytes() const { return arena_.bytes(); }
    // Host-side events that cost a query time without showing up in any kernel: host waits taken inside the engine,
    // job-table uploads (cache misses) and hits, workspace-arena growths, powers-buffer allocations, wraps of the pinned
    // staging area.  Steady state = only hits move.  (apsu_he_debug_counters)
    enum Counter { C_HOST_SYNC = 0, C_JOB_UPLOAD, C_JOB_HIT, C_ARENA_GROW, C_POWERS_ALLOC, C_STAGE_WRAP, C_JOB_REALLOC, C_PIPELINED, C_COUNT };
    void counters_read(uint64_t *out, int capacity) const { for (int i = 0; i < capacity && i < C_COUNT; i++) out[i] = counters_[i]; }

    // ---------------- per-kernel timing with HIP events on the engine's stream (bench.py roofline)
    enum ProfKind { P_NTT_FWD = 0, P_NTT_INV, P_MAC, P_BEHZ_EXT, P_TENSOR, P_BEHZ_FINISH, P_KEYSWITCH, P_MODSWITCH, P_OTHER, P_NTT_FUSED, P_COUNT };
    struct ProfStats { double ms[P_COUNT]; uint64_t launches[P_COUNT]; uint64_t units[P_COUNT]; };
    void profile_enable(int mode);      // 0 off, 1 every kernel class, 2 NTT launches only (cheapest)
    void profile_read(ProfStats *out, bool reset);
    // ---------------- phase timers under the reference's STOPWATCH names (receiver_osn.cpp:167,403,504): device time from HIP
    // events on the engine's streams.  RunQuery = start of a compute_powers call .. end of the last eval_bundles call before the
    // next compute_powers; ComputePowers ends when BOTH streams of a two-stream walk have finished.
    enum Phase { PH_RUN_QUERY = 0, PH_COMPUTE_POWERS, PH_PROCESS_BIN_BUNDLE_CACHE, PH_COUNT };
    struct PhaseSummary { uint64_t count = 0; double sum_ms = 0, min_ms = 0, max_ms = 0; };      // cli/common_utils.cpp:54-76 prints these
    static const char *phase_name(int phase);
    void phase_enable(bool on);
    void phase_read(PhaseSummary *out, bool reset);            // out[PH_COUNT]; waits for the recorded work
    // two-stream ComputePowers: -1 = default policy (on for one or two bundle indices), 0 = off, 1 = on
    void set_two_stream(int mode) { std::lock_guard<std::mutex> g(mu_); two_stream_mode_ = mode < 0 ? -1 : (mode ? 1 : 0); }
    // device-resident evaluation results without the closing stream synchronisation (see apsu_he_set_async_results)
    void set_async_results(bool on) { std::lock_guard<std::mutex> g(mu_); async_results_ = on; }
    // apsu_he_set_query_overlap.  0: off (the second stream waits for everything the main stream has queued).  1: the caller's inputs are
    // complete at call time -- the second stream waits only for the last reader of the powers buffer it reuses, and a ComputePowers
    // that finds an evaluation still running takes the pipelined walk.  2: as 1 without the pipelined walk.  3: as 1 with the
    // pipelined walk taken whether or not the device is busy (tests: the walk's event chain is then exercised deterministically).
    void set_query_overlap(int mode)
    {
        std::lock_guard<std::mutex> g(mu_);
        inputs_ready_ = mode != 0;
        pipe_cp_ = mode == 1 || mode == 3;
        force_pipe_ = mode == 3;
    }
    // tier 1 on device-resident operands: the per-method calls take device pointers and return with their work queued on the
    // engine's stream (no host round trip per Evaluator call); see apsu_he_set_tier1_on_device
    void set_tier1_on_device(bool on) { std::lock_guard<std::mutex> g(mu_); tier1_device_ = on; }
    void wait();                                                  // locked sync() + check_sources()
    void drain();                                                 // locked sync() alone: never throws for a query's data (buffer growth in callers)
    // Has any ComputePowers whose work is complete found a source coefficient outside [0, q) that no call has reported yet?  Clears it.
    // Never throws: for callers that must finish their own protocol first (MultiEngine::eval_all, in front of a collective).
    bool take_bad_source();
    // test hook: copy one computed power to the host (serialised with the other calls on this context)
    void download_power(const Powers &pw, uint32_t bundle_idx, uint32_t power, u64 *out, size_t capacity_words, int *chain_idx,
                        int *is_ntt);
    int device() const { return device_; }
    // Staging of apsu_he_run_query_request (c_api.cpp) kept across calls: a device buffer and a page-locked host buffer of at least
    // `bytes` (grown on demand; hipHostMalloc / hipMalloc per query cost milliseconds).  The caller holds wire_mutex() for the whole query.
    std::mutex &wire_mutex() { return wire_mu_; }
    void wire_stage(size_t bytes, u64 **device, u64 **pinned);

private:
    // arena (bump allocator reset per top-level operation)
    u64 *ws(size_t words);
    void ws_reset(size_t need_bytes_hint = 0);
    template <class T> const T *upload_jobs(const std::vector<T> &v);

    const DevLevel *dlevel(int chain_idx) const { return d_levels_.u() ? reinterpret_cast<const DevLevel *>(d_levels_.p()) + chain_idx : nullptr; }
    const LevelConstants &hlevel(int chain_idx) const { return hp_.level[chain_idx]; }
    const NttTable *tabs() const { return reinterpret_cast<const NttTable *>(d_tabs_.p()); }
    const DevKey *dkey() const { return reinterpret_cast<const DevKey *>(d_key_.p()); }
    const int *map_ct() const { return reinterpret_cast<const int *>(d_map_ct_.p()); }
    const int *map_ext(int chain_idx) const { return reinterpret_cast<const int *>(d_map_ext_.p()) + chain_idx * DMAXE; }
    // the same map for the inverse transform in front of a finish kernel: NTT_MAP_RAW set where that level's finish takes
    // the twist into its own constants (fast_finish), identical to map_ext otherwise
    const int *map_ext_fin(int chain_idx) const { return reinterpret_cast<const int *>(d_map_ext_fin_.p()) + chain_idx * DMAXE; }
    bool fast_finish(int chain_idx) const { const LevelConstants &h = hp_.level[chain_idx]; return h.L == h.nB && h.L <= 3; }
    const int *map_ks(int chain_idx) const { return reinterpret_cast<const int *>(d_map_ks_.p()) + chain_idx * (DMAXL + 1) * DMAXL; }
    const int *map_ksacc(int chain_idx) const { return reinterpret_cast<const int *>(d_map_ksacc_.p()) + chain_idx * (DMAXL + 1); }
    const int *map_ksacc_raw(int chain_idx) const { return reinterpret_cast<const int *>(d_map_ksacc_raw_.p()) + chain_idx * (DMAXL + 1); }

    // device-pointer building blocks
    void d_ntt(u64 *data, size_t count, const int *modmap, int period, bool inverse);
    void d_ntt_ct(u64 *data, size_t polys, int chain_idx, bool inverse) { d_ntt(data, polys * (chain_idx + 1), map_ct(), chain_idx + 1, inverse); }
    // BFV multiply of `njobs` (a, b) pairs given as ext-NTT operands; writes size-3 results
    // ext_out / n_ext: the first n_ext ciphertexts also get their BEHZ extension written to ext_out[b][2][E][n] (fused into
    // the mod-down; returns false when the level has no unrolled extension and the caller must run launch_behz_ext itself)
    // defer_moddown (round 6): the caller's next kernel performs the rounding mod-down itself (k_eval_epilogue<true>): the RAW inverse transforms
    // of the key-switch sums are left at *defer_moddown ([batch][2][L+1][n], workspace) and ct3 is NOT updated.  Refused (nullptr stored) when the
    // level has no RAW mod-down constants (L > 4).
    bool d_relinearize(u64 *ct3, size_t ct_stride, int batch, const RelinKeys &rk, int chain_idx, u64 *ext_out = nullptr, int n_ext = 0,
                       u64 **defer_moddown = nullptr);
    bool fuse_tail_ = true;           // eval_patstock: the last key switch's mod-down inside the epilogue kernel (APSU_HE_FUSE_TAIL=0: its own launch)
    void check_level(int chain_idx) const;
    // BEHZ steps 4-8 for operands given as ext-NTT polynomials [size][E][n]; out: [sa + sb - 1][L][n], coefficient form
    void d_multiply_sized(const u64 *ea, int sa, const u64 *eb, int sb, u64 *out, int chain_idx);
    // ComputePowers / eval / eval_patstock without key switching and with products (ciphertexts of any size): plain
    // compositions of the per-polynomial kernels, not tuned -- no shipped parameter set reaches them
    std::unique_ptr<Powers> compute_powers_nks(const uint32_t *bundle_indices, int nb, const u64 *const *src, bool on_device);
    void eval_bundles_nks(const Bundle *const *bundles, int count, const Powers &pw, const u64 *const *masks, bool masks_on_device,
                          u64 *out, bool out_on_device, u64 *const *out_rows);
    bool nks_ = false;                // no key switching AND the PowersDag has products
    bool nks_oversize_ = false;       // some product exceeds CT_SIZE_MAX polynomials: ComputePowers throws like SEAL's multiply
    std::vector<uint32_t> nks_size_;  // polynomials per target power (index = power; 0 = no target)
    uint32_t nks_S_ = 2;              // polynomials stored per power (largest size, rounded up to even)
    uint32_t result_polys_ = 2;

    HeParams hp_;
    PSUParams psu_;
    bool has_psu_ = false;
    PowersDag dag_;
    int device_ = 0;
    hipStream_t st_ = nullptr;
    std::mutex mu_;                   // ABI calls are serialised per context (thread-safe)

    DevBuf d_tabs_, d_tw_, d_levels_, d_key_, d_map_ct_, d_map_ext_, d_map_ext_fin_, d_map_ks_, d_map_ksacc_, d_map_ksacc_raw_, d_fin_, d_drop_, d_mdtw_;
    DevBuf arena_;
    size_t arena_off_ = 0;
    // Lanes: lane 0 = the main stream, lane 1 = a second stream with its own arena for work that is independent of
    // the main chain (the high-power half of ComputePowers).  st_/arena_/arena_off_/job_seq_ always describe the
    // CURRENT lane; switch_lane() parks them and loads the other lane's, so every helper works on either.
    struct Lane { hipStream_t st = nullptr; DevBuf arena; size_t off = 0, job_seq = 0; };
    Lane lanes_[3];                   // the lanes that are not current (the current one's state lives in st_ / arena_ / arena_off_ / job_seq_): 0 main,
                                      // 1 second stream (high-power chain; a queued query's whole ComputePowers), 2 the evaluation's side work
    int cur_lane_ = 0, overflow_lane_ = 0;
    void switch_lane(int lane);
    hipEvent_t ev_main_ = nullptr;    // main-stream progress marker the second stream waits on
    hipEvent_t ev_fork_ = nullptr, ev_side_ = nullptr, ev_intt_ = nullptr;   // eval_patstock's side lane: start marker on the main stream, end marker on the side stream, main-stream inverse transforms done
    bool inputs_ready_ = false;       // the caller's promise behind apsu_he_set_query_overlap: the second stream then waits for the last reader of its powers buffer only
    bool pipe_cp_ = true;             // queued queries: the whole ComputePowers on the second stream, next to the evaluation in front (set_query_overlap 1 / 3)
    bool force_pipe_ = false;         // ... whether or not an evaluation is still running (set_query_overlap 3)
    bool data_primes_narrow_ = false; // every key prime runs the transform without range control (ntt_is_narrow)
    size_t ntt_latency_limbs_ = 0;    // transform launches of at most this many limbs take the 8-coefficient-per-lane form (APSU_HE_NTT_LATENCY_LIMBS)
    bool eval_side_ = true;           // cf sums + i = 0 finish of eval_patstock on a side stream (APSU_HE_EVAL_SIDE=0: on the main stream)
    bool async_results_ = false;      // eval_bundles with device masks + device output returns once the work is queued
    // Asynchronous evaluations in flight: the host may run at most max_inflight_ queries ahead of the device.  Unbounded
    // run-ahead (20 queued queries = 1300 launches + 200 stream events) makes the HIP runtime block the host inside a
    // launch and drain its queues, which shows up as idle gaps on the device (profiles/r03_shard_probe.txt).
    std::vector<hipEvent_t> inflight_;
    size_t inflight_head_ = 0, inflight_count_ = 0;
    int max_inflight_ = 2;
    void throttle_inflight();
    void mark_inflight();
    // device-resident source ciphertexts are checked while they are gathered (k_copy_sources: every word below its limb's prime, SEAL's
    // is_data_valid_for); the kernel raises this word of page-locked host memory, sync() looks at it and throws std::invalid_argument
    // page-locked words written by k_copy_sources: slot (seq % BAD_SLOTS) holds the sequence number of the last ComputePowers that found
    // a source coefficient outside [0, q) there; bad_reported_ = what has been thrown already
    static constexpr int BAD_SLOTS = 16;
    unsigned *bad_source_ = nullptr;
    unsigned bad_reported_[BAD_SLOTS] = {};
    uint32_t query_seq_ = 0;
    void check_sources();                                         // any query
    void check_sources(const Powers &pw);                         // this query
    bool bad_source_pending(int slot, bool take);
    void *stage_ = nullptr;           // pinned host staging for job arrays
    size_t stage_bytes_ = 0, stage_off_ = 0;
    // job-array cache: the n-th upload of a top-level call usually carries the same bytes as in the
    // previous call of the same shape (workspace addresses are deterministic), so it is kept on the
    // device and the copy is skipped when the content hash matches.
    // Two ways per slot: a caller that still holds the previous query's powers while it computes the next one alternates
    // between two powers buffers, hence between two versions of every table that names them (tools/shard_probe.py).
    struct JobWay { DevBuf buf; std::vector<unsigned char> host; uint64_t stamp = 0; };
    struct JobSlot { JobWay way[4]; };                  // four-way: a stream of queries cycles through up to three powers buffers (compute_powers), whose addresses are in the tables
    uint64_t job_stamp_ = 0;
    std::vector<JobSlot> job_slots_;
    size_t job_seq_ = 0, job_seq_base_ = 0;   // slots [base, ..) belong to the running top-level op
    std::vector<std::unique_ptr<Powers>> powers_pool_;
public:
    void recycle_powers(std::unique_ptr<Powers> p);
private:
    std::vector<DevBuf> retired_;     // arenas replaced while kernels may still reference them
    uint64_t counters_[C_COUNT] = {};
    std::mutex wire_mu_;
    DevBuf wire_dev_;
    void *wire_pinned_ = nullptr;
    size_t wire_pinned_bytes_ = 0;

    // PowersDag schedule (slot order = depth, parents first, power)
    struct Sched {
        std::vector<uint32_t> slot_power;                 // slot -> power
        std::vector<int> slot_of;                         // power -> slot (-1 if absent)
        struct Level { int s0, s1, sp; };                 // slots [s0,s1) ; [s0,sp) are parents of later nodes
        std::vector<Level> levels;
        std::vector<std::array<int, 3>> nodes;           // per non-source slot: {slot, slot_p1, slot_p2}
        std::vector<uint32_t> low_powers, high_powers;    // target powers by final form
    } sched_, sched_low_, sched_high_;
    void mask_generate_impl(uint32_t count, u64 *masks_dev, u64 *values_host, u64 *blocks_host, const std::function<void(u64 *, size_t)> &fill);
    bool fuse_tensor_ = true;         // BEHZ step 4 is formed by the inverse transform's load (k_intt_tensor); off only for n = 32768, whose limb does not fit one workgroup
    bool force_per_term_ = false;     // APSU_HE_EVAL_PER_TERM: eval_patstock's products finished one by one (the fallback of the summed finish)
    bool seed_expand_host_ = false;   // APSU_HE_SEED_EXPAND_HOST: seeded objects expanded by the host codec (the fallback of the device sampler)
    bool tier1_device_ = false;       // tier-1 operands are device memory and calls do not synchronise
    void tier1_done() { if (!tier1_device_ || prof_on_) sync(); }
    bool packed_rows_ = true;         // BinBundle plaintexts are kept bit-packed in HBM (APSU_HE_PACKED_ROWS=0: dense 64-bit words; Bundle::packed)
    size_t slot_bytes(int chain_idx, bool packed) const;          // bytes of one NTT-form plaintext at a level, either format
    void pack_bundle(Bundle &b);      // dense -> packed when this context keeps packed rows (no-op otherwise)
    void unpack_bundle(Bundle &b);    // packed -> dense (images of the other format)
    int mac_kara_ = -1;               // k_mac with three products per term instead of four: -1 by chain length, 0 / 1 forced (APSU_HE_MAC_KARA)
    bool mac_kara(int lvl, uint32_t mean_cnt) const;
    uint64_t mac_units(const std::vector<MacJob> &mj) const;      // bits of database rows per coefficient index (profile unit of P_MAC)
    size_t eval_ws_budget_ = (size_t)6 << 30;
    bool split_ok_ = false;           // the low-power and high-power halves of the PowersDag share no node
    int two_stream_mode_ = -1, two_stream_default_ = -1;   // apsu_he_set_two_stream ; APSU_HE_SPLIT
    void build_schedule();
    void build_schedule_for(Sched &s, const std::vector<char> &member);
    struct DagRun;                    // per-call state of one walk over a schedule
    void run_dag(const Sched &s, DagRun &run, int stage, int nb, const u64 *const *src, bool on_device, const RelinKeys *rk, Powers &pw,
                 bool do_low, bool do_high);
    // eval_bundles in pieces (engine.cpp): per-call context, Paterson-Stockmeyer plan and batch
    struct EvalCall;
    struct PsPlan;
    void eval_plain(EvalCall &c, const std::vector<int> &pl_ids);
    void eval_patstock(EvalCall &c, const std::vector<int> &ps_ids);
    void ps_cf_streams(const EvalCall &c, const PsBatch &g, std::vector<MacStream> &out);
    void ps_tables(EvalCall &c, const PsPlan &plan, PsBatch &g);
    void ps_run(EvalCall &c, const PsPlan &plan, PsBatch &g);
    void finish_bundle(Bundle &b, const u64 *raw);     // raw: [degree+1][n] coefficient-form plaintexts mod t (device)
    DevBuf d_slot_map_;
    DevBuf d_seed_rej_, d_seed_mm_, d_key_level_;   // seed expansion: rejection lists, max_multiple per (level, limb), DevLevel-shaped view of the key level

    // profiling state
    struct ProfRec { hipEvent_t a, b; int kind; uint64_t units; };
    bool prof_on_ = false, prof_open_ = false, prof_ntt_only_ = false;
    std::vector<ProfRec> prof_recs_;
    std::vector<hipEvent_t> prof_pool_;
    ProfStats prof_{};
    void prof_begin(int kind, uint64_t units);
    void prof_end();
    void prof_collect();

    // phase timers
    struct PhaseSpan { hipEvent_t a = nullptr, b = nullptr, b2 = nullptr; int phase = 0; };
    bool phase_on_ = false;
    std::vector<PhaseSpan> phase_spans_;
    std::vector<hipEvent_t> phase_pool_;
    PhaseSummary phase_[PH_COUNT];
    hipEvent_t query_start_ = nullptr, query_end_ = nullptr;     // the open RunQuery span
    hipEvent_t phase_event(hipStream_t st);
    void phase_close_query();
    void phase_collect();
    struct Enter;                     // lock + current-device guard taken by every public entry point
    friend struct EngineAccess;
    friend struct ProfScope;
};

} // namespace apsu_he
