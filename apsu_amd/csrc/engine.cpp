// See engine.h.  Host orchestration of the HIP kernels; mirrors, step for step,
//   Receiver::ComputePowers                 receiver/apsu/receiver_osn.cpp:395-488
//   BatchedPlaintextPolyn::eval             receiver/apsu/bin_bundle.cpp:106-174
//   BatchedPlaintextPolyn::eval_patstock    receiver/apsu/bin_bundle.cpp:192-360
// Exact (rounding-free) steps are batched / re-associated freely; every rounding step
// (drop-limb, BEHZ floor, key-switch mod-down) is applied per term exactly where the reference
// applies it (SURVEY.md §2.4 note N1), so results are bit-identical to the CPU path.
#include "engine.h"
#include "sched_policy.h"
#include "seal_codec.h"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace apsu_he {
static uint32_t packed_row_bits(u64 q);   // width of a bit-packed database row (see Engine::mac_units)

void throw_hip(hipError_t e, const char *file, int line)
{
    throw HipError(std::string("HIP error: ") + hipGetErrorString(e) + " at " + file + ":" + std::to_string(line));
}
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__); } while (0)

namespace { struct ArenaOverflow { size_t need; }; }

// Every public entry point serialises on the context and runs with the context's device current: HIP's current
// device is per host thread, and the reference calls the Evaluator from a thread pool (receiver_osn.cpp:334-364),
// so a worker thread may arrive with another device selected (several contexts on different GPUs in one process).
struct Engine::Enter {
    std::lock_guard<std::mutex> lock;
    int prev = -1;
    explicit Enter(Engine *e) : lock(e->mu_)
    {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != e->device_) { prev = cur; HIP_CHECK(hipSetDevice(e->device_)); }
    }
    ~Enter() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void DevBuf::alloc(size_t bytes)
{
    release();
    if (!bytes) return;
    HIP_CHECK(hipMalloc(&p_, bytes));
    bytes_ = bytes;
}
void DevBuf::release()
{
    if (p_) (void)hipFree(p_);
    p_ = nullptr;
    bytes_ = 0;
}

int Powers::slot_of(uint32_t bundle_idx) const
{
    for (int i = 0; i < nb; i++) if (bundle_indices[i] == bundle_idx) return i;
    return -1;
}

static ShoupConst shoup_const(u64 w, u64 q) { return ShoupConst{ w, (u64)(((u128)w << 64) / q) }; }
static Mod make_mod(u64 q) { ModulusInfo m(q); return Mod{ q, m.ratio[0], m.ratio[1] }; }

// ============================================================================ construction
Engine::Engine(const HeParams &hp, const PSUParams *psu, int device) : hp_(hp), device_(device)
{
    if (psu) { psu_ = *psu; has_psu_ = true; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        throw HipError("no HIP device available: the query-evaluation engine has no CPU fallback");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
    // the NTT keeps one limb in a workgroup's LDS: n = 2^logn coefficients with a compiled pass plan (ntt_core.h)
    // (32768: the limb is split into two LDS-resident halves around one radix-2 stage over global memory, kernels.hip)
    const bool split_ntt = hp_.logn == 15;
    if (plan_passes(hp_.logn) == 0 && !split_ntt)
        throw std::invalid_argument("poly_modulus_degree " + std::to_string(hp_.n) +
                                    " is not supported by the GPU engine (supported: 64, 256, 1024, 2048, 4096, 8192, 16384, 32768)");
    struct Restore { int prev = -1; ~Restore() { if (prev >= 0) (void)hipSetDevice(prev); } } restore;
    { int cur = -1; if (hipGetDevice(&cur) == hipSuccess && cur != device) restore.prev = cur; }
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    {   // the second stream carries short latency-bound chains that must make progress next to a grid-filling kernel
        // of the main stream (the BinBundle inner products): highest dispatch priority
        int least = 0, greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_CHECK(hipStreamCreateWithPriority(&lanes_[1].st, hipStreamNonBlocking, greatest));
        HIP_CHECK(hipStreamCreateWithPriority(&lanes_[2].st, hipStreamNonBlocking, greatest));
    }
    HIP_CHECK(hipEventCreateWithFlags(&ev_main_, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_side_, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_intt_, hipEventDisableTiming));

    const size_t n = hp_.n;
    const int nmod = (int)hp_.ntt.size();
    // twiddles: per modulus [fwd n][dit n][scale n] TwPair
    {
        std::vector<TwPair> tw((size_t)nmod * 3 * n + (split_ntt ? (size_t)nmod * n : 0));
        for (int m = 0; m < nmod; m++) {
            const NttTablesHost &t = hp_.ntt[m];
            for (size_t k = 0; k < n; k++) {
                tw[((size_t)m * 3 + 0) * n + k] = TwPair{ t.fwd[k], t.fwd_q[k] };
                tw[((size_t)m * 3 + 1) * n + k] = TwPair{ t.dit[k], t.dit_q[k] };
                tw[((size_t)m * 3 + 2) * n + k] = TwPair{ t.scale[k], t.scale_q[k] };
            }
            if (split_ntt) {
                // forward twiddles of the two half transforms: stage s of half h is stage s + 1 of the big transform,
                // blocks h 2^s ..: W_h[2^s + b] = W[2^(s+1) + h 2^s + b]
                for (size_t h = 0; h < 2; h++) {
                    TwPair *dst = tw.data() + (size_t)nmod * 3 * n + ((size_t)m * 2 + h) * (n / 2);
                    dst[0] = TwPair{ 0, 0 };
                    for (size_t m2 = 1; m2 < n / 2; m2 <<= 1)
                        for (size_t b = 0; b < m2; b++) {
                            const size_t from = 2 * m2 + h * m2 + b;
                            dst[m2 + b] = TwPair{ t.fwd[from], t.fwd_q[from] };
                        }
                }
            }
        }
        d_tw_.alloc(tw.size() * sizeof(TwPair));
        HIP_CHECK(hipMemcpy(d_tw_.p(), tw.data(), tw.size() * sizeof(TwPair), hipMemcpyHostToDevice));
        std::vector<NttTable> tabs(split_ntt ? (size_t)nmod * 2 : (size_t)nmod);
        const TwPair *base = reinterpret_cast<const TwPair *>(d_tw_.p());
        for (int m = 0; m < nmod; m++) {
            NttTable tb{};
            tb.q = hp_.ntt[m].mod.value;
            tb.ninv = hp_.ntt[m].ninv;
            tb.ninv_q = hp_.ntt[m].ninv_q;
            tb.r1 = hp_.ntt[m].mod.ratio[1];
            tb.r0 = hp_.ntt[m].mod.ratio[0];
            tb.narrow = ntt_is_narrow(hp_.ntt[m].mod.value, split_ntt ? hp_.logn - 1 : hp_.logn) ? 1 : 0;   // stages inside one workgroup
            ntt_fold_params(tb.q, tb.fold_k, tb.fold_c);
            tb.wide_d4 = ntt_wide_d4(tb.q, tb.narrow != 0);
            tb.fwd = base + ((size_t)m * 3 + 0) * n;
            tb.dit = base + ((size_t)m * 3 + 1) * n;
            tb.scale = base + ((size_t)m * 3 + 2) * n;
            if (!split_ntt) { tabs[m] = tb; continue; }
            // split transform: table 2 m + h for half h; ninv / ninv_q carry the first stage's twiddle psi^brv(1)
            tb.ninv = hp_.ntt[m].fwd[1];
            tb.ninv_q = hp_.ntt[m].fwd_q[1];
            for (size_t h = 0; h < 2; h++) {
                tb.fwd = base + (size_t)nmod * 3 * n + ((size_t)m * 2 + h) * (n / 2);
                tabs[(size_t)m * 2 + h] = tb;
            }
        }
        d_tabs_.alloc(tabs.size() * sizeof(NttTable));
        HIP_CHECK(hipMemcpy(d_tabs_.p(), tabs.data(), tabs.size() * sizeof(NttTable), hipMemcpyHostToDevice));
        // Environment switches (read once per context).  Round 5 retired the switches of A/B experiments that were decided in earlier
        // rounds together with their losing code paths (separate tensor / extension kernels where the fused forms apply, inverse
        // transforms with their own twist in front of drop / mod-down kernels, the tensor transform in launch order, canonical products
        // in the tensor-on-load transform, k_mac's other grid orders, the i = 0 block's per-term products as k_mac chains, the gathered
        // transforms' unconditional reduce-on-load; the records are in profiles/r03_ab_*.txt and profiles/r04_ab_*.txt).  What is left
        // either selects a data format, sizes a buffer, or forces a correctness fallback that the engine otherwise takes by itself:
        //   APSU_HE_SPLIT=0/1          default of apsu_he_set_two_stream (profiling scripts: one-stream kernel traces)
        //   APSU_HE_EVAL_SIDE=0        the evaluation's side work (coefficient-form sums, i = 0 finish) stays on the main stream
        //   APSU_HE_PACKED_ROWS=0      BinBundle rows as dense 64-bit words instead of bit-packed (images of either format load anywhere)
        //   APSU_HE_EVAL_WS_BYTES=n    evaluation workspace -> BinBundles per chunk (default 6 GiB)
        //   APSU_HE_ARENA_BYTES=n      initial workspace arena (grows on demand)
        //   APSU_HE_EVAL_PER_TERM=1    eval_patstock's products finished one by one (the fallback of the summed finish)
        //   APSU_HE_MAC_KARA=0/1       three-product k_mac forced off / on (default: by chain length)
        //   APSU_HE_SEED_EXPAND_HOST=1 seeded objects expanded by the host codec (the fallback of the device sampler)
        //   APSU_HE_FUSE_TAIL=0        eval_patstock's last mod-down as its own launch instead of inside the epilogue kernel (round 6)
        //   APSU_HE_NTT_LATENCY_LIMBS=n transform launches of at most n limbs take the latency form (8 coefficients per lane; round 6);
        //                              0 = always the throughput form.  Default: the measured crossover per ring size and kind of launch.
        if (const char *v = std::getenv("APSU_HE_SPLIT")) two_stream_default_ = std::atoi(v) != 0 ? 1 : 0;
        if (split_ntt) fuse_tensor_ = false;                     // the fused load belongs to a whole-limb workgroup
        if (const char *v = std::getenv("APSU_HE_MAC_KARA")) mac_kara_ = std::atoi(v) != 0 ? 1 : 0;
        if (const char *v = std::getenv("APSU_HE_EVAL_SIDE")) eval_side_ = std::atoi(v) != 0;
        // BinBundle plaintexts bit-packed in HBM (12.5 % fewer bytes for 56-bit primes, 22 % for 50-bit ones; k_mac<.., PACKED>): in-process
        // A/B on 16M-4096 -0.146 +- 0.017 ms (-4.2 %) on the whole query, -2.4 % on the N = 8 shard, same bits
        // (profiles/r04_ab_packed_rows.txt).  Only with key switching (the single-prime paths keep dense rows).
        if (const char *v = std::getenv("APSU_HE_PACKED_ROWS")) packed_rows_ = std::atoi(v) != 0;
        if (!hp_.using_keyswitching) packed_rows_ = false;
        if (const char *v = std::getenv("APSU_HE_EVAL_WS_BYTES")) eval_ws_budget_ = std::strtoull(v, nullptr, 10);
        if (const char *v = std::getenv("APSU_HE_EVAL_PER_TERM")) force_per_term_ = std::atoi(v) != 0;
        if (const char *v = std::getenv("APSU_HE_SEED_EXPAND_HOST")) seed_expand_host_ = std::atoi(v) != 0;
        // The latency form of the LDS-resident transform (ntt_core.h plan_k, c = 8): a limb's workgroup has twice the waves, so a launch
        // that gives a CU at most one limb hides that limb's LDS turnarounds and table loads behind three other waves per SIMD.
        // Crossovers: kernels.hip, ntt_use_latency_form (tools/microbench/ntt_forms.hip, profiles/r06_ntt_forms_n8192.txt / _n4096.txt).
        ntt_latency_limbs_ = NTT_FORM_AUTO;
        data_primes_narrow_ = true;
        for (int j = 0; j < hp_.K; j++) data_primes_narrow_ = data_primes_narrow_ && ntt_is_narrow(hp_.key_q[j], hp_.logn);
        if (const char *v = std::getenv("APSU_HE_FUSE_TAIL")) fuse_tail_ = std::atoi(v) != 0;
        if (const char *v = std::getenv("APSU_HE_NTT_LATENCY_LIMBS")) ntt_latency_limbs_ = std::strtoull(v, nullptr, 10);
    }
    // level constants
    {
        const int nl = hp_.first_chain_idx + 1;
        std::vector<DevLevel> lv(nl);
        std::vector<int> map_ext((size_t)nl * DMAXE, 0), map_ks((size_t)nl * (DMAXL + 1) * DMAXL, 0),
            map_ksacc((size_t)nl * (DMAXL + 1), 0);
        const u64 mt = (u64)1 << 32;
        for (int c = 0; c < nl; c++) {
            const LevelConstants &h = hp_.level[c];
            DevLevel &d = lv[c];
            std::memset(&d, 0, sizeof(d));
            const int L = h.L, nB = h.nB, nBsk = nB + 1;
            if (L > DMAXL || nBsk > DMAXB) throw std::invalid_argument("too many RNS limbs");
            d.L = L; d.nB = nB; d.nBsk = nBsk; d.E = L + nBsk;
            std::vector<u64> bsk = h.B;
            bsk.push_back(h.m_sk);
            d.t = hp_.t;
            d.q_mod_t = h.q_mod_t;
            d.threshold = h.upper_half_threshold;
            d.half = h.q[L - 1] >> 1;
            for (int j = 0; j < L; j++) {
                const u64 qj = h.q[j];
                ModulusInfo mj(qj);
                d.q[j] = make_mod(qj);
                d.ext[j] = d.q[j];
                {
                    const int bits = 64 - __builtin_clzll(qj);
                    const int sh = (bits + 1) / 2;                         // both operand halves < 2^sh (sh <= 30)
                    d.mac_shift[j] = (u32)sh;
                    // cross sum takes two products (< 2^(2 sh)) per term, plus one slot for the carried residue
                    const u64 cap = ((u64)1 << (63 - 2 * sh));
                    d.mac_chunk[j] = (u32)std::min<u64>(cap > 2 ? cap - 1 : 2, 1u << 20);
                    // three-product form: one middle product (a0 + a1)(c0 + c1) < 2^(2 sh + 2) per term, the carried residue enters as
                    // (r0, r0 + r1) < 2^(sh + 1): one slot as well
                    const u64 capk = 2 * sh + 2 < 64 ? ((u64)1 << (62 - 2 * sh)) : 0;
                    d.mac_chunk_k[j] = (u32)std::min<u64>(capk > 2 ? capk - 1 : 0, 1u << 20);   // 0: not usable for this modulus
                    // packed row width: the smallest w >= max(bits, 32) whose 2-coefficient group fits a lane's 16-byte window at every
                    // position: the group starts at bit 2 w m, i.e. (2 w m) mod 32 <= 32 - gcd(2 w, 32) into its first dword
                    // (the geometry is there in every context: images of either format load anywhere)
                    const u32 w = hp_.using_keyswitching ? packed_row_bits(qj) : 64;
                    d.mac_bits[j] = w;
                    d.mac_row_off[j] = j ? d.mac_row_off[j - 1] + (u32)(hp_.n * d.mac_bits[j - 1] / 8) : 0;
                    d.mac_mask_hi[j] = w == 64 ? 0xffffffffu : (u32)(((u64)1 << (w - sh)) - 1);
                }
                d.coeff_div_plain[j] = h.coeff_div_plain[j];
                d.incr[j] = h.upper_half_incr[j];
                d.half_mod[j] = d.half % qj;
                if (j + 1 < L) d.inv_q_last[j] = shoup_const(h.inv_q_last[j], qj);
                d.ext_scale[j] = shoup_const(mj.mul(mt % qj, h.inv_punct_q[j]), qj);
                d.q_to_mt[j] = (u32)h.q_to_mtilde[j];
                d.t_inv_punct_q[j] = shoup_const(mj.mul(hp_.t % qj, h.inv_punct_q[j]), qj);
                d.prod_B_q[j] = h.prod_B_mod_q[j];
                d.neg_prod_B_q[j] = (qj - h.prod_B_mod_q[j]) % qj;
                d.s_prod_B_q[j] = shoup_const(d.prod_B_q[j], qj);
                d.s_neg_prod_B_q[j] = shoup_const(d.neg_prod_B_q[j], qj);
                for (int i = 0; i < nB; i++) { d.B_to_q[j][i] = h.B_to_q[j][i]; d.s_B_to_q[j][i] = shoup_const(h.B_to_q[j][i], qj); }
                map_ext[(size_t)c * DMAXE + j] = j;
            }
            d.neg_inv_q_mt = (u32)h.neg_inv_q_mod_mtilde;
            for (int i = 0; i < nBsk; i++) {
                const u64 m = bsk[i];
                d.bsk[i] = make_mod(m);
                d.ext[L + i] = d.bsk[i];
                for (int j = 0; j < L; j++) { d.q_to_bsk[i][j] = h.q_to_bsk[i][j]; d.s_q_to_bsk[i][j] = shoup_const(h.q_to_bsk[i][j], m); }
                d.prod_q_bsk[i] = h.prod_q_mod_bsk[i];
                d.s_prod_q_bsk[i] = shoup_const(h.prod_q_mod_bsk[i], m);
                d.s_fl[i] = shoup_const(i < nB ? ModulusInfo(m).mul(h.inv_prod_q_mod_bsk[i], h.inv_punct_B[i]) : h.inv_prod_q_mod_bsk[i], m);
                d.inv_mt_bsk[i] = shoup_const(h.inv_mtilde_mod_bsk[i], m);
                {
                    const ModulusInfo mi(m);
                    for (int j = 0; j < L; j++) d.s_q_to_bsk_mt[i][j] = shoup_const(mi.mul(h.q_to_bsk[i][j] % m, h.inv_mtilde_mod_bsk[i]), m);
                    d.s_prod_q_bsk_mt[i] = shoup_const(mi.mul(h.prod_q_mod_bsk[i], h.inv_mtilde_mod_bsk[i]), m);
                }
                d.t_bsk[i] = shoup_const(hp_.t % m, m);
                d.inv_prod_q_bsk[i] = shoup_const(h.inv_prod_q_mod_bsk[i], m);
                if (i < nB) {
                    d.inv_punct_B[i] = shoup_const(h.inv_punct_B[i], m);
                    d.B_to_msk[i] = h.B_to_msk[i];
                    d.s_B_to_msk[i] = shoup_const(h.B_to_msk[i], h.m_sk);
                }
                map_ext[(size_t)c * DMAXE + L + i] = hp_.bsk_id(nB, i);
            }
            d.inv_prod_B_msk = shoup_const(h.inv_prod_B_mod_msk, h.m_sk);
            d.msk_half = h.m_sk >> 1;
            // key-switch maps
            for (int I = 0; I <= L; I++) {
                const int id = I == L ? hp_.K - 1 : I;
                for (int J = 0; J < L; J++) map_ks[(size_t)c * (DMAXL + 1) * DMAXL + (size_t)I * L + J] = id;
                map_ksacc[(size_t)c * (DMAXL + 1) + I] = id;
            }
        }
        // per-position constants of the unrolled finish kernels: their own constant times the inverse transform's twist
        // (device.h: fin_q / fin_b), and the matching inverse-NTT maps
        std::vector<int> map_ext_fin = map_ext;
        {
            std::vector<ShoupConst> fin;
            std::vector<std::pair<int, int>> where;                   // (level, ext limb) of each table, in order
            for (int c = 0; c < nl; c++) {
                if (!fast_finish(c)) continue;
                const LevelConstants &h = hp_.level[c];
                const int L = h.L, nB = h.nB;
                for (int e = 0; e < L + nB + 1; e++) {
                    const int id = e < L ? e : hp_.bsk_id(nB, e - L);
                    const NttTablesHost &tb = hp_.ntt[id];
                    const u64 m = tb.mod.value;
                    const u64 cst = e < L ? lv[c].t_inv_punct_q[e].w : lv[c].t_bsk[e - L].w;
                    for (size_t k = 0; k < n; k++) fin.push_back(shoup_const(tb.mod.mul(cst, tb.scale[k]), m));
                    where.push_back({ c, e });
                    map_ext_fin[(size_t)c * DMAXE + e] |= NTT_MAP_RAW;
                }
            }
            if (!fin.empty()) {
                d_fin_.alloc(fin.size() * sizeof(ShoupConst));
                HIP_CHECK(hipMemcpy(d_fin_.p(), fin.data(), fin.size() * sizeof(ShoupConst), hipMemcpyHostToDevice));
                const ShoupConst *base = reinterpret_cast<const ShoupConst *>(d_fin_.p());
                for (size_t i = 0; i < where.size(); i++) {
                    DevLevel &d = lv[where[i].first];
                    const int e = where[i].second;
                    if (e < d.L) d.fin_q[e] = base + i * n;
                    else d.fin_b[e - d.L] = base + i * n;
                }
            }
        }
        // per-position constants of the drop-last-limb consumers of a RAW inverse transform (device.h: drop_tw / last_tw)
        {
            std::vector<ShoupConst> dt;
            std::vector<std::pair<int, int>> where;                   // (level, limb) ; limb = -1: last_tw
            const TwPair *twbase = reinterpret_cast<const TwPair *>(d_tw_.p());
            for (int c = 1; c < nl; c++) {
                const LevelConstants &h = hp_.level[c];
                const int L = h.L;
                for (int j = 0; j + 1 < L; j++) {
                    const NttTablesHost &tb = hp_.ntt[j];
                    for (size_t k = 0; k < n; k++) dt.push_back(shoup_const(tb.mod.mul(h.inv_q_last[j], tb.scale[k]), tb.mod.value));
                    where.push_back({ c, j });
                }
                // the dropped limb's plain twist is the transform's own scale table (same {w, wq} layout)
                lv[c].last_tw = reinterpret_cast<const ShoupConst *>(twbase + ((size_t)(L - 1) * 3 + 2) * n);
            }
            if (!dt.empty()) {
                d_drop_.alloc(dt.size() * sizeof(ShoupConst));
                HIP_CHECK(hipMemcpy(d_drop_.p(), dt.data(), dt.size() * sizeof(ShoupConst), hipMemcpyHostToDevice));
                const ShoupConst *base = reinterpret_cast<const ShoupConst *>(d_drop_.p());
                for (size_t i = 0; i < where.size(); i++) lv[where[i].first].drop_tw[where[i].second] = base + i * n;
            }
        }
        d_levels_.alloc(lv.size() * sizeof(DevLevel));
        HIP_CHECK(hipMemcpy(d_levels_.p(), lv.data(), lv.size() * sizeof(DevLevel), hipMemcpyHostToDevice));
        auto up = [](DevBuf &b, const std::vector<int> &v) {
            b.alloc(v.size() * sizeof(int));
            HIP_CHECK(hipMemcpy(b.p(), v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
        };
        up(d_map_ext_, map_ext); up(d_map_ext_fin_, map_ext_fin); up(d_map_ks_, map_ks); up(d_map_ksacc_, map_ksacc);
        std::vector<int> map_ksacc_raw = map_ksacc;
        for (int &v : map_ksacc_raw) v |= NTT_MAP_RAW;
        up(d_map_ksacc_raw_, map_ksacc_raw);
        std::vector<int> ident(DMAXL + DMAXB + 4);                     // identity over every modulus id (incl. plain modulus)
        for (size_t i = 0; i < ident.size(); i++) ident[i] = (int)i;
        up(d_map_ct_, ident);
    }
    // key-switching constants
    {
        DevKey k;
        std::memset(&k, 0, sizeof(k));
        k.K = hp_.K;
        for (int j = 0; j < hp_.K; j++) k.q[j] = make_mod(hp_.key_q[j]);
        if (hp_.K > 1) {
            const u64 p = hp_.key_q[hp_.K - 1];
            k.p_half = p >> 1;
            for (int j = 0; j < hp_.K - 1; j++) {
                k.p_half_mod[j] = k.p_half % hp_.key_q[j];
                k.inv_p[j] = shoup_const(hp_.inv_p_mod_q[j], hp_.key_q[j]);
            }
            // per-position constants of the mod-down behind a RAW inverse transform (device.h: md_tw / p_tw)
            std::vector<ShoupConst> mt;
            for (int j = 0; j < hp_.K - 1; j++) {
                const NttTablesHost &tb = hp_.ntt[j];
                for (size_t kk = 0; kk < n; kk++) mt.push_back(shoup_const(tb.mod.mul(hp_.inv_p_mod_q[j], tb.scale[kk]), tb.mod.value));
            }
            d_mdtw_.alloc(mt.size() * sizeof(ShoupConst));
            HIP_CHECK(hipMemcpy(d_mdtw_.p(), mt.data(), mt.size() * sizeof(ShoupConst), hipMemcpyHostToDevice));
            for (int j = 0; j < hp_.K - 1; j++) k.md_tw[j] = reinterpret_cast<const ShoupConst *>(d_mdtw_.p()) + (size_t)j * n;
            k.p_tw = reinterpret_cast<const ShoupConst *>(reinterpret_cast<const TwPair *>(d_tw_.p()) + ((size_t)(hp_.K - 1) * 3 + 2) * n);
        }
        d_key_.alloc(sizeof(DevKey));
        HIP_CHECK(hipMemcpy(d_key_.p(), &k, sizeof(DevKey), hipMemcpyHostToDevice));
    }
    if (hp_.batching) {
        d_slot_map_.alloc(hp_.slot_map.size() * sizeof(uint32_t));
        HIP_CHECK(hipMemcpy(d_slot_map_.p(), hp_.slot_map.data(), hp_.slot_map.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (has_psu_) {
        auto targets = create_powers_set(psu_.query_params.ps_low_degree, psu_.table_params.max_items_per_bin);
        if (!dag_.configure(psu_.query_params.query_powers, targets))
            throw std::invalid_argument("failed to configure PowersDag");
        build_schedule();
        // polynomials per target power: 2 with key switching; without it a product has size(parent1) + size(parent2) - 1
        const uint32_t max_power = *dag_.target_powers().rbegin();
        nks_size_.assign(max_power + 1, 0);
        for (uint32_t p : dag_.target_powers()) nks_size_[p] = 2;
        nks_ = !hp_.using_keyswitching && dag_.depth() > 0;
        if (nks_) {
            uint32_t widest = 2;
            for (const auto &nd : sched_.nodes) {                 // slot order: parents come first
                const uint32_t a = nks_size_[sched_.slot_power[nd[1]]], b = nks_size_[sched_.slot_power[nd[2]]];
                const uint32_t sz = std::min<uint32_t>(a + b - 1, 2 * CT_SIZE_MAX);        // (kept finite; anything above the limit throws)
                nks_size_[sched_.slot_power[nd[0]]] = sz;
                widest = std::max(widest, sz);
            }
            nks_oversize_ = widest > CT_SIZE_MAX;
            nks_S_ = (std::min(widest, CT_SIZE_MAX) + 1) & ~1u;
        }
        if (!hp_.using_keyswitching)                              // (eval_patstock alone already leaves three polynomials)
            for (uint32_t deg = 0; deg <= psu_.table_params.max_items_per_bin; deg++) {
                const uint32_t r = result_size_for(deg);
                if (r <= CT_SIZE_MAX) result_polys_ = std::max(result_polys_, r);
            }
    }
    size_t init = 4096 * n * sizeof(u64);          // 256 MiB at n = 8192; grows on demand
    if (const char *env = std::getenv("APSU_HE_ARENA_BYTES")) init = std::strtoull(env, nullptr, 10);
    arena_.alloc(init);
    lanes_[1].arena.alloc(std::max<size_t>(init / 4, (size_t)1 << 20));
    lanes_[2].arena.alloc((size_t)1 << 20);
    stage_bytes_ = 4u << 20;
    HIP_CHECK(hipHostMalloc(&stage_, stage_bytes_));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&bad_source_), BAD_SLOTS * sizeof(unsigned)));
    for (int i = 0; i < BAD_SLOTS; i++) bad_source_[i] = 0;
}

Engine::~Engine()
{
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != device_ && hipSetDevice(device_) == hipSuccess;
    struct Back { bool on; int dev; ~Back() { if (on) (void)hipSetDevice(dev); } } back{ switched, cur };
    // queued work first (apsu_he_set_async_results leaves evaluations in flight), then the buffers it uses
    if (st_) (void)hipStreamSynchronize(st_);
    for (Lane &l : lanes_) if (l.st) (void)hipStreamSynchronize(l.st);
    powers_pool_.clear();
    if (st_) (void)hipStreamDestroy(st_);
    for (Lane &l : lanes_) if (l.st) (void)hipStreamDestroy(l.st);
    for (hipEvent_t e : inflight_) if (e) (void)hipEventDestroy(e);
    for (PhaseSpan &sp : phase_spans_) for (hipEvent_t e : { sp.a, sp.b, sp.b2 }) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : phase_pool_) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : { query_start_, query_end_ }) if (e) (void)hipEventDestroy(e);
    if (ev_main_) (void)hipEventDestroy(ev_main_);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_side_) (void)hipEventDestroy(ev_side_);
    if (ev_intt_) (void)hipEventDestroy(ev_intt_);
    if (stage_) (void)hipHostFree(stage_);
    if (bad_source_) (void)hipHostFree(bad_source_);
    if (wire_pinned_) (void)hipHostFree(wire_pinned_);
}

void Engine::wire_stage(size_t bytes, u64 **device, u64 **pinned)
{
    Enter g(this);                                               // (device guard; the buffers belong to this engine's device)
    if (wire_dev_.bytes() < bytes) { sync(); wire_dev_.alloc(bytes + bytes / 4); }
    if (wire_pinned_bytes_ < bytes) {
        sync();
        if (wire_pinned_) (void)hipHostFree(wire_pinned_);
        wire_pinned_ = nullptr; wire_pinned_bytes_ = 0;
        HIP_CHECK(hipHostMalloc(&wire_pinned_, bytes + bytes / 4));
        wire_pinned_bytes_ = bytes + bytes / 4;
    }
    *device = wire_dev_.u();
    *pinned = static_cast<u64 *>(wire_pinned_);
}

void Engine::sync()
{
    counters_[C_HOST_SYNC]++;
    HIP_CHECK(hipStreamSynchronize(st_));
    for (Lane &l : lanes_) if (l.st) HIP_CHECK(hipStreamSynchronize(l.st));
    if (prof_on_) prof_collect();
}

// Called where a PUBLIC entry point has just waited for the device (apsu_he_sync, a synchronous apsu_he_eval_bundles, apsu_he_powers_download):
// k_copy_sources of a query whose work has completed by now may have found a source coefficient outside [0, q).  (Not inside sync():
// the engine also waits in the middle of its own bookkeeping -- arena growth, job-table reallocation -- where nothing may be thrown.)
bool Engine::bad_source_pending(int slot, bool take)
{
    if (!bad_source_) return false;
    const unsigned v = *const_cast<volatile unsigned *>(bad_source_ + slot);
    if (v == bad_reported_[slot]) return false;
    if (take) bad_reported_[slot] = v;
    return true;
}

static const char *const BAD_SOURCE_TEXT = "a source ciphertext of apsu_he_compute_powers holds a coefficient outside [0, q): the results computed from it "
                                           "are not valid (seal::is_data_valid_for)";

void Engine::check_sources()
{
    bool any = false;
    for (int i = 0; i < BAD_SLOTS; i++) any |= bad_source_pending(i, true);
    if (any) throw std::invalid_argument(BAD_SOURCE_TEXT);
}

// the report of ONE query: only the word its ComputePowers writes, and only that call's sequence number in it (an older or a newer
// query's report stays where it is for the call that waits for that query, or for apsu_he_sync)
void Engine::check_sources(const Powers &pw)
{
    const int slot = (int)(pw.seq % BAD_SLOTS);
    if (!bad_source_ || !pw.seq) return;
    const unsigned v = *const_cast<volatile unsigned *>(bad_source_ + slot);
    if (v == pw.seq && bad_reported_[slot] != v) {
        bad_reported_[slot] = v;
        throw std::invalid_argument(BAD_SOURCE_TEXT);
    }
}

bool Engine::take_bad_source()
{
    Enter g(this);
    bool any = false;
    for (int i = 0; i < BAD_SLOTS; i++) any |= bad_source_pending(i, true);
    return any;
}

void Engine::wait()
{
    Enter g(this);
    sync();
    check_sources();
}

void Engine::drain()
{
    Enter g(this);
    sync();
}

void Engine::switch_lane(int lane)
{
    if (lane == cur_lane_) return;
    for (Lane *l : { &lanes_[cur_lane_], &lanes_[lane] }) {      // park the current lane in its (empty) place, take the other one out of its
        std::swap(st_, l->st);
        std::swap(arena_, l->arena);
        std::swap(arena_off_, l->off);
        std::swap(job_seq_, l->job_seq);
    }
    cur_lane_ = lane;
}

// ---- profiling: one HIP event pair per launch on the engine's stream, resolved at the next sync
void Engine::profile_enable(int mode)
{
    Enter g(this);
    HIP_CHECK(hipStreamSynchronize(st_));
    for (Lane &l : lanes_) if (l.st) HIP_CHECK(hipStreamSynchronize(l.st));
    prof_collect();
    prof_on_ = mode != 0;
    prof_ntt_only_ = mode == 2;
}

void Engine::profile_read(ProfStats *out, bool reset)
{
    Enter g(this);
    HIP_CHECK(hipStreamSynchronize(st_));
    for (Lane &l : lanes_) if (l.st) HIP_CHECK(hipStreamSynchronize(l.st));
    prof_collect();
    if (out) *out = prof_;
    if (reset) prof_ = ProfStats{};
}

void Engine::prof_begin(int kind, uint64_t units)
{
    if (!prof_on_ || (prof_ntt_only_ && kind > P_NTT_INV && kind != P_NTT_FUSED)) return;
    ProfRec r{ nullptr, nullptr, kind, units };
    for (hipEvent_t *e : { &r.a, &r.b }) {
        if (!prof_pool_.empty()) { *e = prof_pool_.back(); prof_pool_.pop_back(); }
        else HIP_CHECK(hipEventCreate(e));
    }
    HIP_CHECK(hipEventRecord(r.a, st_));
    prof_recs_.push_back(r);
    prof_open_ = true;
}

void Engine::prof_end()
{
    if (!prof_open_ || prof_recs_.empty()) return;
    HIP_CHECK(hipEventRecord(prof_recs_.back().b, st_));
    prof_open_ = false;
}

void Engine::prof_collect()
{
    // a sync can happen inside an open scope (job-buffer growth): keep that record for later
    ProfRec open_rec{};
    const bool keep = prof_open_ && !prof_recs_.empty();
    if (keep) { open_rec = prof_recs_.back(); prof_recs_.pop_back(); }
    for (auto &r : prof_recs_) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            prof_.ms[r.kind] += ms;
            prof_.launches[r.kind] += 1;
            prof_.units[r.kind] += r.units;
        }
        prof_pool_.push_back(r.a);
        prof_pool_.push_back(r.b);
    }
    prof_recs_.clear();
    if (keep) prof_recs_.push_back(open_rec);
}

// ---- phase timers
const char *Engine::phase_name(int phase)
{
    static const char *names[PH_COUNT] = { "Receiver::RunQuery", "Receiver::ComputePowers", "Receiver::ProcessBinBundleCache" };
    return phase >= 0 && phase < PH_COUNT ? names[phase] : "";
}

hipEvent_t Engine::phase_event(hipStream_t st)
{
    hipEvent_t e = nullptr;
    if (!phase_pool_.empty()) { e = phase_pool_.back(); phase_pool_.pop_back(); }
    else HIP_CHECK(hipEventCreate(&e));
    HIP_CHECK(hipEventRecord(e, st));
    return e;
}

void Engine::phase_close_query()
{
    if (query_start_ && query_end_) phase_spans_.push_back(PhaseSpan{ query_start_, query_end_, nullptr, PH_RUN_QUERY });
    else { if (query_start_) phase_pool_.push_back(query_start_); if (query_end_) phase_pool_.push_back(query_end_); }
    query_start_ = query_end_ = nullptr;
}

void Engine::phase_collect()
{
    for (PhaseSpan &sp : phase_spans_) {
        float ms = 0, ms2 = 0;
        if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            if (sp.b2 && hipEventSynchronize(sp.b2) == hipSuccess && hipEventElapsedTime(&ms2, sp.a, sp.b2) == hipSuccess) ms = std::max(ms, ms2);
            PhaseSummary &p = phase_[sp.phase];
            p.min_ms = p.count ? std::min(p.min_ms, (double)ms) : ms;
            p.max_ms = p.count ? std::max(p.max_ms, (double)ms) : ms;
            p.sum_ms += ms;
            p.count++;
        }
        for (hipEvent_t e : { sp.a, sp.b, sp.b2 }) if (e) phase_pool_.push_back(e);
    }
    phase_spans_.clear();
}

void Engine::phase_enable(bool on)
{
    Enter g(this);
    if (!on && phase_on_) { phase_close_query(); phase_collect(); }
    phase_on_ = on;
}

void Engine::phase_read(PhaseSummary *out, bool reset)
{
    Enter g(this);
    phase_close_query();
    phase_collect();
    if (out) for (int i = 0; i < PH_COUNT; i++) out[i] = phase_[i];
    if (reset) for (int i = 0; i < PH_COUNT; i++) phase_[i] = PhaseSummary{};
}

struct ProfScope {
    Engine *e;
    ProfScope(Engine *e_, int kind, uint64_t units) : e(e_) { e->prof_begin(kind, units); }
    ~ProfScope() { e->prof_end(); }
};
#define PROF(kind, units) ProfScope prof_scope_(this, kind, units)
// element-wise classes: units = ALGORITHMIC bytes of the launch (compulsory operand reads + result writes, 8 bytes per word;
// level constants and the relinearisation keys -- shared by every coefficient, cache-resident -- not counted)
#define PROFW(kind, words) ProfScope prof_scope_(this, kind, (uint64_t)(words) * 8)
// P_MAC's profile unit: BITS of database rows streamed per coefficient index, summed over terms, streams and limbs (64 per limb of
// a dense row, mac_bits of a bit-packed one) -- bytes streamed = units * n / 8
static uint32_t packed_row_bits(u64 q)
{
    const int bits = 64 - __builtin_clzll(q);
    for (u32 c = (u32)std::max(bits, 32); c < 64; c++) {
        u32 g = 2 * c, r = 32;
        while (r) { const u32 t2 = g % r; g = r; r = t2; }
        if ((32 - g) + 2 * c <= 128) return c;
    }
    return 64;
}
uint64_t Engine::mac_units(const std::vector<MacJob> &mj) const
{
    uint64_t u = 0;
    for (auto &j : mj) {
        uint64_t w = 0;
        for (u32 l = j.limb0; l < j.limb0 + j.nl; l++) w += j.packed ? packed_row_bits(hp_.key_q[l]) : 64;
        u += (uint64_t)j.cnt * j.ng * w;
    }
    return u;
}
// mean number of terms per (stream, limb) chain of a launch: the three-product form pays for long chains only
static uint32_t mac_mean_cnt(const std::vector<MacJob> &mj)
{
    uint64_t u = 0, c = 0;
    for (auto &j : mj) { u += (uint64_t)j.cnt * j.ng * j.nl; c += (uint64_t)j.ng * j.nl; }
    return c ? (uint32_t)(u / c) : 0;
}

// single-stream description of a multiply-accumulate; group_mac() packs streams that share the
// ciphertext powers and the term count into MacJobs of up to MAC_G streams
struct MacStream { const u64 *pt; const u64 *pw; u64 *out; u32 cnt, pt_stride, pw_stride, pw_poly_stride, out_poly_stride, limb0, nl; u32 packed = 0; };
// where slot `slot` of a BinBundle's NTT-form plaintexts (lifted = false) or of its pre-lifted coefficient-form ones starts, and the
// distance between consecutive slots in the unit k_mac takes it in (words for dense rows, BYTES for bit-packed ones)
static const u64 *bundle_slot(const Bundle &b, bool lifted, size_t slot, size_t dense_words)
{
    const DevBuf &buf = lifted ? b.lifted : b.ntt;
    if (!b.packed) return buf.u() + slot * dense_words;
    return reinterpret_cast<const u64 *>(static_cast<const char *>(buf.p()) + slot * (lifted ? b.lifted_slot_bytes : b.ntt_slot_bytes));
}
static u32 bundle_stride(const Bundle &b, bool lifted, size_t dense_words)
{
    return b.packed ? (u32)(lifted ? b.lifted_slot_bytes : b.ntt_slot_bytes) : (u32)dense_words;
}
static bool mac_packed(const std::vector<MacJob> &mj) { return !mj.empty() && mj[0].packed != 0; }
static std::vector<MacJob> group_mac(const std::vector<MacStream> &ss)
{
    std::vector<MacJob> jobs;
    std::vector<char> used(ss.size(), 0);
    // streams are generated bundle-major; match each unused stream with later ones of equal key
    std::vector<size_t> order(ss.size());
    for (size_t i = 0; i < ss.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        if (ss[a].pw != ss[b].pw) return ss[a].pw < ss[b].pw;
        if (ss[a].limb0 != ss[b].limb0) return ss[a].limb0 < ss[b].limb0;
        if (ss[a].nl != ss[b].nl) return ss[a].nl < ss[b].nl;
        return ss[a].cnt < ss[b].cnt;
    });
    for (size_t x = 0; x < order.size();) {
        const MacStream &f = ss[order[x]];
        MacJob j{};
        j.pw = f.pw; j.cnt = f.cnt; j.pt_stride = f.pt_stride; j.pw_stride = f.pw_stride; j.pw_poly_stride = f.pw_poly_stride;
        j.out_poly_stride = f.out_poly_stride; j.limb0 = f.limb0; j.nl = f.nl; j.packed = f.packed;
        u32 g = 0;
        while (x < order.size() && g < (u32)MAC_G) {
            const MacStream &s = ss[order[x]];
            if (s.pw != f.pw || s.cnt != f.cnt || s.packed != f.packed || s.pt_stride != f.pt_stride || s.pw_stride != f.pw_stride ||
                s.pw_poly_stride != f.pw_poly_stride || s.out_poly_stride != f.out_poly_stride || s.limb0 != f.limb0 || s.nl != f.nl) break;
            j.pt[g] = s.pt; j.out[g] = s.out; g++; x++;
        }
        j.ng = g;
        for (u32 r = g; r < (u32)MAC_G; r++) { j.pt[r] = j.pt[0]; j.out[r] = j.out[0]; }
        jobs.push_back(j);
    }
    return jobs;
}

void Engine::check_level(int chain_idx) const
{
    if (chain_idx < 0 || chain_idx > hp_.first_chain_idx) throw std::invalid_argument("chain_idx is not a data level");
}

// ============================================================================ arena
u64 *Engine::ws(size_t words)
{
    size_t bytes = (words * sizeof(u64) + 255) & ~(size_t)255;
    if (arena_off_ + bytes > arena_.bytes()) { overflow_lane_ = cur_lane_; throw ArenaOverflow{ arena_off_ + bytes }; }
    u64 *p = reinterpret_cast<u64 *>(static_cast<char *>(arena_.p()) + arena_off_);
    arena_off_ += bytes;
    return p;
}

void Engine::ws_reset(size_t need)
{
    if (need) {                                   // grow the lane whose bump allocator overflowed
        switch_lane(overflow_lane_);
        if (need > arena_.bytes()) {
            counters_[C_ARENA_GROW]++;
            sync();
            arena_.release();
            arena_.alloc(need + need / 4);
        }
    }
    switch_lane(0);
    arena_off_ = 0;
    job_seq_ = job_seq_base_;
    lanes_[1].off = 0;
    lanes_[1].job_seq = job_seq_base_ + 128;      // lane 1 uses the upper half of the call's job-cache slots,
    lanes_[2].off = 0;
    lanes_[2].job_seq = job_seq_base_ + 240;      // lane 2 (a handful of tables) its last sixteen
}

template <class T> const T *Engine::upload_jobs(const std::vector<T> &v)
{
    if (v.empty()) return nullptr;
    const size_t bytes = v.size() * sizeof(T);
    // lanes 1 and 2 run next to other lanes' kernels: their tables must stay inside their own slot ranges (ws_reset; a pipelined
    // ComputePowers uses 12 of lane 1's 112)
    if ((cur_lane_ == 1 && job_seq_ >= job_seq_base_ + 240) || (cur_lane_ == 2 && job_seq_ >= job_seq_base_ + 256))
        throw std::logic_error("job-table slots of a side lane exhausted");
    if (job_seq_ >= job_slots_.size()) job_slots_.resize(job_seq_ + 1);
    JobSlot &slot = job_slots_[job_seq_++];
    for (JobWay &w : slot.way)
        if (w.host.size() == bytes && std::memcmp(w.host.data(), v.data(), bytes) == 0) {
            counters_[C_JOB_HIT]++;
            w.stamp = ++job_stamp_;
            return reinterpret_cast<const T *>(w.buf.p());                                            // unchanged since an earlier call
        }
    counters_[C_JOB_UPLOAD]++;
    JobWay *lru = &slot.way[0];
    for (JobWay &w : slot.way) if (w.stamp < lru->stamp) lru = &w;                                    // least recently used
    JobWay &way = *lru;
    way.stamp = ++job_stamp_;
    if (way.buf.bytes() < bytes) {
        counters_[C_JOB_REALLOC]++;
        sync();                                   // the old buffer may still be read by queued kernels
        way.buf.alloc(bytes + bytes / 2);
    }
    // through a pinned staging area so the copy is truly asynchronous and the std::vector may die
    const size_t aligned = (bytes + 63) & ~(size_t)63;
    if (stage_off_ + aligned > stage_bytes_) {
        counters_[C_STAGE_WRAP]++;
        sync();                                   // everything staged so far has been consumed
        if (aligned > stage_bytes_) {
            if (stage_) (void)hipHostFree(stage_);
            stage_bytes_ = aligned * 2;
            HIP_CHECK(hipHostMalloc(&stage_, stage_bytes_));
        }
        stage_off_ = 0;
    }
    char *hp = static_cast<char *>(stage_) + stage_off_;
    std::memcpy(hp, v.data(), bytes);
    stage_off_ += aligned;
    HIP_CHECK(hipMemcpyAsync(way.buf.p(), hp, bytes, hipMemcpyHostToDevice, st_));
    way.host.assign(reinterpret_cast<const unsigned char *>(v.data()), reinterpret_cast<const unsigned char *>(v.data()) + bytes);
    return reinterpret_cast<const T *>(way.buf.p());
}

void Engine::throttle_inflight()
{
    while (inflight_count_ >= (size_t)std::max(1, max_inflight_)) {
        HIP_CHECK(hipEventSynchronize(inflight_[inflight_head_]));
        inflight_head_ = (inflight_head_ + 1) % inflight_.size();
        inflight_count_--;
    }
}

void Engine::mark_inflight()
{
    const size_t cap = (size_t)std::max(1, max_inflight_) + 1;
    if (inflight_.size() != cap) {                               // (re)build the ring; nothing pending survives a resize
        for (size_t i = 0; i < inflight_count_; i++) (void)hipEventSynchronize(inflight_[(inflight_head_ + i) % inflight_.size()]);
        for (hipEvent_t e : inflight_) (void)hipEventDestroy(e);
        inflight_.assign(cap, nullptr);
        for (hipEvent_t &e : inflight_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        inflight_head_ = inflight_count_ = 0;
    }
    throttle_inflight();
    const size_t at = (inflight_head_ + inflight_count_) % inflight_.size();
    HIP_CHECK(hipEventRecord(inflight_[at], st_));
    inflight_count_++;
}

void Engine::recycle_powers(std::unique_ptr<Powers> p)
{
    Enter g(this);
    if (!p) return;
    // keep the most recently released buffers: a context that changes its batch shape must not be left with a pool
    // full of buffers of the old shape (every call would then allocate and free ~100 MB)
    if (powers_pool_.size() >= 6) powers_pool_.erase(powers_pool_.begin());
    powers_pool_.push_back(std::move(p));
}

// run `fn` with the arena, growing it and retrying when the bump allocator overflows
template <class F> static void with_arena(Engine *e, F &&fn, void (Engine::*reset)(size_t))
{
    size_t need = 0;
    for (int attempt = 0; attempt < 40; attempt++) {
        (e->*reset)(need);
        try { fn(); return; }
        catch (const ArenaOverflow &o) { need = std::max(o.need * 2, need); }
    }
    throw std::runtime_error("workspace arena could not be sized");
}
struct EngineAccess {
    template <class F> static void run(Engine *e, F &&fn) { with_arena(e, fn, &Engine::ws_reset); }
};
#define WITH_ARENA(...) EngineAccess::run(this, [&]() __VA_ARGS__)
#define TIER1_SLOTS() job_seq_base_ = 512

// ============================================================================ device building blocks
// Does the multiply-accumulate launch of a level use its three-product form?  It needs carry-free chunks of at least 7 terms for
// every limb, and it pays for long chains only: macbench (profiles/r04_mac_kara.txt) has it 1.2 % slower at 44 terms per chain
// (16M-4096) and 2 % faster at 150 (256M-4096 has 310).  APSU_HE_MAC_KARA=0/1 forces it off / on wherever it is usable.
bool Engine::mac_kara(int lvl, uint32_t mean_cnt) const
{
    if (mac_kara_ == 0 || (mac_kara_ < 0 && mean_cnt < 96)) return false;
    for (int j = 0; j <= lvl; j++) {
        const int bits = 64 - __builtin_clzll(hp_.key_q[j]), sh = (bits + 1) / 2;
        if (62 - 2 * sh < 3) return false;
    }
    return true;
}

void Engine::d_ntt(u64 *data, size_t count, const int *modmap, int period, bool inverse)
{
    PROF(inverse ? P_NTT_INV : P_NTT_FWD, count);
    // (launches over the data primes alone -- map_ct() and its suffixes -- may take the narrow-moduli build, kernels.hip launch_ntt)
    const bool narrow_only = data_primes_narrow_ && modmap >= map_ct() && (modmap - map_ct()) + period <= hp_.K;
    launch_ntt(hp_.logn, inverse, data, count, tabs(), modmap, period, st_, ntt_latency_limbs_, narrow_only);
}

bool Engine::d_relinearize(u64 *ct3, size_t ct_stride, int batch, const RelinKeys &rk, int chain_idx, u64 *ext_out, int n_ext, u64 **defer_moddown)
{
    const int L = chain_idx + 1;
    const size_t n = hp_.n;
    u64 *tdec = ws((size_t)batch * (L + 1) * L * n);
    {   // tdec[b][I][J] = NTT_I(c2[b][J] mod m_I): the decomposition is the NTT's load (no separate pass)
        std::vector<const u64 *> src((size_t)batch * (L + 1) * L);
        for (int b = 0; b < batch; b++)
            for (int I = 0; I <= L; I++)
                for (int J = 0; J < L; J++) src[((size_t)b * (L + 1) + I) * L + J] = ct3 + (size_t)b * ct_stride + ((size_t)2 * L + J) * n;
        // sources are residues of q_0 .. q_{L-1}, targets q_0 .. q_{L-1} and the special prime: with SEAL's narrow coefficient primes
        // the lazy transform takes them as they are (16M-4096: 56-bit sources into 56- and 50-bit targets)
        u64 max_src = 0;
        for (int J = 0; J < L; J++) max_src = std::max(max_src, hp_.key_q[J]);
        bool nored = hp_.logn <= 14;
        for (int I = 0; I <= L && nored; I++) nored = ntt_gather_nored_ok(hp_.key_q[I < L ? I : hp_.K - 1], max_src, hp_.logn);
        PROF(P_NTT_FWD, src.size());
        launch_ntt_gather(hp_.logn, upload_jobs(src), tdec, src.size(), tabs(), map_ks(chain_idx), (L + 1) * L, st_, nored, ntt_latency_limbs_);
    }
    u64 *acc = ws((size_t)batch * 2 * (L + 1) * n);
    // the inverse transform leaves its twist to the mod-down kernel, whose own constants absorb it (unrolled sizes)
    const bool raw = L <= 4;
    const int *amap = raw ? map_ksacc_raw(chain_idx) : map_ksacc(chain_idx);
    // (the inner product formed by the inverse transform's load, the way the BEHZ tensor product is, was measured in round 3:
    //  2 % SLOWER on the whole query -- six operand streams per output and tdec read twice; tools/microbench/intt_ks_experiment.hip)
    { PROFW(P_KEYSWITCH, (size_t)batch * n * ((size_t)L * (L + 1) + 2 * (L + 1))); launch_ks_inner(dkey(), L, tdec, rk.data.u(), acc, n, batch, st_); }
    d_ntt(acc, (size_t)batch * 2 * (L + 1), amap, L + 1, true);
    if (defer_moddown) {
        *defer_moddown = raw ? acc : nullptr;
        if (raw) return false;                                   // the caller's epilogue kernel takes it from here
    }
    const bool fuse_ext = ext_out && n_ext > 0 && hlevel(chain_idx).L == hlevel(chain_idx).nB && L <= 3;
    { PROFW(P_KEYSWITCH, (size_t)batch * n * (2 * (L + 1) + 4 * L) + (fuse_ext ? (size_t)n_ext * 2 * (hlevel(chain_idx).L + hlevel(chain_idx).nB + 1) * n : 0));
      launch_ks_moddown(dkey(), L, acc, ct3, ct_stride, n, batch, st_, dlevel(chain_idx), fuse_ext ? ext_out : nullptr, fuse_ext ? n_ext : 0, raw); }
    return fuse_ext;
}

// ============================================================================ tier 1
// Tier-1 operands are host pointers by default; with apsu_he_set_tier1_on_device they are device (or page-locked) memory and the
// calls only queue their work -- unified addressing lets one copy kind serve both
#define H2D(dst, src, words) HIP_CHECK(hipMemcpyAsync(dst, src, (words) * sizeof(u64), tier1_device_ ? hipMemcpyDefault : hipMemcpyHostToDevice, st_))
#define D2H(dst, src, words) HIP_CHECK(hipMemcpyAsync(dst, src, (words) * sizeof(u64), tier1_device_ ? hipMemcpyDefault : hipMemcpyDeviceToHost, st_))
#define D2D(dst, src, words) HIP_CHECK(hipMemcpyAsync(dst, src, (words) * sizeof(u64), hipMemcpyDeviceToDevice, st_))

void Engine::transform_to_ntt(u64 *ct, int polys, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    const size_t w = (size_t)polys * (chain_idx + 1) * hp_.n;
    WITH_ARENA({
        u64 *d = ws(w);
        H2D(d, ct, w);
        d_ntt_ct(d, polys, chain_idx, false);
        D2H(ct, d, w);
        tier1_done();
    });
}

void Engine::transform_from_ntt(u64 *ct, int polys, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    const size_t w = (size_t)polys * (chain_idx + 1) * hp_.n;
    WITH_ARENA({
        u64 *d = ws(w);
        H2D(d, ct, w);
        d_ntt_ct(d, polys, chain_idx, true);
        D2H(ct, d, w);
        tier1_done();
    });
}

void Engine::multiply_plain_ntt(const u64 *ct, const u64 *pt_ntt, u64 *out, int polys, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    const size_t n = hp_.n, L = chain_idx + 1, w = polys * L * n;
    WITH_ARENA({
        u64 *d = ws(w), *p = ws(L * n), *o = ws(w);
        H2D(d, ct, w);
        H2D(p, pt_ntt, L * n);
        launch_dyadic_plain(dlevel(chain_idx), d, p, o, polys, n, 1, 0, st_);
        D2H(out, o, w);
        tier1_done();
    });
}

static bool is_monomial(const u64 *pt, size_t count)
{
    size_t nz = 0;
    for (size_t k = 0; k < count && nz < 2; k++) nz += pt[k] != 0;
    return nz == 1;
}

void Engine::transform_plain_to_ntt(const u64 *pt, size_t pt_coeffs, u64 *out, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    const size_t n = hp_.n, L = chain_idx + 1;
    if (pt_coeffs > n) throw std::invalid_argument("plaintext has too many coefficients");
    WITH_ARENA({
        u64 *p = ws(n), *o = ws(L * n);
        HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(u64), st_));
        H2D(p, pt, pt_coeffs);
        launch_lift(dlevel(chain_idx), p, o, n, 1, nullptr, st_);
        d_ntt_ct(o, 1, chain_idx, false);
        D2H(out, o, L * n);
        tier1_done();
    });
}

// multiply_plain on coefficient-form ct and plaintext (bin_bundle.cpp:334): lift, NTT both,
// dyadic product, INTT.  SEAL's monomial shortcut (no lift) is honoured.
void Engine::multiply_plain(const u64 *ct, const u64 *pt, size_t pt_coeffs, u64 *out, int polys, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    const size_t n = hp_.n, L = chain_idx + 1, w = polys * L * n;
    if (pt_coeffs > n) throw std::invalid_argument("plaintext has too many coefficients");
    const unsigned char mono = !tier1_device_ && is_monomial(pt, pt_coeffs) ? 1 : 0;      // (device operands: flagged by a kernel below)
    WITH_ARENA({
        u64 *d = ws(w), *p = ws(n), *pl = ws(L * n), *o = ws(w), *flag = ws(1);
        H2D(d, ct, w);
        HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(u64), st_));
        H2D(p, pt, pt_coeffs);
        if (tier1_device_) launch_flag_monomial(p, n, 1, reinterpret_cast<unsigned char *>(flag), st_);
        else HIP_CHECK(hipMemcpyAsync(flag, &mono, 1, hipMemcpyHostToDevice, st_));
        launch_lift(dlevel(chain_idx), p, pl, n, 1, reinterpret_cast<const unsigned char *>(flag), st_);
        d_ntt_ct(pl, 1, chain_idx, false);
        d_ntt_ct(d, polys, chain_idx, false);
        launch_dyadic_plain(dlevel(chain_idx), d, pl, o, polys, n, 1, 0, st_);
        d_ntt_ct(o, polys, chain_idx, true);
        D2H(out, o, w);
        tier1_done();
    });
}

void Engine::add(u64 *acc, const u64 *x, int polys, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    const size_t w = (size_t)polys * (chain_idx + 1) * hp_.n;
    WITH_ARENA({
        u64 *a = ws(w), *b = ws(w);
        H2D(a, acc, w);
        H2D(b, x, w);
        launch_add(dlevel(chain_idx), a, b, polys, hp_.n, 1, st_);
        D2H(acc, a, w);
        tier1_done();
    });
}

void Engine::add_plain(u64 *ct, const u64 *pt, size_t pt_coeffs, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    const size_t n = hp_.n, L = chain_idx + 1;
    if (pt_coeffs > n) throw std::invalid_argument("plaintext has too many coefficients");
    WITH_ARENA({
        u64 *c0 = ws(L * n), *p = ws(n);
        H2D(c0, ct, L * n);
        HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(u64), st_));
        H2D(p, pt, pt_coeffs);
        std::vector<PlainJob> jobs{ PlainJob{ c0, p } };
        { PROF(P_OTHER, 0); launch_add_plain(dlevel(chain_idx), upload_jobs(jobs), n, 1, st_); }
        D2H(ct, c0, L * n);
        tier1_done();
    });
}

void Engine::multiply(const u64 *a, const u64 *b, u64 *out3, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    const size_t n = hp_.n, L = chain_idx + 1;
    const int E = hlevel(chain_idx).L + hlevel(chain_idx).nB + 1;
    const bool square = (a == b);
    WITH_ARENA({
        const int nop = square ? 1 : 2;
        u64 *in = ws((size_t)nop * 2 * L * n);
        H2D(in, a, 2 * L * n);
        if (!square) H2D(in + 2 * L * n, b, 2 * L * n);
        u64 *ext = ws((size_t)nop * 2 * E * n);
        { PROF(P_BEHZ_EXT, 0); launch_behz_ext(dlevel(chain_idx), hlevel(chain_idx).L, hlevel(chain_idx).nB, in, L * n, 1, ext, n, nop * 2, st_); }
        d_ntt(ext, (size_t)nop * 2 * E, map_ext(chain_idx), E, false);
        u64 *d = ws((size_t)3 * E * n), *o = ws(3 * L * n);
        std::vector<TensorJob> tj{ TensorJob{ ext, square ? ext : ext + (size_t)2 * E * n, d } };
        { PROF(P_TENSOR, 0); launch_tensor(dlevel(chain_idx), upload_jobs(tj), n, 1, st_); }
        d_ntt(d, (size_t)3 * E, map_ext_fin(chain_idx), E, true);        // the finish below applies the twist where it can
        std::vector<FinishJob> fj{ FinishJob{ d, o, 1, 0 } };
        { PROF(P_BEHZ_FINISH, 0); launch_behz_finish(dlevel(chain_idx), hlevel(chain_idx).L, hlevel(chain_idx).nB, upload_jobs(fj), false, n, 1, st_); }
        D2H(out3, o, 3 * L * n);
        tier1_done();
    });
}

void Engine::relinearize(u64 *ct3, const RelinKeys &rk, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    if (!hp_.using_keyswitching) throw std::logic_error("parameters do not support key switching");
    const size_t n = hp_.n, L = chain_idx + 1;
    WITH_ARENA({
        u64 *d = ws(3 * L * n);
        H2D(d, ct3, 3 * L * n);
        d_relinearize(d, 3 * L * n, 1, rk, chain_idx);
        D2H(ct3, d, 2 * L * n);
        tier1_done();
    });
}

void Engine::mod_switch_to_next(u64 *ct, int polys, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    if (chain_idx == 0) throw std::invalid_argument("end of modulus switching chain reached");
    const size_t n = hp_.n, L = chain_idx + 1;
    WITH_ARENA({
        u64 *d = ws(polys * L * n), *o = ws(polys * (L - 1) * n);
        H2D(d, ct, polys * L * n);
        { PROF(P_MODSWITCH, 0); launch_modswitch(dlevel(chain_idx), d, polys * L * n, polys, o, n, 1, st_); }
        D2H(ct, o, polys * (L - 1) * n);
        tier1_done();
    });
}

void Engine::clear_irrelevant_bits(u64 *ct, int polys)
{
    Enter g(this);
    TIER1_SLOTS();
    const size_t w = (size_t)polys * hp_.n;
    WITH_ARENA({
        u64 *d = ws(w);
        H2D(d, ct, w);
        launch_clear_bits(d, w, hp_.irrelevant_bit_count, st_);
        D2H(ct, d, w);
        tier1_done();
    });
}

// ============================================================================ tier 2: uploads
std::unique_ptr<RelinKeys> Engine::upload_relin_keys(const u64 *rk_host)
{
    Enter g(this);
    if (!hp_.using_keyswitching) throw std::logic_error("parameters do not support key switching");
    auto rk = std::make_unique<RelinKeys>();
    const size_t w = (size_t)(hp_.K - 1) * 2 * hp_.K * hp_.n;
    rk->data.alloc(w * sizeof(u64));
    HIP_CHECK(hipMemcpy(rk->data.p(), rk_host, w * sizeof(u64), hipMemcpyHostToDevice));
    return rk;
}

static void bundle_shape(const PSUParams &psu, const HeParams &hp, uint32_t degree, Bundle &b)
{
    const uint32_t ps = psu.query_params.ps_low_degree;
    b.degree = degree;
    b.use_ps = ps > 1 && ps < degree;                       // receiver_osn.cpp:520-522
    const uint32_t h = ps + 1;
    b.H = ps ? degree / h : 0;
    b.r = ps ? degree % h : 0;
    b.pt_level = std::min(hp.first_chain_idx, ps ? 2 : 1);  // bin_bundle.cpp:385-389
    b.ntt_count = 0;
    for (uint32_t i = 0; i <= degree; i++)
        if ((!ps && i != 0) || (ps && (i % h) != 0)) b.ntt_count++;
    if (!b.use_ps && ps && degree > ps)
        throw std::invalid_argument("ps_low_degree == 1 leaves coefficient-form plaintexts that eval() cannot multiply");
}

// bytes of one NTT-form plaintext slot at a level: dense 64-bit words, or bit-packed rows (the same widths as DevLevel::mac_bits)
size_t Engine::slot_bytes(int chain_idx, bool packed) const
{
    const size_t n = hp_.n;
    if (!packed) return (size_t)(chain_idx + 1) * n * sizeof(u64);
    size_t b = 0;
    for (int j = 0; j <= chain_idx; j++) {
        const u32 w = packed_row_bits(hp_.key_q[j]);
        b += n * w / 8;
    }
    return b;
}

void Engine::pack_bundle(Bundle &b)
{
    if (!packed_rows_ || b.packed) return;
    const size_t n = hp_.n;
    const int high = hp_.clamp_chain_idx(1);
    b.ntt_slot_bytes = slot_bytes(b.pt_level, true);
    b.lifted_slot_bytes = slot_bytes(high, true);
    const size_t H = b.lifted.bytes() / ((size_t)(high + 1) * n * sizeof(u64));
    if (b.ntt_count) {
        DevBuf pk;
        pk.alloc(b.ntt_count * b.ntt_slot_bytes + 16);
        HIP_CHECK(hipMemsetAsync(static_cast<char *>(pk.p()) + b.ntt_count * b.ntt_slot_bytes, 0, 16, st_));
        launch_pack_rows(dlevel(b.pt_level), b.pt_level + 1, b.ntt.u(), pk.p(), b.ntt_slot_bytes, n, b.ntt_count, st_);
        sync();
        b.ntt = std::move(pk);
    }
    if (H) {
        DevBuf pk;
        pk.alloc(H * b.lifted_slot_bytes + 16);
        HIP_CHECK(hipMemsetAsync(static_cast<char *>(pk.p()) + H * b.lifted_slot_bytes, 0, 16, st_));
        launch_pack_rows(dlevel(high), high + 1, b.lifted.u(), pk.p(), b.lifted_slot_bytes, n, H, st_);
        sync();
        b.lifted = std::move(pk);
    }
    b.packed = true;
}

void Engine::unpack_bundle(Bundle &b)
{
    if (!b.packed) return;
    const size_t n = hp_.n;
    const int high = hp_.clamp_chain_idx(1);
    if (b.ntt_count) {
        DevBuf dn;
        dn.alloc(b.ntt_count * (size_t)(b.pt_level + 1) * n * sizeof(u64));
        launch_unpack_rows(dlevel(b.pt_level), b.pt_level + 1, b.ntt.p(), b.ntt_slot_bytes, dn.u(), n, b.ntt_count, st_);
        sync();
        b.ntt = std::move(dn);
    }
    const size_t H = b.use_ps ? b.H : 0;
    if (H && b.lifted.bytes()) {
        DevBuf dn;
        dn.alloc(H * (size_t)(high + 1) * n * sizeof(u64));
        launch_unpack_rows(dlevel(high), high + 1, b.lifted.p(), b.lifted_slot_bytes, dn.u(), n, H, st_);
        sync();
        b.lifted = std::move(dn);
    }
    b.packed = false;
    b.ntt_slot_bytes = b.lifted_slot_bytes = 0;
}

std::unique_ptr<Bundle> Engine::upload_bundle(uint32_t bundle_idx, uint32_t cache_idx, uint32_t n_coeffs,
                                              const u64 *const *coeff_ptrs, const unsigned char *is_ntt)
{
    Enter g(this);
    TIER1_SLOTS();
    if (!has_psu_) throw std::logic_error("context was created without PSUParams");
    if (!n_coeffs) throw std::invalid_argument("batched_coeffs is empty");
    if (n_coeffs - 1 > psu_.table_params.max_items_per_bin) throw std::invalid_argument("degree exceeds max_items_per_bin");
    if (bundle_idx >= psu_.bundle_idx_count) throw std::invalid_argument("bundle_idx out of range");
    auto b = std::make_unique<Bundle>();
    b->bundle_idx = bundle_idx;
    b->cache_idx = cache_idx;
    bundle_shape(psu_, hp_, n_coeffs - 1, *b);
    const uint32_t ps = psu_.query_params.ps_low_degree, h = ps + 1;
    const size_t n = hp_.n, Lpt = b->pt_level + 1;
    const int high = hp_.clamp_chain_idx(1);
    const size_t Lh = high + 1;
    for (uint32_t i = 0; i < n_coeffs; i++) {
        bool want = (!ps && i != 0) || (ps && (i % h) != 0);
        if ((is_ntt[i] != 0) != want) throw std::invalid_argument("plaintext NTT form does not match the BinBundle layout rule");
    }
    b->ntt.alloc(b->ntt_count * Lpt * n * sizeof(u64));
    b->a0.alloc(n * sizeof(u64));
    HIP_CHECK(hipMemcpy(b->a0.p(), coeff_ptrs[0], n * sizeof(u64), hipMemcpyHostToDevice));
    size_t slot = 0;
    std::vector<unsigned char> mono;
    std::vector<const u64 *> cf;
    for (uint32_t i = 1; i < n_coeffs; i++) {
        if (is_ntt[i]) {
            HIP_CHECK(hipMemcpy(b->ntt.u() + slot * Lpt * n, coeff_ptrs[i], Lpt * n * sizeof(u64), hipMemcpyHostToDevice));
            slot++;
        } else {
            cf.push_back(coeff_ptrs[i]);
            mono.push_back(is_monomial(coeff_ptrs[i], n) ? 1 : 0);
        }
    }
    if (b->use_ps) {
        // K4: pre-lift and pre-NTT the coefficient-form plaintexts a_{i*h} at the high level; this
        // is what multiply_plain (bin_bundle.cpp:334) recomputes on every call in the reference.
        const size_t H = cf.size();
        b->lifted.alloc(H * Lh * n * sizeof(u64));
        WITH_ARENA({
            u64 *raw = ws(H * n), *flags = ws((H + 7) / 8 + 1);
            for (size_t i = 0; i < H; i++) H2D(raw + i * n, cf[i], n);
            HIP_CHECK(hipMemcpyAsync(flags, mono.data(), H, hipMemcpyHostToDevice, st_));
            launch_lift(dlevel(high), raw, b->lifted.u(), n, (int)H, reinterpret_cast<const unsigned char *>(flags), st_);
            d_ntt_ct(b->lifted.u(), H, high, false);
            sync();
        });
    }
    pack_bundle(*b);
    return b;
}

// ============================================================================ tier 2: ComputePowers
void Engine::build_schedule_for(Sched &s, const std::vector<char> &member)
{
    s = Sched{};
    const auto &nodes = dag_.nodes();
    const uint32_t max_power = *dag_.target_powers().rbegin();
    std::vector<char> is_parent(max_power + 1, 0);
    uint32_t depth = 0;
    for (auto &kv : nodes) {
        if (!member[kv.first]) continue;
        depth = std::max(depth, kv.second.depth);
        if (!kv.second.is_source()) { is_parent[kv.second.parents.first] = 1; is_parent[kv.second.parents.second] = 1; }
    }
    std::vector<PowersDag::PowersNode> order;
    for (auto &kv : nodes) if (member[kv.first]) order.push_back(kv.second);
    std::stable_sort(order.begin(), order.end(), [&](const auto &a, const auto &b) {
        if (a.depth != b.depth) return a.depth < b.depth;
        if (is_parent[a.power] != is_parent[b.power]) return is_parent[a.power] > is_parent[b.power];
        return a.power < b.power;
    });
    s.slot_of.assign(max_power + 1, -1);
    for (size_t i = 0; i < order.size(); i++) { s.slot_power.push_back(order[i].power); s.slot_of[order[i].power] = (int)i; }
    s.levels.assign(depth + 1, Sched::Level{ 0, 0, 0 });
    for (uint32_t d = 0; d <= depth; d++) {
        int s0 = -1, s1 = -1, sp = -1;
        for (size_t i = 0; i < order.size(); i++) {
            if (order[i].depth != d) continue;
            if (s0 < 0) s0 = (int)i;
            s1 = (int)i + 1;
            if (is_parent[order[i].power]) sp = (int)i + 1;
        }
        if (s0 < 0) s0 = s1 = 0;                      // no node of this subset at this depth
        if (sp < 0) sp = s0;
        s.levels[d] = Sched::Level{ s0, s1, sp };
    }
    for (size_t i = 0; i < order.size(); i++)
        if (!order[i].is_source())
            s.nodes.push_back({ (int)i, s.slot_of[order[i].parents.first], s.slot_of[order[i].parents.second] });
    const uint32_t ps = psu_.query_params.ps_low_degree;
    for (uint32_t p : dag_.target_powers()) {
        if (!member[p]) continue;
        if (!ps || p <= ps) s.low_powers.push_back(p);
        else s.high_powers.push_back(p);
    }
}

void Engine::build_schedule()
{
    const auto &nodes = dag_.nodes();
    const uint32_t max_power = *dag_.target_powers().rbegin();
    build_schedule_for(sched_, std::vector<char>(max_power + 1, 1));
    // The powers that stay at the low level and the Paterson-Stockmeyer high powers usually descend from different
    // query powers (all four reference parameter sets).  Then ComputePowers is two independent chains: the second one
    // runs on its own stream, next to the first and to the BinBundle inner products that only need the low powers.
    split_ok_ = false;
    if (sched_.high_powers.empty() || sched_.low_powers.empty() || dag_.depth() == 0) return;
    auto closure = [&](const std::vector<uint32_t> &targets) {
        std::vector<char> m(max_power + 1, 0);
        std::vector<uint32_t> stack(targets.begin(), targets.end());
        while (!stack.empty()) {
            const uint32_t p = stack.back();
            stack.pop_back();
            if (m[p]) continue;
            m[p] = 1;
            const auto &nd = nodes.at(p);
            if (!nd.is_source()) { stack.push_back(nd.parents.first); stack.push_back(nd.parents.second); }
        }
        return m;
    };
    const auto ml = closure(sched_.low_powers), mh = closure(sched_.high_powers);
    for (uint32_t p = 0; p <= max_power; p++) if (ml[p] && mh[p]) return;
    build_schedule_for(sched_low_, ml);
    build_schedule_for(sched_high_, mh);
    split_ok_ = true;
}

// One walk over a (sub)schedule of the PowersDag on the current lane: sources, the level-synchronous products, and the
// final per-power conversions (receiver_osn.cpp:395-488).
//   stage 0: workspace + sources;  stage d >= 1: the products of depth d;  stage -1: final conversions.
// The stages of one walk must run in order on one lane; `run` carries the walk's buffers between them, so two walks
// (the halves of a split DAG) can be interleaved level by level on two lanes.
struct Engine::DagRun { u64 *pwf = nullptr, *ext = nullptr, *dbuf = nullptr; size_t arena_mark = 0; int ext_done = -1; };   // ext_done: level whose parents are already extended

void Engine::run_dag(const Sched &s, DagRun &run, int stage, int nb, const u64 *const *src, bool on_device, const RelinKeys *rk,
                     Powers &pwr, bool do_low, bool do_high)
{
    Powers *pw = &pwr;
    const size_t n = hp_.n;
    const int first = hp_.first_chain_idx, high = hp_.clamp_chain_idx(1);
    const int low_target = pw->low_level;
    const size_t Lf = first + 1, Ef = dlevel(first) ? (size_t)(hlevel(first).L + hlevel(first).nB + 1) : 0;
    const size_t P = s.slot_power.size();
    const size_t slot_w = 3 * Lf * n;                       // (c0, c1, c2 scratch) per power and bundle index
    const size_t Lh = high + 1, Eh = hlevel(high).L + hlevel(high).nB + 1;
    auto slot_ptr = [&](int slot, int b) { return run.pwf + ((size_t)slot * nb + b) * slot_w; };
    auto ext_ptr = [&](int slot, int b) { return run.ext + ((size_t)slot * nb + b) * 2 * Ef * n; };
    {
        // sources (receiver_osn.cpp:304-317)
        if (stage == 0) {
            run.pwf = ws(P * nb * slot_w);
            int si = 0;
            std::vector<CtJob> cj;                               // device-resident sources: one gather launch
            for (auto &kv : dag_.nodes()) {
                if (!kv.second.is_source()) continue;
                if (s.slot_of[kv.first] < 0) { si++; continue; }        // belongs to the other half of a split DAG
                for (int b = 0; b < nb; b++) {
                    const u64 *sp = src[(size_t)b * dag_.source_count() + si];
                    if (on_device) cj.push_back(CtJob{ sp, slot_ptr(s.slot_of[kv.first], b) });
                    else {                                       // host sources: a plain copy, then the same check in place (round 6)
                        H2D(slot_ptr(s.slot_of[kv.first], b), sp, 2 * Lf * n);
                        cj.push_back(CtJob{ slot_ptr(s.slot_of[kv.first], b), slot_ptr(s.slot_of[kv.first], b) });
                    }
                }
                si++;
            }
            if (!cj.empty()) { PROF(P_OTHER, 0); launch_copy_sources(upload_jobs(cj), 2 * Lf * n, (int)cj.size(), dlevel(first), (int)Lf, n, bad_source_ + query_seq_ % BAD_SLOTS, query_seq_, st_); }
            if (s.levels.size() > 1) {
                run.ext = ws(P * nb * 2 * Ef * n);
                size_t max_nodes = 0;
                for (size_t d = 1; d < s.levels.size(); d++) max_nodes = std::max(max_nodes, (size_t)(s.levels[d].s1 - s.levels[d].s0));
                run.dbuf = ws(max_nodes * nb * 3 * Ef * n);
            }
            run.arena_mark = arena_off_;
            return;
        }
        if (stage > 0) {
            if ((size_t)stage >= s.levels.size()) return;
            u64 *dbuf = run.dbuf;
            {
                const size_t d = (size_t)stage;
                arena_off_ = run.arena_mark;
                // extend + NTT the parents that became available at depth d-1
                const auto &pl = s.levels[d - 1];
                const int npar = pl.sp - pl.s0;
                if (npar > 0) {
                    // (parents that came out of a key switch were extended by its mod-down kernel: run.ext_done)
                    if (run.ext_done != (int)d - 1) { PROFW(P_BEHZ_EXT, (size_t)npar * nb * 2 * n * (Lf + Ef)); launch_behz_ext(dlevel(first), hlevel(first).L, hlevel(first).nB, slot_ptr(pl.s0, 0), slot_w, 2, ext_ptr(pl.s0, 0), n, npar * nb, st_); }
                    d_ntt(ext_ptr(pl.s0, 0), (size_t)npar * nb * 2 * Ef, map_ext(first), (int)Ef, false);
                }
                const auto &cl = s.levels[d];
                const int nn = cl.s1 - cl.s0;
                if (nn <= 0) return;
                std::vector<TensorJob> tj;
                std::vector<FinishJob> fj;
                for (auto &nd : s.nodes) {
                    if (nd[0] < cl.s0 || nd[0] >= cl.s1) continue;
                    for (int b = 0; b < nb; b++) {
                        u64 *dd = dbuf + ((size_t)(nd[0] - cl.s0) * nb + b) * 3 * Ef * n;
                        tj.push_back(TensorJob{ ext_ptr(nd[1], b), ext_ptr(nd[2], b), dd });
                        fj.push_back(FinishJob{ dd, slot_ptr(nd[0], b), 1, 0 });
                    }
                }
                if (fuse_tensor_) {                                                                                              // :422/:424
                    PROF(P_NTT_FUSED, tj.size() * 3 * Ef);
                    launch_intt_tensor(hp_.logn, upload_jobs(tj), (int)tj.size(), (int)Ef, Ef * n, nullptr, 0, tabs(), map_ext_fin(first), (int)Ef, st_, ntt_latency_limbs_);
                } else {
                    { PROF(P_TENSOR, 0); launch_tensor(dlevel(first), upload_jobs(tj), n, (int)tj.size(), st_); }
                    d_ntt(dbuf, (size_t)nn * nb * 3 * Ef, map_ext_fin(first), (int)Ef, true);
                }
                { PROFW(P_BEHZ_FINISH, fj.size() * 3 * n * (Ef + Lf)); launch_behz_finish(dlevel(first), hlevel(first).L, hlevel(first).nB, upload_jobs(fj), false, n, (int)fj.size(), st_); }
                if (hp_.using_keyswitching && nn > 0) {                                                                          // :431
                    const int npar_here = ((size_t)d + 1 < s.levels.size()) ? (cl.sp - cl.s0) : 0;
                    if (d_relinearize(slot_ptr(cl.s0, 0), slot_w, nn * nb, *rk, first, npar_here ? ext_ptr(cl.s0, 0) : nullptr, npar_here * nb))
                        run.ext_done = (int)d;
                }
            }
            return;
        }
        arena_off_ = run.arena_mark;
        // final per-power conversions (receiver_osn.cpp:459-487)
        auto convert = [&](const std::vector<uint32_t> &powers, int target, u64 *out) {
            if (powers.empty()) return;
            const int cnt = (int)powers.size() * nb;
            std::vector<CtJob> jobs;
            int lvl = first;
            u64 *cur = nullptr;
            auto dst_for = [&](int level_after) -> u64 * {
                return level_after == target ? out : ws((size_t)cnt * 2 * (level_after + 1) * n);
            };
            if (first == target) {
                for (size_t i = 0; i < powers.size(); i++)
                    for (int b = 0; b < nb; b++)
                        jobs.push_back(CtJob{ slot_ptr(s.slot_of[powers[i]], b), out + ((size_t)b * powers.size() + i) * 2 * Lf * n });
                { PROF(P_OTHER, 0); launch_copy_jobs(upload_jobs(jobs), 2 * Lf * n, cnt, st_); }
                return;
            }
            cur = dst_for(first - 1);
            for (size_t i = 0; i < powers.size(); i++)
                for (int b = 0; b < nb; b++)
                    jobs.push_back(CtJob{ slot_ptr(s.slot_of[powers[i]], b), cur + ((size_t)b * powers.size() + i) * 2 * (Lf - 1) * n });
            { PROFW(P_MODSWITCH, (size_t)cnt * 2 * n * (2 * Lf - 1)); launch_modswitch_jobs(dlevel(first), upload_jobs(jobs), 2, n, cnt, st_); }                     // :463,471,478
            lvl = first - 1;
            while (lvl > target) {
                u64 *nxt = dst_for(lvl - 1);
                { PROFW(P_MODSWITCH, (size_t)cnt * 2 * n * (2 * lvl + 1)); launch_modswitch(dlevel(lvl), cur, (size_t)2 * (lvl + 1) * n, 2, nxt, n, cnt, st_); }
                cur = nxt;
                lvl--;
            }
        };
        if (do_low && first == low_target && !s.low_powers.empty()) {
            // no level change: the forward NTT (:467,475) reads the slots and writes the packed output directly
            const size_t np = s.low_powers.size();
            std::vector<const u64 *> srcp(np * nb * 2 * Lf);
            for (int b = 0; b < nb; b++)
                for (size_t i = 0; i < np; i++)
                    for (size_t pl = 0; pl < 2 * Lf; pl++)
                        srcp[(((size_t)b * np + i) * 2 * Lf) + pl] = slot_ptr(s.slot_of[s.low_powers[i]], b) + pl * n;
            // (limb j of a slot is already a canonical residue of q_j: nothing to reduce)
            bool nored = hp_.logn <= 14;
            for (size_t j = 0; j < Lf && nored; j++) nored = ntt_gather_nored_ok(hp_.key_q[j], hp_.key_q[j], hp_.logn);
            PROF(P_NTT_FWD, srcp.size());
            launch_ntt_gather(hp_.logn, upload_jobs(srcp), pw->low.u(), srcp.size(), tabs(), map_ct(), (int)Lf, st_, nored, ntt_latency_limbs_);
        } else if (do_low) {
            convert(s.low_powers, low_target, pw->low.u());
            d_ntt_ct(pw->low.u(), (size_t)pw->n_low * nb * 2, low_target, false);                      // :467,475
        }
        if (do_high && pw->n_high) {
            convert(s.high_powers, high, pw->high.u());
            // derived form used by eval_patstock's ct x ct products and coefficient-form plaintext products
            { PROFW(P_BEHZ_EXT, (size_t)pw->n_high * nb * 2 * n * (Lh + Eh)); launch_behz_ext(dlevel(high), hlevel(high).L, hlevel(high).nB, pw->high.u(), Lh * n, 1, pw->hext.u(), n, (int)(pw->n_high * nb * 2), st_); }
            d_ntt(pw->hext.u(), (size_t)pw->n_high * nb * 2 * Eh, map_ext(high), (int)Eh, false);
        }
    }
}

std::unique_ptr<Powers> Engine::compute_powers(const uint32_t *bundle_indices, int nb, const u64 *const *src, bool on_device,
                                               const RelinKeys *rk)
{
    Enter g(this);
    if (!has_psu_) throw std::logic_error("context was created without PSUParams");
    if (nb <= 0) throw std::invalid_argument("no bundle indices given");
    if (hp_.using_keyswitching && dag_.depth() > 0 && !rk) throw std::invalid_argument("relinearization keys are required");
    // one coefficient prime = no key switching: the reference then leaves every product unrelinearised (receiver_osn.cpp:427-432)
    // and multiplies the longer ciphertexts further: a path of its own (every shipped single-prime set has depth 0)
    if (nks_) return compute_powers_nks(bundle_indices, nb, src, on_device);
    const Sched &s = sched_;
    const size_t n = hp_.n;
    job_seq_base_ = 0;                                       // job-cache slots 0..255: ComputePowers
    const int high = hp_.clamp_chain_idx(1), low = hp_.clamp_chain_idx(2);
    const uint32_t ps = psu_.query_params.ps_low_degree;
    const int low_target = ps ? low : high;                 // receiver_osn.cpp:459-487
    const size_t Ll_ = low_target + 1, Lh_ = high + 1, Eh_ = hlevel(high).L + hlevel(high).nB + 1;
    const size_t need_low = s.low_powers.size() * nb * 2 * Ll_ * n * sizeof(u64);
    const size_t need_high = s.high_powers.size() * nb * 2 * Lh_ * n * sizeof(u64);
    const size_t need_hext = s.high_powers.size() * nb * 2 * Eh_ * n * sizeof(u64);
    std::unique_ptr<Powers> pw;
    // (round 4) a pooled buffer whose last reader -- the evaluation of the query in front -- is still running would make this
    // query's second stream wait for that evaluation's end; with the caller's overlap promise the engine rather keeps TWO buffers of a shape and
    // takes the one whose reader is done (a caller that frees its powers right after queueing the evaluation, as the reference's
    // RunQuery does, then gets the alternation for free: 68 MB more at 16M-4096)
    {
        // (the rule itself: sched_policy.h, pick_pooled_buffer -- enumerated on the CPU tier)
        std::vector<PoolEntryState> st(powers_pool_.size());
        for (size_t i = 0; i < powers_pool_.size(); i++) {
            Powers &c = *powers_pool_[i];
            st[i].fits = c.low.bytes() == need_low && c.high.bytes() == need_high && c.hext.bytes() == need_hext;
            st[i].last_use_set = c.last_use_set;
            st[i].last_use_done = st[i].fits && inputs_ready_ && c.last_use_set && hipEventQuery(c.last_use) == hipSuccess;
        }
        const int pick = pick_pooled_buffer(st.data(), st.size(), inputs_ready_);
        if (pick >= 0) {
            pw = std::move(powers_pool_[pick]);
            powers_pool_.erase(powers_pool_.begin() + pick);
        }
    }
    const bool recycled = (bool)pw;
    // The ordering rules -- which walk, behind which events -- are a pure function of this state (sched_policy.h, plan_walk; every
    // state is enumerated against the no-unordered-writer invariant in tests/test_host_logic.py).
    WalkState wst{};
    wst.recycled = recycled;
    wst.last_use_set = recycled && pw->last_use_set;
    wst.last_use_done = wst.last_use_set && hipEventQuery(pw->last_use) == hipSuccess;
    wst.high_async = recycled && pw->high_async && pw->high_ready;
    wst.split_ok = split_ok_;
    wst.prof_on = prof_on_;
    wst.split_mode = two_stream_mode_ >= 0 ? two_stream_mode_ : two_stream_default_;       // API override, then environment
    wst.pipe_cp = pipe_cp_;
    wst.force_pipe = force_pipe_;
    wst.inputs_ready = inputs_ready_;
    wst.on_device = on_device;
    // "busy": an evaluation queued earlier is still running (a query that finds the device idle takes the split walk: the merged
    // chain is ~3 % slower for one query alone, profiles/r02_pipe_sweep.txt; a stream of queued queries takes the pipelined one)
    if (pipe_cp_ && inflight_count_ > 0 && !inflight_.empty())
        wst.device_busy = hipEventQuery(inflight_[(inflight_head_ + inflight_count_ - 1) % inflight_.size()]) == hipErrorNotReady;
    const WalkPlan plan = plan_walk(wst);
    // a pooled buffer whose high half was produced on the second stream and never consumed: the main stream must not
    // overwrite it before those kernels have finished
    if (plan.main_waits_high_ready) HIP_CHECK(hipStreamWaitEvent(st_, pw->high_ready, 0));
    if (!pw) pw = std::make_unique<Powers>();
    if (++query_seq_ == 0) query_seq_ = 1;                   // (0 = "never computed")
    pw->seq = query_seq_;
    pw->nb = nb;
    pw->bundle_indices.assign(bundle_indices, bundle_indices + nb);
    pw->low_level = low_target;
    pw->high_level = high;
    pw->n_low = (uint32_t)s.low_powers.size();
    pw->n_high = (uint32_t)s.high_powers.size();
    if (!recycled) {
        counters_[C_POWERS_ALLOC]++;
        pw->low.alloc(need_low);
        if (pw->n_high) {
            pw->high.alloc(need_high);
            pw->hext.alloc(need_hext);
        }
    }

    struct LaneGuard { Engine *e; ~LaneGuard() { e->switch_lane(0); } } lane_guard{ this };
    PhaseSpan cp_span;
    if (phase_on_) {
        phase_close_query();
        cp_span.phase = PH_COMPUTE_POWERS;
        cp_span.a = phase_event(st_);
        query_start_ = phase_event(st_);
    }
    // Two-stream walk: the high-power half of the DAG runs on the second stream next to the low-power half and to the
    // BinBundle inner products.  Measured on 16M-4096 (tools/pipe_sweep.py, tools/rank_cost.py): 3.84 ->
    // 3.65 ms for the whole query (four bundle indices), 0.95 -> 0.86 ms per rank with one bundle index.  Default: on
    // whenever the PowersDag splits; APSU_HE_SPLIT=0/1 (read at apsu_he_create) or apsu_he_set_two_stream force it.  Event profiling always takes
    // the one-stream walk: a launch bracketed by events next to another stream's kernels measures the sharing, not the kernel.
    const bool split = plan.walk != WALK_ONE_STREAM;         // (host inputs are uploaded per lane and end with a sync)
    pw->high_async = split;
    if (split && !pw->high_ready) HIP_CHECK(hipEventCreateWithFlags(&pw->high_ready, hipEventDisableTiming));
    // Pipelined queries (round 4): the WHOLE walk on the second stream -- one merged chain of launches -- next to the evaluation of
    // the query in front, the main stream only evaluates: -2.5 % on the rate of queued 16M-4096 queries, -7.7 % on the N = 8 shard
    // (profiles/r04_ab_pipe_cp.txt).  Needs the caller's apsu_he_set_query_overlap promise; only while an evaluation queued earlier is
    // still running and only into a buffer nobody reads any more (plan_walk).  `last_use` is consumed here and set again by
    // eval_bundles alone: a buffer that was computed and given back without an evaluation keeps no mark and takes the conservative order.
    const bool pipe = plan.walk == WALK_PIPELINED;
    pw->low_async = pipe;
    WITH_ARENA({
        for (hipEvent_t *e : { &cp_span.b, &cp_span.b2 }) if (*e) { phase_pool_.push_back(*e); *e = nullptr; }   // a retry after arena growth
        if (pipe) {
            switch_lane(1);
            if (plan.side_waits_last_use) HIP_CHECK(hipStreamWaitEvent(st_, pw->last_use, 0));
            DagRun r;
            run_dag(sched_, r, 0, nb, src, on_device, rk, *pw, true, true);
            for (int d = 1; d < (int)sched_.levels.size(); d++) run_dag(sched_, r, d, nb, src, on_device, rk, *pw, true, true);
            run_dag(sched_, r, -1, nb, src, on_device, rk, *pw, true, true);
            HIP_CHECK(hipEventRecord(pw->high_ready, st_));
            switch_lane(0);
        } else if (!split) {
            DagRun r;
            run_dag(sched_, r, 0, nb, src, on_device, rk, *pw, true, true);
            for (int d = 1; d < (int)sched_.levels.size(); d++) run_dag(sched_, r, d, nb, src, on_device, rk, *pw, true, true);
            run_dag(sched_, r, -1, nb, src, on_device, rk, *pw, true, true);
        } else {
            // everything queued so far on the main stream (the previous query's evaluation may still read a pooled
            // Powers buffer) precedes the second stream's work; the two walks are queued level by level so that
            // both streams have work from the start
            DagRun rl, rh;
            const int depth = (int)std::max(sched_low_.levels.size(), sched_high_.levels.size());
            // (round 4: with device-resident inputs that the caller has declared complete -- apsu_he_set_query_overlap; lane 1 reads
            //  the sources and the relinearisation keys, which a caller may otherwise still be producing on the main stream -- the
            //  second stream waits for the LAST READER of this powers buffer -- the
            //  evaluation that used it before it went back to the pool -- not for everything the main stream has queued: the next
            //  query's high-power chain then runs next to the tail of the query in front of it, whose launches leave CUs idle)
            //  (plan.side_waits_main / side_waits_last_use; a pooled buffer whose reader left no mark waits for all)
            if (plan.side_waits_main) HIP_CHECK(hipEventRecord(ev_main_, st_));
            run_dag(sched_low_, rl, 0, nb, src, on_device, rk, *pw, true, false);
            switch_lane(1);
            if (plan.side_waits_main) HIP_CHECK(hipStreamWaitEvent(st_, ev_main_, 0));
            else if (plan.side_waits_last_use) HIP_CHECK(hipStreamWaitEvent(st_, pw->last_use, 0));
            run_dag(sched_high_, rh, 0, nb, src, on_device, rk, *pw, false, true);
            for (int d = 1; d < depth; d++) {
                switch_lane(0);
                run_dag(sched_low_, rl, d, nb, src, on_device, rk, *pw, true, false);
                switch_lane(1);
                run_dag(sched_high_, rh, d, nb, src, on_device, rk, *pw, false, true);
            }
            switch_lane(0);
            run_dag(sched_low_, rl, -1, nb, src, on_device, rk, *pw, true, false);
            switch_lane(1);
            run_dag(sched_high_, rh, -1, nb, src, on_device, rk, *pw, false, true);
            // (starting the high-power chain only when the low-power chain -- the evaluation's critical path -- has finished was
            //  measured in round 3: level on the whole query, 21 % slower on the N = 8 shard; profiles/r03_ab_fusions.txt)
            HIP_CHECK(hipEventRecord(pw->high_ready, st_));
            if (phase_on_ && cp_span.a && !cp_span.b2) cp_span.b2 = phase_event(st_);
            switch_lane(0);
        }
        if (phase_on_ && cp_span.a && !cp_span.b) cp_span.b = phase_event(st_);
        // device-resident inputs: no sync, consumers (eval_bundles, powers_download) are ordered on / synchronise
        // with the engine's streams.  Host inputs: the caller's buffers must have been consumed before returning.
        if (!on_device) sync();
    });
    if (phase_on_ && cp_span.a && cp_span.b) phase_spans_.push_back(cp_span);
    if (pipe) counters_[C_PIPELINED]++;                      // (once per call: the walk above is queued again after an arena growth)
    pw->last_use_set = false;                               // consumed above; only an evaluation of THESE powers sets it again
    return pw;
}


void Engine::download_power(const Powers &pw, uint32_t bundle_idx, uint32_t power, u64 *out, size_t capacity_words, int *chain_idx,
                            int *is_ntt)
{
    Enter g(this);
    if (!has_psu_) throw std::invalid_argument("context has no PSUParams");
    const int b = pw.slot_of(bundle_idx);
    if (b < 0) throw std::invalid_argument("bundle index not present");
    const uint32_t ps = psu_.query_params.ps_low_degree;
    const bool low = !ps || power <= ps;
    const int lvl = low ? pw.low_level : pw.high_level;
    uint32_t idx;
    if (low) {
        if (power < 1 || power > pw.n_low) throw std::invalid_argument("power not available");
        idx = power - 1;
    } else {
        if (power % (ps + 1) != 0 || power / (ps + 1) > pw.n_high) throw std::invalid_argument("power not available");
        idx = power / (ps + 1) - 1;
    }
    const size_t words = (size_t)power_size(power) * (lvl + 1) * hp_.n;               // the stored slot may be zero-padded beyond that
    if (capacity_words < words) throw std::invalid_argument("output buffer too small");
    sync();
    check_sources(pw);
    const u64 *src = (low ? pw.low.u() : pw.high.u()) + ((size_t)b * (low ? pw.n_low : pw.n_high) + idx) * pw.polys * (lvl + 1) * hp_.n;
    HIP_CHECK(hipMemcpy(out, src, words * sizeof(u64), hipMemcpyDeviceToHost));
    if (chain_idx) *chain_idx = lvl;
    if (is_ntt) *is_ntt = low ? 1 : 0;
}

// ============================================================================ tier 2: BinBundle construction
// raw = the batched polynomial's coefficient-form plaintexts [degree+1][n] mod t on the device.  Applies the
// layout rule of the BatchedPlaintextPolyn ctor (bin_bundle.cpp:385-420): NTT-form coefficients are lifted
// and transformed at pt_level; coefficient-form a_{i*h} are pre-lifted (honouring SEAL's monomial
// shortcut of multiply_plain) and pre-NTT'd at the high level; a_0 stays raw.
void Engine::finish_bundle(Bundle &b, const u64 *raw)
{
    const uint32_t ps = psu_.query_params.ps_low_degree, h = ps + 1, degree = b.degree;
    const size_t n = hp_.n, Lpt = b.pt_level + 1;
    const int high = hp_.clamp_chain_idx(1);
    const size_t Lh = high + 1;
    b.ntt.alloc(b.ntt_count * Lpt * n * sizeof(u64));
    b.a0.alloc(n * sizeof(u64));
    const size_t H = b.use_ps ? b.H : 0;
    if (H) b.lifted.alloc(H * Lh * n * sizeof(u64));
    D2D(b.a0.u(), raw, n);
    unsigned char *flags = reinterpret_cast<unsigned char *>(ws((degree + 1 + 7) / 8 + 1));
    launch_flag_monomial(raw, n, (int)degree + 1, flags, st_);
    size_t slot = 0, hi = 0;
    uint32_t d = 1;
    while (d <= degree) {
        const bool is_ntt = (!ps && d != 0) || (ps && (d % h) != 0);
        uint32_t e = d;
        while (e + 1 <= degree && (((!ps) || ((e + 1) % h) != 0) == is_ntt)) e++;
        const size_t run = e - d + 1;
        if (is_ntt) {
            launch_lift(dlevel(b.pt_level), raw + (size_t)d * n, b.ntt.u() + slot * Lpt * n, n, (int)run, nullptr, st_);
            slot += run;
        } else if (H) {
            launch_lift(dlevel(high), raw + (size_t)d * n, b.lifted.u() + hi * Lh * n, n, (int)run, flags + d, st_);
            hi += run;
        }
        d = e + 1;
    }
    d_ntt_ct(b.ntt.u(), b.ntt_count, b.pt_level, false);
    if (H) d_ntt_ct(b.lifted.u(), H, high, false);
}

std::unique_ptr<Bundle> Engine::random_bundle(uint32_t bundle_idx, uint32_t cache_idx, uint32_t degree, u64 seed)
{
    Enter g(this);
    TIER1_SLOTS();
    if (!has_psu_) throw std::logic_error("context was created without PSUParams");
    if (degree > psu_.table_params.max_items_per_bin) throw std::invalid_argument("degree exceeds max_items_per_bin");
    auto b = std::make_unique<Bundle>();
    b->bundle_idx = bundle_idx;
    b->cache_idx = cache_idx;
    bundle_shape(psu_, hp_, degree, *b);
    const size_t n = hp_.n;
    // coefficient d, index k of the batched polynomial = splitmix64 stream at offset d*n + k (mod t), coefficient form
    WITH_ARENA({
        u64 *raw = ws((size_t)(degree + 1) * n);
        launch_fill_random(raw, (size_t)(degree + 1) * n, seed, hp_.t, st_);
        finish_bundle(*b, raw);
        sync();
    });
    pack_bundle(*b);
    return b;
}

std::unique_ptr<Bundle> Engine::build_bundle(uint32_t bundle_idx, uint32_t cache_idx, const u64 *roots, const uint32_t *counts,
                                             uint32_t bins, uint32_t stride)
{
    Enter g(this);
    TIER1_SLOTS();
    if (!has_psu_) throw std::logic_error("context was created without PSUParams");
    if (!hp_.batching) throw std::logic_error("plain_modulus does not support batching");
    const size_t n = hp_.n;
    if (bins > n) throw std::invalid_argument("more bins than batching slots");
    if (bundle_idx >= psu_.bundle_idx_count) throw std::invalid_argument("bundle_idx out of range");
    uint32_t degree = 0;
    for (uint32_t s = 0; s < bins; s++) {
        if (counts[s] > stride) throw std::invalid_argument("bin count exceeds stride");
        degree = std::max(degree, counts[s]);
    }
    // a bin may hold at most max_items_per_bin - 1 items (receiver_db.cpp:388-389: insertion requires size < max)
    if (degree > psu_.table_params.max_items_per_bin) throw std::invalid_argument("bin size exceeds max_items_per_bin");
    for (uint32_t s = 0; s < bins; s++)
        for (uint32_t r = 0; r < counts[s]; r++)
            if (roots[(size_t)s * stride + r] >= hp_.t) throw std::invalid_argument("field element is not reduced modulo plain_modulus");
    auto b = std::make_unique<Bundle>();
    b->bundle_idx = bundle_idx;
    b->cache_idx = cache_idx;
    bundle_shape(psu_, hp_, degree, *b);
    const int tid = hp_.plain_id();
    WITH_ARENA({
        u64 *droots = ws((size_t)bins * stride + 1);
        uint32_t *dcounts = reinterpret_cast<uint32_t *>(ws((bins + 1) / 2 + 1));
        if (bins) {
            H2D(droots, roots, (size_t)bins * stride);
            HIP_CHECK(hipMemcpyAsync(dcounts, counts, bins * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
        }
        u64 *poly = ws((size_t)(degree + 1) * n);                  // [d][slot] slot values of the batched polynomial
        launch_polyn_with_roots(droots, dcounts, bins, stride, degree, make_mod(hp_.t), poly, n, st_);
        // BatchEncoder::encode (bin_bundle.cpp:409): slot permutation, then inverse negacyclic NTT mod t
        u64 *raw = ws((size_t)(degree + 1) * n);
        launch_scatter_slots(poly, reinterpret_cast<const uint32_t *>(d_slot_map_.p()), raw, n, (int)degree + 1, st_);
        d_ntt(raw, degree + 1, map_ct() + tid, 1, true);
        finish_bundle(*b, raw);
        sync();
    });
    pack_bundle(*b);
    return b;
}

// ---- N2: BinBundle image ------------------------------------------------------------------------------------
void Engine::algebraize_items(const unsigned char *items, size_t count, bool items_on_device, u64 *out, bool out_on_device)
{
    Enter g(this);
    if (!has_psu_) throw std::logic_error("context was created without PSUParams");
    if (!count) return;
    TIER1_SLOTS();
    const u32 felts = psu_.item_params.felts_per_item, bpf = psu_.item_bit_count_per_felt, bits = psu_.item_bit_count;
    WITH_ARENA({
        const unsigned char *src = items;
        if (!items_on_device) {
            u64 *d = ws((count * 16 + 7) / 8);
            HIP_CHECK(hipMemcpyAsync(d, items, count * 16, hipMemcpyHostToDevice, st_));
            src = reinterpret_cast<const unsigned char *>(d);
        }
        u64 *dst = out_on_device ? out : ws(count * felts);
        { PROF(P_OTHER, 0); launch_algebraize(src, count, felts, bpf, bits, dst, st_); }
        if (!out_on_device) D2H(out, dst, count * felts);
        sync();
    });
}

namespace {
struct ImageHeader {                     // little-endian, 256 bytes
    char magic[8];                       // "APSUHEB2"
    uint64_t header_bytes, total_bytes;
    uint64_t n, t, K, q[8];
    uint32_t ps_low_degree, max_items_per_bin;
    uint32_t bundle_idx, cache_idx, degree, use_ps, H, r, pt_level, row_format;   // row_format: 0 dense 64-bit words, 1 bit-packed rows (was `reserved`)
    uint64_t ntt_count, ntt_bytes, lifted_bytes, a0_bytes;
    uint64_t checksum;                   // checksum64 over the payload
    unsigned char pad[256 - 8 - 16 - 88 - 8 - 32 - 32 - 8];
};
static_assert(sizeof(ImageHeader) == 256, "image header layout");
uint64_t fnv1a64(const unsigned char *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
// Payload checksum: four interleaved FNV-1a-style lanes over 64-bit little-endian words, folded together with the tail bytes and
// the length at the end.  (Byte-serial FNV-1a, the "APSUHEB1" images of round 2, is one dependent multiply per byte: under
// 1 GB/s, minutes for a 75 GiB database on every load; this runs at memory speed.)
uint64_t checksum64(const unsigned char *p, size_t n)
{
    const uint64_t P = 1099511628211ull;
    uint64_t h[4] = { 1469598103934665603ull, 0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull };
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        for (int k = 0; k < 4; k++) h[k] = (h[k] ^ w[k]) * P;
    }
    uint64_t r = fnv1a64(p + i, n - i);
    for (int k = 0; k < 4; k++) { r = (r ^ h[k]) * P; r ^= r >> 31; }
    return r ^ (uint64_t)n;
}
}

size_t Engine::bundle_image_size(const Bundle &b) const { return sizeof(ImageHeader) + b.ntt.bytes() + b.lifted.bytes() + b.a0.bytes(); }

size_t Engine::save_bundle(const Bundle &b, unsigned char *buf, size_t capacity)
{
    Enter g(this);
    if (!has_psu_) throw std::logic_error("context was created without PSUParams");
    const size_t total = bundle_image_size(b);
    if (capacity < total) throw std::invalid_argument("image buffer too small");
    sync();
    ImageHeader hd;
    std::memset(&hd, 0, sizeof(hd));
    std::memcpy(hd.magic, "APSUHEB2", 8);
    hd.header_bytes = sizeof(hd); hd.total_bytes = total;
    hd.n = hp_.n; hd.t = hp_.t; hd.K = hp_.K;
    for (int j = 0; j < hp_.K && j < 8; j++) hd.q[j] = hp_.key_q[j];
    hd.ps_low_degree = psu_.query_params.ps_low_degree; hd.max_items_per_bin = psu_.table_params.max_items_per_bin;
    hd.bundle_idx = b.bundle_idx; hd.cache_idx = b.cache_idx; hd.degree = b.degree; hd.use_ps = b.use_ps; hd.H = b.H; hd.r = b.r;
    hd.pt_level = (uint32_t)b.pt_level; hd.ntt_count = b.ntt_count; hd.row_format = b.packed ? 1 : 0;
    hd.ntt_bytes = b.ntt.bytes(); hd.lifted_bytes = b.lifted.bytes(); hd.a0_bytes = b.a0.bytes();
    unsigned char *p = buf + sizeof(hd);
    if (hd.ntt_bytes) HIP_CHECK(hipMemcpy(p, b.ntt.p(), hd.ntt_bytes, hipMemcpyDeviceToHost));
    p += hd.ntt_bytes;
    if (hd.lifted_bytes) HIP_CHECK(hipMemcpy(p, b.lifted.p(), hd.lifted_bytes, hipMemcpyDeviceToHost));
    p += hd.lifted_bytes;
    HIP_CHECK(hipMemcpy(p, b.a0.p(), hd.a0_bytes, hipMemcpyDeviceToHost));
    hd.checksum = checksum64(buf + sizeof(hd), total - sizeof(hd));
    std::memcpy(buf, &hd, sizeof(hd));
    return total;
}

std::unique_ptr<Bundle> Engine::load_bundle(const unsigned char *buf, size_t size)
{
    Enter g(this);
    if (!has_psu_) throw std::logic_error("context was created without PSUParams");
    ImageHeader hd;
    if (size < sizeof(hd)) throw std::invalid_argument("BinBundle image is truncated");
    std::memcpy(&hd, buf, sizeof(hd));
    if (std::memcmp(hd.magic, "APSUHEB2", 8) != 0 || hd.header_bytes != sizeof(hd)) throw std::invalid_argument("not a BinBundle image");
    if (hd.total_bytes != size || hd.total_bytes != sizeof(hd) + hd.ntt_bytes + hd.lifted_bytes + hd.a0_bytes)
        throw std::invalid_argument("BinBundle image size mismatch");
    bool same = hd.n == hp_.n && hd.t == hp_.t && hd.K == (uint64_t)hp_.K && hd.ps_low_degree == psu_.query_params.ps_low_degree &&
                hd.max_items_per_bin == psu_.table_params.max_items_per_bin;
    for (int j = 0; same && j < hp_.K && j < 8; j++) same = hd.q[j] == hp_.key_q[j];
    if (!same) throw std::invalid_argument("BinBundle image was built for different parameters");
    if (checksum64(buf + sizeof(hd), size - sizeof(hd)) != hd.checksum) throw std::invalid_argument("BinBundle image is corrupt (checksum)");
    auto b = std::make_unique<Bundle>();
    b->bundle_idx = hd.bundle_idx; b->cache_idx = hd.cache_idx;
    bundle_shape(psu_, hp_, hd.degree, *b);                      // re-derive and cross-check the shape
    const size_t n = hp_.n, Lh = hp_.clamp_chain_idx(1) + 1;
    if (hd.row_format > 1 || (hd.row_format == 1 && !hp_.using_keyswitching)) throw std::invalid_argument("BinBundle image header is inconsistent");
    const bool img_packed = hd.row_format == 1;
    const size_t ntt_slot = slot_bytes(b->pt_level, img_packed), lift_slot = slot_bytes(hp_.clamp_chain_idx(1), img_packed);
    const size_t want_ntt = b->ntt_count ? b->ntt_count * ntt_slot + (img_packed ? 16 : 0) : 0;
    const size_t want_lift = b->use_ps && b->H ? (size_t)b->H * lift_slot + (img_packed ? 16 : 0) : 0;
    (void)Lh;
    if (b->H != hd.H || b->r != hd.r || (uint32_t)b->use_ps != hd.use_ps || (uint32_t)b->pt_level != hd.pt_level || b->ntt_count != hd.ntt_count ||
        hd.ntt_bytes != want_ntt || hd.a0_bytes != n * sizeof(u64) || hd.lifted_bytes != want_lift || hd.bundle_idx >= psu_.bundle_idx_count)
        throw std::invalid_argument("BinBundle image header is inconsistent");
    const unsigned char *p = buf + sizeof(hd);
    b->ntt.alloc(hd.ntt_bytes); b->lifted.alloc(hd.lifted_bytes); b->a0.alloc(hd.a0_bytes);
    if (hd.ntt_bytes) HIP_CHECK(hipMemcpy(b->ntt.p(), p, hd.ntt_bytes, hipMemcpyHostToDevice));
    p += hd.ntt_bytes;
    if (hd.lifted_bytes) HIP_CHECK(hipMemcpy(b->lifted.p(), p, hd.lifted_bytes, hipMemcpyHostToDevice));
    p += hd.lifted_bytes;
    HIP_CHECK(hipMemcpy(b->a0.p(), p, hd.a0_bytes, hipMemcpyHostToDevice));
    b->packed = img_packed;
    if (img_packed) { b->ntt_slot_bytes = ntt_slot; b->lifted_slot_bytes = lift_slot; }
    // an image of the other row format is converted to this context's (APSU_HE_PACKED_ROWS)
    if (packed_rows_ && !b->packed) pack_bundle(*b);
    else if (!packed_rows_ && b->packed) unpack_bundle(*b);
    return b;
}

size_t Engine::download_coeff(const Bundle &b, uint32_t d, u64 *out, size_t capacity, int *kind)
{
    Enter g(this);
    sync();
    const uint32_t ps = psu_.query_params.ps_low_degree, h = ps + 1;
    const size_t n = hp_.n;
    if (d > b.degree) throw std::invalid_argument("degree out of range");
    const u64 *src;
    size_t words;
    int k;
    const bool is_ntt = (!ps && d != 0) || (ps && (d % h) != 0);
    if (d == 0) { src = b.a0.u(); words = n; k = 0; }
    else if (is_ntt) { src = b.ntt.u() + (size_t)(d - (ps ? d / h : 0) - 1) * (b.pt_level + 1) * n; words = (size_t)(b.pt_level + 1) * n; k = 1; }
    else {
        if (!b.use_ps) throw std::invalid_argument("coefficient is not stored for this bundle");
        const size_t Lh = hp_.clamp_chain_idx(1) + 1;
        src = b.lifted.u() + (size_t)(d / h - 1) * Lh * n; words = Lh * n; k = 2;
    }
    if (capacity < words) throw std::invalid_argument("output buffer too small");
    if (b.packed && k != 0) {                                    // one bit-packed slot -> dense words
        const int lvl = k == 1 ? b.pt_level : hp_.clamp_chain_idx(1);
        const size_t sb = k == 1 ? b.ntt_slot_bytes : b.lifted_slot_bytes;
        const size_t slot = k == 1 ? (size_t)(d - (ps ? d / h : 0) - 1) : (size_t)(d / h - 1);
        const char *base = static_cast<const char *>(k == 1 ? b.ntt.p() : b.lifted.p());
        DevBuf tmp(words * sizeof(u64));
        launch_unpack_rows(dlevel(lvl), lvl + 1, base + slot * sb, sb, tmp.u(), n, 1, st_);
        sync();
        HIP_CHECK(hipMemcpy(out, tmp.p(), words * sizeof(u64), hipMemcpyDeviceToHost));
    } else HIP_CHECK(hipMemcpy(out, src, words * sizeof(u64), hipMemcpyDeviceToHost));
    if (kind) *kind = k;
    return words;
}

// ============================================================================ N3: seeded objects expanded on the device
void Engine::seed_expand(int chain_idx, int count, const u64 *seeds, u64 *const *dst)
{
    Enter g(this);
    if (count <= 0) return;
    if (!seeds || !dst) throw std::invalid_argument("null argument");
    const bool key_level = chain_idx < 0 || chain_idx == hp_.K - 1;
    if (!key_level) check_level(chain_idx);
    const int L = key_level ? hp_.K : chain_idx + 1;
    // The device list holds 8192 rejected words per object.  The rejection rate is ~q / 2^64 per word, so ~60-bit primes with a large
    // L * n may legitimately exceed it (and L may exceed the device tables): those objects are expanded by the host's
    // sample_poly_uniform (seal_codec.cpp) and uploaded -- slower, same words.  APSU_HE_SEED_EXPAND_HOST=1 forces that path (tests).
    auto host_expand = [&] {
        std::vector<u64> q(hp_.key_q.begin(), hp_.key_q.begin() + L), buf((size_t)L * hp_.n);
        for (int i = 0; i < count; i++) {
            if (!dst[i]) throw std::invalid_argument("null destination");
            sealio::sample_poly_uniform(seeds + (size_t)i * 8, q.data(), (size_t)L, hp_.n, buf.data());
            HIP_CHECK(hipMemcpyAsync(dst[i], buf.data(), buf.size() * sizeof(u64), hipMemcpyHostToDevice, st_));
            sync();                                              // buf is reused
        }
    };
    if (L > DMAXL || seed_expand_host_) { host_expand(); return; }
    const DevLevel *lv = nullptr;
    if (key_level && hp_.K - 1 > hp_.first_chain_idx) {
        // the key level is not a data level: a DevLevel-shaped view that carries its moduli only
        if (!d_key_level_.p()) {
            std::vector<unsigned char> raw(sizeof(DevLevel), 0);
            DevLevel *d = reinterpret_cast<DevLevel *>(raw.data());
            d->L = hp_.K;
            for (int j = 0; j < hp_.K; j++) d->q[j] = make_mod(hp_.key_q[j]);
            d_key_level_.alloc(sizeof(DevLevel));
            HIP_CHECK(hipMemcpy(d_key_level_.p(), raw.data(), sizeof(DevLevel), hipMemcpyHostToDevice));
        }
        lv = reinterpret_cast<const DevLevel *>(d_key_level_.p());
    } else lv = dlevel(key_level ? hp_.K - 1 : chain_idx);
    TIER1_SLOTS();
    job_seq_ = job_seq_base_;
    std::vector<u64> mm(L);
    for (int j = 0; j < L; j++) mm[j] = ~(u64)0 - (~(u64)0 % hp_.key_q[j]) - 1;          // util/rlwe.cpp: max_multiple
    std::vector<SeedJob> jobs(count);
    for (int i = 0; i < count; i++) {
        for (int k = 0; k < 8; k++) jobs[i].seed.w[k] = seeds[(size_t)i * 8 + k];
        jobs[i].dst = dst[i];
        if (!dst[i]) throw std::invalid_argument("null destination");
    }
    const size_t need = ((size_t)count * (1 + 8192) + 1) * sizeof(u32);
    if (d_seed_rej_.bytes() < need) {
        sync();
        d_seed_rej_.alloc(need * 2);
        HIP_CHECK(hipMemsetAsync(d_seed_rej_.p(), 0, d_seed_rej_.bytes(), st_));
    }
    u32 *rej = reinterpret_cast<u32 *>(d_seed_rej_.p());
    int *overflow = reinterpret_cast<int *>(rej + (size_t)count * (1 + 8192));
    HIP_CHECK(hipMemsetAsync(overflow, 0, sizeof(int), st_));
    { PROF(P_OTHER, 0); launch_seed_expand(upload_jobs(jobs), count, lv, L, upload_jobs(mm), hp_.n, rej, overflow, st_); }
    int h_overflow = 0;
    HIP_CHECK(hipMemcpyAsync(&h_overflow, overflow, sizeof(int), hipMemcpyDeviceToHost, st_));
    sync();
    if (h_overflow) {
        HIP_CHECK(hipMemsetAsync(d_seed_rej_.p(), 0, d_seed_rej_.bytes(), st_));
        sync();
        host_expand();                                           // more rejected words than the device list holds: the host redoes these objects
    }
}

// ============================================================================ N4: masks, packing, loopback decrypt
static uint32_t plain_modulus_len(u64 t)
{
    // receiver_osn.cpp:54-57: smallest len with (1 << len) - 1 >= plain_modulus
    uint32_t len = 1;
    while ((((u64)1 << len) - 1) < t) len++;
    return len;
}

void Engine::mask_generate(u64 seed, uint32_t count, u64 *masks_dev, u64 *values_host, u64 *blocks_host)
{
    mask_generate_impl(count, masks_dev, values_host, blocks_host, [&](u64 *vals, size_t words) { launch_fill_random(vals, words, seed, hp_.t, st_); });
}

void Engine::mask_generate_blake2xb(const u64 seed[8], u64 first_value, uint32_t count, u64 *masks_dev, u64 *values_host, u64 *blocks_host)
{
    Blake2xbSeed sd;
    for (int i = 0; i < 8; i++) sd.w[i] = seed[i];
    mask_generate_impl(count, masks_dev, values_host, blocks_host,
                       [&](u64 *vals, size_t words) { launch_fill_blake2xb(vals, words, sd, first_value, hp_.t, st_); });
}

void Engine::mask_generate_impl(uint32_t count, u64 *masks_dev, u64 *values_host, u64 *blocks_host, const std::function<void(u64 *, size_t)> &fill)
{
    Enter g(this);
    if (!has_psu_) throw std::logic_error("context was created without PSUParams");
    if (!hp_.batching) throw std::logic_error("plain_modulus does not support batching");
    if (!count) return;
    const size_t n = hp_.n;
    const uint32_t items = psu_.items_per_bundle, felts = psu_.item_params.felts_per_item;
    const int tid = hp_.plain_id();
    TIER1_SLOTS();
    WITH_ARENA({
        u64 *vals = ws((size_t)count * n);
        { PROF(P_OTHER, 0); fill(vals, (size_t)count * n); }                                                       // :248-251
        // BatchEncoder::encode (:271): slot permutation, inverse negacyclic NTT mod t
        { PROF(P_OTHER, 0); launch_scatter_slots(vals, reinterpret_cast<const uint32_t *>(d_slot_map_.p()), masks_dev, n, (int)count, st_); }
        d_ntt(masks_dev, count, map_ct() + tid, 1, true);
        if (blocks_host) {                                                                                        // :256-266
            u64 *blk = ws((size_t)count * items * 2);
            { PROF(P_OTHER, 0); launch_pack_blocks(vals, n, items, felts, plain_modulus_len(hp_.t), blk, (int)count, st_); }
            D2H(blocks_host, blk, (size_t)count * items * 2);
        }
        if (values_host) D2H(values_host, vals, (size_t)count * n);
        sync();
    });
}

void Engine::decrypt_decode(const u64 *sk_ntt_host, const u64 *cts, bool on_device, uint32_t count, u64 *values_host, u64 *blocks_host)
{
    Enter g(this);
    if (!has_psu_) throw std::logic_error("context was created without PSUParams");
    if (!hp_.batching) throw std::logic_error("plain_modulus does not support batching");
    if (!count) return;
    const size_t n = hp_.n;
    const uint32_t items = psu_.items_per_bundle, felts = psu_.item_params.felts_per_item;
    const int tid = hp_.plain_id();
    for (size_t k = 0; k < n; k++)
        if (sk_ntt_host[k] >= hp_.key_q[0]) throw std::invalid_argument("secret key is not reduced modulo q_0");
    TIER1_SLOTS();
    WITH_ARENA({
        u64 *sk = ws(n);
        H2D(sk, sk_ntt_host, n);
        const u64 *ct = cts;
        if (!on_device) {
            u64 *c = ws((size_t)count * 2 * n);
            H2D(c, cts, (size_t)count * 2 * n);
            ct = c;
        }
        // c1 -> NTT, (.) s, INTT  (dot_product_ct_sk_array at one limb)
        u64 *v = ws((size_t)count * n);
        std::vector<CtJob> cj;
        for (uint32_t i = 0; i < count; i++) cj.push_back(CtJob{ ct + ((size_t)i * 2 + 1) * n, v + (size_t)i * n });
        { PROF(P_OTHER, 0); launch_copy_jobs(upload_jobs(cj), n, (int)count, st_); }
        d_ntt(v, count, map_ct(), 1, false);
        { PROF(P_OTHER, 0); launch_dyadic_plain(dlevel(0), v, sk, v, 1, n, (int)count, 0, st_); }
        d_ntt(v, count, map_ct(), 1, true);
        // x = c0 + v; m = round(t x / q_0) mod t
        u64 *pt = ws((size_t)count * n);
        { PROF(P_OTHER, 0); launch_decrypt_round(ct, 2 * n, v, hp_.key_q[0], hp_.t, pt, n, (int)count, st_); }
        // BatchEncoder::decode: forward NTT mod t, slot gather
        d_ntt(pt, count, map_ct() + tid, 1, false);
        u64 *vals = ws((size_t)count * n);
        { PROF(P_OTHER, 0); launch_gather_slots(pt, reinterpret_cast<const uint32_t *>(d_slot_map_.p()), vals, n, (int)count, st_); }
        if (blocks_host) {                                                                      // sender_osn.cpp:684-690
            u64 *blk = ws((size_t)count * items * 2);
            { PROF(P_OTHER, 0); launch_pack_blocks(vals, n, items, felts, plain_modulus_len(hp_.t), blk, (int)count, st_); }
            D2H(blocks_host, blk, (size_t)count * items * 2);
        }
        if (values_host) D2H(values_host, vals, (size_t)count * n);
        sync();
    });
}

// ============================================================================ tier 2: BinBundle evaluation
// ---- the evaluation of one chunk of BinBundles, in pieces (eval_bundles below is the driver) ------------------------------
// What the pieces share: the call's arguments, the levels, the placement of the powers and the chunk's result / mask rows.
struct Engine::EvalCall {
    const Bundle *const *bundles; const Powers &pw; const RelinKeys *rk;
    const u64 *const *masks; bool masks_on_device; u64 *const *out_rows;
    size_t n; uint32_t l; int high, low; size_t Ll, Lh, Eh; u32 low_term_stride;
    std::vector<int> bslot;                                  // powers slot of every BinBundle of the call
    int c0 = 0;                                              // first BinBundle of the chunk being evaluated
    u64 *res = nullptr, *mask_d = nullptr;                   // the chunk's result rows / staged masks (when they are not the caller's)
    // powers are stored bundle-index major ([idx][power][2][L][n]): one index's powers are contiguous
    const u64 *low_ptr(uint32_t power, int b) const { return pw.low.u() + (((size_t)b * pw.n_low + (power - 1)) * 2) * Ll * n; }
    const u64 *hext_ptr(uint32_t i, int b) const { return pw.hext.u() + (((size_t)b * pw.n_high + (i - 1)) * 2) * Eh * n; }
    u64 *res_ptr(int i) const { return out_rows ? out_rows[c0 + i] : res + (size_t)i * 2 * n; }
    const u64 *mask_ptr(int i) const { return masks_on_device ? masks[c0 + i] : mask_d + (size_t)i * n; }
};
// which forms of the Paterson-Stockmeyer steps this call takes (decided once per chunk in eval_patstock)
struct Engine::PsPlan { bool i0_fast, need_vlast, raw_drop, raw_i0, async_high, late_high; };

// One batch: every Paterson-Stockmeyer BinBundle of the chunk, ordered by bundle index (the shared powers of
// one index then stay in the same L2).  (Cutting the batch into groups whose database scans overlap the previous
// group's VALU-bound tail on a second stream was built and measured in rounds 2 and 3, also with an LDS-DMA
// multiply-accumulate that leaves room for an NTT workgroup per CU: slower to level, the chip is power-bound.
// profiles/r03_eval_pipeline.txt; the code is in git history at 22dbbd1, engine.cpp:1600-1745.)
struct PsBatch {
    std::vector<int> ids;                               // positions in this chunk
    std::vector<int> nin, in_off;                       // inner polynomials per BinBundle, prefix offsets
    int NI = 0;
    u64 *inner = nullptr, *ssum = nullptr, *vlast = nullptr, *term = nullptr, *cf = nullptr;
    std::vector<int> imap;                              // modulus of every limb polynomial of the merged block
    const MacJob *mac_jobs = nullptr;
    int n_mac = 0;
    uint64_t units = 0; uint32_t mean_cnt = 0; bool mac_is_packed = false;
    const TermJob *term_jobs = nullptr;                 // the i = 0 block's per-term products on the dropped limb (k_term_product)
    size_t n_term = 0; bool term_packed = false;
};

// BatchedPlaintextPolyn::eval (bin_bundle.cpp:106-174) for the BinBundles pl_ids of the chunk
void Engine::eval_plain(EvalCall &c, const std::vector<int> &pl_ids)
{
    const size_t n = c.n, Ll = c.Ll;
    const int low = c.low, c0 = c.c0;
    const Bundle *const *bundles = c.bundles;
    const std::vector<int> &bslot = c.bslot;
    const u32 low_term_stride = c.low_term_stride;
    auto low_ptr = [&](uint32_t power, int b) { return c.low_ptr(power, b); };
    auto res_ptr = [&](int i) { return c.res_ptr(i); };
    auto mask_ptr = [&](int i) { return c.mask_ptr(i); };
    const int Bp = (int)pl_ids.size();
    const int lvl = low;                                   // level of powers[1]
    const size_t Lv = lvl + 1;
    u64 *acc = ws((size_t)Bp * 2 * Lv * n);
    std::vector<MacStream> ms;
    std::vector<EpiJob> ej;
    for (int x = 0; x < Bp; x++) {
        const Bundle &b = *bundles[c0 + pl_ids[x]];
        u64 *o = acc + (size_t)x * 2 * Lv * n;
        if (b.degree) ms.push_back(MacStream{ bundle_slot(b, false, 0, Lv * n), low_ptr(1, bslot[c0 + pl_ids[x]]), o, b.degree,
                                              bundle_stride(b, false, Lv * n), low_term_stride, (u32)(Ll * n), (u32)(Lv * n), 0, (u32)Lv,
                                              (u32)b.packed });   // :140-149
        else HIP_CHECK(hipMemsetAsync(o, 0, 2 * Lv * n * sizeof(u64), st_));
        ej.push_back(EpiJob{ o, nullptr, nullptr, b.a0.u(), mask_ptr(pl_ids[x]), res_ptr(pl_ids[x]) });
    }
    { auto mj = group_mac(ms); PROF(P_MAC, mac_units(mj)); launch_mac(dlevel(lvl), (int)Lv, upload_jobs(mj), n, (int)mj.size(), st_, mac_kara(lvl, mac_mean_cnt(mj)), mac_packed(mj)); }
    d_ntt_ct(acc, (size_t)Bp * 2, lvl, true);                                                 // :154
    // :159 add_plain(a_0), :162 add_plain(mask), :168-170 mod switch to the last level, :171 clear bits
    { PROFW(P_MODSWITCH, (size_t)Bp * n * (2 * Lv + 4)); launch_eval_epilogue(dlevel(0), lvl, upload_jobs(ej), Lv * n, hp_.irrelevant_bit_count, n, Bp, st_); }
}

// the sums of the pre-lifted coefficient-form plaintexts, sum_i lift(a_{i*h}) (.) C^{i*h} (bin_bundle.cpp:328-337), as MAC streams
void Engine::ps_cf_streams(const EvalCall &c, const PsBatch &g, std::vector<MacStream> &out)
{
    const size_t n = c.n, Lh = c.Lh, Eh = c.Eh;
    const int c0 = c.c0;
    const Bundle *const *bundles = c.bundles;
    const std::vector<int> &bslot = c.bslot;
    auto hext_ptr = [&](uint32_t i, int b) { return c.hext_ptr(i, b); };
    for (size_t x = 0; x < g.ids.size(); x++) {
        const Bundle &b = *bundles[c0 + g.ids[x]];
        out.push_back(MacStream{ bundle_slot(b, true, 0, Lh * n), hext_ptr(1, bslot[c0 + g.ids[x]]), g.cf + x * 2 * Lh * n, b.H,
                                 bundle_stride(b, true, Lh * n), (u32)((size_t)2 * Eh * n), (u32)(Eh * n), (u32)(Lh * n), 0, (u32)Lh,
                                 (u32)b.packed });
    }
}

// BatchedPlaintextPolyn::eval_patstock (bin_bundle.cpp:192-360) for the BinBundles ps_ids of the chunk: the plan, the batch, its
// job tables (ps_tables) and the launch sequence (ps_run)
void Engine::eval_patstock(EvalCall &c, const std::vector<int> &ps_ids)
{
    const size_t Ll = c.Ll;
    const uint32_t l = c.l;
    const int high = c.high, low = c.low, c0 = c.c0;
    const Powers &pw = c.pw;
    const std::vector<int> &bslot = c.bslot;
    // i = 0 block (:314-324): every term C^j (.) a_j is INTT'd and rounded to the high level ON ITS OWN before
    // the sum (note N1).  With one dropped limb the sum of the rounded terms is
    //   (sum_j c_j[m] + l*half - sum_j ((c_j[last] + half) mod q_last)) * q_last^-1  mod q_m,
    // where the first sum is exact and may be taken in the NTT domain.  So only the LAST limb of each term
    // needs its own inverse transform (2 per term instead of 2*L_low), bit-identical to the reference.
    const bool i0_fast = (low - high <= 1) && ((unsigned __int128)(l + 1) * hlevel(low).q[Ll - 1] < ((unsigned __int128)1 << 64));
    const bool need_vlast = i0_fast && low > high;
    // RAW inverse transforms (no twist, no final reduction) where the consumer's own constants absorb the twist:
    // the inner polynomials when the fused drop + extension kernel takes them, the i = 0 block's sums and last limbs
    const bool raw_drop = low == high + 1 && hlevel(high).L == hlevel(high).nB && hlevel(high).L <= 3;
    const bool raw_i0 = need_vlast;
    PsBatch g;
    g.ids = ps_ids;
    std::stable_sort(g.ids.begin(), g.ids.end(), [&](int a, int b) { return bslot[c0 + a] < bslot[c0 + b]; });
    // high powers still in flight on the second stream (split ComputePowers): everything that needs only the low
    // powers goes first, the cf products (which read the high powers) come later
    const bool async_high = pw.high_async && pw.high_ready;
    const bool late_high = async_high;
    const PsPlan plan{ i0_fast, need_vlast, raw_drop, raw_i0, async_high, late_high };
    ps_tables(c, plan, g);
    ps_run(c, plan, g);
}

// phase A: workspace and the job array of the multiply-accumulate
void Engine::ps_tables(EvalCall &c, const PsPlan &plan, PsBatch &g)
{
    const size_t n = c.n, Ll = c.Ll, Lh = c.Lh;
    const uint32_t l = c.l;
    const int c0 = c.c0;
    const Bundle *const *bundles = c.bundles;
    const std::vector<int> &bslot = c.bslot;
    const u32 low_term_stride = c.low_term_stride;
    auto low_ptr = [&](uint32_t power, int b) { return c.low_ptr(power, b); };
    const bool i0_fast = plan.i0_fast, need_vlast = plan.need_vlast, raw_drop = plan.raw_drop, raw_i0 = plan.raw_i0;
    const bool late_high = plan.late_high;
    const int Bs = (int)g.ids.size();
    // inner polynomials i = 1..H (block H only if r > 0)                     :248-304
    g.nin.resize(Bs); g.in_off.resize(Bs);
    for (int x = 0; x < Bs; x++) {
        const Bundle &b = *bundles[c0 + g.ids[x]];
        g.nin[x] = (int)b.H - (b.r == 0 ? 1 : 0);
        g.in_off[x] = g.NI;
        g.NI += g.nin[x];
    }
    // Every dyadic multiply-accumulate of the evaluation reads only the query powers and the database, so
    // all of a group run as ONE launch, and their results share ONE inverse-NTT launch:
    //   inner [NI][2][Ll]   sum_j C^j (.) a_{i*h+j}                                  :258-264
    //   ssum  [Bs][2][Lh]   sum_j C^j (.) a_j on the limbs that survive the switch  (i = 0 block, fast form)
    //   vlast [Bs*l][2][1]  C^j (.) a_j on the dropped limb, per term               (i = 0 block, fast form)
    //   term  [Bs*l][2][Ll] C^j (.) a_j, per term                                   (i = 0 block, general form)
    //   cf    [Bs][2][Lh]   sum_i lift(a_{i*h}) (.) C^{i*h}                          :328-337 (exact)
    const size_t w_inner = (size_t)g.NI * 2 * Ll * n, w_ssum = i0_fast ? (size_t)Bs * 2 * Lh * n : 0;
    const size_t w_vlast = need_vlast ? (size_t)Bs * l * 2 * n : 0, w_term = i0_fast ? 0 : (size_t)Bs * l * 2 * Ll * n;
    const size_t w_cf = (size_t)Bs * 2 * Lh * n;
    g.inner = ws(w_inner + w_ssum + w_vlast + w_term + (late_high ? 0 : w_cf));
    g.ssum = g.inner + w_inner; g.vlast = g.ssum + w_ssum; g.term = g.vlast + w_vlast;
    g.cf = late_high ? nullptr : g.term + w_term;
    std::vector<MacStream> ms;
    auto map_push = [&](size_t polys, int first_limb, int limbs) {
        for (size_t p = 0; p < polys; p++) for (int j = 0; j < limbs; j++) g.imap.push_back(first_limb + j);
    };
    for (int x = 0; x < Bs; x++) {
        const Bundle &b = *bundles[c0 + g.ids[x]];
        const int bs = bslot[c0 + g.ids[x]];
        for (int i = 1; i <= g.nin[x]; i++) {
            const u32 cnt = (u32)i < b.H ? l : b.r;
            ms.push_back(MacStream{ bundle_slot(b, false, (size_t)i * l, Ll * n), low_ptr(1, bs),
                                    g.inner + ((size_t)g.in_off[x] + i - 1) * 2 * Ll * n, cnt,
                                    bundle_stride(b, false, Ll * n), low_term_stride, (u32)(Ll * n), (u32)(Ll * n), 0, (u32)Ll,
                                    (u32)b.packed });
        }
    }
    map_push((size_t)g.NI * 2, 0, (int)Ll);
    if (raw_drop) for (int &v : g.imap) v |= NTT_MAP_RAW;          // consumed by the fused drop + extension only
    if (i0_fast) {
        for (int x = 0; x < Bs; x++) {
            const Bundle &b = *bundles[c0 + g.ids[x]];
            ms.push_back(MacStream{ bundle_slot(b, false, 0, Ll * n), low_ptr(1, bslot[c0 + g.ids[x]]), g.ssum + (size_t)x * 2 * Lh * n, l,
                                    bundle_stride(b, false, Ll * n), low_term_stride, (u32)(Ll * n), (u32)(Lh * n), 0, (u32)Lh,
                                    (u32)b.packed });
        }
        map_push((size_t)Bs * 2, 0, (int)Lh);
        if (raw_i0) for (size_t x = g.imap.size() - (size_t)Bs * 2 * Lh; x < g.imap.size(); x++) g.imap[x] |= NTT_MAP_RAW;
    }
    std::vector<TermJob> tj;
    if (need_vlast || !i0_fast) {
        for (int x = 0; x < Bs; x++) {
            const Bundle &b = *bundles[c0 + g.ids[x]];
            const int bs = bslot[c0 + g.ids[x]];
            if (i0_fast) {
                if (x == 0) g.term_packed = b.packed;
                else if (g.term_packed != (bool)b.packed) throw std::logic_error("BinBundles of one evaluation differ in their row format");
            }
            for (u32 j = 1; j <= l; j++) {
                if (i0_fast)                                 // the dropped limb of every term by itself: k_term_product (not k_mac chains of length one)
                    tj.push_back(TermJob{ bundle_slot(b, false, j - 1, Ll * n), low_ptr(j, bs), g.vlast + ((size_t)x * l + j - 1) * 2 * n });
                else
                    ms.push_back(MacStream{ bundle_slot(b, false, j - 1, Ll * n), low_ptr(j, bs),
                                            g.term + ((size_t)x * l + j - 1) * 2 * Ll * n, 1, bundle_stride(b, false, Ll * n), low_term_stride,
                                            (u32)(Ll * n), (u32)(Ll * n), 0, (u32)Ll, (u32)b.packed });
            }
        }
        if (i0_fast) {
            map_push((size_t)Bs * l * 2, (int)Ll - 1, 1);
            if (raw_i0) for (size_t x = g.imap.size() - (size_t)Bs * l * 2; x < g.imap.size(); x++) g.imap[x] |= NTT_MAP_RAW;
        } else map_push((size_t)Bs * l * 2, 0, (int)Ll);
    }
    if (!late_high) { ps_cf_streams(c, g, ms); map_push((size_t)Bs * 2, 0, (int)Lh); }
    auto mj = group_mac(ms);
    g.mac_jobs = upload_jobs(mj);
    g.n_mac = (int)mj.size();
    g.units = mac_units(mj);
    g.mean_cnt = mac_mean_cnt(mj);
    g.mac_is_packed = mac_packed(mj);
    if (!tj.empty()) { g.term_jobs = upload_jobs(tj); g.n_term = tj.size(); }
}

// ---- phase B: the multiply-accumulate (the level-`low` constants serve every limb: levels share their leading
// primes) and everything behind it
void Engine::ps_run(EvalCall &c, const PsPlan &plan, PsBatch &g)
{
    const size_t n = c.n, Ll = c.Ll, Lh = c.Lh, Eh = c.Eh;
    const uint32_t l = c.l;
    const int high = c.high, low = c.low, c0 = c.c0;
    const Powers &pw = c.pw;
    const Bundle *const *bundles = c.bundles;
    const std::vector<int> &bslot = c.bslot;
    auto hext_ptr = [&](uint32_t i, int b) { return c.hext_ptr(i, b); };
    auto res_ptr = [&](int i) { return c.res_ptr(i); };
    auto mask_ptr = [&](int i) { return c.mask_ptr(i); };
    const bool i0_fast = plan.i0_fast, raw_drop = plan.raw_drop, raw_i0 = plan.raw_i0;
    const bool async_high = plan.async_high, late_high = plan.late_high;
    auto cf_streams = [&](const PsBatch &gg, std::vector<MacStream> &out) { ps_cf_streams(c, gg, out); };
    const RelinKeys *rk = c.rk;
    const int Bs = (int)g.ids.size(), NI = g.NI;
    const std::vector<int> &nin = g.nin, &in_off = g.in_off;
    u64 *inner = g.inner, *ssum = g.ssum, *vlast = g.vlast, *term = g.term;
    { PROF(P_MAC, g.units); launch_mac(dlevel(low), (int)Ll, g.mac_jobs, n, g.n_mac, st_, mac_kara(low, g.mean_cnt), g.mac_is_packed); }
    // Side lane (rounds 4, 5): see below.  side_tp (round 5): the i = 0 block's per-term products and THEIR inverse transforms leave the main
    // stream too -- they feed only the i = 0 finish, which runs on the side lane anyway -- so the main chain behind k_mac starts with the
    // inner polynomials alone: -0.017 ms (-0.5 %) on the latency of the 16M-4096 query over four order-balanced A/B runs, -1.7 % on the
    // N = 8 shard, same bits (profiles/r05_ab_side_term_product.txt)
    const bool side_tp = eval_side_ && plan.late_high && !prof_on_ && i0_fast && low != high && lanes_[2].st && cur_lane_ == 0 &&
                         (size_t)Bs * l <= 4096 && g.n_term && plan.need_vlast;
    const size_t n_vlast = side_tp ? (size_t)Bs * l * 2 : 0;
    const int *imap_dev = upload_jobs(g.imap);
    if (side_tp) HIP_CHECK(hipEventRecord(ev_fork_, st_));    // behind k_mac (the powers and the database are read-only from here on)
    else if (g.n_term) { PROF(P_MAC, (uint64_t)g.n_term * (g.term_packed ? packed_row_bits(hp_.key_q[Ll - 1]) : 64)); launch_term_product(dlevel(low), g.term_jobs, g.n_term, n, (int)Ll - 1, (u32)(Ll * n), (u32)n, g.term_packed, st_); }
    d_ntt(inner, g.imap.size() - n_vlast, imap_dev, (int)g.imap.size(), true);               // :268,297,320,333
    if (side_tp) HIP_CHECK(hipEventRecord(ev_intt_, st_));

    // Side lane (round 4).  Two pieces of the evaluation hang off nothing that follows on the main stream: the sums of the
    // coefficient-form products (they read the high powers and the database, :328-337) and the i = 0 block's finish (it reads the
    // inverse transforms above).  Both are launches that cannot fill the chip (224 workgroups of 28-term chains; one pass over
    // the per-term last limbs) and used to sit in the tail of the main stream, where nothing could hide them.  They run on a
    // third stream (lane 2; the first of them waits for the high-power chain's event) next to the drop / extension / transform
    // launches, and the epilogue waits for them.  (With pipelined queries the next query's ComputePowers fills the same holes and the
    // lane is level, profiles/r04_ab_eval_side.txt; it still serves a query that runs alone.)
    const bool side = eval_side_ && late_high && !prof_on_ && i0_fast && low != high && lanes_[2].st && cur_lane_ == 0;
    // (the i = 0 finish only while it is small: 256M-4096's reads 4 GB of per-term limbs, a bandwidth-bound pass that gains nothing
    //  from running next to the transforms -- measured +0.9 % there, -1.2 % at 16M-4096, -4.2 % on its N = 8 shard; profiles/r04_ab_eval_side.txt)
    const bool side_i0 = side && (size_t)Bs * l <= 4096;
    u64 *i0_side = nullptr;
    if (side) {
        g.cf = ws((size_t)Bs * 2 * Lh * n);
        if (side_i0) i0_side = ws((size_t)Bs * 2 * Lh * n);
    }
    bool side_ran = false;
    auto run_side = [&]() {
        side_ran = true;
        std::vector<MacStream> cs;
        cf_streams(g, cs);
        std::vector<I0Job> ij;
        if (side_i0)
            for (int x = 0; x < Bs; x++)
                ij.push_back(I0Job{ ssum + (size_t)x * 2 * Lh * n, vlast + (size_t)x * l * 2 * n, i0_side + (size_t)x * 2 * Lh * n, (int)l, 1 });
        if (!side_tp) HIP_CHECK(hipEventRecord(ev_fork_, st_));
        switch_lane(2);
        struct Back { Engine *e; ~Back() { e->switch_lane(0); } } back{ this };
        HIP_CHECK(hipStreamWaitEvent(st_, ev_fork_, 0));
        if (side_tp) {
            launch_term_product(dlevel(low), g.term_jobs, g.n_term, n, (int)Ll - 1, (u32)(Ll * n), (u32)n, g.term_packed, st_);
            d_ntt(vlast, n_vlast, imap_dev + (g.imap.size() - n_vlast), (int)n_vlast, true);
        }
        if (async_high) HIP_CHECK(hipStreamWaitEvent(st_, pw.high_ready, 0));   // the cf sums read the high powers (second stream)
        { auto mj = group_mac(cs); launch_mac(dlevel(high), (int)Lh, upload_jobs(mj), n, (int)mj.size(), st_, mac_kara(high, mac_mean_cnt(mj)), mac_packed(mj)); }
        d_ntt_ct(g.cf, (size_t)Bs * 2, high, true);
        if (side_tp) HIP_CHECK(hipStreamWaitEvent(st_, ev_intt_, 0));            // the i = 0 finish also reads the sums the main stream transforms back
        if (side_i0) launch_i0_finish(dlevel(low), upload_jobs(ij), n, Bs, st_, raw_i0);
        HIP_CHECK(hipEventRecord(ev_side_, st_));
    };
    // the side lane starts behind the inverse transforms above, next to the drop / extension / transform launches (starting it behind the
    // tensor-on-load transform instead, next to the finish and the key switch of the sums, measured level: profiles/r04_ab_eval_side.txt)
    if (side) run_side();
    const bool late_cf = late_high && !side;                 // the cf sums on the main stream, behind the wait for the high powers

    // mod switch to the high level (:269,298), then ct x ct with the high powers (:272,301): extend, NTT,
    // tensor, INTT, finish (+ sum over i, :273,303).  A single drop is folded into the extension's pass.
    u64 *ext = ws((size_t)NI * 2 * Eh * n);
    bool fused_drop = false;
    if (low == high + 1) {
        PROFW(P_BEHZ_EXT, (size_t)NI * 2 * n * (Ll + Eh));
        fused_drop = launch_drop_behz_ext(dlevel(high), hlevel(high).L, hlevel(high).nB, inner, Ll * n, 1, ext, n, NI * 2, st_, raw_drop);
    }
    if (!fused_drop) {
        u64 *innerh = inner;
        for (int lv = low; lv > high; lv--) {
            u64 *nxt = ws((size_t)NI * 2 * lv * n);
            { PROFW(P_MODSWITCH, (size_t)NI * 2 * n * (2 * lv + 1)); launch_modswitch(dlevel(lv), innerh, (size_t)2 * (lv + 1) * n, 2, nxt, n, NI, st_); }
            innerh = nxt;
        }
        { PROFW(P_BEHZ_EXT, (size_t)NI * 2 * n * (Lh + Eh)); launch_behz_ext(dlevel(high), hlevel(high).L, hlevel(high).nB, innerh, Lh * n, 1, ext, n, NI * 2, st_); }
    }
    d_ntt(ext, (size_t)NI * 2 * Eh, map_ext(high), (int)Eh, false);
    if (async_high) HIP_CHECK(hipStreamWaitEvent(st_, pw.high_ready, 0));
    u64 *result = ws((size_t)Bs * 3 * Lh * n);                                                  // :238-240
    // The products of one BinBundle are summed (:273,303).  Each keeps its own rounding (note N1), but only
    // the q limbs are needed per term for that: the Bsk limbs are summed in the NTT domain by the tensor
    // kernel and finished once per BinBundle (see behz_finish_coeff).  Bit-identical, 6 instead of 15
    // inverse transforms per term at L = 2.
    int max_terms = 0;
    for (int x = 0; x < Bs; x++) max_terms = std::max(max_terms, nin[x]);
    // the summed finish adds per-term canonical residues of EVERY q limb as plain integers: the widest limb bounds it
    u64 q_widest = 0;
    for (size_t j = 0; j < Lh; j++) q_widest = std::max(q_widest, hlevel(high).q[j]);
    const bool summed = !force_per_term_ && Lh <= 4 &&
                        (unsigned __int128)max_terms * q_widest < ((unsigned __int128)1 << 63);
    const size_t w_cf = (size_t)Bs * 2 * Lh * n;
    if (summed) {
        const size_t nBskh = Eh - Lh;
        u64 *dq = ws((size_t)NI * 3 * Lh * n + (size_t)Bs * 3 * nBskh * n + (late_cf ? w_cf : 0));
        u64 *bsum = dq + (size_t)NI * 3 * Lh * n;
        if (late_cf) {                                      // the cf sums join this inverse-NTT launch
            g.cf = bsum + (size_t)Bs * 3 * nBskh * n;
            std::vector<MacStream> cs;
            cf_streams(g, cs);
            auto mj = group_mac(cs); PROF(P_MAC, mac_units(mj)); launch_mac(dlevel(high), (int)Lh, upload_jobs(mj), n, (int)mj.size(), st_, mac_kara(high, mac_mean_cnt(mj)), mac_packed(mj));
        }
        std::vector<TensorSumJob> tj;
        std::vector<FinishSumJob> fj;
        std::vector<int> dmap;
        for (int x = 0; x < Bs; x++) {
            if (!nin[x]) { HIP_CHECK(hipMemsetAsync(result + (size_t)x * 3 * Lh * n, 0, 3 * Lh * n * sizeof(u64), st_)); continue; }
            const size_t job = (size_t)in_off[x];
            tj.push_back(TensorSumJob{ ext + job * 2 * Eh * n, hext_ptr(1, bslot[c0 + g.ids[x]]), dq + job * 3 * Lh * n,
                                       bsum + (size_t)x * 3 * nBskh * n, nin[x], 0 });
            fj.push_back(FinishSumJob{ dq + job * 3 * Lh * n, bsum + (size_t)x * 3 * nBskh * n, result + (size_t)x * 3 * Lh * n, nin[x], 0 });
        }
        // (the finish applies the inverse transform's twist itself where it is the unrolled kernel: raw output)
        const int rawf = fast_finish(high) ? NTT_MAP_RAW : 0;
        for (size_t p = 0; p < (size_t)NI * 3; p++) for (size_t j = 0; j < Lh; j++) dmap.push_back((int)j | rawf);
        for (size_t p = 0; p < (size_t)Bs * 3; p++) for (size_t i = 0; i < nBskh; i++) dmap.push_back(hp_.bsk_id(hlevel(high).nB, (int)i) | rawf);
        if (late_cf) for (size_t p = 0; p < (size_t)Bs * 2; p++) for (size_t j = 0; j < Lh; j++) dmap.push_back((int)j);
        if (fuse_tensor_) {
            // per-term q limbs: product formed by the inverse transform's load; the Bsk sums (and cf) join the launch
            std::vector<TensorJob> pj;
            for (int x = 0; x < Bs; x++)
                for (int i = 0; i < nin[x]; i++) {
                    const size_t job = (size_t)in_off[x] + i;
                    pj.push_back(TensorJob{ ext + job * 2 * Eh * n, hext_ptr(1 + i, bslot[c0 + g.ids[x]]), dq + job * 3 * Lh * n });
                }
            // (only the Bsk limbs: four operand limbs per term in, three sums per BinBundle out)
            { PROFW(P_TENSOR, ((size_t)NI * 4 + (size_t)Bs * 3) * (Eh - Lh) * n); launch_tensor_sum(dlevel(high), (int)Eh, upload_jobs(tj), n, (int)tj.size(), (int)Lh, st_); }
            PROF(P_NTT_FUSED, dmap.size());
            launch_intt_tensor(hp_.logn, upload_jobs(pj), (int)pj.size(), (int)Lh, Eh * n, bsum, dmap.size() - pj.size() * 3 * Lh,
                               tabs(), upload_jobs(dmap), (int)dmap.size(), st_, ntt_latency_limbs_);
        } else {
            { PROFW(P_TENSOR, ((size_t)NI * 4 * Eh + (size_t)NI * 3 * Lh + (size_t)Bs * 3 * (Eh - Lh)) * n); launch_tensor_sum(dlevel(high), (int)Eh, upload_jobs(tj), n, (int)tj.size(), 0, st_); }
            d_ntt(dq, dmap.size(), upload_jobs(dmap), (int)dmap.size(), true);
        }
        { PROFW(P_BEHZ_FINISH, ((size_t)NI * 3 * Lh + (size_t)Bs * 3 * (Eh - Lh) + (size_t)Bs * 3 * Lh) * n); launch_behz_finish_sum(dlevel(high), hlevel(high).L, hlevel(high).nB, upload_jobs(fj), n, (int)fj.size(), st_); }
    } else {
        if (late_cf) {
            g.cf = ws(w_cf);
            std::vector<MacStream> cs;
            cf_streams(g, cs);
            { auto mj = group_mac(cs); PROF(P_MAC, mac_units(mj)); launch_mac(dlevel(high), (int)Lh, upload_jobs(mj), n, (int)mj.size(), st_, mac_kara(high, mac_mean_cnt(mj)), mac_packed(mj)); }
            d_ntt_ct(g.cf, (size_t)Bs * 2, high, true);
        }
        u64 *dbuf = ws((size_t)NI * 3 * Eh * n);
        // every (BinBundle, block) product is finished (x t, floor, Bsk -> q) by its own threads, then the
        // per-term results are summed per BinBundle (:273,303): the roundings stay per term (note N1)
        u64 *tbuf = ws((size_t)NI * 3 * Lh * n);
        std::vector<TensorJob> tj;
        std::vector<FinishJob> fj;
        std::vector<SumJob> sj;
        for (int x = 0; x < Bs; x++) {
            const int bs = bslot[c0 + g.ids[x]];
            for (int i = 1; i <= nin[x]; i++) {
                const size_t job = (size_t)in_off[x] + i - 1;
                tj.push_back(TensorJob{ ext + job * 2 * Eh * n, hext_ptr(i, bs), dbuf + job * 3 * Eh * n });
                fj.push_back(FinishJob{ dbuf + job * 3 * Eh * n, tbuf + job * 3 * Lh * n, 1, 0 });
            }
            sj.push_back(SumJob{ tbuf + (size_t)in_off[x] * 3 * Lh * n, result + (size_t)x * 3 * Lh * n, nin[x], 0 });
        }
        if (fuse_tensor_ && tj.size() == (size_t)NI) {
            PROF(P_NTT_FUSED, tj.size() * 3 * Eh);
            launch_intt_tensor(hp_.logn, upload_jobs(tj), (int)tj.size(), (int)Eh, Eh * n, nullptr, 0, tabs(), map_ext_fin(high), (int)Eh, st_, ntt_latency_limbs_);
        } else {
            if (!tj.empty()) { PROF(P_TENSOR, 0); launch_tensor(dlevel(high), upload_jobs(tj), n, (int)tj.size(), st_); }
            d_ntt(dbuf, (size_t)NI * 3 * Eh, map_ext_fin(high), (int)Eh, true);
        }
        { PROF(P_BEHZ_FINISH, 0); launch_behz_finish(dlevel(high), hlevel(high).L, hlevel(high).nB, upload_jobs(fj), false, n, (int)fj.size(), st_); }
        { PROF(P_BEHZ_FINISH, 0); launch_sum_jobs(dlevel(high), (int)Lh, upload_jobs(sj), 3, n, Bs, st_); }
    }
    // (round 6: the mod-down of this last key switch is performed by the epilogue kernel below, on the way in: one launch fewer at the
    //  end of every query, the updated (c0, c1) never go to memory; APSU_HE_FUSE_TAIL=0 keeps the launch)
    u64 *ks_acc = nullptr;
    d_relinearize(result, 3 * Lh * n, Bs, *rk, high, nullptr, 0, fuse_tail_ ? &ks_acc : nullptr);   // :308-310

    // i = 0 block, reduced to one exact [2][Lh][n] addend per BinBundle
    u64 *i0 = nullptr;
    if (side && !side_ran) run_side();                        // (paths without the fused tensor launch)
    if (side) HIP_CHECK(hipStreamWaitEvent(st_, ev_side_, 0));
    if (side_i0) {
        i0 = i0_side;
    } else if (i0_fast && low == high) {
        i0 = ssum;
    } else if (i0_fast) {
        i0 = ws((size_t)Bs * 2 * Lh * n);
        std::vector<I0Job> ij;
        for (int x = 0; x < Bs; x++)
            ij.push_back(I0Job{ ssum + (size_t)x * 2 * Lh * n, vlast + (size_t)x * l * 2 * n, i0 + (size_t)x * 2 * Lh * n, (int)l, 1 });
        { PROFW(P_MODSWITCH, (size_t)Bs * n * (4 * Lh + 2 * l)); launch_i0_finish(dlevel(low), upload_jobs(ij), n, Bs, st_, raw_i0); }
    } else {
        u64 *termh = term;
        for (int lv = low; lv > high; lv--) {
            u64 *nxt = ws((size_t)Bs * l * 2 * lv * n);
            { PROFW(P_MODSWITCH, (size_t)Bs * l * 2 * n * (2 * lv + 1)); launch_modswitch(dlevel(lv), termh, (size_t)2 * (lv + 1) * n, 2, nxt, n, Bs * (int)l, st_); }
            termh = nxt;
        }
        i0 = ws((size_t)Bs * 2 * Lh * n);
        HIP_CHECK(hipMemsetAsync(i0, 0, (size_t)Bs * 2 * Lh * n * sizeof(u64), st_));
        { PROF(P_OTHER, 0); launch_add_many(dlevel(high), i0, 2 * Lh * n, termh, (int)l, 2, n, Bs, st_); }
    }

    // :340-343 the two exact addends, :345 add_plain(a_0), :346 add_plain(mask), :354-356 mod switch to the last
    // level, :357 clear bits — one pass over the result
    std::vector<EpiJob> ej;
    for (int x = 0; x < Bs; x++)
        ej.push_back(EpiJob{ result + (size_t)x * 3 * Lh * n, i0 + (size_t)x * 2 * Lh * n, g.cf + (size_t)x * 2 * Lh * n,
                             bundles[c0 + g.ids[x]]->a0.u(), mask_ptr(g.ids[x]), res_ptr(g.ids[x]),
                             ks_acc ? ks_acc + (size_t)x * 2 * (Lh + 1) * n : nullptr });
    { PROFW(P_MODSWITCH, (size_t)Bs * n * (7 * Lh + 4 + (ks_acc ? 2 * (Lh + 1) : 0)));
      launch_eval_epilogue(dlevel(0), high, upload_jobs(ej), Lh * n, hp_.irrelevant_bit_count, n, Bs, st_, ks_acc ? dkey() : nullptr); }

}

void Engine::eval_bundles(const Bundle *const *bundles, int count, const Powers &pw, const RelinKeys *rk,
                          const u64 *const *masks, bool masks_on_device, u64 *out, bool out_on_device, u64 *const *out_rows)
{
    if (out_rows && !out_on_device) throw std::invalid_argument("out_rows are device-accessible destinations");
    Enter g(this);
    if (!has_psu_) throw std::logic_error("context was created without PSUParams");
    if (count <= 0) return;
    const size_t n = hp_.n;
    job_seq_base_ = 256;                                     // job-cache slots 256..: eval_bundles
    if (pw.low_async && pw.high_ready) HIP_CHECK(hipStreamWaitEvent(st_, pw.high_ready, 0));   // (pipelined walk: every power comes from the second stream)
    // the powers' last reader (see Powers::last_use): marked on the main stream when this call leaves, also by an exception --
    // kernels that read the powers may have been queued by then
    struct LastUse {
        Engine *e; const Powers &p;
        ~LastUse()
        {
            if (e->cur_lane_ != 0) e->switch_lane(0);
            if (!p.last_use && hipEventCreateWithFlags(&p.last_use, hipEventDisableTiming) != hipSuccess) { p.last_use = nullptr; p.last_use_set = false; return; }
            p.last_use_set = hipEventRecord(p.last_use, e->st_) == hipSuccess;
        }
    } last_use_mark{ this, pw };
    PhaseSpan ev_span;
    if (phase_on_) { ev_span.phase = PH_PROCESS_BIN_BUNDLE_CACHE; ev_span.a = phase_event(st_); }
    const uint32_t ps = psu_.query_params.ps_low_degree, l = ps;
    const int high = pw.high_level, low = pw.low_level;
    const size_t Ll = low + 1, Lh = high + 1;
    const size_t Eh = hlevel(high).L + hlevel(high).nB + 1;

    // validation mirrors bin_bundle.cpp:116-118,204-213 ("not enough ciphertext powers available")
    std::vector<int> bslot(count);
    bool any_ps = false;
    for (int i = 0; i < count; i++) {
        const Bundle &b = *bundles[i];
        bslot[i] = pw.slot_of(b.bundle_idx);
        if (bslot[i] < 0) throw std::invalid_argument("no ciphertext powers for this bundle index");
        if (b.use_ps) {
            any_ps = true;
            if (b.H > pw.n_high || l > pw.n_low) throw std::invalid_argument("not enough ciphertext powers available");
            if (b.pt_level != low) throw std::logic_error("plaintext level does not match the low powers");
        } else {
            if (b.degree > pw.n_low) throw std::invalid_argument("not enough ciphertext powers available");
            if (b.degree && b.pt_level != low) throw std::logic_error("plaintext level does not match the powers");
        }
    }
    // Without key switching the reference leaves eval_patstock's result unrelinearised (bin_bundle.cpp:238-240,308-310) and
    // the powers themselves may be longer than two polynomials: a path of its own
    if (pw.polys != (nks_ ? nks_S_ : 2u)) throw std::invalid_argument("ciphertext powers do not belong to this context");
    if (nks_ || result_polys_ > 2) {                        // every BinBundle of such a set: the rows are result_polys_ polynomials apart
        eval_bundles_nks(bundles, count, pw, masks, masks_on_device, out, out_on_device, out_rows);
        if (phase_on_ && ev_span.a) {
            ev_span.b = phase_event(st_);
            phase_spans_.push_back(ev_span);
            if (query_start_) { if (query_end_) phase_pool_.push_back(query_end_); query_end_ = phase_event(st_); }
        }
        return;
    }
    if (any_ps && !rk) throw std::invalid_argument("relinearization keys are required");

    EvalCall c{ bundles, pw, rk, masks, masks_on_device, out_rows, n, l, high, low, Ll, Lh, Eh, (u32)((size_t)2 * Ll * n), std::move(bslot) };

    // workspace budget -> chunk size
    size_t per_bundle_words = 0;
    for (int i = 0; i < count; i++) {
        const Bundle &b = *bundles[i];
        size_t w = 16 * n;
        if (b.use_ps) w += ((size_t)b.H * (2 * Ll + 2 * Lh + 2 * Eh + 3 * Eh) + (size_t)l * (2 * Ll + 2 * Lh) + 3 * Lh * (Lh + 4) + 8 * Lh) * n;
        else w += (size_t)(6 * Ll + 8) * n;
        per_bundle_words = std::max(per_bundle_words, w);
    }
    const size_t budget = eval_ws_budget_;
    int chunk = (int)std::max<size_t>(1, budget / (per_bundle_words * sizeof(u64)));
    chunk = std::min(chunk, count);

    bool synced = false;
    for (int c0 = 0; c0 < count; c0 += chunk) {
        const int B = std::min(chunk, count - c0);
        WITH_ARENA({
            // final [B][2][1][n]: written in place when the caller's buffer is on the device
            u64 *res = out_rows ? nullptr : (out_on_device ? out + (size_t)c0 * 2 * n : ws((size_t)B * 2 * n));
            u64 *mask_d = nullptr;
            if (!masks_on_device) {
                mask_d = ws((size_t)B * n);
                for (int i = 0; i < B; i++) H2D(mask_d + (size_t)i * n, masks[c0 + i], n);
            }
            c.c0 = c0; c.res = res; c.mask_d = mask_d;

            // split the chunk into Paterson-Stockmeyer and plain evaluations
            std::vector<int> ps_ids, pl_ids;
            for (int i = 0; i < B; i++) (bundles[c0 + i]->use_ps ? ps_ids : pl_ids).push_back(i);

            // ---------------------------------------------------------------- plain: bin_bundle.cpp:106-174
            if (!pl_ids.empty()) eval_plain(c, pl_ids);
            // ---------------------------------------------------------------- Paterson-Stockmeyer: bin_bundle.cpp:192-360
            if (!ps_ids.empty()) eval_patstock(c, ps_ids);
            if (!out_on_device) D2H(out + (size_t)c0 * 2 * n, res, (size_t)B * 2 * n);
            // device-resident masks and results: nothing of the caller's is read or written by the host, so the call may
            // return with the work queued (stream order protects the workspace, the job tables and the pooled powers)
            if (phase_on_ && c0 + B >= count) {                  // the last chunk closes the span (before any host wait)
                ev_span.b = phase_event(st_);
                phase_spans_.push_back(ev_span);
                if (query_start_) { if (query_end_) phase_pool_.push_back(query_end_); query_end_ = phase_event(st_); }
            }
            if (!(async_results_ && out_on_device && masks_on_device && !prof_on_)) { sync(); inflight_count_ = 0; synced = true; }
            else mark_inflight();
        });
    }
    if (synced) check_sources(pw);
}

// ============================================================================ no key switching: ciphertexts of any size
// One coefficient prime = SEALContext::using_keyswitching() false: Receiver::ComputePowers skips relinearize_inplace
// (receiver_osn.cpp:416,430-432) and eval_patstock does too (bin_bundle.cpp:308-310).  Every product then has
// size(a) + size(b) - 1 polynomials (Evaluator::bfv_multiply; Ciphertext::resize throws above SEAL_CIPHERTEXT_SIZE_MAX = 16),
// add_inplace takes the longer operand's size, multiply_plain / the transforms act on every polynomial.  There is one level
// (chain index 0, one limb), so no modulus switching.  Stored powers are zero-padded to a common even polynomial count
// (Powers::polys): the multiply-accumulate kernel handles a pair of polynomials per job, and a zero polynomial contributes
// zero to every sum, so only the SIZES are bookkeeping (power_size / result_size) -- the numbers are exact.
// Plain compositions of the per-polynomial kernels; nothing here is tuned (no shipped parameter set has products with one prime).

uint32_t Engine::power_size(uint32_t power) const
{
    if (!has_psu_) throw std::invalid_argument("context has no PSUParams");
    if (power >= nks_size_.size() || !nks_size_[power]) throw std::invalid_argument("power not available");
    return nks_size_[power];
}

// size of BatchedPlaintextPolyn::eval / eval_patstock's result for a BinBundle of this degree (bin_bundle.cpp:132-134 starts at
// size 2, :238-240 at size 3); values above CT_SIZE_MAX mean SEAL throws inside the evaluation
uint32_t Engine::result_size_for(uint32_t degree) const
{
    if (hp_.using_keyswitching) return 2;
    const uint32_t ps = psu_.query_params.ps_low_degree, h = ps + 1;
    auto sz = [&](uint32_t p) { return p < nks_size_.size() ? nks_size_[p] : 0u; };
    if (!(ps > 1 && ps < degree)) {                          // receiver_osn.cpp:520-522 -> eval
        uint32_t r = 2;
        for (uint32_t d = 1; d <= degree; d++) r = std::max(r, sz(d));
        return r;
    }
    const uint32_t H = degree / h, rem = degree % h;
    uint32_t r = 3, low_all = 2;
    for (uint32_t j = 1; j <= ps; j++) low_all = std::max(low_all, sz(j));
    for (uint32_t i = 1; i <= H; i++) {
        const uint32_t cnt = i < H ? ps : rem;
        if (!cnt) break;
        uint32_t s_in = 2;
        for (uint32_t j = 1; j <= cnt; j++) s_in = std::max(s_in, sz(j));
        r = std::max(r, s_in + sz(i * h) - 1);               // :272,301
    }
    r = std::max(r, low_all);                                // :314-324
    for (uint32_t i = 1; i <= H; i++) r = std::max(r, sz(i * h));    // :328-337
    return r;
}

uint32_t Engine::result_size(const Bundle &b) const
{
    if (!has_psu_) throw std::invalid_argument("context has no PSUParams");
    return result_size_for(b.degree);
}

void Engine::d_multiply_sized(const u64 *ea, int sa, const u64 *eb, int sb, u64 *out, int chain_idx)
{
    const size_t n = hp_.n, L = chain_idx + 1;
    const size_t E = hlevel(chain_idx).L + hlevel(chain_idx).nB + 1;
    const int so = sa + sb - 1, so3 = (so + 2) / 3 * 3;        // the finish kernels take polynomials three at a time
    u64 *d = ws((size_t)so3 * E * n), *o = ws((size_t)so3 * L * n);
    if (so3 > so) HIP_CHECK(hipMemsetAsync(d + (size_t)so * E * n, 0, (size_t)(so3 - so) * E * n * sizeof(u64), st_));
    std::vector<TensorConvJob> tj{ TensorConvJob{ ea, eb, d, sa, sb } };
    { PROF(P_TENSOR, 0); launch_tensor_conv(dlevel(chain_idx), upload_jobs(tj), n, 1, st_); }
    d_ntt(d, (size_t)so3 * E, map_ext_fin(chain_idx), (int)E, true);
    std::vector<FinishJob> fj;
    for (int t = 0; t < so3 / 3; t++) fj.push_back(FinishJob{ d + (size_t)3 * t * E * n, o + (size_t)3 * t * L * n, 1, 0 });
    { PROF(P_BEHZ_FINISH, 0); launch_behz_finish(dlevel(chain_idx), hlevel(chain_idx).L, hlevel(chain_idx).nB, upload_jobs(fj), false, n, (int)fj.size(), st_); }
    D2D(out, o, (size_t)so * L * n);
}

void Engine::multiply_sized(const u64 *a, int sa, const u64 *b, int sb, u64 *out, int chain_idx)
{
    Enter g(this);
    TIER1_SLOTS();
    check_level(chain_idx);
    if (sa < 2 || sb < 2 || sa + sb - 1 > (int)CT_SIZE_MAX) throw std::invalid_argument("invalid size");      // Ciphertext::resize
    const size_t n = hp_.n, L = chain_idx + 1;
    const size_t E = hlevel(chain_idx).L + hlevel(chain_idx).nB + 1;
    const bool square = (a == b && sa == sb);
    WITH_ARENA({
        u64 *in = ws((size_t)(sa + (square ? 0 : sb)) * L * n);
        H2D(in, a, (size_t)sa * L * n);
        if (!square) H2D(in + (size_t)sa * L * n, b, (size_t)sb * L * n);
        const int np = sa + (square ? 0 : sb);
        u64 *ext = ws((size_t)np * E * n);
        { PROF(P_BEHZ_EXT, 0); launch_behz_ext(dlevel(chain_idx), hlevel(chain_idx).L, hlevel(chain_idx).nB, in, L * n, 1, ext, n, np, st_); }
        d_ntt(ext, (size_t)np * E, map_ext(chain_idx), (int)E, false);
        u64 *o = ws((size_t)(sa + sb - 1) * L * n);
        d_multiply_sized(ext, sa, square ? ext : ext + (size_t)sa * E * n, sb, o, chain_idx);
        D2H(out, o, (size_t)(sa + sb - 1) * L * n);
        sync();
    });
}

std::unique_ptr<Powers> Engine::compute_powers_nks(const uint32_t *bundle_indices, int nb, const u64 *const *src, bool on_device)
{
    if (nks_oversize_)          // Evaluator::multiply -> Ciphertext::resize: std::invalid_argument("invalid size or poly_modulus_degree")
        throw std::invalid_argument("invalid size: a product of unrelinearized ciphertext powers exceeds SEAL's largest ciphertext");
    const Sched &s = sched_;
    const size_t n = hp_.n, S = nks_S_, P = s.slot_power.size();
    const size_t E = hlevel(0).L + hlevel(0).nB + 1;
    job_seq_base_ = 0;
    auto pw = std::make_unique<Powers>();
    if (++query_seq_ == 0) query_seq_ = 1;
    pw->seq = query_seq_;
    pw->nb = nb;
    pw->bundle_indices.assign(bundle_indices, bundle_indices + nb);
    pw->low_level = pw->high_level = 0;
    pw->n_low = (uint32_t)s.low_powers.size();
    pw->n_high = (uint32_t)s.high_powers.size();
    pw->polys = (uint32_t)S;
    counters_[C_POWERS_ALLOC]++;
    pw->low.alloc((size_t)pw->n_low * nb * S * n * sizeof(u64));
    if (pw->n_high) {
        pw->high.alloc((size_t)pw->n_high * nb * S * n * sizeof(u64));
        pw->hext.alloc((size_t)pw->n_high * nb * S * E * n * sizeof(u64));
    }
    PhaseSpan cp_span;
    if (phase_on_) {
        phase_close_query();
        cp_span.phase = PH_COMPUTE_POWERS;
        cp_span.a = phase_event(st_);
        query_start_ = phase_event(st_);
    }
    WITH_ARENA({
        if (cp_span.b) { phase_pool_.push_back(cp_span.b); cp_span.b = nullptr; }                  // a retry after arena growth
        // slot-major like the key-switching walk: [slot][bundle index][S polys]; coefficient form and extended + NTT form
        u64 *coef = ws(P * nb * S * n), *ext = ws(P * nb * S * E * n);
        HIP_CHECK(hipMemsetAsync(coef, 0, P * nb * S * n * sizeof(u64), st_));
        auto coef_ptr = [&](int slot, int b) { return coef + ((size_t)slot * nb + b) * S * n; };
        auto ext_ptr = [&](int slot, int b) { return ext + ((size_t)slot * nb + b) * S * E * n; };
        int si = 0;
        std::vector<CtJob> chk;
        for (auto &kv : dag_.nodes()) {                                                            // receiver_osn.cpp:304-317
            if (!kv.second.is_source()) continue;
            for (int b = 0; b < nb; b++) {
                const u64 *sp = src[(size_t)b * dag_.source_count() + si];
                if (on_device) D2D(coef_ptr(s.slot_of[kv.first], b), sp, 2 * n);
                else H2D(coef_ptr(s.slot_of[kv.first], b), sp, 2 * n);
                chk.push_back(CtJob{ coef_ptr(s.slot_of[kv.first], b), coef_ptr(s.slot_of[kv.first], b) });
            }
            si++;
        }
        // the sources against their (single) prime, in place: the same report as on the key-switching path (round 6)
        launch_copy_sources(upload_jobs(chk), 2 * n, (int)chk.size(), dlevel(0), 1, n, bad_source_ + query_seq_ % BAD_SLOTS, query_seq_, st_);
        // BEHZ extension + NTT of every polynomial of the slots [s0, s1) (zero padding extends to zero)
        auto extend = [&](int s0, int s1) {
            if (s1 <= s0) return;
            { PROF(P_BEHZ_EXT, 0); launch_behz_ext(dlevel(0), hlevel(0).L, hlevel(0).nB, coef_ptr(s0, 0), S * n, (int)S, ext_ptr(s0, 0), n, (s1 - s0) * nb, st_); }
            d_ntt(ext_ptr(s0, 0), (size_t)(s1 - s0) * nb * S * E, map_ext(0), (int)E, false);
        };
        extend(s.levels[0].s0, s.levels[0].s1);
        for (size_t d = 1; d < s.levels.size(); d++) {
            const size_t mark = arena_off_;
            for (const auto &nd : s.nodes) {                                                       // :418-433, one product per node and index
                if (nd[0] < s.levels[d].s0 || nd[0] >= s.levels[d].s1) continue;
                const int sa = (int)nks_size_[s.slot_power[nd[1]]], sb = (int)nks_size_[s.slot_power[nd[2]]];
                for (int b = 0; b < nb; b++) {
                    d_multiply_sized(ext_ptr(nd[1], b), sa, ext_ptr(nd[2], b), sb, coef_ptr(nd[0], b), 0);
                    arena_off_ = mark;                                                             // stream order makes the reuse safe
                }
            }
            extend(s.levels[d].s0, s.levels[d].s1);
        }
        // :458-487: no level below the first one; low powers (all of them without Paterson-Stockmeyer) go to NTT form
        for (size_t i = 0; i < s.low_powers.size(); i++)
            for (int b = 0; b < nb; b++)
                D2D(pw->low.u() + ((size_t)b * pw->n_low + i) * S * n, coef_ptr(s.slot_of[s.low_powers[i]], b), S * n);
        d_ntt_ct(pw->low.u(), (size_t)pw->n_low * nb * S, 0, false);
        for (size_t i = 0; i < s.high_powers.size(); i++)
            for (int b = 0; b < nb; b++) {
                D2D(pw->high.u() + ((size_t)b * pw->n_high + i) * S * n, coef_ptr(s.slot_of[s.high_powers[i]], b), S * n);
                D2D(pw->hext.u() + ((size_t)b * pw->n_high + i) * S * E * n, ext_ptr(s.slot_of[s.high_powers[i]], b), S * E * n);
            }
        if (phase_on_ && cp_span.a) cp_span.b = phase_event(st_);
        sync();                                              // the workspace is reused by the next call; nothing here is latency-critical
    });
    if (phase_on_ && cp_span.a && cp_span.b) phase_spans_.push_back(cp_span);
    pw->last_use_set = false;                               // (the path synchronises; kept for the pool's bookkeeping)
    check_sources(*pw);                                      // this path has just waited: the report comes from the call that took the sources
    return pw;
}

void Engine::eval_bundles_nks(const Bundle *const *bundles, int count, const Powers &pw, const u64 *const *masks, bool masks_on_device,
                              u64 *out, bool out_on_device, u64 *const *out_rows)
{
    const size_t n = hp_.n, S = pw.polys, R = result_polys_;
    const size_t E = hlevel(0).L + hlevel(0).nB + 1;
    const uint32_t l = psu_.query_params.ps_low_degree, h = l + 1;
    auto even = [](uint32_t v) { return (v + 1) & ~1u; };
    auto sz = [&](uint32_t p) { return nks_size_[p]; };
    std::vector<uint32_t> rs(count);
    for (int i = 0; i < count; i++) {
        rs[i] = result_size_for(bundles[i]->degree);
        if (rs[i] > CT_SIZE_MAX)                             // Evaluator::multiply inside eval_patstock would throw
            throw std::invalid_argument("invalid size: a Paterson-Stockmeyer product exceeds SEAL's largest ciphertext");
    }
    WITH_ARENA({
        u64 *res = out_rows ? nullptr : (out_on_device ? out : ws((size_t)count * R * n));
        auto res_ptr = [&](int i) { return out_rows ? out_rows[i] : res + (size_t)i * R * n; };
        u64 *mask_d = nullptr;
        if (!masks_on_device) {
            mask_d = ws((size_t)count * n);
            for (int i = 0; i < count; i++) H2D(mask_d + (size_t)i * n, masks[i], n);
        }
        auto low_ptr = [&](int b) { return pw.low.u() + (size_t)b * pw.n_low * S * n; };            // power 1 of bundle-index slot b
        auto hext_ptr = [&](uint32_t i, int b) { return pw.hext.u() + ((size_t)b * pw.n_high + (i - 1)) * S * E * n; };
        const size_t base_mark = arena_off_;
        for (int x = 0; x < count; x++) {
            arena_off_ = base_mark;                                                                 // stream order makes the reuse safe
            const Bundle &b = *bundles[x];
            const int bs = pw.slot_of(b.bundle_idx);
            const uint32_t RS = even(rs[x]);
            u64 *result = ws((size_t)RS * n);                                                       // coefficient form
            std::vector<MacStream> ms;
            // every stream: pairs of polynomials (2k, 2k+1) of the powers, terms 1..cnt of a run of NTT-form plaintexts
            auto low_streams = [&](const u64 *pt, u32 cnt, uint32_t polys, u64 *acc) {
                for (uint32_t pp = 0; pp < even(polys) / 2; pp++)
                    ms.push_back(MacStream{ pt, low_ptr(bs) + (size_t)pp * 2 * n, acc + (size_t)pp * 2 * n, cnt, (u32)n, (u32)(S * n), (u32)n, (u32)n, 0, 1 });
            };
            auto size_upto = [&](uint32_t cnt) { uint32_t v = 2; for (uint32_t j = 1; j <= cnt; j++) v = std::max(v, sz(j)); return v; };
            if (!b.use_ps) {                                                                        // bin_bundle.cpp:106-174
                if (b.degree) {
                    low_streams(b.ntt.u(), b.degree, rs[x], result);                                // :140-149
                    { auto mj = group_mac(ms); PROF(P_MAC, mac_units(mj)); launch_mac(dlevel(0), 1, upload_jobs(mj), n, (int)mj.size(), st_, mac_kara(0, mac_mean_cnt(mj))); }
                    d_ntt_ct(result, RS, 0, true);                                                  // :154
                } else {
                    HIP_CHECK(hipMemsetAsync(result, 0, (size_t)RS * n * sizeof(u64), st_));
                }
            } else {                                                                                // bin_bundle.cpp:192-360
                const uint32_t H = b.H, nin = H - (b.r == 0 ? 1 : 0);
                // one multiply-accumulate launch and one inverse transform for: the inner polynomials (:258-264,287-293), the
                // i = 0 block sum_j C^j a_j (:314-324; the sum of the terms' inverse transforms is the transform of the sum,
                // and there is no modulus switch to round per term), and sum_i C^{ih} a_{ih} (:328-337, exact)
                std::vector<uint32_t> s_in(nin + 1, 2);
                std::vector<u64 *> inner(nin + 1, nullptr);
                size_t words = 0;
                for (uint32_t i = 1; i <= nin; i++) { s_in[i] = size_upto(i < H ? l : b.r); words += (size_t)even(s_in[i]) * n; }
                const uint32_t s_low = size_upto(l);
                uint32_t s_high = 2;
                for (uint32_t i = 1; i <= H; i++) s_high = std::max(s_high, sz(i * h));
                u64 *blk = ws(words + (size_t)(even(s_low) + even(s_high)) * n);
                u64 *cur = blk;
                for (uint32_t i = 1; i <= nin; i++) {
                    inner[i] = cur;
                    low_streams(b.ntt.u() + (size_t)i * l * n, i < H ? l : b.r, s_in[i], cur);
                    cur += (size_t)even(s_in[i]) * n;
                }
                u64 *lowsum = cur, *cf = cur + (size_t)even(s_low) * n;
                low_streams(b.ntt.u(), l, s_low, lowsum);
                for (uint32_t pp = 0; pp < even(s_high) / 2; pp++)
                    ms.push_back(MacStream{ b.lifted.u(), hext_ptr(1, bs) + (size_t)pp * 2 * E * n, cf + (size_t)pp * 2 * n, H, (u32)n,
                                            (u32)(S * E * n), (u32)(E * n), (u32)n, 0, 1 });
                { auto mj = group_mac(ms); PROF(P_MAC, mac_units(mj)); launch_mac(dlevel(0), 1, upload_jobs(mj), n, (int)mj.size(), st_, mac_kara(0, mac_mean_cnt(mj))); }
                d_ntt_ct(blk, (words / n) + even(s_low) + even(s_high), 0, true);                   // :268,297,321 and the product's transform
                HIP_CHECK(hipMemsetAsync(result, 0, (size_t)RS * n * sizeof(u64), st_));            // :238-240
                for (uint32_t i = 1; i <= nin; i++) {                                               // :272-273,301-303
                    const size_t mark = arena_off_;
                    const int sa = (int)s_in[i], sb = (int)sz(i * h), so = sa + sb - 1;
                    u64 *iext = ws((size_t)sa * E * n), *prod = ws((size_t)so * n);
                    { PROF(P_BEHZ_EXT, 0); launch_behz_ext(dlevel(0), hlevel(0).L, hlevel(0).nB, inner[i], n, 1, iext, n, sa, st_); }
                    d_ntt(iext, (size_t)sa * E, map_ext(0), (int)E, false);
                    d_multiply_sized(iext, sa, hext_ptr(i, bs), sb, prod, 0);
                    { PROF(P_OTHER, 0); launch_add(dlevel(0), result, prod, so, n, 1, st_); }
                    arena_off_ = mark;
                }
                { PROF(P_OTHER, 0); launch_add(dlevel(0), result, lowsum, (int)s_low, n, 1, st_); }
                { PROF(P_OTHER, 0); launch_add(dlevel(0), result, cf, (int)s_high, n, 1, st_); }
            }
            // :159-171 / :345-357: add_plain(a_0), add_plain(mask) on the first polynomial; one level, so no modulus switch;
            // clear the irrelevant bits of every polynomial.  Row layout: rs[x] polynomials, zeros up to R.
            u64 *row = res_ptr(x);
            const u64 *mk = masks_on_device ? masks[x] : mask_d + (size_t)x * n;
            std::vector<EpiJob> ej{ EpiJob{ result, nullptr, nullptr, b.a0.u(), mk, row } };
            { PROF(P_MODSWITCH, 0); launch_eval_epilogue(dlevel(0), 0, upload_jobs(ej), n, hp_.irrelevant_bit_count, n, 1, st_); }
            if (rs[x] > 2) {
                D2D(row + 2 * n, result + 2 * n, (size_t)(rs[x] - 2) * n);
                { PROF(P_OTHER, 0); launch_clear_bits(row + 2 * n, (size_t)(rs[x] - 2) * n, hp_.irrelevant_bit_count, st_); }
            }
            if (R > rs[x]) HIP_CHECK(hipMemsetAsync(row + (size_t)rs[x] * n, 0, (R - rs[x]) * n * sizeof(u64), st_));
        }
        if (!out_rows && !out_on_device) D2H(out, res, (size_t)count * R * n);
        sync();
        inflight_count_ = 0;
    });
}

} // namespace apsu_he
