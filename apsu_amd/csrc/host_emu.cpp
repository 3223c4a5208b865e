// CPU emulation of the product's host logic and of one NTT workgroup, for the no-GPU test tier.
// The NTT emulation executes the SAME pass functions (ntt_core.h) the gfx950 kernel runs, with
// the workgroup's threads stepped sequentially between barriers, so the index algebra, twiddle
// addressing, LDS padding and lazy ranges are checked on the CPU.  This is NOT a fallback path:
// it is not reachable from the C ABI and is only loaded by tests.
#define APSU_HOST_EMU 1
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "ntt_core.h"
#include "blake2x.h"
#include "params.h"
#include "powers_dag.h"
#include "sched_policy.h"

using namespace apsu_he;

// inverse transform whose first pass forms the dyadic tensor product while it loads (k_intt_tensor)
template <int LOGN, int MODE, int PASS, int C = 16> static void emu_pass_tensor(u64 *lds, u64 *glob, int T, const NttTable &tab, const SrcTensor &ops)
{
    if constexpr (PASS < plan_passes(LOGN, C)) {
        if constexpr (PASS == 0)           // as in ntt_body: products staged into the LDS image with coalesced loads
            for (int tid = 0; tid < T; tid++)
                for (int e = 2 * tid; e < (1 << LOGN); e += 2 * T) *reinterpret_cast<u64x2 *>(lds + lds_slot(e)) = src_load2(ops, glob, e, tab);
        for (int tid = 0; tid < T; tid++) {
            if constexpr (PASS == 0) ntt_pass<LOGN, true, MODE, 0, false, false, SrcTensor, true, TwInline, NoHook, C>(lds, glob, tid, T, tab, ops);   // (the source's input bound rides along, as in ntt_body)
            else ntt_pass<LOGN, true, MODE, PASS, 0, false, SrcPlain, false, TwInline, NoHook, C>(lds, glob, tid, T, tab);
        }
        emu_pass_tensor<LOGN, MODE, PASS + 1, C>(lds, glob, T, tab, ops);
    }
}

template <int LOGN, int C = 16> static void emu_intt_tensor(u64 *out, const NttTable &tab, int T, const SrcTensor &ops)
{
    std::vector<u64> lds(lds_slots(1 << LOGN));
    if (tab.narrow) emu_pass_tensor<LOGN, NTT_NARROW, 0, C>(lds.data(), out, T, tab, ops);
    else if (tab.wide_d4) emu_pass_tensor<LOGN, NTT_WIDE_NEAR, 0, C>(lds.data(), out, T, tab, ops);
    else emu_pass_tensor<LOGN, NTT_WIDE, 0, C>(lds.data(), out, T, tab, ops);
}

template <int LOGN, bool INV, int MODE, int PASS, int C = 16> static void emu_pass(u64 *lds, u64 *glob, int T, const NttTable &tab)
{
    if constexpr (PASS < plan_passes(LOGN, C)) {
        // in-place global reads/writes of a pass touch disjoint 16-coefficient sets per work item, so
        // stepping the threads sequentially is equivalent to the barrier-separated parallel execution
        if constexpr (INV && PASS == 0) {
            // as in k_ntt since round 5 (SrcStaged): the inverse stages its limb into the LDS image with coalesced loads and runs its first pass from there
            for (int tid = 0; tid < T; tid++)
                for (int e = 2 * tid; e < (1 << LOGN); e += 2 * T) *reinterpret_cast<u64x2 *>(lds + lds_slot(e)) = src_load2(SrcStaged(), glob, e, tab);
            for (int tid = 0; tid < T; tid++) ntt_pass<LOGN, true, MODE, 0, false, false, SrcPlain, true, TwInline, NoHook, C>(lds, glob, tid, T, tab);
        } else
            for (int tid = 0; tid < T; tid++) ntt_pass<LOGN, INV, MODE, PASS, 0, false, SrcPlain, false, TwInline, NoHook, C>(lds, glob, tid, T, tab);
        emu_pass<LOGN, INV, MODE, PASS + 1, C>(lds, glob, T, tab);
    }
}

template <int LOGN, bool INV, int MODE, int C = 16> static void emu_ntt_n(u64 *data, const NttTable &tab, int T)
{
    constexpr int N = 1 << LOGN;
    std::vector<u64> lds(lds_slots(N));
    emu_pass<LOGN, INV, MODE, 0, C>(lds.data(), data, T, tab);
    if (!INV) for (int e = 0; e < N; e++) data[e] = ntt_fwd_finish<MODE>(lds[lds_slot(e)], tab);
}

// C: coefficients per work item (16 = the throughput form, 8 = the latency form of round 6; ntt_core.h plan_k)
template <int LOGN, bool INV, int C = 16> static void emu_ntt(u64 *data, const NttTable &tab, int T)
{
    if (tab.narrow) emu_ntt_n<LOGN, INV, NTT_NARROW, C>(data, tab, T);
    else if (tab.wide_d4) emu_ntt_n<LOGN, INV, NTT_WIDE_NEAR, C>(data, tab, T);
    else emu_ntt_n<LOGN, INV, NTT_WIDE, C>(data, tab, T);
}

static thread_local std::string g_err;

extern "C" {

const char *emu_last_error() { return g_err.c_str(); }

// NTT of one limb with the product's tables for modulus q (n = 2^logn); coeffs = coefficients per work item (16, or 8 where the ring
// size has the latency form: logn 12 and 13)
int emu_ntt_limb_c(int logn, int inverse, uint64_t q, uint64_t *data, int threads, int coeffs);
int emu_ntt_limb(int logn, int inverse, uint64_t q, uint64_t *data, int threads) { return emu_ntt_limb_c(logn, inverse, q, data, threads, 16); }
int emu_ntt_limb_c(int logn, int inverse, uint64_t q, uint64_t *data, int threads, int coeffs)
{
    try {
        if (coeffs != 16 && !(coeffs == 8 && plan_has_latency_form(logn))) throw std::invalid_argument("no pass schedule for this ring size and coefficients per work item");
        size_t n = (size_t)1 << logn;
        HeParams hp;   // only need tables: build for this single modulus via Create with K=1
        // plain modulus irrelevant for the tables; pick any value < q
        hp = HeParams::Create(n, { q }, 65537 < q ? 65537 : 3);
        const NttTablesHost &t = hp.ntt[0];
        std::vector<TwPair> fwd(n), dit(n), sc(n);
        for (size_t k = 0; k < n; k++) {
            fwd[k] = { t.fwd[k], t.fwd_q[k] };
            dit[k] = { t.dit[k], t.dit_q[k] }; sc[k] = { t.scale[k], t.scale_q[k] };
        }
        NttTable tab{ q, t.ninv, t.ninv_q, t.mod.ratio[1], fwd.data(), dit.data(), sc.data(),
                      ntt_is_narrow(q, logn) ? 1 : 0, 0, 0, 0 };
        ntt_fold_params(q, tab.fold_k, tab.fold_c);
        tab.wide_d4 = ntt_wide_d4(q, tab.narrow != 0);
        if (coeffs == 8) {
            if (logn == 13) { if (inverse) emu_ntt<13, true, 8>(data, tab, threads); else emu_ntt<13, false, 8>(data, tab, threads); }
            else { if (inverse) emu_ntt<12, true, 8>(data, tab, threads); else emu_ntt<12, false, 8>(data, tab, threads); }
            return 0;
        }
#define CASE(L) case L: if (inverse) emu_ntt<L, true>(data, tab, threads); else emu_ntt<L, false>(data, tab, threads); break;
        switch (logn) { CASE(14) CASE(13) CASE(12) CASE(11) CASE(10) CASE(8) CASE(6) default: throw std::invalid_argument("unsupported logn"); }
#undef CASE
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// INTT(x0*y0 (+ x1*y1)) of one limb through the fused loader; x1 = y1 = NULL for a single product.  Returns -2 when the
// modulus does not admit the 128-bit fold reduction (the engine then keeps the separate tensor kernel).
int emu_intt_tensor_limb_c(int logn, uint64_t q, const uint64_t *x0, const uint64_t *y0, const uint64_t *x1, const uint64_t *y1,
                           uint64_t *out, int threads, int coeffs);
int emu_intt_tensor_limb(int logn, uint64_t q, const uint64_t *x0, const uint64_t *y0, const uint64_t *x1, const uint64_t *y1,
                         uint64_t *out, int threads)
{
    return emu_intt_tensor_limb_c(logn, q, x0, y0, x1, y1, out, threads, 16);
}
int emu_intt_tensor_limb_c(int logn, uint64_t q, const uint64_t *x0, const uint64_t *y0, const uint64_t *x1, const uint64_t *y1,
                           uint64_t *out, int threads, int coeffs)
{
    try {
        if ((coeffs & 0xff) != 16 && !((coeffs & 0xff) == 8 && plan_has_latency_form(logn))) throw std::invalid_argument("no pass schedule for this ring size and coefficients per work item");
        size_t n = (size_t)1 << logn;
        HeParams hp = HeParams::Create(n, { q }, 65537 < q ? 65537 : 3);
        const NttTablesHost &t = hp.ntt[0];
        std::vector<TwPair> fwd(n), dit(n), sc(n);
        for (size_t k = 0; k < n; k++) {
            fwd[k] = { t.fwd[k], t.fwd_q[k] };
            dit[k] = { t.dit[k], t.dit_q[k] }; sc[k] = { t.scale[k], t.scale_q[k] };
        }
        NttTable tab{ q, t.ninv, t.ninv_q, t.mod.ratio[1], fwd.data(), dit.data(), sc.data(), ntt_is_narrow(q, logn) ? 1 : 0, 0, 0, 0 };
        ntt_fold_params(q, tab.fold_k, tab.fold_c);
        tab.wide_d4 = ntt_wide_d4(q, tab.narrow != 0);
        if (!ntt_fold128_ok(tab.fold_k, tab.fold_c)) return -2;
        // coeffs | 0x100: the products enter as the fold's last word (< 4q) where the engine would take them so (k_intt_tensor: ntt_lazy_input_ok)
        const bool lazy = (coeffs & 0x100) != 0;
        coeffs &= 0xff;
        if (lazy && !ntt_lazy_input_ok(tab, logn)) return -3;
        const SrcTensor ops{ x0, y0, x1, y1, lazy };
        if (coeffs == 8) {
            if (logn == 13) emu_intt_tensor<13, 8>(out, tab, threads, ops); else emu_intt_tensor<12, 8>(out, tab, threads, ops);
            return 0;
        }
#define CASE(L) case L: emu_intt_tensor<L>(out, tab, threads, ops); break;
        switch (logn) { CASE(14) CASE(13) CASE(12) CASE(11) CASE(10) CASE(8) CASE(6) default: throw std::invalid_argument("unsupported logn"); }
#undef CASE
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// `count` 32-bit outputs of the Blake2xb generator (blake2x.h, the code k_fill_blake2xb runs) starting at output `first`
int emu_blake2xb_values(const uint64_t *seed, uint64_t first, uint32_t *out, int count)
{
    Blake2xbSeed sd;
    for (int i = 0; i < 8; i++) sd.w[i] = seed[i];
    u64 blk[8];
    u64 have = ~(u64)0;
    for (int i = 0; i < count; i++) {
        const u64 g = first + (u64)i;
        if (g / 16 != have) { have = g / 16; blake2xb_stream_block(sd, have, blk); }
        out[i] = blake2xb_stream_u32(blk, (unsigned)(g % 16));
    }
    return 0;
}

// ntt_reduce128_fold on explicit (hi, lo) pairs; returns 0 when the modulus does not admit it
int emu_reduce128(uint64_t q, const uint64_t *hi, const uint64_t *lo, uint64_t *out, int count)
{
    NttTable tab{};
    tab.q = q;
    ntt_fold_params(q, tab.fold_k, tab.fold_c);
    if (!ntt_fold128_ok(tab.fold_k, tab.fold_c)) return 0;
    for (int i = 0; i < count; i++) out[i] = ntt_reduce128_fold(hi[i], lo[i], tab);
    return (int)tab.fold_k;
}

// ntt_reduce_any (the fold / Barrett final reduction of the NTT kernels) on explicit values; returns fold_k
int emu_reduce_any(uint64_t q, const uint64_t *x, uint64_t *out, int count)
{
    ModulusInfo m(q);
    NttTable tab{};
    tab.q = q; tab.r1 = m.ratio[1];
    ntt_fold_params(q, tab.fold_k, tab.fold_c);
    for (int i = 0; i < count; i++) out[i] = ntt_reduce_any(x[i], tab);
    return (int)tab.fold_k;
}

// The scheduler's ordering rules (sched_policy.h), for exhaustive enumeration by tests/test_host_logic.py.
// bits: 0 recycled, 1 last_use_set, 2 last_use_done, 3 high_async, 4 split_ok, 5 prof_on, 6 pipe_cp, 7 force_pipe, 8 inputs_ready,
// 9 on_device, 10 device_busy.  Returns walk | main_waits_high_ready << 2 | side_waits_last_use << 3 | side_waits_main << 4 | consumes_last_use << 5.
int emu_plan_walk(unsigned bits, int split_mode)
{
    WalkState s{};
    s.recycled = bits & 1; s.last_use_set = bits & 2; s.last_use_done = bits & 4; s.high_async = bits & 8; s.split_ok = bits & 16;
    s.prof_on = bits & 32; s.pipe_cp = bits & 64; s.force_pipe = bits & 128; s.inputs_ready = bits & 256; s.on_device = bits & 512;
    s.device_busy = bits & 1024; s.split_mode = split_mode;
    const WalkPlan p = plan_walk(s);
    return p.walk | (p.main_waits_high_ready ? 4 : 0) | (p.side_waits_last_use ? 8 : 0) | (p.side_waits_main ? 16 : 0) | (p.consumes_last_use ? 32 : 0);
}
// entry i: bit 0 fits, bit 1 last_use_set, bit 2 last_use_done
int emu_pick_pooled_buffer(const unsigned char *entries, int count, int inputs_ready)
{
    std::vector<PoolEntryState> st(count);
    for (int i = 0; i < count; i++) st[i] = PoolEntryState{ (entries[i] & 1) != 0, (entries[i] & 2) != 0, (entries[i] & 4) != 0 };
    return pick_pooled_buffer(st.data(), st.size(), inputs_ready != 0);
}

// PSUParams::Load + HeParams: returns derived numbers for comparison with the oracle
int emu_params_info(const char *json, uint64_t *out, int cap)
{
    try {
        PSUParams p = PSUParams::Load(json);
        HeParams hp = HeParams::FromPSUParams(p);
        std::vector<u64> v;
        v.push_back(hp.n); v.push_back(hp.K); v.push_back(hp.first_chain_idx); v.push_back(hp.t);
        for (u64 q : hp.key_q) v.push_back(q);
        for (int j = 0; j < hp.K; j++) v.push_back(hp.ntt[j].psi);
        const LevelConstants &lv = hp.level[hp.first_chain_idx];
        v.push_back(lv.nB); v.push_back(lv.m_sk); v.push_back(lv.gamma);
        for (u64 b : lv.B) v.push_back(b);
        v.push_back(p.bundle_idx_count); v.push_back(p.items_per_bundle); v.push_back(p.item_bit_count);
        v.push_back(hp.irrelevant_bit_count);
        for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
        return (int)v.size();
    } catch (const std::invalid_argument &e) { g_err = e.what(); return -1;
    } catch (const std::exception &e) { g_err = e.what(); return -2; }
}

// PowersDag::configure on explicit sets; nodes: [power, depth, p1, p2] ascending by power
int emu_powers_dag(const uint32_t *sources, int ns, const uint32_t *targets, int nt, uint32_t *nodes)
{
    PowersDag d;
    if (!d.configure(std::set<uint32_t>(sources, sources + ns), std::set<uint32_t>(targets, targets + nt))) return -1;
    int i = 0;
    for (auto &kv : d.nodes()) {
        nodes[4 * i + 0] = kv.second.power; nodes[4 * i + 1] = kv.second.depth;
        nodes[4 * i + 2] = kv.second.parents.first; nodes[4 * i + 3] = kv.second.parents.second;
        i++;
    }
    return (int)d.depth();
}

int emu_create_powers_set(uint32_t ps_low, uint32_t target, uint32_t *out, int cap)
{
    try {
        auto s = create_powers_set(ps_low, target);
        int i = 0;
        for (uint32_t p : s) { if (i < cap) out[i] = p; i++; }
        return i;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// level constants flattened for diffing against the oracle (test_constants)
int emu_level_constants(uint64_t n, const uint64_t *q, int k, uint64_t t, int chain_idx, uint64_t *out, int cap)
{
    try {
        HeParams hp = HeParams::Create((size_t)n, std::vector<u64>(q, q + k), t);
        const LevelConstants &lv = hp.level.at(chain_idx);
        std::vector<u64> v;
        auto put = [&](const std::vector<u64> &a) { for (u64 x : a) v.push_back(x); };
        v.push_back(lv.L); v.push_back(lv.nB); v.push_back(lv.m_sk); v.push_back(lv.gamma);
        put(lv.B); put(lv.coeff_div_plain); v.push_back(lv.q_mod_t); v.push_back(lv.upper_half_threshold);
        put(lv.upper_half_incr); put(lv.inv_q_last); put(lv.inv_punct_q);
        for (auto &r : lv.q_to_bsk) put(r);
        put(lv.q_to_mtilde); v.push_back(lv.neg_inv_q_mod_mtilde);
        put(lv.prod_q_mod_bsk); put(lv.inv_prod_q_mod_bsk); put(lv.inv_mtilde_mod_bsk); put(lv.inv_punct_B);
        for (auto &r : lv.B_to_q) put(r);
        put(lv.B_to_msk); v.push_back(lv.inv_prod_B_mod_msk); put(lv.prod_B_mod_q); put(hp.inv_p_mod_q);
        for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
        return (int)v.size();
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

} // extern "C"
