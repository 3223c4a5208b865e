// A/B of the transform workgroup body (apsu_amd/csrc/ntt_wg.h) outside the engine: the barrier-per-pass schedule of rounds 1-3
// (WS = false) against the wave-private schedule (WS = true), for the plain forward / inverse transforms, the gathered forward
// transform with and without its reduce-on-load, and the tensor-on-load inverse.  Every variant's output is compared bit for bit
// with the WS = false, RED = 1 form (which the engine's parity suite pins against the oracle) before it is timed.
// Limb counts: >= 1 GiB of distinct limbs (HBM) and the launch sizes of a 16M-4096 ComputePowers (cache-resident, launch-bound).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../apsu_amd/csrc ntt_variants.hip ../../apsu_amd/csrc/params.cpp
//        ../../apsu_amd/csrc/powers_dag.cpp -o _bin/ntt_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "ntt_wg.h"
#include "params.h"

using namespace apsu_he;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int LOGN = 13, T = 512, N = 1 << LOGN;

template <bool INV, int WS, bool RAW>
__global__ __launch_bounds__(T, 4) void k_plain(u64 *__restrict__ data, const NttTable *__restrict__ tabs, const int *__restrict__ modmap, int period)
{
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const size_t g = blockIdx.x;
    const NttTable tab = tabs[modmap[g % (size_t)period]];
    u64 *p = data + g * N;
    if (tab.narrow) ntt_body<LOGN, INV, NTT_NARROW, T, 0, RAW, SrcPlain, (WS > 0), false, (WS == 2 ? 12 : WS == 3 ? 40 : 0)>(lds, p, tab, threadIdx.x);
    else if (tab.wide_d4) ntt_body<LOGN, INV, NTT_WIDE_NEAR, T, 0, RAW, SrcPlain, (WS > 0), false, (WS == 2 ? 12 : WS == 3 ? 40 : 0)>(lds, p, tab, threadIdx.x);
    else ntt_body<LOGN, INV, NTT_WIDE, T, 0, RAW, SrcPlain, (WS > 0), false, (WS == 2 ? 12 : WS == 3 ? 40 : 0)>(lds, p, tab, threadIdx.x);
}

// round 5: the plain INVERSE transform with its limb STAGED into the LDS image by coalesced 16-byte loads (the path the tensor-on-load transform takes)
// instead of the first pass reading 16 contiguous coefficients per lane straight from global memory (128 bytes per lane: 64 different lines per load
// instruction, every line touched by eight consecutive instructions -- 512 L1 look-ups per wave instead of 64)
// (SrcStaged is the library's since the variant was adopted: ntt_core.h)
template <bool RAW>
__global__ __launch_bounds__(T, 4) void k_inv_staged(u64 *__restrict__ data, const NttTable *__restrict__ tabs, const int *__restrict__ modmap, int period)
{
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const size_t g = blockIdx.x;
    const NttTable tab = tabs[modmap[g % (size_t)period]];
    u64 *p = data + g * N;
    if (tab.narrow) ntt_body<LOGN, true, NTT_NARROW, T, 0, RAW, SrcStaged>(lds, p, tab, threadIdx.x, nullptr, SrcStaged());
    else if (tab.wide_d4) ntt_body<LOGN, true, NTT_WIDE_NEAR, T, 0, RAW, SrcStaged>(lds, p, tab, threadIdx.x, nullptr, SrcStaged());
    else ntt_body<LOGN, true, NTT_WIDE, T, 0, RAW, SrcStaged>(lds, p, tab, threadIdx.x, nullptr, SrcStaged());
}

template <int RED, int WS>
__global__ __launch_bounds__(T, 4) void k_gather(const u64 *const *__restrict__ src, u64 *__restrict__ data, const NttTable *__restrict__ tabs,
                                                 const int *__restrict__ modmap, int period)
{
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const size_t g = blockIdx.x;
    const NttTable tab = tabs[modmap[g % (size_t)period]];
    u64 *p = data + g * N;
    if (tab.narrow) ntt_body<LOGN, false, NTT_NARROW, T, RED, false, SrcPlain, (WS > 0), false, (WS == 2 ? 12 : WS == 3 ? 40 : 0)>(lds, p, tab, threadIdx.x, src[g]);
    else if (tab.wide_d4) ntt_body<LOGN, false, NTT_WIDE_NEAR, T, 1, false, SrcPlain, (WS > 0), false, (WS == 2 ? 12 : WS == 3 ? 40 : 0)>(lds, p, tab, threadIdx.x, src[g]);
    else ntt_body<LOGN, false, NTT_WIDE, T, 1, false, SrcPlain, (WS > 0), false, (WS == 2 ? 12 : WS == 3 ? 40 : 0)>(lds, p, tab, threadIdx.x, src[g]);
}

// product jb: operands a, b = two polynomials of `limbs` limbs each; workgroup g = (jb, polynomial pl of 3, limb e)
template <int WS>
__global__ __launch_bounds__(T, 4) void k_tensor(const u64 *__restrict__ A, const u64 *__restrict__ B, u64 *__restrict__ D, int limbs,
                                                 const NttTable *__restrict__ tabs, const int *__restrict__ modmap, int period)
{
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const size_t g = blockIdx.x, per = (size_t)3 * limbs, jb = g / per;
    const int r = (int)(g - jb * per), pl = r / limbs, e = r - pl * limbs;
    const NttTable tab = tabs[modmap[g % (size_t)period]];
    const size_t ps = (size_t)limbs * N;
    const u64 *a0 = A + jb * 2 * ps + (size_t)e * N, *a1 = a0 + ps, *b0 = B + jb * 2 * ps + (size_t)e * N, *b1 = b0 + ps;
    SrcTensor ops;
    if (pl == 0) ops = SrcTensor{ a0, b0, nullptr, nullptr, false };
    else if (pl == 1) ops = SrcTensor{ a0, b1, a1, b0, false };
    else ops = SrcTensor{ a1, b1, nullptr, nullptr, false };
    u64 *p = D + g * N;
    if (tab.narrow) ntt_body<LOGN, true, NTT_NARROW, T, 0, false, SrcTensor, (WS > 0), false, (WS == 2 ? 12 : WS == 3 ? 40 : 0)>(lds, p, tab, threadIdx.x, nullptr, ops);
    else if (tab.wide_d4) ntt_body<LOGN, true, NTT_WIDE_NEAR, T, 0, false, SrcTensor, (WS > 0), false, (WS == 2 ? 12 : WS == 3 ? 40 : 0)>(lds, p, tab, threadIdx.x, nullptr, ops);
    else ntt_body<LOGN, true, NTT_WIDE, T, 0, false, SrcTensor, (WS > 0), false, (WS == 2 ? 12 : WS == 3 ? 40 : 0)>(lds, p, tab, threadIdx.x, nullptr, ops);
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
    template <class F> double us(F f, int reps)
    {
        f();
        CHECK(hipDeviceSynchronize());
        double best = 1e30, sum = 0;
        for (int i = 0; i < reps; i++) {
            CHECK(hipEventRecord(a));
            f();
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
            sum += ms;
        }
        (void)best;
        return sum / reps * 1e3;
    }
};

int main(int argc, char **argv)
{
    const size_t big = argc > 1 ? (size_t)atol(argv[1]) : 16380;          // limbs of the streaming runs (multiple of 7 * 12 not needed)
    // 16M-4096: key primes 56, 56, 56, 50 bits; HeParams::Create adds the 61-bit BEHZ primes (ids K ..)
    const std::vector<u64> kq = { 0xfffffffff70001ULL, 0xfffffffff78001ULL, 0xfffffffffb4001ULL, 0x3ffffffffc001ULL };
    HeParams hp = HeParams::Create(N, kq, 4079617);
    const int nmod = (int)hp.ntt.size(), K = hp.K;
    std::vector<TwPair> tw((size_t)nmod * 3 * N);
    for (int m = 0; m < nmod; m++)
        for (size_t k = 0; k < (size_t)N; k++) {
            const NttTablesHost &t = hp.ntt[m];
            tw[((size_t)m * 3 + 0) * N + k] = TwPair{ t.fwd[k], t.fwd_q[k] };
            tw[((size_t)m * 3 + 1) * N + k] = TwPair{ t.dit[k], t.dit_q[k] };
            tw[((size_t)m * 3 + 2) * N + k] = TwPair{ t.scale[k], t.scale_q[k] };
        }
    TwPair *d_tw;
    CHECK(hipMalloc(&d_tw, tw.size() * sizeof(TwPair)));
    CHECK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(TwPair), hipMemcpyHostToDevice));
    std::vector<NttTable> tabs(nmod);
    for (int m = 0; m < nmod; m++) {
        NttTable tb{};
        tb.q = hp.ntt[m].mod.value; tb.ninv = hp.ntt[m].ninv; tb.ninv_q = hp.ntt[m].ninv_q;
        tb.r1 = hp.ntt[m].mod.ratio[1]; tb.r0 = hp.ntt[m].mod.ratio[0];
        tb.narrow = ntt_is_narrow(tb.q, LOGN) ? 1 : 0;
        ntt_fold_params(tb.q, tb.fold_k, tb.fold_c);
        tb.wide_d4 = ntt_wide_d4(tb.q, tb.narrow != 0);
        tb.fwd = d_tw + ((size_t)m * 3 + 0) * N; tb.dit = d_tw + ((size_t)m * 3 + 1) * N; tb.scale = d_tw + ((size_t)m * 3 + 2) * N;
        tabs[m] = tb;
        printf("modulus %d: %d bits, %s\n", m, hp.ntt[m].mod.bits, tb.narrow ? "narrow" : (tb.wide_d4 ? "wide-near" : "wide"));
    }
    NttTable *d_tabs;
    CHECK(hipMalloc(&d_tabs, nmod * sizeof(NttTable)));
    CHECK(hipMemcpy(d_tabs, tabs.data(), nmod * sizeof(NttTable), hipMemcpyHostToDevice));
    // modulus maps: the three data primes (narrow); the extended base of the first level: q0 q1 q2 | B0 B1 B2 | m_sk (3 narrow + 4 wide-near);
    // the key switch: target (q0 q1 q2 p) x source (3)
    const std::vector<int> map_q = { 0, 1, 2 }, map_ext = { 0, 1, 2, K + 2, K + 3, K + 4, K + 0 };
    std::vector<int> map_ks;
    for (int I = 0; I < 4; I++) for (int J = 0; J < 3; J++) map_ks.push_back(I);
    auto up = [&](const std::vector<int> &m) { int *d; CHECK(hipMalloc(&d, m.size() * sizeof(int))); CHECK(hipMemcpy(d, m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice)); return d; };
    int *d_map_q = up(map_q), *d_map_ext = up(map_ext), *d_map_ks = up(map_ks);

    const size_t words = big * N;
    std::vector<u64> host(words);
    std::mt19937_64 rng(0x41505355);
    for (size_t i = 0; i < words; i++) host[i] = rng() % kq[0];           // below every 56- and 61-bit modulus; the 50-bit key prime only ever sees them as gathered sources
    u64 *d_in, *d_a, *d_b;
    CHECK(hipMalloc(&d_in, words * 8)); CHECK(hipMalloc(&d_a, words * 8)); CHECK(hipMalloc(&d_b, words * 8));
    CHECK(hipMemcpy(d_in, host.data(), words * 8, hipMemcpyHostToDevice));
    std::vector<const u64 *> srcp(big);
    for (size_t g = 0; g < big; g++) srcp[g] = d_in + ((g * 7919) % big) * N;                     // a scattered gather
    const u64 **d_src;
    CHECK(hipMalloc(&d_src, big * sizeof(u64 *)));
    CHECK(hipMemcpy(d_src, srcp.data(), big * sizeof(u64 *), hipMemcpyHostToDevice));
    std::vector<u64> ra(words), rb(words);
    auto same = [&](const char *what) {
        CHECK(hipMemcpy(ra.data(), d_a, words * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(rb.data(), d_b, words * 8, hipMemcpyDeviceToHost));
        const bool ok = std::memcmp(ra.data(), rb.data(), words * 8) == 0;
        printf("%-58s %s\n", what, ok ? "same bits" : "MISMATCH");
        if (!ok) exit(2);
    };
    auto copy_in = [&](u64 *dst) { CHECK(hipMemcpy(dst, d_in, words * 8, hipMemcpyDeviceToDevice)); };
    // ---- correctness: new schedule == old schedule, bit for bit
    for (int which = 0; which < 2; which++) {
        const int *map = which ? d_map_ext : d_map_q; const int period = which ? 7 : 3;
        copy_in(d_a); copy_in(d_b);
        hipLaunchKernelGGL((k_plain<false, 0, false>), dim3(big), dim3(T), 0, 0, d_a, d_tabs, map, period);
        hipLaunchKernelGGL((k_plain<false, 1, false>), dim3(big), dim3(T), 0, 0, d_b, d_tabs, map, period);
        same(which ? "forward, extended base (narrow + wide-near)" : "forward, data primes (narrow)");
        copy_in(d_b);
        hipLaunchKernelGGL((k_plain<false, 2, false>), dim3(big), dim3(T), 0, 0, d_b, d_tabs, map, period);
        same("  ... with stagger");
        hipLaunchKernelGGL((k_plain<true, 0, false>), dim3(big), dim3(T), 0, 0, d_a, d_tabs, map, period);
        hipLaunchKernelGGL((k_plain<true, 2, false>), dim3(big), dim3(T), 0, 0, d_b, d_tabs, map, period);
        same(which ? "inverse, extended base (stagger)" : "inverse, data primes (stagger)");
        CHECK(hipMemcpy(ra.data(), d_a, words * 8, hipMemcpyDeviceToHost));
        if (std::memcmp(ra.data(), host.data(), words * 8)) { printf("inverse(forward(x)) != x\n"); return 2; }
        copy_in(d_a); copy_in(d_b);
        hipLaunchKernelGGL((k_plain<true, 0, true>), dim3(big), dim3(T), 0, 0, d_a, d_tabs, map, period);
        hipLaunchKernelGGL((k_plain<true, 1, true>), dim3(big), dim3(T), 0, 0, d_b, d_tabs, map, period);
        same(which ? "inverse RAW, extended base" : "inverse RAW, data primes");
        copy_in(d_b);
        hipLaunchKernelGGL((k_plain<true, 2, true>), dim3(big), dim3(T), 0, 0, d_b, d_tabs, map, period);
        same("  ... with stagger");
    }
    for (int which = 0; which < 2; which++) {                     // staged inverse == the library's inverse
        const int *map = which ? d_map_ext : d_map_q; const int period = which ? 7 : 3;
        copy_in(d_a); copy_in(d_b);
        hipLaunchKernelGGL((k_plain<true, 1, true>), dim3(big), dim3(T), 0, 0, d_a, d_tabs, map, period);
        hipLaunchKernelGGL((k_inv_staged<true>), dim3(big), dim3(T), 0, 0, d_b, d_tabs, map, period);
        same(which ? "inverse RAW staged, extended base" : "inverse RAW staged, data primes");
        copy_in(d_a); copy_in(d_b);
        hipLaunchKernelGGL((k_plain<true, 1, false>), dim3(big), dim3(T), 0, 0, d_a, d_tabs, map, period);
        hipLaunchKernelGGL((k_inv_staged<false>), dim3(big), dim3(T), 0, 0, d_b, d_tabs, map, period);
        same(which ? "inverse staged, extended base" : "inverse staged, data primes");
    }
    hipLaunchKernelGGL((k_gather<1, 0>), dim3(big), dim3(T), 0, 0, d_src, d_a, d_tabs, d_map_ks, 12);
    hipLaunchKernelGGL((k_gather<1, 2>), dim3(big), dim3(T), 0, 0, d_src, d_b, d_tabs, d_map_ks, 12);
    same("gathered forward, reduce on load (stagger)");
    hipLaunchKernelGGL((k_gather<2, 1>), dim3(big), dim3(T), 0, 0, d_src, d_b, d_tabs, d_map_ks, 12);
    same("gathered forward, NO reduce on load");
    hipLaunchKernelGGL((k_gather<2, 2>), dim3(big), dim3(T), 0, 0, d_src, d_b, d_tabs, d_map_ks, 12);
    same("gathered forward, NO reduce on load (stagger)");
    const int limbs = 7;
    const size_t prods = big / (3 * limbs) / 2 * 2, tgrid = prods * 3 * limbs;                     // operands: prods x 2 polys x 7 limbs <= big limbs
    {
        u64 *d_out2;
        CHECK(hipMalloc(&d_out2, tgrid * N * 8));
        hipLaunchKernelGGL((k_tensor<0>), dim3(tgrid), dim3(T), 0, 0, d_in, d_in + (big / 3) * N, d_a, limbs, d_tabs, d_map_ext, 7);
        hipLaunchKernelGGL((k_tensor<2>), dim3(tgrid), dim3(T), 0, 0, d_in, d_in + (big / 3) * N, d_out2, limbs, d_tabs, d_map_ext, 7);
        CHECK(hipMemcpy(ra.data(), d_a, tgrid * N * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(rb.data(), d_out2, tgrid * N * 8, hipMemcpyDeviceToHost));
        const bool ok = std::memcmp(ra.data(), rb.data(), tgrid * N * 8) == 0;
        printf("%-58s %s\n", "tensor-on-load inverse", ok ? "same bits" : "MISMATCH");
        if (!ok) return 2;
        CHECK(hipFree(d_out2));
    }
    // ---- timing
    Timer tm;
    const int reps = 7;
    auto line = [&](const char *name, size_t count, double t0, double t1, double t2, double t3) {
        const double by = (double)count * 16 * N;
        printf("%-38s %6zu limbs  old %7.1f us %5.0f GB/s | wave-private %7.1f us %5.0f GB/s %+5.1f %% | stagger 12: %7.1f us %+5.1f %% | stagger 40: %7.1f us %+5.1f %%\n", name, count,
               t0, by / t0 / 1e3, t1, by / t1 / 1e3, (t1 / t0 - 1) * 100, t2, (t2 / t0 - 1) * 100, t3, (t3 / t0 - 1) * 100);
    };
#define PLAIN3(NAME, INV, RAW, MAP, PER) line(NAME, count, \
        tm.us([&] { hipLaunchKernelGGL((k_plain<INV, 0, RAW>), g, dim3(T), 0, 0, d_a, d_tabs, MAP, PER); }, reps), \
        tm.us([&] { hipLaunchKernelGGL((k_plain<INV, 1, RAW>), g, dim3(T), 0, 0, d_a, d_tabs, MAP, PER); }, reps), \
        tm.us([&] { hipLaunchKernelGGL((k_plain<INV, 2, RAW>), g, dim3(T), 0, 0, d_a, d_tabs, MAP, PER); }, reps), \
        tm.us([&] { hipLaunchKernelGGL((k_plain<INV, 3, RAW>), g, dim3(T), 0, 0, d_a, d_tabs, MAP, PER); }, reps))
    const size_t sizes[] = { big, big, 6840, 1872, 624, 416 };
    for (size_t count : sizes) {
        if (count > big) continue;
        const dim3 g((unsigned)count);
        printf("\n");
        PLAIN3("forward, data primes", false, false, d_map_q, 3);
        PLAIN3("forward, extended base", false, false, d_map_ext, 7);
        PLAIN3("inverse, data primes", true, false, d_map_q, 3);
        PLAIN3("inverse RAW, data primes", true, true, d_map_q, 3);
        PLAIN3("inverse, extended base", true, false, d_map_ext, 7);
        if (getenv("STAGED")) {
            auto st = [&](const char *name, bool raw, const int *map, int per) {
                const double a = raw ? tm.us([&] { hipLaunchKernelGGL((k_plain<true, 1, true>), g, dim3(T), 0, 0, d_a, d_tabs, map, per); }, reps)
                                     : tm.us([&] { hipLaunchKernelGGL((k_plain<true, 1, false>), g, dim3(T), 0, 0, d_a, d_tabs, map, per); }, reps);
                const double b = raw ? tm.us([&] { hipLaunchKernelGGL((k_inv_staged<true>), g, dim3(T), 0, 0, d_a, d_tabs, map, per); }, reps)
                                     : tm.us([&] { hipLaunchKernelGGL((k_inv_staged<false>), g, dim3(T), 0, 0, d_a, d_tabs, map, per); }, reps);
                const double by = (double)count * 16 * N;
                printf("STAGED %-34s %6zu limbs  library %7.1f us %5.0f GB/s | staged %7.1f us %5.0f GB/s %+5.1f %%\n", name, count, a, by / a / 1e3, b, by / b / 1e3, (b / a - 1) * 100);
            };
            st("inverse RAW, data primes", true, d_map_q, 3);
            st("inverse, data primes", false, d_map_q, 3);
            st("inverse RAW, extended base", true, d_map_ext, 7);
        }
        line("gather, reduce on load", count, tm.us([&] { hipLaunchKernelGGL((k_gather<1, 0>), g, dim3(T), 0, 0, d_src, d_a, d_tabs, d_map_ks, 12); }, reps),
             tm.us([&] { hipLaunchKernelGGL((k_gather<1, 1>), g, dim3(T), 0, 0, d_src, d_a, d_tabs, d_map_ks, 12); }, reps),
             tm.us([&] { hipLaunchKernelGGL((k_gather<1, 2>), g, dim3(T), 0, 0, d_src, d_a, d_tabs, d_map_ks, 12); }, reps),
             tm.us([&] { hipLaunchKernelGGL((k_gather<1, 3>), g, dim3(T), 0, 0, d_src, d_a, d_tabs, d_map_ks, 12); }, reps));
        line("gather, old: reduce; new: NO reduce", count, tm.us([&] { hipLaunchKernelGGL((k_gather<1, 0>), g, dim3(T), 0, 0, d_src, d_a, d_tabs, d_map_ks, 12); }, reps),
             tm.us([&] { hipLaunchKernelGGL((k_gather<2, 1>), g, dim3(T), 0, 0, d_src, d_a, d_tabs, d_map_ks, 12); }, reps),
             tm.us([&] { hipLaunchKernelGGL((k_gather<2, 2>), g, dim3(T), 0, 0, d_src, d_a, d_tabs, d_map_ks, 12); }, reps),
             tm.us([&] { hipLaunchKernelGGL((k_gather<2, 3>), g, dim3(T), 0, 0, d_src, d_a, d_tabs, d_map_ks, 12); }, reps));
        const size_t tg = count / 21 * 21;
        if (tg && tg <= tgrid) {
            const dim3 gt((unsigned)tg);
            line("tensor-on-load inverse, extended base", tg,
                 tm.us([&] { hipLaunchKernelGGL((k_tensor<0>), gt, dim3(T), 0, 0, d_in, d_in + (big / 3) * N, d_a, limbs, d_tabs, d_map_ext, 7); }, reps),
                 tm.us([&] { hipLaunchKernelGGL((k_tensor<1>), gt, dim3(T), 0, 0, d_in, d_in + (big / 3) * N, d_a, limbs, d_tabs, d_map_ext, 7); }, reps),
                 tm.us([&] { hipLaunchKernelGGL((k_tensor<2>), gt, dim3(T), 0, 0, d_in, d_in + (big / 3) * N, d_a, limbs, d_tabs, d_map_ext, 7); }, reps),
                 tm.us([&] { hipLaunchKernelGGL((k_tensor<3>), gt, dim3(T), 0, 0, d_in, d_in + (big / 3) * N, d_a, limbs, d_tabs, d_map_ext, 7); }, reps));
        }
    }
    printf("\nok\n");
    return 0;
}
