// Round 6: the FORMS of the LDS-resident transform workgroup (apsu_amd/csrc/ntt_wg.h) side by side, outside the engine:
//   F16   16 coefficients per lane, T = n / 16 threads per limb, <= 128 VGPRs (4 waves per SIMD)        -- the throughput form of rounds 1-5
//   F8     8 coefficients per lane, T = n / 8,  <= 128 VGPRs: twice the waves per limb                   -- the latency form
//   F8o    the same with <= 64 VGPRs (__launch_bounds__(T, 8)): two 1024-thread workgroups per CU at n = 8192, 8 waves per SIMD
// for the plain forward / inverse (staged; with its twist and RAW) transforms, the gathered forward transform without reduce-on-load and the
// tensor-on-load inverse, over launch sizes from one limb to >= 1 GiB.  Every form's output is compared bit for bit with F16's before it is timed.
// Build (RING_LOGN = 13 or 12):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRING_LOGN=13 -I../../apsu_amd/csrc ntt_forms.hip ../../apsu_amd/csrc/params.cpp ../../apsu_amd/csrc/powers_dag.cpp -o _bin/ntt_forms13
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

#ifndef RING_LOGN
#define RING_LOGN 13
#endif
constexpr int LOGN = RING_LOGN, N = 1 << LOGN;

#define BODY(INV, RED, RAW, SRC, C, ...) \
    if (tab.narrow) ntt_body<LOGN, INV, NTT_NARROW, N / C, RED, RAW, SRC, true, false, 0, -1, C>(__VA_ARGS__); \
    else if (tab.wide_d4) ntt_body<LOGN, INV, NTT_WIDE_NEAR, N / C, (RED == 2 ? 1 : RED), RAW, SRC, true, false, 0, -1, C>(__VA_ARGS__); \
    else ntt_body<LOGN, INV, NTT_WIDE, N / C, (RED == 2 ? 1 : RED), RAW, SRC, true, false, 0, -1, C>(__VA_ARGS__);

template <bool INV, bool RAW, int C, int MINW>
__global__ __launch_bounds__(N / C, MINW) void k_plain(u64 *__restrict__ data, const NttTable *__restrict__ tabs, const int *__restrict__ modmap, int period)
{
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const size_t g = blockIdx.x;
    const NttTable tab = tabs[modmap[g % (size_t)period]];
    u64 *p = data + g * N;
    using SRC = std::conditional_t<INV, SrcStaged, SrcPlain>;
    BODY(INV, 0, RAW, SRC, C, lds, p, tab, threadIdx.x, nullptr, SRC())
}

template <int C, int MINW>
__global__ __launch_bounds__(N / C, MINW) void k_gather(const u64 *const *__restrict__ src, u64 *__restrict__ data, const NttTable *__restrict__ tabs,
                                                        const int *__restrict__ modmap, int period)
{
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const size_t g = blockIdx.x;
    const NttTable tab = tabs[modmap[g % (size_t)period]];
    u64 *p = data + g * N;
    BODY(false, 2, false, SrcPlain, C, lds, p, tab, threadIdx.x, src[g])
}

// product jb: operands a, b = two polynomials of `limbs` limbs each; workgroup g = (jb, polynomial pl of 3, limb e)
template <int C, int MINW>
__global__ __launch_bounds__(N / C, MINW) void k_tensor(const u64 *__restrict__ A, const u64 *__restrict__ B, u64 *__restrict__ D, int limbs,
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
    BODY(true, 0, true, SrcTensor, C, lds, p, tab, threadIdx.x, nullptr, ops)
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
    // mean over `reps` back-to-back launches bracketed by ONE event pair (the event pair's own ~4 us is shared by all of them)
    template <class F> double us(F f, int reps)
    {
        f(); f();
        CHECK(hipDeviceSynchronize());
        double best = 1e30;
        for (int round = 0; round < 3; round++) {
            CHECK(hipEventRecord(a));
            for (int i = 0; i < reps; i++) f();
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        return best / reps * 1e3;
    }
};

int main(int argc, char **argv)
{
    const size_t big = argc > 1 ? (size_t)atol(argv[1]) : (LOGN == 13 ? 16380 : 32760);          // limbs of the streaming runs (>= 1 GiB)
    const std::vector<u64> kq13 = { 0xfffffffff70001ULL, 0xfffffffff78001ULL, 0xfffffffffb4001ULL, 0x3ffffffffc001ULL };   // 16M-4096
    const std::vector<u64> kq12 = { 0xffffffffc001ULL, 0xffffee001ULL, 0x1ffc001ULL };                                      // 1M-1024-com
    const std::vector<u64> &kq = LOGN == 13 ? kq13 : kq12;
    HeParams hp = HeParams::Create(N, kq, LOGN == 13 ? 4079617 : 188417);
    const int nmod = (int)hp.ntt.size(), K = hp.K, Ld = K - 1;
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
    }
    NttTable *d_tabs;
    CHECK(hipMalloc(&d_tabs, nmod * sizeof(NttTable)));
    CHECK(hipMemcpy(d_tabs, tabs.data(), nmod * sizeof(NttTable), hipMemcpyHostToDevice));
    // modulus maps: the data primes (narrow); the extended base of the first level q.. | B.. | m_sk; the key switch: target (q.., p) x source
    std::vector<int> map_q, map_ext, map_ks;
    for (int j = 0; j < Ld; j++) map_q.push_back(j);
    map_ext = map_q;
    for (int j = 0; j < Ld; j++) map_ext.push_back(K + 2 + j);
    map_ext.push_back(K + 0);
    for (int I = 0; I <= Ld; I++) for (int J = 0; J < Ld; J++) map_ks.push_back(I);
    const int E = (int)map_ext.size();
    auto up = [&](const std::vector<int> &m) { int *d; CHECK(hipMalloc(&d, m.size() * sizeof(int))); CHECK(hipMemcpy(d, m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice)); return d; };
    int *d_map_q = up(map_q), *d_map_ext = up(map_ext), *d_map_ks = up(map_ks);
    printf("n = %d: data primes %d, extended base %d limbs; forms: F16 = %d threads x 16, F8 = %d threads x 8 (<= 128 VGPRs), F8o = the same at <= 64 VGPRs\n",
           N, Ld, E, N / 16, N / 8);

    const size_t words = big * N;
    std::vector<u64> host(words);
    std::mt19937_64 rng(0x41505355);
    u64 qmin = kq[0];
    for (int j = 0; j < Ld; j++) qmin = std::min(qmin, kq[j]);
    for (size_t i = 0; i < words; i++) host[i] = rng() % qmin;
    u64 *d_in, *d_a, *d_b;
    CHECK(hipMalloc(&d_in, words * 8)); CHECK(hipMalloc(&d_a, words * 8)); CHECK(hipMalloc(&d_b, words * 8));
    CHECK(hipMemcpy(d_in, host.data(), words * 8, hipMemcpyHostToDevice));
    std::vector<const u64 *> srcp(big);
    for (size_t g = 0; g < big; g++) srcp[g] = d_in + ((g * 7919) % big) * N;
    const u64 **d_src;
    CHECK(hipMalloc(&d_src, big * sizeof(u64 *)));
    CHECK(hipMemcpy(d_src, srcp.data(), big * sizeof(u64 *), hipMemcpyHostToDevice));
    std::vector<u64> ra(words), rb(words);
    auto same = [&](const char *what, size_t w = 0) {
        if (!w) w = words;
        CHECK(hipMemcpy(ra.data(), d_a, w * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(rb.data(), d_b, w * 8, hipMemcpyDeviceToHost));
        const bool ok = std::memcmp(ra.data(), rb.data(), w * 8) == 0;
        printf("  %-66s %s\n", what, ok ? "same bits" : "MISMATCH");
        if (!ok) exit(2);
    };
    // RAW outputs are lazy representatives (the consumer's constants absorb twist and reduction): two forms whose first inverse pass covers
    // different stages may leave different representatives of the same residues -- compare modulo the limb's modulus
    auto same_mod = [&](const char *what, const std::vector<int> &map, size_t w = 0) {
        if (!w) w = words;
        CHECK(hipMemcpy(ra.data(), d_a, w * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(rb.data(), d_b, w * 8, hipMemcpyDeviceToHost));
        bool ok = true;
        for (size_t i = 0; i < w && ok; i++) {
            const u64 q = tabs[map[(i / N) % map.size()]].q;
            ok = ra[i] % q == rb[i] % q;
        }
        printf("  %-66s %s\n", what, ok ? "same residues" : "MISMATCH");
        if (!ok) exit(2);
    };
    auto copy_in = [&](u64 *dst) { CHECK(hipMemcpy(dst, d_in, words * 8, hipMemcpyDeviceToDevice)); };
    const int limbs = E;
    const size_t prods = big / (3 * limbs) / 2 * 2, tgrid = prods * 3 * limbs;
    // ---- correctness: F8 / F8o == F16, bit for bit
#define LAUNCH_PLAIN(INV, RAW, C, MINW, buf, cnt, MAP, PER) hipLaunchKernelGGL((k_plain<INV, RAW, C, MINW>), dim3((unsigned)(cnt)), dim3(N / C), 0, 0, buf, d_tabs, MAP, PER)
    for (int which = 0; which < 2; which++) {
        const int *map = which ? d_map_ext : d_map_q; const int period = which ? E : Ld;
        const char *base = which ? "extended base" : "data primes";
        char msg[128];
        copy_in(d_a); copy_in(d_b);
        LAUNCH_PLAIN(false, false, 16, 4, d_a, big, map, period); LAUNCH_PLAIN(false, false, 8, 4, d_b, big, map, period);
        snprintf(msg, sizeof msg, "forward, %s: F8", base); same(msg);
        copy_in(d_b); LAUNCH_PLAIN(false, false, 8, 8, d_b, big, map, period);
        snprintf(msg, sizeof msg, "forward, %s: F8o", base); same(msg);
        LAUNCH_PLAIN(true, false, 16, 4, d_a, big, map, period); LAUNCH_PLAIN(true, false, 8, 4, d_b, big, map, period);
        snprintf(msg, sizeof msg, "inverse, %s: F8", base); same(msg);
        CHECK(hipMemcpy(ra.data(), d_a, words * 8, hipMemcpyDeviceToHost));
        if (std::memcmp(ra.data(), host.data(), words * 8)) { printf("inverse(forward(x)) != x\n"); return 2; }
        copy_in(d_a); copy_in(d_b);
        LAUNCH_PLAIN(true, true, 16, 4, d_a, big, map, period); LAUNCH_PLAIN(true, true, 8, 4, d_b, big, map, period);
        snprintf(msg, sizeof msg, "inverse RAW, %s: F8", base); same_mod(msg, which ? map_ext : map_q);
        copy_in(d_b); LAUNCH_PLAIN(true, true, 8, 8, d_b, big, map, period);
        snprintf(msg, sizeof msg, "inverse RAW, %s: F8o", base); same_mod(msg, which ? map_ext : map_q);
        copy_in(d_a); copy_in(d_b);
        LAUNCH_PLAIN(true, false, 16, 4, d_a, big, map, period); LAUNCH_PLAIN(true, false, 8, 8, d_b, big, map, period);
        snprintf(msg, sizeof msg, "inverse (with twist), %s: F8o", base); same(msg);
    }
    hipLaunchKernelGGL((k_gather<16, 4>), dim3(big), dim3(N / 16), 0, 0, d_src, d_a, d_tabs, d_map_ks, (Ld + 1) * Ld);
    hipLaunchKernelGGL((k_gather<8, 4>), dim3(big), dim3(N / 8), 0, 0, d_src, d_b, d_tabs, d_map_ks, (Ld + 1) * Ld);
    same("gathered forward (no reduce on load): F8");
    hipLaunchKernelGGL((k_gather<8, 8>), dim3(big), dim3(N / 8), 0, 0, d_src, d_b, d_tabs, d_map_ks, (Ld + 1) * Ld);
    same("gathered forward (no reduce on load): F8o");
    {
        hipLaunchKernelGGL((k_tensor<16, 4>), dim3(tgrid), dim3(N / 16), 0, 0, d_in, d_in + (big / 3) * N, d_a, limbs, d_tabs, d_map_ext, E);
        hipLaunchKernelGGL((k_tensor<8, 4>), dim3(tgrid), dim3(N / 8), 0, 0, d_in, d_in + (big / 3) * N, d_b, limbs, d_tabs, d_map_ext, E);
        same_mod("tensor-on-load inverse: F8", map_ext, tgrid * N);
        hipLaunchKernelGGL((k_tensor<8, 8>), dim3(tgrid), dim3(N / 8), 0, 0, d_in, d_in + (big / 3) * N, d_b, limbs, d_tabs, d_map_ext, E);
        same_mod("tensor-on-load inverse: F8o", map_ext, tgrid * N);
    }
    // ---- timing
    Timer tm;
    auto line = [&](const char *name, size_t count, double t0, double t1, double t2) {
        const double by = (double)count * 16 * N;
        printf("%-30s %6zu limbs  F16 %7.1f us %5.0f GB/s | F8 %7.1f us %5.0f GB/s %+6.1f %% | F8o %7.1f us %5.0f GB/s %+6.1f %%\n", name, count,
               t0, by / t0 / 1e3, t1, by / t1 / 1e3, (t1 / t0 - 1) * 100, t2, by / t2 / 1e3, (t2 / t0 - 1) * 100);
    };
#define PLAIN3(NAME, INV, RAW, MAP, PER) line(NAME, count, \
        tm.us([&] { LAUNCH_PLAIN(INV, RAW, 16, 4, d_a, count, MAP, PER); }, reps), \
        tm.us([&] { LAUNCH_PLAIN(INV, RAW, 8, 4, d_a, count, MAP, PER); }, reps), \
        tm.us([&] { LAUNCH_PLAIN(INV, RAW, 8, 8, d_a, count, MAP, PER); }, reps))
    const size_t sizes[] = { big, 6840, 3825, 2048, 1536, 1024, 768, 512, 384, 256, 168, 56, 6 };
    for (size_t count : sizes) {
        if (count > big) continue;
        const int reps = count >= 2048 ? 8 : 40;
        printf("\n");
        PLAIN3("forward, data primes", false, false, d_map_q, Ld);
        PLAIN3("forward, extended base", false, false, d_map_ext, E);
        PLAIN3("inverse, data primes", true, false, d_map_q, Ld);
        PLAIN3("inverse RAW, data primes", true, true, d_map_q, Ld);
        PLAIN3("inverse RAW, extended base", true, true, d_map_ext, E);
        const dim3 g((unsigned)count);
        line("gather (no reduce on load)", count,
             tm.us([&] { hipLaunchKernelGGL((k_gather<16, 4>), g, dim3(N / 16), 0, 0, d_src, d_a, d_tabs, d_map_ks, (Ld + 1) * Ld); }, reps),
             tm.us([&] { hipLaunchKernelGGL((k_gather<8, 4>), g, dim3(N / 8), 0, 0, d_src, d_a, d_tabs, d_map_ks, (Ld + 1) * Ld); }, reps),
             tm.us([&] { hipLaunchKernelGGL((k_gather<8, 8>), g, dim3(N / 8), 0, 0, d_src, d_a, d_tabs, d_map_ks, (Ld + 1) * Ld); }, reps));
        const size_t tg = count / (3 * limbs) * (3 * limbs);
        if (tg && tg <= tgrid) {
            const dim3 gt((unsigned)tg);
            line("tensor-on-load inverse RAW", tg,
                 tm.us([&] { hipLaunchKernelGGL((k_tensor<16, 4>), gt, dim3(N / 16), 0, 0, d_in, d_in + (big / 3) * N, d_a, limbs, d_tabs, d_map_ext, E); }, reps),
                 tm.us([&] { hipLaunchKernelGGL((k_tensor<8, 4>), gt, dim3(N / 8), 0, 0, d_in, d_in + (big / 3) * N, d_a, limbs, d_tabs, d_map_ext, E); }, reps),
                 tm.us([&] { hipLaunchKernelGGL((k_tensor<8, 8>), gt, dim3(N / 8), 0, 0, d_in, d_in + (big / 3) * N, d_a, limbs, d_tabs, d_map_ext, E); }, reps));
        }
    }
    printf("\nok\n");
    return 0;
}
