// A/B of the "touch the successor's limb" prefetch of the transform workgroup (apsu_amd/csrc/ntt_wg.h, NEXTPASS): the workgroup
// g touches the limb of workgroup g + ahead (one 128-byte line per thread) in front of its pass number NEXTPASS, so that the
// workgroup which takes over its slot finds its input in the XCD's L2 instead of waiting for HBM in its first pass.
// Variants: off | in front of pass 0 | pass 2 | pass 3 (the last), ahead = 256 / 512 / 1024 workgroups; every variant's output is
// compared bit for bit with the variant without the touch.  Limb counts: HBM-sized batches and the launch sizes of a 16M-4096 query.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../apsu_amd/csrc ntt_next.hip ../../apsu_amd/csrc/params.cpp
//        ../../apsu_amd/csrc/powers_dag.cpp -o _bin/ntt_next
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

template <bool INV, bool RAW, int NP>
__global__ __launch_bounds__(T, 4) void k_plain(u64 *__restrict__ data, const NttTable *__restrict__ tabs, const int *__restrict__ modmap, int period, int ahead)
{
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const size_t g = blockIdx.x;
    const NttTable tab = tabs[modmap[g % (size_t)period]];
    u64 *p = data + g * N;
    const void *next = nullptr;
    if (NP >= 0 && g + (size_t)ahead < gridDim.x) next = reinterpret_cast<const char *>(data + (g + (size_t)ahead) * N) + (size_t)threadIdx.x * 128;
    if (tab.narrow) ntt_body<LOGN, INV, NTT_NARROW, T, 0, RAW, SrcPlain, true, false, 0, NP>(lds, p, tab, threadIdx.x, nullptr, SrcPlain(), next);
    else if (tab.wide_d4) ntt_body<LOGN, INV, NTT_WIDE_NEAR, T, 0, RAW, SrcPlain, true, false, 0, NP>(lds, p, tab, threadIdx.x, nullptr, SrcPlain(), next);
    else ntt_body<LOGN, INV, NTT_WIDE, T, 0, RAW, SrcPlain, true, false, 0, NP>(lds, p, tab, threadIdx.x, nullptr, SrcPlain(), next);
}

// the two workgroups a CU holds start together and, doing equal work, stay in lockstep: their global-memory phases coincide.
// Variant: the second resident workgroup of every CU (first generation only: blocks 256 .. 511) sleeps `shift` x 64 cycles once.
template <bool INV, bool RAW>
__global__ __launch_bounds__(T, 4) void k_shift(u64 *__restrict__ data, const NttTable *__restrict__ tabs, const int *__restrict__ modmap, int period, int shift)
{
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const size_t g = blockIdx.x;
    if (g >= 256 && g < 512) for (int i = 0; i < shift; i += 100) __builtin_amdgcn_s_sleep(100);
    const NttTable tab = tabs[modmap[g % (size_t)period]];
    u64 *p = data + g * N;
    if (tab.narrow) ntt_body<LOGN, INV, NTT_NARROW, T, 0, RAW>(lds, p, tab, threadIdx.x);
    else if (tab.wide_d4) ntt_body<LOGN, INV, NTT_WIDE_NEAR, T, 0, RAW>(lds, p, tab, threadIdx.x);
    else ntt_body<LOGN, INV, NTT_WIDE, T, 0, RAW>(lds, p, tab, threadIdx.x);
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
    template <class F> double us(F f, int reps)
    {
        f();
        CHECK(hipDeviceSynchronize());
        double sum = 0;
        for (int i = 0; i < reps; i++) {
            CHECK(hipEventRecord(a));
            f();
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            sum += ms;
        }
        return sum / reps * 1e3;
    }
};

int main(int argc, char **argv)
{
    const size_t big = argc > 1 ? (size_t)atol(argv[1]) : 16380;
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
    }
    NttTable *d_tabs;
    CHECK(hipMalloc(&d_tabs, nmod * sizeof(NttTable)));
    CHECK(hipMemcpy(d_tabs, tabs.data(), nmod * sizeof(NttTable), hipMemcpyHostToDevice));
    const std::vector<int> map_q = { 0, 1, 2 }, map_ext = { 0, 1, 2, K + 2, K + 3, K + 4, K + 0 };
    auto up = [&](const std::vector<int> &m) { int *d; CHECK(hipMalloc(&d, m.size() * sizeof(int))); CHECK(hipMemcpy(d, m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice)); return d; };
    int *d_map_q = up(map_q), *d_map_ext = up(map_ext);

    const size_t words = big * N;
    std::vector<u64> host(words);
    std::mt19937_64 rng(0x41505355);
    for (size_t i = 0; i < words; i++) host[i] = rng() % kq[0];
    u64 *d_in, *d_a, *d_b;
    CHECK(hipMalloc(&d_in, words * 8)); CHECK(hipMalloc(&d_a, words * 8)); CHECK(hipMalloc(&d_b, words * 8));
    CHECK(hipMemcpy(d_in, host.data(), words * 8, hipMemcpyHostToDevice));
    std::vector<u64> ra(words), rb(words);
    auto same = [&](const char *what) {
        CHECK(hipMemcpy(ra.data(), d_a, words * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(rb.data(), d_b, words * 8, hipMemcpyDeviceToHost));
        const bool ok = std::memcmp(ra.data(), rb.data(), words * 8) == 0;
        printf("%-58s %s\n", what, ok ? "same bits" : "MISMATCH");
        if (!ok) exit(2);
    };
    auto copy_in = [&](u64 *dst) { CHECK(hipMemcpy(dst, d_in, words * 8, hipMemcpyDeviceToDevice)); };
    const dim3 gb((unsigned)big);
#define SAME3(INV, RAW, NAME) \
    copy_in(d_a); hipLaunchKernelGGL((k_plain<INV, RAW, -1>), gb, dim3(T), 0, 0, d_a, d_tabs, d_map_ext, 7, 0); \
    copy_in(d_b); hipLaunchKernelGGL((k_plain<INV, RAW, 0>), gb, dim3(T), 0, 0, d_b, d_tabs, d_map_ext, 7, 512); same(NAME ", touch in front of pass 0"); \
    copy_in(d_b); hipLaunchKernelGGL((k_plain<INV, RAW, 2>), gb, dim3(T), 0, 0, d_b, d_tabs, d_map_ext, 7, 512); same(NAME ", touch in front of pass 2"); \
    copy_in(d_b); hipLaunchKernelGGL((k_plain<INV, RAW, 3>), gb, dim3(T), 0, 0, d_b, d_tabs, d_map_ext, 7, 300); same(NAME ", touch in front of pass 3");
    SAME3(false, false, "forward")
    SAME3(true, false, "inverse")
    SAME3(true, true, "inverse RAW")

    Timer tm;
    const int reps = 9;
    const int aheads[] = { 256, 512, 1024 };
#define ROW(NAME, INV, RAW, MAP, PER) do { \
        const double by = (double)count * 16 * N; \
        const double t0 = tm.us([&] { hipLaunchKernelGGL((k_plain<INV, RAW, -1>), g, dim3(T), 0, 0, d_a, d_tabs, MAP, PER, 0); }, reps); \
        printf("%-26s %6zu limbs  off %7.1f us %5.0f GB/s |", NAME, count, t0, by / t0 / 1e3); \
        for (int ah : aheads) { \
            const double t1 = tm.us([&] { hipLaunchKernelGGL((k_plain<INV, RAW, 0>), g, dim3(T), 0, 0, d_a, d_tabs, MAP, PER, ah); }, reps); \
            const double t2 = tm.us([&] { hipLaunchKernelGGL((k_plain<INV, RAW, 2>), g, dim3(T), 0, 0, d_a, d_tabs, MAP, PER, ah); }, reps); \
            const double t3 = tm.us([&] { hipLaunchKernelGGL((k_plain<INV, RAW, 3>), g, dim3(T), 0, 0, d_a, d_tabs, MAP, PER, ah); }, reps); \
            printf(" +%d: p0 %+5.1f %% p2 %+5.1f %% p3 %+5.1f %% |", ah, (t1 / t0 - 1) * 100, (t2 / t0 - 1) * 100, (t3 / t0 - 1) * 100); \
        } \
        const double t9 = tm.us([&] { hipLaunchKernelGGL((k_plain<INV, RAW, -1>), g, dim3(T), 0, 0, d_a, d_tabs, MAP, PER, 0); }, reps); \
        printf(" off again %+5.1f %%\n", (t9 / t0 - 1) * 100); \
    } while (0)
    const size_t sizes[] = { big, 6840, 3825, 1872, 1248, 624 };
    for (size_t count : sizes) {
        if (count > big) continue;
        const dim3 g((unsigned)count);
        printf("\n");
        ROW("forward, data primes", false, false, d_map_q, 3);
        ROW("forward, extended base", false, false, d_map_ext, 7);
        ROW("inverse, data primes", true, false, d_map_q, 3);
        ROW("inverse RAW, data primes", true, true, d_map_q, 3);
        ROW("inverse, extended base", true, false, d_map_ext, 7);
    }
    printf("\nphase-shifted first generation (sleep of blocks 256..511, x 64 cycles)\n");
    const int shifts[] = { 0, 100, 200, 300, 400 };
#define SROW(NAME, INV, RAW, MAP, PER) do { \
        const double by = (double)count * 16 * N; \
        printf("%-26s %6zu limbs |", NAME, count); \
        double t0 = 0; \
        for (int sh : shifts) { \
            const double t1 = tm.us([&] { hipLaunchKernelGGL((k_shift<INV, RAW>), g, dim3(T), 0, 0, d_a, d_tabs, MAP, PER, sh); }, reps); \
            if (!sh) t0 = t1; \
            printf(" shift %3d: %7.1f us %5.0f GB/s %+5.1f %% |", sh, t1, by / t1 / 1e3, (t1 / t0 - 1) * 100); \
        } \
        printf("\n"); \
    } while (0)
    for (size_t count : sizes) {
        if (count > big || count < 1000) continue;
        const dim3 g((unsigned)count);
        SROW("forward, data primes", false, false, d_map_q, 3);
        SROW("forward, extended base", false, false, d_map_ext, 7);
        SROW("inverse RAW, data primes", true, true, d_map_q, 3);
        SROW("inverse, extended base", true, false, d_map_ext, 7);
    }
    printf("\nok\n");
    return 0;
}
