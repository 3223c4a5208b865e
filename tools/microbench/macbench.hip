#define APSU_MAC_TILED_EXPERIMENT 1
// Stand-alone timing of the engine's k_mac on a synthetic DB, to bisect its HBM efficiency (see readbw.hip for the ceilings).
#include "../../apsu_amd/csrc/kernels.hip"
#ifndef MACBENCH_PLAIN      // -DMACBENCH_PLAIN: k_mac only (the ring / persistent variants are written for four streams per job)
#include "mac_ring.hip"
#include "mac_persist.hip"
#endif
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
namespace apsu_he { void throw_hip(hipError_t e, const char* f, int l) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), f, l); abort(); } }
using namespace apsu_he;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_fillrand(u64* p, size_t words, u64 mask) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
        u64 z = i * 0x9e3779b97f4a7c15ULL + 0x1234; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; p[i] = (z ^ (z >> 31)) & mask; }
}

// Round 6, PLACEMENT mode with PROBE=1: what does ADDRESS TRANSLATION cost on this copy of the database?  65 536 lanes each read one 8-byte word
// from a different page per iteration (page = `stride` bytes; the next page a large odd step further), the addresses independent of the data:
// nothing is reused, every access is a new line and, at 4 KiB ... 2 MiB strides, a new page -- the time per iteration is the translation
// path's (UTCL1 miss -> UTCL2 -> page walk), which depends on how large the fragments are the driver mapped this copy with.
__global__ __launch_bounds__(256) void k_tlb_probe(const u64 *__restrict__ base, size_t npages, size_t stride_words, int iters, u64 *__restrict__ sink)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t p = (gid * 7919) % npages;
    u64 acc = 0;
    for (int i = 0; i < iters; i++) {
        acc += __builtin_nontemporal_load(base + p * stride_words + (gid & 7));
        p += 104729; if (p >= npages) p -= npages; if (p >= npages) p %= npages;
    }
    if (acc == 0x123456789abcdefULL) sink[0] = acc;
}

// ---- bisect kernels: same loads as k_mac<2,2>, different bodies
template <int MODE>
__global__ __launch_bounds__(256) void k_morph(const MacJob *__restrict__ jobs, size_t n, u64 *sink)
{
    const size_t k = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const MacJob *__restrict__ jp = jobs + blockIdx.z / 2;
    const int g0 = (blockIdx.z % 2) * 2;
    const int j = blockIdx.y;
    const u64 *p0 = jp->pw + (size_t)j * n + k, *p1 = p0 + jp->pw_poly_stride;
    const u64 *a0 = jp->pt[g0] + (size_t)j * n + k, *a1 = jp->pt[g0 + 1] + (size_t)j * n + k;
    const u32 cnt = jp->cnt, pts = jp->pt_stride, pws = jp->pw_stride;
    u64 acc = 0;
    if (MODE == 0) {            // simple loop, compiler-scheduled
        for (u32 i = 0; i < cnt; i++) {
            const u64x2 c0 = ldg16(p0 + (size_t)i * pws), c1 = ldg16(p1 + (size_t)i * pws);
            const u64x2 x = ldg16_nt(a0 + (size_t)i * pts), y = ldg16_nt(a1 + (size_t)i * pts);
            acc += (x[0] * c0[0]) ^ (x[1] * c1[1]) ^ (y[0] * c0[1]) ^ (y[1] * c1[0]);
        }
    } else if (MODE == 1) {     // unroll 4
#pragma unroll 4
        for (u32 i = 0; i < cnt; i++) {
            const u64x2 c0 = ldg16(p0 + (size_t)i * pws), c1 = ldg16(p1 + (size_t)i * pws);
            const u64x2 x = ldg16_nt(a0 + (size_t)i * pts), y = ldg16_nt(a1 + (size_t)i * pts);
            acc += (x[0] * c0[0]) ^ (x[1] * c1[1]) ^ (y[0] * c0[1]) ^ (y[1] * c1[0]);
        }
    } else if (MODE == 2) {     // no power loads at all
#pragma unroll 4
        for (u32 i = 0; i < cnt; i++) {
            const u64x2 x = ldg16_nt(a0 + (size_t)i * pts), y = ldg16_nt(a1 + (size_t)i * pts);
            acc += (x[0] * 3) ^ (x[1] * 5) ^ (y[0] * 7) ^ (y[1] * 11);
        }
    } else if (MODE == 3) {     // 4 streams per block (G = 4): half the power loads per DB byte
        const u64 *a2 = jp->pt[(g0 + 2) & 3] + (size_t)j * n + k, *a3 = jp->pt[(g0 + 3) & 3] + (size_t)j * n + k;
        if (g0) return;
#pragma unroll 2
        for (u32 i = 0; i < cnt; i++) {
            const u64x2 c0 = ldg16(p0 + (size_t)i * pws), c1 = ldg16(p1 + (size_t)i * pws);
            const u64x2 x = ldg16_nt(a0 + (size_t)i * pts), y = ldg16_nt(a1 + (size_t)i * pts);
            const u64x2 z = ldg16_nt(a2 + (size_t)i * pts), w = ldg16_nt(a3 + (size_t)i * pts);
            acc += (x[0] * c0[0]) ^ (x[1] * c1[1]) ^ (y[0] * c0[1]) ^ (y[1] * c1[0]) ^ (z[0] * c0[0]) ^ (z[1] * c1[1]) ^ (w[0] * c0[1]) ^ (w[1] * c1[0]);
        }
    }
    if (acc == 0x1234567) sink[0] = acc;
}
int main(int argc, char** argv) {
    const size_t n = 8192, L = 3, ptw = L * n;
    const int terms = getenv("TERMS") ? atoi(getenv("TERMS")) : 44, streams = 784 - 784 % MAC_G, nb = argc > 1 ? atoi(argv[1]) : 4; const size_t pad = argc > 2 ? atoi(argv[2]) : 0, ppad = argc > 3 ? atoi(argv[3]) : 0;   // pad: words added to the power stride; ppad: to the poly stride       // nb: bundle indices interleaved in the powers layout
    const size_t skew = getenv("SKEW") ? atoi(getenv("SKEW")) : 0;   // extra words between consecutive streams (breaks the 8.25 MB stride)
    const size_t words = (size_t)streams * terms * ptw;
    u64 *db, *pw, *out; DevLevel* lv; MacJob* dj;
    CHECK(hipMalloc(&db, (words + (size_t)streams * skew) * 8)); const size_t pstride = nb * 2 * L * n + pad + 2 * ppad; CHECK(hipMalloc(&pw, (size_t)terms * pstride * 8)); CHECK(hipMalloc(&out, (size_t)streams * 2 * L * n * 8));
    k_fillrand<<<4096, 256>>>(db, words + (size_t)streams * skew, ((u64)1 << 55) - 1); k_fillrand<<<1024, 256>>>(pw, (size_t)terms * pstride, ((u64)1 << 55) - 1);
    DevLevel h; memset(&h, 0, sizeof(h)); h.L = 3;
    u64 q[3] = { 0xfffffffff70001ULL, 0xfffffffff78001ULL, 0xfffffffffb4001ULL };
    for (int j = 0; j < 3; j++) { unsigned __int128 all = ~(unsigned __int128)0; unsigned __int128 r = all / q[j]; h.q[j] = Mod{ q[j], (u64)r, (u64)(r >> 64) }; h.mac_shift[j] = 28; h.mac_chunk[j] = 127; h.mac_chunk_k[j] = 63; }
    CHECK(hipMalloc(&lv, sizeof(h))); CHECK(hipMemcpy(lv, &h, sizeof(h), hipMemcpyHostToDevice));
    std::vector<MacJob> jobs;
    for (int s = 0; s < streams; s += MAC_G) {
        MacJob j{}; j.pw = pw; j.cnt = terms; j.ng = MAC_G; j.pt_stride = ptw; j.pw_stride = pstride; j.pw_poly_stride = L * n + ppad; j.nl = 3; j.out_poly_stride = L * n; j.limb0 = 0;
        for (int g = 0; g < MAC_G; g++) { j.pt[g] = db + (size_t)(s + g) * (terms * ptw + skew); j.out[g] = out + (size_t)(s + g) * 2 * L * n; }
        jobs.push_back(j);
    }
    CHECK(hipMalloc(&dj, jobs.size() * sizeof(MacJob))); CHECK(hipMemcpy(dj, jobs.data(), jobs.size() * sizeof(MacJob), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    {
        std::vector<float> t;
        for (int rep = 0; rep < 16; rep++) {
            CHECK(hipEventRecord(e0)); launch_mac(lv, 3, dj, n, (int)jobs.size(), 0); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("k_mac<%d,%d> ring=%d nb=%d pad=%zu: min %.3f ms (%.0f GB/s)  median %.3f ms (%.0f GB/s)  max %.3f\n", APSU_MAC_G, APSU_MAC_C, 2, nb, pad,
               t[0], words * 8 / (t[0] * 1e-3) / 1e9, t[t.size() / 2], words * 8 / (t[t.size() / 2] * 1e-3) / 1e9, t.back());
    }
#ifdef APSU_MAC_STAMPS
    if (getenv("STAMPS")) {
        // build with -DAPSU_MAC_STAMPS: the per-workgroup timeline of k_mac (in-kernel shader-clock stamps of lane 0)
        const size_t nwg = (size_t)16 * 3 * jobs.size();
        unsigned long long *dst; CHECK(hipMalloc(&dst, nwg * 8 * sizeof(unsigned long long))); CHECK(hipMemset(dst, 0, nwg * 8 * sizeof(unsigned long long)));
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(apsu_he::g_mac_stamps), &dst, sizeof(dst)));
        for (int rep = 0; rep < 3; rep++) { launch_mac(lv, 3, dj, n, (int)jobs.size(), 0); CHECK(hipDeviceSynchronize()); }
        std::vector<unsigned long long> h(nwg * 8);
        CHECK(hipMemcpy(h.data(), dst, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long *nul = nullptr; CHECK(hipMemcpyToSymbol(HIP_SYMBOL(apsu_he::g_mac_stamps), &nul, sizeof(nul)));
        const char *names[5] = { "setup (descriptor, pointers)", "first term arrives + consumed", "steady loop", "fold", "store + drain" };
        std::vector<double> ph[5], life, start;
        unsigned long long t0 = ~0ull;
        for (size_t w = 0; w < nwg; w++) if (h[w * 8 + 6] && h[w * 8 + 6] < t0) t0 = h[w * 8 + 6];
        for (size_t w = 0; w < nwg; w++) {
            const unsigned long long *s2 = &h[w * 8];
            if (!s2[5]) continue;
            for (int i = 0; i < 5; i++) ph[i].push_back((double)(s2[i + 1] - s2[i]));
            life.push_back((double)(s2[7] - s2[6]) / 100.0);     // us (100 MHz)
            start.push_back((double)(s2[6] - t0) / 100.0);
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
        auto p90 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() * 9 / 10]; };
        double tot = 0; for (int i = 0; i < 5; i++) tot += med(ph[i]);
        printf("workgroups stamped: %zu of %zu; median lifetime %.1f us (p90 %.1f); last start at %.1f us\n", life.size(), nwg, med(life), p90(life), *std::max_element(start.begin(), start.end()));
        for (int i = 0; i < 5; i++) printf("  %-32s median %9.0f cycles (%4.1f %%)   p90 %9.0f\n", names[i], med(ph[i]), 100 * med(ph[i]) / tot, p90(ph[i]));
        std::vector<int> hist(100, 0);
        for (double st2 : start) { size_t b = (size_t)(st2 / 20.0); if (b < hist.size()) hist[b]++; }
        printf("  starts per 20 us:"); for (size_t b = 0; b < hist.size() && b < 80; b++) printf(" %d", hist[b]); printf("\n");
    }
#endif
    // (the grid-order comparison that lived here -- ORDER=1: orders 0 / 1 / 2 with real powers and with one term's powers -- is recorded in
    //  profiles/r04_mac_grid_order.txt; the library keeps order 1 and launch_mac no longer takes an order)
    if (getenv("PLACEMENT")) {
        // Does the scan's speed depend on WHERE the database lies?  (Identical binaries run one after the other alternate between 1.12 and
        // 1.26 ms, profiles/r05_mac_ring3.txt.)  Several copies of the same database in ONE process, the same kernel on each in turn;
        // SKEW_LIST: the same with the streams of one copy moved apart by extra words (relative alignment of the concurrently read rows).
        const int copies = atoi(getenv("PLACEMENT"));
        std::vector<u64 *> dbs{ db };
        for (int c = 1; c < copies; c++) { u64 *d2; CHECK(hipMalloc(&d2, (words + (size_t)streams * 8192) * 8)); k_fillrand<<<4096, 256>>>(d2, words + (size_t)streams * 8192, ((u64)1 << 55) - 1); dbs.push_back(d2); }
        std::vector<size_t> skews{ 0 };
        if (getenv("SKEW_LIST")) { skews.clear(); char *t = strdup(getenv("SKEW_LIST")); for (char *q2 = strtok(t, ","); q2; q2 = strtok(nullptr, ",")) skews.push_back((size_t)atoll(q2)); }
        for (int pass = 0; pass < 3; pass++)
            for (size_t c = 0; c < dbs.size(); c++)
                for (size_t sk : skews) {
                    if (c == 0 && sk > skew) continue;              // the first copy was allocated without room for a skew
                    std::vector<MacJob> jc = jobs;
                    for (size_t x = 0; x < jc.size(); x++) for (int g = 0; g < MAC_G; g++) jc[x].pt[g] = dbs[c] + (x * MAC_G + g) * (terms * ptw + sk);
                    MacJob *djc; CHECK(hipMalloc(&djc, jc.size() * sizeof(MacJob))); CHECK(hipMemcpy(djc, jc.data(), jc.size() * sizeof(MacJob), hipMemcpyHostToDevice));
                    std::vector<float> t;
                    for (int rep = 0; rep < 8; rep++) {
                        CHECK(hipEventRecord(e0)); launch_mac(lv, 3, djc, n, (int)jc.size(), 0); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) t.push_back(ms);
                    }
                    std::sort(t.begin(), t.end());
                    printf("pass %d  copy %zu at %p  skew %5zu words: median %.3f ms (%.0f GB/s)\n", pass, c, (void *)dbs[c], sk, t[t.size() / 2], words * 8 / (t[t.size() / 2] * 1e-3) / 1e9);
                    if (getenv("ROTATE")) {                          // the same copy, every workgroup starting at another term (MacJob::pad = 3)
                        for (auto &x : jc) x.pad = 3;
                        CHECK(hipMemcpy(djc, jc.data(), jc.size() * sizeof(MacJob), hipMemcpyHostToDevice));
                        std::vector<float> tr;
                        for (int rep = 0; rep < 8; rep++) {
                            CHECK(hipEventRecord(e0)); launch_mac(lv, 3, djc, n, (int)jc.size(), 0); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) tr.push_back(ms);
                        }
                        std::sort(tr.begin(), tr.end());
                        printf("        rotated start terms on copy %zu: median %.3f ms (%.0f GB/s)  %+.1f %%\n", c, tr[tr.size() / 2], words * 8 / (tr[tr.size() / 2] * 1e-3) / 1e9,
                               100.0 * (tr[tr.size() / 2] / t[t.size() / 2] - 1));
                    }
                    CHECK(hipFree(djc));
                    if (getenv("PROBE") && pass == 2 && sk == 0) {
                        for (size_t stride : { (size_t)4096, (size_t)65536, (size_t)(2u << 20) }) {
                            const size_t npages = words * 8 / stride;
                            std::vector<float> tp;
                            for (int rep = 0; rep < 7; rep++) {
                                CHECK(hipEventRecord(e0)); k_tlb_probe<<<256, 256>>>(dbs[c], npages, stride / 8, 64, out); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) tp.push_back(ms);
                            }
                            std::sort(tp.begin(), tp.end());
                            printf("        probe copy %zu: one word per %7zu-byte page, 65536 lanes x 64 pages: median %.1f us (%.1f ns per wave-access)\n", c, stride, tp[tp.size() / 2] * 1e3,
                                   tp[tp.size() / 2] * 1e6 / 64.0);
                        }
                    }
                }
    }
    if (getenv("TILED")) {
        // Is the scan bound by bytes or by the NUMBER of separate pieces it reads?  Row layout (the engine's): a workgroup's G streams
        // are G pieces of 4 KiB (dense) / 3.5 KiB (56-bit packed) per term, each in another stream's slot.  Tiled layout: the G pieces
        // of one (term, limb, block) adjacent -- one piece of 16 / 14 KiB per workgroup and term.  Timing only (same bytes, other sums).
        DevLevel hp = h;
        const u32 kbits = getenv("KBITS") ? atoi(getenv("KBITS")) : 56;
        for (int j = 0; j < 3; j++) { hp.mac_bits[j] = kbits; hp.mac_mask_hi[j] = (1u << (kbits - 28)) - 1; hp.mac_row_off[j] = (u32)(j * n * kbits / 8); }
        DevLevel *lvp; CHECK(hipMalloc(&lvp, sizeof(hp))); CHECK(hipMemcpy(lvp, &hp, sizeof(hp), hipMemcpyHostToDevice));
        const size_t slot_b = 3 * n * kbits / 8;
        std::vector<MacJob> jd = jobs, jdt = jobs, jp = jobs, jpt = jobs, jpb = jobs;
        for (size_t x = 0; x < jobs.size(); x++) {
            const size_t s0 = x * MAC_G;
            jdt[x].pad = 1; jdt[x].pt_stride = (u32)(ptw * MAC_G); jdt[x].pt[0] = db + s0 * terms * ptw;
            for (int g = 0; g < MAC_G; g++) jp[x].pt[g] = reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(db) + (s0 + g) * terms * slot_b);
            jp[x].packed = 1; jp[x].pt_stride = (u32)slot_b;
            jpb[x] = jp[x]; jpb[x].pad = 2; jpb[x].pt_stride = (u32)(256 * 2 * kbits / 8);      // block-major: a term = the next 3.5 KiB tile
            jpt[x] = jp[x]; jpt[x].pad = 1; jpt[x].pt_stride = (u32)(slot_b * MAC_G); jpt[x].pt[0] = reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(db) + s0 * terms * slot_b);
        }
        auto upj = [&](const std::vector<MacJob> &v) { MacJob *d; if (hipMalloc(&d, v.size() * sizeof(MacJob)) != hipSuccess) abort(); if (hipMemcpy(d, v.data(), v.size() * sizeof(MacJob), hipMemcpyHostToDevice) != hipSuccess) abort(); return d; };
        MacJob *d_jd = upj(jd), *d_jdt = upj(jdt), *d_jp = upj(jp), *d_jpt = upj(jpt), *d_jpb = upj(jpb);
        const double coefs = (double)streams * terms * ptw;
        for (int pass = 0; pass < 2; pass++)
            for (int v = 0; v < 5; v++) {
                const bool packed = v >= 2; MacJob *dv = v == 0 ? d_jd : v == 1 ? d_jdt : v == 2 ? d_jp : v == 3 ? d_jpt : d_jpb;
                std::vector<float> t;
                for (int rep = 0; rep < 10; rep++) {
                    CHECK(hipEventRecord(e0)); launch_mac(packed ? lvp : lv, 3, dv, n, (int)jobs.size(), 0, false, packed); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) t.push_back(ms);
                }
                std::sort(t.begin(), t.end());
                const double by = coefs * (packed ? kbits / 8.0 : 8.0), med = t[t.size() / 2];
                printf("terms %d  %-22s %s: median %.3f ms  %.0f GB/s  %.3f Tcoef/s\n", terms, packed ? (kbits == 56 ? "packed 56 bits" : "packed") : "dense", v == 4 ? "block-major" : (v & 1) ? "tiled" : "rows ", med,
                       by / (med * 1e-3) / 1e9, coefs / (med * 1e-3) / 1e12);
            }
    }
#ifndef MACBENCH_PLAIN
    if (getenv("PERSIST")) {
        // long-lived workgroups (k_mac_p) against one workgroup per unit (k_mac): separate output, bit-compared; A B A B timing
        u64 *out2; CHECK(hipMalloc(&out2, (size_t)streams * 2 * L * n * 8));
        std::vector<MacJob> jobs2 = jobs;
        for (size_t x = 0; x < jobs2.size(); x++) for (int g = 0; g < MAC_G; g++) jobs2[x].out[g] = out2 + (jobs[x].out[g] - out);
        MacJob *dj2; CHECK(hipMalloc(&dj2, jobs2.size() * sizeof(MacJob))); CHECK(hipMemcpy(dj2, jobs2.data(), jobs2.size() * sizeof(MacJob), hipMemcpyHostToDevice));
        const bool kara = getenv("KARA2") != nullptr;
        for (int r : { 0, 16, 24, 32, 48, 64, 96 }) {
            std::vector<float> ta, tb;
            for (int rep = 0; rep < 14; rep++) {
                float ms;
                CHECK(hipEventRecord(e0)); launch_mac(lv, 3, dj, n, (int)jobs.size(), 0, kara); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) ta.push_back(ms);
                CHECK(hipEventRecord(e0)); launch_mac_persist(lv, dj2, n, (int)jobs2.size(), 0, kara, r); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) tb.push_back(ms);
            }
            std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
            const size_t ow = (size_t)streams * 2 * L * n;
            std::vector<u64> h1(ow), h2(ow);
            CHECK(hipMemcpy(h1.data(), out, ow * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2.data(), out2, ow * 8, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t x = 0; x < ow; x++) bad += h1[x] != h2[x];
            printf("persistent R=%2d: median %.3f ms (%.0f GB/s)   per unit: %.3f ms (%.0f GB/s)   %+.1f %%   %zu words differ\n", r, tb[tb.size() / 2],
                   words * 8 / (tb[tb.size() / 2] * 1e-3) / 1e9, ta[ta.size() / 2], words * 8 / (ta[ta.size() / 2] * 1e-3) / 1e9, (tb[tb.size() / 2] / ta[ta.size() / 2] - 1) * 100, bad);
            CHECK(hipMemset(out2, 0, ow * 8));
        }
    }
#endif
    if (getenv("B2B")) {
        // is part of a launch's time a fixed cost per launch?  N launches queued back to back (no host wait in between) against N x one launch
        for (int nl : { 1, 2, 4 }) {
            std::vector<float> t;
            for (int rep = 0; rep < 8; rep++) {
                CHECK(hipEventRecord(e0));
                for (int k = 0; k < nl; k++) launch_mac(lv, 3, dj, n, (int)jobs.size(), 0);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            printf("%d launches back to back: median %.3f ms = %.3f ms per launch (%.0f GB/s)\n", nl, t[t.size() / 2], t[t.size() / 2] / nl, words * 8 / (t[t.size() / 2] / nl * 1e-3) / 1e9);
        }
    }
    if (getenv("KARA")) {
        // three-product accumulation: same jobs, separate output, bit-compared with the four-product form; A B A B timing
        u64 *out2; CHECK(hipMalloc(&out2, (size_t)streams * 2 * L * n * 8));
        std::vector<MacJob> jobs2 = jobs;
        for (size_t x = 0; x < jobs2.size(); x++) for (int g = 0; g < MAC_G; g++) jobs2[x].out[g] = out2 + (jobs[x].out[g] - out);
        MacJob *dj2; CHECK(hipMalloc(&dj2, jobs2.size() * sizeof(MacJob))); CHECK(hipMemcpy(dj2, jobs2.data(), jobs2.size() * sizeof(MacJob), hipMemcpyHostToDevice));
        std::vector<float> ta, tb;
        for (int rep = 0; rep < 24; rep++) {
            float ms;
            CHECK(hipEventRecord(e0)); launch_mac(lv, 3, dj, n, (int)jobs.size(), 0, false); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 4) ta.push_back(ms);
            CHECK(hipEventRecord(e0)); launch_mac(lv, 3, dj2, n, (int)jobs2.size(), 0, true); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 4) tb.push_back(ms);
        }
        std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
        printf("four products : min %.3f ms (%.0f GB/s)  median %.3f ms (%.0f GB/s)\n", ta[0], words * 8 / (ta[0] * 1e-3) / 1e9, ta[ta.size() / 2], words * 8 / (ta[ta.size() / 2] * 1e-3) / 1e9);
        printf("three products: min %.3f ms (%.0f GB/s)  median %.3f ms (%.0f GB/s)\n", tb[0], words * 8 / (tb[0] * 1e-3) / 1e9, tb[tb.size() / 2], words * 8 / (tb[tb.size() / 2] * 1e-3) / 1e9);
        const size_t ow = (size_t)streams * 2 * L * n;
        std::vector<u64> h1(ow), h2(ow);
        CHECK(hipMemcpy(h1.data(), out, ow * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2.data(), out2, ow * 8, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t x = 0; x < ow; x++) bad += h1[x] != h2[x];
        printf("three vs four products: %zu of %zu output words differ\n", bad, ow);
        if (getenv("TERMS")) return 0;
    }
#ifndef MACBENCH_PLAIN
    if (getenv("RING")) {
        // LDS-DMA ring variant: same jobs, separate output, bit-compared with k_mac's
        u64 *out2; CHECK(hipMalloc(&out2, (size_t)streams * 2 * L * n * 8));
        std::vector<MacJob> jobs2 = jobs;
        for (size_t x = 0; x < jobs2.size(); x++) for (int g = 0; g < MAC_G; g++) jobs2[x].out[g] = out2 + (jobs[x].out[g] - out);
        MacJob *dj2; CHECK(hipMalloc(&dj2, jobs2.size() * sizeof(MacJob))); CHECK(hipMemcpy(dj2, jobs2.data(), jobs2.size() * sizeof(MacJob), hipMemcpyHostToDevice));
        for (unsigned nt = 0; nt < 2; nt++) {
            std::vector<float> t;
            for (int rep = 0; rep < 16; rep++) {
                CHECK(hipEventRecord(e0)); launch_mac_ring(lv, 3, dj2, n, (int)jobs2.size(), 0, nt, getenv("RING_D") ? atoi(getenv("RING_D")) : 3, getenv("RING_LDS_KB") ? atoi(getenv("RING_LDS_KB")) * 1024 : 0); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            printf("k_mac_ring<%d,T=%d> nt=%u: min %.3f ms (%.0f GB/s)  median %.3f ms (%.0f GB/s)  max %.3f\n", MAC_G, APSU_MAC_RING_T, nt,
                   t[0], words * 8 / (t[0] * 1e-3) / 1e9, t[t.size() / 2], words * 8 / (t[t.size() / 2] * 1e-3) / 1e9, t.back());
        }
        const size_t ow = (size_t)streams * 2 * L * n;
        std::vector<u64> h1(ow), h2(ow);
        CHECK(hipMemcpy(h1.data(), out, ow * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2.data(), out2, ow * 8, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t x = 0; x < ow; x++) bad += h1[x] != h2[x];
        printf("ring vs k_mac: %zu of %zu output words differ\n", bad, ow);
        // once more k_mac, after the ring runs (same clocks / thermal state)
        std::vector<float> t;
        for (int rep = 0; rep < 12; rep++) { CHECK(hipEventRecord(e0)); launch_mac(lv, 3, dj, n, (int)jobs.size(), 0); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms); }
        std::sort(t.begin(), t.end());
        printf("k_mac again: min %.3f median %.3f\n", t[0], t[t.size() / 2]);
    }
#endif
    if (!getenv("MORPH") || MAC_G != 4) return 0;
    for (int rep = 0; rep < 2; rep++) {
#define MORPH(M, NAME) { CHECK(hipEventRecord(e0)); hipLaunchKernelGGL((k_morph<M>), dim3(16, 3, (unsigned)jobs.size() * 2), dim3(256), 0, 0, dj, n, out); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("%-40s %.3f ms  %.0f GB/s\n", NAME, ms, words * 8 / (ms * 1e-3) / 1e9); }
        MORPH(0, "morph simple loop")
        MORPH(1, "morph unroll 4")
        MORPH(2, "morph no power loads")
        MORPH(3, "morph G=4")
    }
    return 0;
}
