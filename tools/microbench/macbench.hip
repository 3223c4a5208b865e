// Stand-alone timing of the engine's k_mac on a synthetic DB, to bisect its HBM efficiency (see readbw.hip for the ceilings).
#include "../../apsu_amd/csrc/kernels.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
namespace apsu_he { void throw_hip(hipError_t e, const char* f, int l) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), f, l); abort(); } }
using namespace apsu_he;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_fillrand(u64* p, size_t words, u64 mask) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
        u64 z = i * 0x9e3779b97f4a7c15ULL + 0x1234; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; p[i] = (z ^ (z >> 31)) & mask; }
}
int main(int argc, char** argv) {
    const size_t n = 8192, L = 3, ptw = L * n;
    const int terms = 44, streams = 784, nb = argc > 1 ? atoi(argv[1]) : 4;       // nb: bundle indices interleaved in the powers layout
    const size_t words = (size_t)streams * terms * ptw;
    u64 *db, *pw, *out; DevLevel* lv; MacJob* dj;
    CHECK(hipMalloc(&db, words * 8)); CHECK(hipMalloc(&pw, (size_t)terms * nb * 2 * L * n * 8)); CHECK(hipMalloc(&out, (size_t)streams * 2 * L * n * 8));
    k_fillrand<<<4096, 256>>>(db, words, ((u64)1 << 55) - 1); k_fillrand<<<1024, 256>>>(pw, (size_t)terms * nb * 2 * L * n, ((u64)1 << 55) - 1);
    DevLevel h; memset(&h, 0, sizeof(h)); h.L = 3;
    u64 q[3] = { 0xfffffffff70001ULL, 0xfffffffff78001ULL, 0xfffffffffb4001ULL };
    for (int j = 0; j < 3; j++) { unsigned __int128 all = ~(unsigned __int128)0; unsigned __int128 r = all / q[j]; h.q[j] = Mod{ q[j], (u64)r, (u64)(r >> 64) }; h.mac_shift[j] = 28; h.mac_chunk[j] = 127; }
    CHECK(hipMalloc(&lv, sizeof(h))); CHECK(hipMemcpy(lv, &h, sizeof(h), hipMemcpyHostToDevice));
    std::vector<MacJob> jobs;
    for (int s = 0; s < streams; s += MAC_G) {
        MacJob j{}; j.pw = pw; j.cnt = terms; j.ng = MAC_G; j.pt_stride = ptw; j.pw_stride = nb * 2 * L * n; j.pw_poly_stride = L * n; j.out_poly_stride = L * n; j.limb0 = 0;
        for (int g = 0; g < MAC_G; g++) { j.pt[g] = db + (size_t)(s + g) * terms * ptw; j.out[g] = out + (size_t)(s + g) * 2 * L * n; }
        jobs.push_back(j);
    }
    CHECK(hipMalloc(&dj, jobs.size() * sizeof(MacJob))); CHECK(hipMemcpy(dj, jobs.data(), jobs.size() * sizeof(MacJob), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0)); launch_mac(lv, 3, dj, n, (int)jobs.size(), 0); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("k_mac<%d,%d> nb=%d: %.3f ms  %.0f GB/s (DB bytes)\n", APSU_MAC_G, APSU_MAC_C, nb, ms, words * 8 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
