// Read-only HBM streaming ceiling on MI355X for the access shapes of k_mac.
// A: flat grid-stride 16-B loads over 6 GiB.  B: MAC-like: each block owns a 4 KiB column of a "plaintext" and walks
// `terms` plaintexts at a 192 KiB stride (the DB layout), 2 streams per block, with and without the nt hint.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_flat(const u64x2* __restrict__ p, size_t n16, u64* out) {
    u64 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { u64x2 v = p[i]; acc += v[0] ^ v[1]; }
    if (acc == 0x1234567) out[0] = acc;
}
template <bool NT, int UNROLL>
__global__ void k_maclike(const u64* __restrict__ db, size_t pt_words, int terms, int streams_per_job, u64* out) {
    // grid: x = 16 column blocks, y = 3 limbs, z = jobs ; block = 256 threads x 2 coefficients
    const size_t k = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t limb = blockIdx.y, n = 8192;
    u64 acc = 0;
    for (int s = 0; s < streams_per_job; s++) {
        const u64* base = db + ((size_t)(blockIdx.z * streams_per_job + s) * terms) * pt_words + limb * n + k;
        for (int i = 0; i < terms; i += UNROLL) {
            u64x2 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const u64x2* q = (const u64x2*)(base + (size_t)(i + u) * pt_words);
                v[u] = NT ? __builtin_nontemporal_load(q) : *q;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc += v[u][0] ^ v[u][1];
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}
// all streams of a job interleaved in the term loop (what k_mac does)
template <bool NT, int S>
__global__ void k_maclike_interleaved(const u64* __restrict__ db, size_t pt_words, int terms, u64* out) {
    const size_t k = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t limb = blockIdx.y, n = 8192;
    u64 acc = 0;
    const u64* base[S];
#pragma unroll
    for (int s = 0; s < S; s++) base[s] = db + ((size_t)(blockIdx.z * S + s) * terms) * pt_words + limb * n + k;
    for (int i = 0; i < terms; i++) {
        u64x2 v[S];
#pragma unroll
        for (int s = 0; s < S; s++) { const u64x2* q = (const u64x2*)(base[s] + (size_t)i * pt_words); v[s] = NT ? __builtin_nontemporal_load(q) : *q; }
#pragma unroll
        for (int s = 0; s < S; s++) acc += v[s][0] ^ v[s][1];
    }
    if (acc == 0x1234567) out[0] = acc;
}

// interleaved streams + two shared "power" loads per term from a small region (L2/MALL resident)
template <bool NT, int S>
__global__ void k_maclike_pw(const u64* __restrict__ db, const u64* __restrict__ pw, size_t pt_words, int terms, u64* out) {
    const size_t k = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t limb = blockIdx.y, n = 8192;
    u64 acc = 0;
    const u64* base[S];
#pragma unroll
    for (int s = 0; s < S; s++) base[s] = db + ((size_t)(blockIdx.z * S + s) * terms) * pt_words + limb * n + k;
    const u64* p0 = pw + limb * n + k;                           // power j: [2][3][n] at stride 6n
    for (int i = 0; i < terms; i++) {
        u64x2 v[S];
        const u64x2 c0 = *(const u64x2*)(p0 + (size_t)i * 6 * n), c1 = *(const u64x2*)(p0 + (size_t)i * 6 * n + 3 * n);
#pragma unroll
        for (int s = 0; s < S; s++) { const u64x2* q = (const u64x2*)(base[s] + (size_t)i * pt_words); v[s] = NT ? __builtin_nontemporal_load(q) : *q; }
#pragma unroll
        for (int s = 0; s < S; s++) acc += (v[s][0] * c0[0]) ^ (v[s][1] * c1[1]);
    }
    if (acc == 0x1234567) out[0] = acc;
}

__global__ void k_fillrand(u64* p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
        u64 z = i * 0x9e3779b97f4a7c15ULL + 0x1234; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; p[i] = (z ^ (z >> 31)) >> 8;
    }
}

int main() {
    const size_t n = 8192, L = 3, pt_words = L * n;            // 192 KiB plaintext
    const int terms = 44, streams = 784 * 1;                   // 784 inner polynomials x 44 terms = 6.3 GiB
    const size_t words = (size_t)streams * terms * pt_words;
    u64 *db, *out;
    CHECK(hipMalloc(&db, words * 8)); CHECK(hipMalloc(&out, 64));
    k_fillrand<<<4096, 256>>>(db, words); CHECK(hipDeviceSynchronize());   // random 56-bit data like the real DB
    u64* pw; CHECK(hipMalloc(&pw, (size_t)44 * 6 * n * 8)); CHECK(hipMemset(pw, 2, (size_t)44 * 6 * n * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto report = [&](const char* name, float ms) { printf("%-44s %.3f ms  %.0f GB/s\n", name, ms, words * 8 / (ms * 1e-3) / 1e9); };
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0)); k_flat<<<256 * 8, 256>>>((const u64x2*)db, words / 2, out); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report("flat grid-stride 16B (2048 blocks)", ms);
        CHECK(hipEventRecord(e0)); k_flat<<<256 * 32, 256>>>((const u64x2*)db, words / 2, out); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report("flat grid-stride 16B (8192 blocks)", ms);
#define RUN(NAME, KERN, GRIDZ, ...) CHECK(hipEventRecord(e0)); KERN<<<dim3(16, 3, GRIDZ), 256>>>(__VA_ARGS__); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report(NAME, ms);
        RUN("mac-like 1 stream/job, unroll 1", (k_maclike<false, 1>), streams, db, pt_words, terms, 1, out)
        RUN("mac-like 1 stream/job, unroll 4", (k_maclike<false, 4>), streams, db, pt_words, terms, 1, out)
        RUN("mac-like 1 stream/job, unroll 4, nt", (k_maclike<true, 4>), streams, db, pt_words, terms, 1, out)
        RUN("mac-like 4 streams interleaved", (k_maclike_interleaved<false, 4>), streams / 4, db, pt_words, terms, out)
        RUN("mac-like 4 streams interleaved, nt", (k_maclike_interleaved<true, 4>), streams / 4, db, pt_words, terms, out)
        RUN("mac-like 2 streams interleaved, nt", (k_maclike_interleaved<true, 2>), streams / 2, db, pt_words, terms, out)
        RUN("mac-like 4 streams + 2 power loads, nt", (k_maclike_pw<true, 4>), streams / 4, db, pw, pt_words, terms, out)
        RUN("mac-like 2 streams + 2 power loads, nt", (k_maclike_pw<true, 2>), streams / 2, db, pw, pt_words, terms, out)
        RUN("mac-like 4 streams + 2 power loads", (k_maclike_pw<false, 4>), streams / 4, db, pw, pt_words, terms, out)
        for (int lds_kb : {0, 20, 40, 80}) {   // dynamic LDS limits resident blocks per CU: 160/lds
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL((k_maclike_pw<true, 2>), dim3(16, 3, streams / 2), dim3(256), lds_kb * 1024, 0, db, pw, pt_words, terms, out);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            char nm[64]; snprintf(nm, 64, "S=2 + pw, nt, %d KiB LDS/block", lds_kb); if (rep) report(nm, ms);
        }
    }
    return 0;
}
