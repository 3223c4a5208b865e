// Does streaming-read bandwidth depend on WHERE an allocation lies?  (k_mac runs 10 % faster on one copy of the database than on another
// in the same process: profiles/r05_mac_placement.txt.)  N slabs of S GiB each, every one read by the same flat 16-byte grid-stride
// kernel and by a k_mac-shaped access pattern (512 concurrent rows of 56 KiB, advancing 168 KiB per step); three passes.
//   hipcc --offload-arch=gfx950 -O3 -o _bin/slabbw slabbw.hip ;  ./slabbw [slabs] [GiB per slab]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef uint64_t u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_fill(u64 *p, size_t words) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = i * 0x9e3779b97f4a7c15ULL; }
__global__ void k_flat(const u64x2 *__restrict__ p, size_t n16, u64 *out)
{
    u64 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const u64x2 v = __builtin_nontemporal_load(p + i); acc += v[0] ^ v[1]; }
    if (acc == 0x1234567) out[0] = acc;
}
// k_mac's shape: block (x: 16 column blocks, y: stream) walks `terms` rows of 56 KiB, 168 KiB apart, 4 streams per block interleaved
__global__ void k_rows(const char *__restrict__ base, size_t stream_bytes, int terms, u64 *out)
{
    u64 acc = 0;
    const char *p[4];
    for (int g = 0; g < 4; g++) p[g] = base + ((size_t)blockIdx.y * 4 + g) * stream_bytes + ((size_t)blockIdx.x * 256 + threadIdx.x) * 14;
    for (int i = 0; i < terms; i++)
        for (int g = 0; g < 4; g++) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(reinterpret_cast<uintptr_t>(p[g] + (size_t)i * 172032) & ~(uintptr_t)3);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(q));
            acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        }
    if (acc == 0x1234567) out[0] = acc;
}
// block-major layout: the `terms` rows of one (stream, block of 512 coefficients) CONTIGUOUS (terms x 3.5 KiB = 154 KiB per stream and
// workgroup read front to back) instead of 168 KiB apart
__global__ void k_rows_blockmajor(const char *__restrict__ base, size_t stream_bytes, int terms, u64 *out)
{
    u64 acc = 0;
    const char *p[4];
    for (int g = 0; g < 4; g++) p[g] = base + ((size_t)blockIdx.y * 4 + g) * stream_bytes + (size_t)blockIdx.x * terms * 3584 + (size_t)threadIdx.x * 14;
    for (int i = 0; i < terms; i++)
        for (int g = 0; g < 4; g++) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(reinterpret_cast<uintptr_t>(p[g] + (size_t)i * 3584) & ~(uintptr_t)3);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(q));
            acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        }
    if (acc == 0x1234567) out[0] = acc;
}
int main(int argc, char **argv)
{
    const int slabs = argc > 1 ? atoi(argv[1]) : 16;
    const size_t gib = argc > 2 ? (size_t)atoi(argv[2]) : 2, bytes = gib << 30;
    std::vector<u64 *> s(slabs);
    u64 *out; CHECK(hipMalloc(&out, 64));
    for (int i = 0; i < slabs; i++) { CHECK(hipMalloc(&s[i], bytes + 4096)); k_fill<<<4096, 256>>>(s[i], bytes / 8); }
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int terms = 44; const size_t stream_bytes = (size_t)terms * 172032; const int streams4 = (int)(bytes / stream_bytes / 4);
    for (int pass = 0; pass < 3; pass++)
        for (int i = 0; i < slabs; i++) {
            float best_a = 1e9f, best_b = 1e9f, best_c = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                float ms;
                CHECK(hipEventRecord(e0)); k_flat<<<256 * 16, 256>>>(reinterpret_cast<const u64x2 *>(s[i]), bytes / 16, out); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&ms, e0, e1)); best_a = std::min(best_a, ms);
                CHECK(hipEventRecord(e0)); k_rows<<<dim3(16, streams4), 256>>>(reinterpret_cast<const char *>(s[i]), stream_bytes, terms, out); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&ms, e0, e1)); best_b = std::min(best_b, ms);
                // (a stream covers 16 blocks x terms x 3.5 KiB = the same bytes as one limb of `terms` rows: stream_bytes / 3 per limb)
                CHECK(hipEventRecord(e0)); k_rows_blockmajor<<<dim3(16, streams4 * 3), 256>>>(reinterpret_cast<const char *>(s[i]), stream_bytes / 3, terms, out); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&ms, e0, e1)); best_c = std::min(best_c, ms);
            }
            const double rows_bytes = (double)streams4 * 4 * terms * 16 * 256 * 14;
            const double bm_bytes = (double)streams4 * 3 * 4 * terms * 16 * 256 * 14;
            printf("pass %d slab %2d at %p: flat read %.0f GB/s   k_mac-shaped rows %.0f GB/s   block-major rows %.0f GB/s\n", pass, i, (void *)s[i], bytes / (best_a * 1e-3) / 1e9,
                   rows_bytes / (best_b * 1e-3) / 1e9, bm_bytes / (best_c * 1e-3) / 1e9);
        }
    return 0;
}
