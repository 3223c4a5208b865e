// Microbenchmark: the inverse NTT's final twist (x * n^-1 psi^-j mod q, canonical result) per coefficient, in registers:
//   0  exact Shoup product + conditional subtraction (k_ntt today)
//   1  approximate-quotient product ([0,4q), 9 multiplies) + 2^k - c fold to [0,2q) + conditional subtraction
// Build: hipcc --offload-arch=gfx950 -O3 twist.hip -o twist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;
#define HD __device__ __forceinline__
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Cst { u64 q, nq; u32 k, c; };
HD u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }
HD u64 twist0(u64 x, u64 w, u64 wq, const Cst &t) { u64 r = x * w - mulhi64(x, wq) * t.q; return r >= t.q ? r - t.q : r; }
HD u64 mul_lazy4(u64 x, u64 w, u64 wq, u64 nq)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), a0 = (u32)wq, a1 = (u32)(wq >> 32);
    const u64 t1 = (u64)x1 * a0, t2 = (u64)x0 * a1;
    const u64 h = (u64)x1 * a1 + (t1 >> 32) + (t2 >> 32);
    const u32 h0 = (u32)h, h1 = (u32)(h >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    const u64 lo = (u64)x0 * w0 + (u64)h0 * n0;
    const u64 mid = (u64)x0 * w1 + (u64)x1 * w0 + (u64)h0 * n1 + (u64)h1 * n0;
    return lo + (mid << 32);
}
HD u64 twist1(u64 x, u64 w, u64 wq, const Cst &t)
{
    const u64 v = mul_lazy4(x, w, wq, t.nq);                      // [0, 4q)
    const u32 sh = t.k - 32, hi = (u32)(v >> 32);
    const u64 low = ((u64)(hi & ((1u << sh) - 1)) << 32) | (u32)v;
    const u64 r = (u64)(hi >> sh) * t.c + low;                    // < 2q
    return r >= t.q ? r - t.q : r;
}
template <int V> __global__ __launch_bounds__(512, 4) void kern(u64 *data, const u64 *tw, Cst t, int iters)
{
    u64 r[16], w[16], wq[16];
    u64 *p = data + ((size_t)blockIdx.x * 512 + threadIdx.x) * 16;
#pragma unroll
    for (int i = 0; i < 16; i++) { r[i] = p[i]; w[i] = tw[2 * ((threadIdx.x + i) & 255)]; wq[i] = tw[2 * ((threadIdx.x + i) & 255) + 1]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) r[i] = (V == 0 ? twist0(r[i], w[i], wq[i], t) : twist1(r[i], w[i], wq[i], t)) + it;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = r[i];
}
template <int V> int run(const char *name, u64 *d, const u64 *tw, const Cst &t)
{
    const int iters = 300, grid = 512;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<V><<<grid, 512>>>(d, tw, t, iters); CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0)); kern<V><<<grid, 512>>>(d, tw, t, iters); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms  %.1f G twists/s\n", name, best, (double)grid * 512 * 16 * iters / (best * 1e-3) / 1e9);
    return 0;
}
int main()
{
    const u64 q = 0xfffffffff70001ull;
    Cst t{ q, 0 - q, 56, 0x8ffff };
    std::vector<u64> h((size_t)512 * 512 * 16), tw(512);
    u64 z = 88172645463325252ull;
    auto rnd = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return z; };
    for (auto &v : h) v = rnd() % q;
    for (size_t i = 0; i < tw.size(); i += 2) { tw[i] = rnd() % q; tw[i + 1] = (u64)(((unsigned __int128)tw[i] << 64) / q); }
    u64 *d, *dt; CHECK(hipMalloc(&d, h.size() * 8)); CHECK(hipMalloc(&dt, tw.size() * 8));
    CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dt, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    run<0>("exact Shoup + csub (k_ntt today)", d, dt, t);
    run<1>("approximate quotient + fold + csub", d, dt, t);
    return 0;
}
