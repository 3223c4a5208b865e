// Microbenchmark: cost of one lazy 64-bit modular butterfly in registers, as k_ntt executes it (16 coefficients per
// lane, radix-8 passes, wave-uniform twiddles), without any memory traffic.  Variants:
//   0  Shoup-lazy butterfly of ntt_core.h (approximate quotient, 9 multiplies, 64-bit subtract with borrow)
//   1  the same with the subtraction written as an add of the complement (no carry chain through VCC)
//   2  fold reduction for q = 2^k - c (full 128-bit product, two folds, 7 multiplies, no quotient word)
//   3  variant 0 + the wide-modulus range fix (csub on the top bit) per butterfly
// Prints ns per butterfly per lane-slot and the implied limb-transform rate at n = 8192.
// Build: hipcc --offload-arch=gfx950 -O3 bfly.hip -o bfly
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;
#define HD __device__ __forceinline__
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Cst { u64 q, nq, q4, q4p1, ones, n4, q2; u32 ks, mhi, c, pad; };

HD void bfly0(u64 &x, u64 &y, u64 w, u64 wq, const Cst &k)
{
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), a0 = (u32)wq, a1 = (u32)(wq >> 32);
    const u64 t1 = (u64)y1 * a0, t2 = (u64)y0 * a1;
    const u64 h = (u64)y1 * a1 + (t1 >> 32) + (t2 >> 32);
    const u32 h0 = (u32)h, h1 = (u32)(h >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)k.nq, n1 = (u32)(k.nq >> 32);
    const u64 lo = (u64)h0 * n0 + ((u64)y0 * w0 + x);
    const u32 mid = y0 * w1 + y1 * w0 + h0 * n1 + h1 * n0;
    const u64 s = lo + ((u64)mid << 32);
    y = ((x << 1) + k.q4) - s;
    x = s;
}
HD void bfly1(u64 &x, u64 &y, u64 w, u64 wq, const Cst &k)
{
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), a0 = (u32)wq, a1 = (u32)(wq >> 32);
    const u64 t1 = (u64)y1 * a0, t2 = (u64)y0 * a1;
    const u64 h = (u64)y1 * a1 + (t1 >> 32) + (t2 >> 32);
    const u32 h0 = (u32)h, h1 = (u32)(h >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)k.nq, n1 = (u32)(k.nq >> 32);
    const u64 lo = (u64)h0 * n0 + ((u64)y0 * w0 + x);
    const u32 mid = y0 * w1 + y1 * w0 + h0 * n1 + h1 * n0;
    const u64 s = lo + ((u64)mid << 32);
    y = ((x << 1) + k.q4p1) + (s ^ k.ones);          // ones = ~0 read at run time: the compiler cannot fold it back into a subtract
    x = s;
}
HD void bfly2(u64 &x, u64 &y, u64 w, u64, const Cst &k)
{
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    const u64 a = (u64)y0 * w0;
    const u64 mid = (u64)y1 * w0 + ((u64)y0 * w1 + (a >> 32));
    const u64 top = (u64)y1 * w1 + (mid >> 32);
    const u32 P0 = (u32)a, P1 = (u32)mid, P2 = (u32)top, P3 = (u32)(top >> 32);
    const u32 Ph0 = __builtin_amdgcn_alignbit(P2, P1, k.ks), Ph1 = __builtin_amdgcn_alignbit(P3, P2, k.ks);
    const u64 Pl = ((u64)(P1 & k.mhi) << 32) | P0;
    const u64 g = (u64)Ph0 * k.c + Pl;
    const u64 e = (u64)Ph1 * k.c + (g >> 32);
    const u32 e0 = (u32)e, e1 = (u32)(e >> 32);
    const u32 Fh = __builtin_amdgcn_alignbit(e1, e0, k.ks);
    const u64 Fl = ((u64)(e0 & k.mhi) << 32) | (u32)g;
    const u64 v = (u64)Fh * k.c + Fl;
    const u64 s = x + v;
    y = (x + k.q2) - v;
    x = s;
}
// 4: the four cross products as ONE v_mad_u64_u32 chain that accumulates straight into the high word of the low product
//    (4 multiply-adds instead of 4 v_mul_lo_u32 + 2 adds + the 64-bit shift-add); twiddle and -q words as scalar operands
HD u64 mad_vs(u32 a, u32 b_sgpr, u64 c)
{
    u64 d; unsigned long long cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(b_sgpr), "v"(c));
    return d;
}
HD void bfly4(u64 &x, u64 &y, u64 w, u64 wq, const Cst &k)
{
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), a0 = (u32)wq, a1 = (u32)(wq >> 32);
    const u64 t1 = (u64)y1 * a0, t2 = (u64)y0 * a1;
    const u64 h = (u64)y1 * a1 + (t1 >> 32) + (t2 >> 32);
    const u32 h0 = (u32)h, h1 = (u32)(h >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)k.nq, n1 = (u32)(k.nq >> 32);
    const u64 lo = (u64)h0 * n0 + ((u64)y0 * w0 + x);
    u64 acc = lo >> 32;
    acc = mad_vs(y0, w1, acc);
    acc = mad_vs(y1, w0, acc);
    acc = mad_vs(h0, n1, acc);
    acc = mad_vs(h1, n0, acc);
    const u64 s = (u64)(u32)lo | (acc << 32);
    y = ((x << 1) + k.q4) - s;
    x = s;
}
// 5: the same chain written in C (what the compiler makes of it)
HD void bfly5(u64 &x, u64 &y, u64 w, u64 wq, const Cst &k)
{
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), a0 = (u32)wq, a1 = (u32)(wq >> 32);
    const u64 t1 = (u64)y1 * a0, t2 = (u64)y0 * a1;
    const u64 h = (u64)y1 * a1 + (t1 >> 32) + (t2 >> 32);
    const u32 h0 = (u32)h, h1 = (u32)(h >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)k.nq, n1 = (u32)(k.nq >> 32);
    const u64 lo = (u64)h0 * n0 + ((u64)y0 * w0 + x);
    u64 acc = lo >> 32;
    acc = (u64)y0 * w1 + acc; acc = (u64)y1 * w0 + acc; acc = (u64)h0 * n1 + acc; acc = (u64)h1 * n0 + acc;
    const u64 s = (u64)(u32)lo | (acc << 32);
    y = ((x << 1) + k.q4) - s;
    x = s;
}
// 6: the chain started from zero and its low word added to the high word of the low product (no register-pair shuffling)
HD u64 mad_vs0(u32 a, u32 b_sgpr)
{
    u64 d; unsigned long long cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "s"(b_sgpr));
    return d;
}
HD void bfly6(u64 &x, u64 &y, u64 w, u64 wq, const Cst &k)
{
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), a0 = (u32)wq, a1 = (u32)(wq >> 32);
    const u64 t1 = (u64)y1 * a0, t2 = (u64)y0 * a1;
    const u64 h = (u64)y1 * a1 + (t1 >> 32) + (t2 >> 32);
    const u32 h0 = (u32)h, h1 = (u32)(h >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)k.nq, n1 = (u32)(k.nq >> 32);
    const u64 lo = (u64)h0 * n0 + ((u64)y0 * w0 + x);
    u64 acc = mad_vs0(y0, w1);
    acc = mad_vs(y1, w0, acc);
    acc = mad_vs(h0, n1, acc);
    acc = mad_vs(h1, n0, acc);
    const u32 shi = (u32)(lo >> 32) + (u32)acc;
    const u64 s = (u64)(u32)lo | ((u64)shi << 32);
    y = ((x << 1) + k.q4) - s;
    x = s;
}
// 7: the chain in C with an empty assembly statement on the 64-bit sum after every step: the compiler must keep the sum whole,
//    so it uses v_mad_u64_u32 with the sum as addend, but it still chooses registers and operand kinds (scalar / vector) itself
#define OPAQUE64(v) asm("" : "+v"(v))
HD void bfly7(u64 &x, u64 &y, u64 w, u64 wq, const Cst &k)
{
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), a0 = (u32)wq, a1 = (u32)(wq >> 32);
    const u64 t1 = (u64)y1 * a0, t2 = (u64)y0 * a1;
    const u64 h = (u64)y1 * a1 + (t1 >> 32) + (t2 >> 32);
    const u32 h0 = (u32)h, h1 = (u32)(h >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)k.nq, n1 = (u32)(k.nq >> 32);
    const u64 lo = (u64)h0 * n0 + ((u64)y0 * w0 + x);
    u64 acc = lo >> 32;
    acc = (u64)y0 * w1 + acc; OPAQUE64(acc);
    acc = (u64)y1 * w0 + acc; OPAQUE64(acc);
    acc = (u64)h0 * n1 + acc; OPAQUE64(acc);
    acc = (u64)h1 * n0 + acc; OPAQUE64(acc);
    const u64 s = (u64)(u32)lo | (acc << 32);
    y = ((x << 1) + k.q4) - s;
    x = s;
}
// 8: variant 7 + the second high word joined by a multiply-add with an opaque factor 1 (one instruction instead of a move and a 64-bit add)
HD void bfly8(u64 &x, u64 &y, u64 w, u64 wq, const Cst &k)
{
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), a0 = (u32)wq, a1 = (u32)(wq >> 32);
    const u32 t1h = (u32)(((u64)y1 * a0) >> 32), t2h = (u32)(((u64)y0 * a1) >> 32);
    u32 one = 1; asm("" : "+v"(one));
    u64 h = (u64)y1 * a1 + t1h; OPAQUE64(h);
    h = (u64)t2h * one + h; OPAQUE64(h);
    const u32 h0 = (u32)h, h1 = (u32)(h >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)k.nq, n1 = (u32)(k.nq >> 32);
    const u64 lo = (u64)h0 * n0 + ((u64)y0 * w0 + x);
    u64 acc = lo >> 32;
    acc = (u64)y0 * w1 + acc; OPAQUE64(acc);
    acc = (u64)y1 * w0 + acc; OPAQUE64(acc);
    acc = (u64)h0 * n1 + acc; OPAQUE64(acc);
    acc = (u64)h1 * n0 + acc; OPAQUE64(acc);
    const u64 s = (u64)(u32)lo | (acc << 32);
    y = ((x << 1) + k.q4) - s;
    x = s;
}
HD void bfly3(u64 &x, u64 &y, u64 w, u64 wq, const Cst &k)
{
    x = x + ((u64)((int64_t)x >> 63) & k.n4);
    bfly0(x, y, w, wq, k);
}

template <int V>
__global__ __launch_bounds__(512, 4) void kern(u64 *data, const u64 *tw, Cst k, int iters)
{
    u64 r[16];
    u64 *p = data + ((size_t)blockIdx.x * 512 + threadIdx.x) * 16;
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = p[i];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int st = 0; st < 3; st++) {
            const int bit = 4 >> st;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (j & bit) continue;
                const u64 w = tw[((it & 7) * 24 + st * 8 + j) * 2], wq = tw[((it & 7) * 24 + st * 8 + j) * 2 + 1];   // wave-uniform -> scalar loads
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    u64 &x = r[g * 8 + j], &y = r[g * 8 + (j | bit)];
                    if (V == 0) bfly0(x, y, w, wq, k);
                    if (V == 1) bfly1(x, y, w, wq, k);
                    if (V == 2) bfly2(x, y, w, wq, k);
                    if (V == 3) bfly3(x, y, w, wq, k);
                    if (V == 4) bfly4(x, y, w, wq, k);
                    if (V == 5) bfly5(x, y, w, wq, k);
                    if (V == 6) bfly6(x, y, w, wq, k);
                    if (V == 7) bfly7(x, y, w, wq, k);
                    if (V == 8) bfly8(x, y, w, wq, k);
                }
            }
        }
        if (V != 3) {            // keep the narrow variants in range like the real transform's pass boundaries do not need to: cheap mask
#pragma unroll
            for (int i = 0; i < 16; i++) r[i] &= 0x03ffffffffffffffull;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = r[i];
}

template <int V> int run(const char *name, u64 *data, const u64 *tw, const Cst &k)
{
    const int iters = 400, grid = 512;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<V><<<grid, 512>>>(data, tw, k, iters);
    CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        kern<V><<<grid, 512>>>(data, tw, k, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bflies = (double)grid * 512 * 24 * iters;
    const double per_s = bflies / (best * 1e-3);
    // one limb transform at n = 8192 is 53248 butterflies and 131072 algorithmic bytes
    printf("%-28s %.3f ms  %.1f G butterflies/s  => butterfly-bound NTT rate %.2f TB/s (n = 8192)\n", name, best, per_s / 1e9,
           per_s / 53248.0 * 131072.0 / 1e12);
    return 0;
}

int main()
{
    const u64 q = 0xfffffffff70001ull;   // 2^56 - 0x8ffff
    Cst k{};
    k.q = q; k.nq = 0 - q; k.q4 = q << 2; k.q4p1 = (q << 2) + 1; k.ones = ~0ull; k.n4 = 0 - (q << 2); k.q2 = q << 1;
    k.ks = 56 - 32; k.mhi = (1u << (56 - 32)) - 1; k.c = 0x8ffff;
    std::vector<u64> h((size_t)512 * 512 * 16), tw(8 * 24 * 2);
    u64 z = 88172645463325252ull;
    auto rnd = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return z; };
    for (auto &v : h) v = rnd() % q;
    for (size_t i = 0; i < tw.size(); i += 2) { tw[i] = rnd() % q; tw[i + 1] = (u64)(((unsigned __int128)tw[i] << 64) / q); }
    u64 *d, *dt;
    CHECK(hipMalloc(&d, h.size() * 8)); CHECK(hipMalloc(&dt, tw.size() * 8));
    CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dt, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    run<0>("shoup-lazy (k_ntt today)", d, dt, k);
    run<1>("shoup-lazy, complement add", d, dt, k);
    run<2>("fold 2^k - c (7 multiplies)", d, dt, k);
    run<3>("shoup-lazy + top-bit csub", d, dt, k);
    run<4>("cross terms as a mad chain (asm)", d, dt, k);
    run<5>("cross terms as a mad chain (C)", d, dt, k);
    run<6>("mad chain from zero + add (asm)", d, dt, k);
    run<0>("shoup-lazy (k_ntt today), again", d, dt, k);
    run<4>("cross terms as a mad chain (asm), again", d, dt, k);
    run<6>("mad chain from zero + add (asm), again", d, dt, k);
    run<7>("mad chain in C behind opaque sums", d, dt, k);
    run<0>("shoup-lazy (k_ntt today), once more", d, dt, k);
    run<7>("mad chain in C behind opaque sums, again", d, dt, k);
    run<4>("cross terms as a mad chain (asm), once more", d, dt, k);
    run<8>("variant 7 + high words joined by a mad", d, dt, k);
    run<7>("mad chain in C behind opaque sums, third", d, dt, k);
    run<8>("variant 7 + high words joined by a mad, again", d, dt, k);
    return 0;
}
