// Microbenchmark: issue rate of the integer/FP64 instructions a 64-bit modular
// butterfly is built from, on gfx950.  Prints cycles per wave-instruction per SIMD
// at 1, 2, 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 intmul.hip -o intmul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

template <int OP>
__global__ void kern(uint64_t* out, uint64_t seed, uint64_t* cycles)
{
    uint32_t a[UNROLL], b = (uint32_t)seed + threadIdx.x;
    uint64_t c[UNROLL];
    double d[UNROLL], e = 1.0000001 + threadIdx.x * 1e-9;
    for (int i = 0; i < UNROLL; i++) { a[i] = threadIdx.x * 2654435761u + i; c[i] = seed * (i + 1) + threadIdx.x; d[i] = 1.0 + i; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) { asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b)); }
            if (OP == 1) { asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b)); }
            if (OP == 2) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b) : "vcc"); }
            if (OP == 3) { asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(e)); }
            if (OP == 4) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b)); }
            if (OP == 5) { asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b)); }
            if (OP == 6) { asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b) : "vcc"); }
            if (OP == 7) { asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(b) : "vcc"); }
            if (OP == 8) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[i]) : "v"(c[(i+1)%UNROLL])); }
            if (OP == 9) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e)); }
            if (OP == 10) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(b) : "vcc"); }
            if (OP == 11) { asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[i]), "v"(b)); }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t acc = 0;
    for (int i = 0; i < UNROLL; i++) acc += a[i] + c[i] + (uint64_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
int run(const char* name)
{
    uint64_t *out, *cyc;
    CHECK(hipMalloc(&out, 256 * 1024 * 8));
    CHECK(hipMalloc(&cyc, 1024 * 8));
    for (int wps : {1, 2, 4}) {          // waves per SIMD (block = 4*wps waves, one block per CU)
        int threads = 64 * 4 * wps;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        kern<OP><<<256, threads>>>(out, 12345, cyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        kern<OP><<<256, threads>>>(out, 12345, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> h(256);
        CHECK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += v; avg /= 256;
        double insts_per_simd = (double)ITERS * UNROLL * wps;
        // s_memtime ticks at 100MHz-ish constant clock? report both
        printf("%-16s waves/SIMD=%d  wall=%.3f ms  memtime_ticks=%.0f  ns/inst/SIMD=%.3f (=> cycles@2.4GHz %.2f)\n",
               name, wps, ms, avg, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
    }
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
    return 0;
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz LDS/block=%zu\n", p.name, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
    run<0>("v_mul_lo_u32"); run<1>("v_mul_hi_u32"); run<2>("v_mad_u64_u32"); run<3>("v_fma_f64");
    run<4>("v_add_u32"); run<5>("v_mul_u32_u24"); run<6>("v_add_co_u32"); run<7>("v_addc_co_u32");
    run<8>("v_lshl_add_u64"); run<9>("v_mul_f64"); run<10>("v_cndmask_b32"); run<11>("v_mad_u32_u24");
    return 0;
}
