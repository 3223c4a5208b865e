// LDS-DMA ring variant of the engine's dyadic multiply-accumulate (k_mac), kept OUT of the library: measured neutral to
// slower (profiles/r03_mac_ring.txt).  Included by macbench.hip behind kernels.hip; `RING=1 ./macbench` runs it against k_mac
// on the same jobs and compares the outputs bit for bit.
#pragma once
namespace apsu_he {
// The same multiply-accumulate with the operands staged through a per-wave LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4): D terms of G plaintext streams + 2 power polynomials (1 KiB per stream and wave) are in
// flight per wave without holding a single VGPR, against one term for the register ping-pong of k_mac.  Every wave owns
// its ring (no barriers); the only ordering is the wave's own vmcnt (LDS-DMA completes in issue order) and lgkmcnt.
// The LDS reads and the counted waits are inline asm: the compiler would otherwise wait for ALL outstanding DMA at the
// first LDS read (cdna_hip_programming.md section 5, "Async global->LDS copy").
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int NS, int D> __device__ __forceinline__ void wait_terms(int terms_behind)
{
    // `terms_behind` younger terms (NS loads each) may stay in flight
    switch (terms_behind) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<NS>(); break;
    case 2: wait_vmcnt<2 * NS>(); break;
    case 3: wait_vmcnt<3 * NS>(); break;
    default: wait_vmcnt<(D > 4 ? 4 : 0) * NS>(); break;
    }
}

template <int G, int D, int T>
__global__ __launch_bounds__(T) void k_mac_ring(const DevLevel *__restrict__ lv, const MacJob *__restrict__ jobs, size_t n, unsigned aux_nt)
{
    constexpr int C = 2, NS = G + 2;                            // streams per term: two power polynomials + G plaintexts
    constexpr int SLOT = NS * 1024, WAVE_RING = D * SLOT;       // bytes
    static_assert(D >= 2 && D <= 5 && (D - 1) * NS < 64, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char mac_ring[];
    const size_t k = ((size_t)blockIdx.x * T + threadIdx.x) * C;
    if (k >= n) return;
    constexpr int SPLIT = MAC_G / G;
    const MacJob *__restrict__ jp = jobs + blockIdx.z / SPLIT;
    struct { const u64 *pw; u32 cnt, ng, pt_stride, pw_stride, pw_poly_stride, out_poly_stride, limb0; } job =
        { jp->pw, jp->cnt, jp->ng, jp->pt_stride, jp->pw_stride, jp->pw_poly_stride, jp->out_poly_stride, jp->limb0 };
    const int g0 = (blockIdx.z % SPLIT) * G;
    if (g0 >= (int)job.ng || blockIdx.y >= jp->nl) return;
    const int j = blockIdx.y + job.limb0;
    const Mod m = lv->q[j];
    const u32 s = lv->mac_shift[j], chunk = lv->mac_chunk[j];
    const u32 lomask = (1u << s) - 1;
    // running source pointers: stream 0,1 = the two power polynomials, 2.. = plaintext streams
    const u64 *src[NS];
    src[0] = job.pw + (size_t)j * n + k;
    src[1] = src[0] + job.pw_poly_stride;
#pragma unroll
    for (int g = 0; g < G; g++) src[2 + g] = jp->pt[g0 + g < (int)job.ng ? g0 + g : g0] + (size_t)j * n + k;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    lds_byte *const ring = (lds_byte *)mac_ring + wave * WAVE_RING;         // wave-uniform
    const u32 rd_base = (u32)(size_t)ring + lane * 16;                       // this lane's 16 bytes of stream 0, slot 0

    u64 s00[G][C][2], sx[G][C][2], s11[G][C][2];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int c = 0; c < C; c++)
#pragma unroll
            for (int p = 0; p < 2; p++) s00[g][c][p] = sx[g][c][p] = s11[g][c][p] = 0;

    u32 wslot = 0, rslot = 0;                                   // ring slots of the next issue / next read (wave-uniform)
    auto issue = [&]() {
        lds_byte *dst = ring + wslot * SLOT;
#pragma unroll
        for (int x = 0; x < NS; x++) {
            const auto *g = (const __attribute__((address_space(1))) void *)src[x];
            if (x < 2) __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void *)(dst + x * 1024), 16, 0, 0);
            else if (aux_nt) __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void *)(dst + x * 1024), 16, 0, 2);
            else __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void *)(dst + x * 1024), 16, 0, 0);
            src[x] += x < 2 ? job.pw_stride : job.pt_stride;
        }
        wslot = wslot + 1 == D ? 0 : wslot + 1;
    };
    auto consume = [&]() {
        const u32 a = rd_base + rslot * SLOT;
        u64x2 v[NS];
        // one asm block: the compiler must not touch a destination before the wait (it does not track asm results)
        static_assert(NS == 6 || NS == 4 || NS == 3, "streams per term");
        if constexpr (NS == 6)
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %6 offset:2048\n\t"
                         "ds_read_b128 %3, %6 offset:3072\n\tds_read_b128 %4, %6 offset:4096\n\tds_read_b128 %5, %6 offset:5120\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]) : "v"(a) : "memory");
        else if constexpr (NS == 4)
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                         "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a) : "memory");
        else
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(a) : "memory");
        rslot = rslot + 1 == D ? 0 : rslot + 1;
        u32 clo[2][C], chi[2][C];
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int c = 0; c < C; c++) { clo[p][c] = (u32)v[p][c] & lomask; chi[p][c] = (u32)(v[p][c] >> s); }
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int c = 0; c < C; c++) {
                const u32 alo = (u32)v[2 + g][c] & lomask, ahi = (u32)(v[2 + g][c] >> s);
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    s00[g][c][p] += (u64)alo * clo[p][c];
                    sx[g][c][p] += (u64)alo * chi[p][c];
                    sx[g][c][p] += (u64)ahi * clo[p][c];
                    s11[g][c][p] += (u64)ahi * chi[p][c];
                }
            }
    };
    auto fold = [&]() {
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    u128p acc{ s00[g][c][p], 0 };
                    add128(acc, u128p{ sx[g][c][p] << s, sx[g][c][p] >> (64 - s) });
                    add128(acc, u128p{ s11[g][c][p] << (2 * s), s11[g][c][p] >> (64 - 2 * s) });
                    s00[g][c][p] = barrett128(acc, m);
                    sx[g][c][p] = s11[g][c][p] = 0;
                }
    };

    const u32 cnt = job.cnt;
    const u32 ahead = cnt < (u32)(D - 1) ? cnt : (u32)(D - 1);
    for (u32 i = 0; i < ahead; i++) issue();
    u32 in_chunk = 0, i = 0;
    for (; i + (D - 1) < cnt; i++) {                            // steady state: D - 1 younger terms stay in flight
        issue();
        wait_vmcnt<(D - 1) * NS>();
        consume();
        if (++in_chunk >= chunk) { fold(); in_chunk = 1; }       // the folded residue counts as one term
    }
    for (; i < cnt; i++) {                                       // drain: everything has been issued
        wait_terms<NS, D>((int)(cnt - 1 - i));
        consume();
        if (++in_chunk >= chunk) { fold(); in_chunk = 1; }
    }
    fold();
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g0 + g < (int)job.ng) {
            u64 *o = jp->out[g0 + g] + (size_t)blockIdx.y * n + k;
            u64x2 r0, r1;
            r0[0] = s00[g][0][0]; r0[1] = s00[g][1][0];
            r1[0] = s00[g][0][1]; r1[1] = s00[g][1][1];
            *reinterpret_cast<u64x2 *>(o) = r0;
            *reinterpret_cast<u64x2 *>(o + job.out_poly_stride) = r1;
        }
    }
}

#ifndef APSU_MAC_RING_T
#define APSU_MAC_RING_T 256
#endif
template <int D>
static void launch_mac_ring_d(const DevLevel *lv, int nlimbs, const MacJob *jobs, size_t n, int njobs, hipStream_t st, unsigned nt, unsigned min_lds)
{
    constexpr int G = MAC_G, T = APSU_MAC_RING_T;
    constexpr unsigned ring = (T / 64) * D * (G + 2) * 1024;
    const unsigned lds = std::max(ring, min_lds);               // min_lds > 80 KiB: one workgroup per CU, 72 KiB left for an NTT workgroup
    static unsigned attr_for = 0;
    if (attr_for < lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mac_ring<G, D, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_for = lds;
    }
    hipLaunchKernelGGL((k_mac_ring<G, D, T>), dim3((unsigned)((n / 2 + T - 1) / T), (unsigned)nlimbs, (unsigned)njobs), dim3(T), lds, st, lv, jobs, n, nt);
}
void launch_mac_ring(const DevLevel *lv, int nlimbs, const MacJob *jobs, size_t n, int njobs, hipStream_t st, unsigned nt, int depth, unsigned min_lds)
{
    if (!njobs || !nlimbs) return;
    switch (depth) {
    case 2: launch_mac_ring_d<2>(lv, nlimbs, jobs, n, njobs, st, nt, min_lds); break;
    case 4: launch_mac_ring_d<4>(lv, nlimbs, jobs, n, njobs, st, nt, min_lds); break;
    default: launch_mac_ring_d<3>(lv, nlimbs, jobs, n, njobs, st, nt, min_lds); break;
    }
    KERNEL_CHECK();
}

} // namespace apsu_he
