// Experiment record (round 4): k_mac with workgroups that live for many (job, limb) units and keep their two-deep load pipeline
// running across units.  Included by macbench.hip (PERSIST=1); NOT part of the library: bit-identical to k_mac and 4-9 % slower
// at 44 and 150 terms per chain (profiles/r04_mac_persist.txt).
#pragma once
namespace apsu_he {
// a wave-uniform value the compiler cannot prove uniform (loaded through a pointer): pin it to scalar registers
template <class T> __device__ __forceinline__ T *uniform_ptr(T *p)
{
    const u64 v = (u64)(uintptr_t)p;
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
    return (T *)(uintptr_t)(((u64)hi << 32) | lo);
}
__device__ __forceinline__ u32 uniform_u32(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// the job table seen through the CONSTANT address space: a load with a wave-uniform address from it is a scalar load (s_load, tracked
// by lgkmcnt), so reading the next unit's descriptor does not wait for -- and thereby drain -- the vector loads in flight
typedef const MacJob __attribute__((address_space(4))) *MacJobC;
template <class T> __device__ __forceinline__ T *from_const(const T __attribute__((address_space(4))) *p) { return (T *)(uintptr_t)p; }

// ---- k_mac_p: the same sums with workgroups that live for many units (round 4) -----------------------------------------
// k_mac starts one workgroup per (coefficient block, limb, job).  Each pays a cold start (descriptor, stream pointers, first
// terms: three dependent memory round trips) and a drain (fold, store) during which its CU streams at most half of what it
// could -- the kernel's register budget allows two workgroups per CU and a term takes only ~1.2 us.  Measured with macbench
// (profiles/r04_mac_units.txt): ~15 us per workgroup that do not depend on the chain length, i.e. 22 % of a 44-term chain.
// Here the grid is (coefficient blocks, R) with R = what the chip holds at once; workgroup (x, r) walks the units
// (job z, limb y) for z = r, r + R, ... and keeps its two-deep load pipeline running ACROSS units: while it finishes one
// chain the first term of the next chain is already in flight, and the fold + store of a chain overlap those loads.  The
// coefficient block x stays fixed per workgroup, so the XCD placement of the shared powers (workgroup id mod 8 = x mod 8)
// is the one k_mac relies on.  Arithmetic, chunking and results are k_mac's.
template <int G, int C, bool KARA = false>
__global__ __launch_bounds__(EW_T, 2) void k_mac_p(const DevLevel *__restrict__ lv, const MacJob *__restrict__ jobs, size_t n, u32 nz)
{
    static_assert(C == 1 || C == 2, "coefficients per lane");
    const size_t k = ((size_t)blockIdx.x * EW_T + threadIdx.x) * C;
    if (k >= n) return;
    constexpr int SPLIT = MAC_G / G;
    struct Unit {                         // wave-uniform: bases of this unit's limb (the lane adds its coefficient offset k)
        const u64 *p0;                    // powers, polynomial 0 (polynomial 1 is pw_poly_stride words on)
        const u64 *pt[G];                 // plaintext streams
        const MacJob *jp;                 // (the output pointers are read when the chain is stored)
        u32 cnt, pt_stride, pw_stride, pw_poly_stride, g0, live, j, y;      // live: streams g < live exist
    };
    // the unit at or behind (z, y) in this workgroup's walk; false at the end of the walk
    auto find = [&](u32 &z, u32 &y, Unit &u) -> bool {
        while (z < nz) {
            const MacJobC jp = (MacJobC)(uintptr_t)(jobs + z / SPLIT);
            const u32 g0 = (z % SPLIT) * G, ng = jp->ng, nl = jp->nl;
            if (g0 < ng && y < nl) {
                const u32 j = y + jp->limb0;
                const size_t off = (size_t)j * n;
                u.p0 = jp->pw + off;
                u.pw_poly_stride = jp->pw_poly_stride;
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const u32 gi = g0 + g < ng ? g0 + g : g0;                   // missing streams alias a real one
                    u.pt[g] = jp->pt[gi] + off;
                }
                u.jp = jobs + z / SPLIT;
                u.cnt = jp->cnt; u.pt_stride = jp->pt_stride; u.pw_stride = jp->pw_stride;
                u.g0 = g0; u.live = ng - g0 < (u32)G ? ng - g0 : (u32)G; u.j = j; u.y = y;
                return true;
            }
            z += gridDim.y; y = 0;
        }
        return false;
    };
    struct Term { u64 c[2][C]; u64 a[G][C]; };
    auto load_term = [&](const Unit &u, u32 i, Term &t) {
        const u64 *c0 = u.p0 + (size_t)i * u.pw_stride, *c1 = c0 + u.pw_poly_stride;               // uniform term bases
        if (C == 2) {
            const u64x2 v0 = ldg16(c0 + k), v1 = ldg16(c1 + k);
            t.c[0][0] = v0[0]; t.c[0][C - 1] = v0[1]; t.c[1][0] = v1[0]; t.c[1][C - 1] = v1[1];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const u64x2 a = ldg16_nt(u.pt[g] + (size_t)i * u.pt_stride + k);
                t.a[g][0] = a[0]; t.a[g][C - 1] = a[1];
            }
        } else {
            t.c[0][0] = c0[k]; t.c[1][0] = c1[k];
#pragma unroll
            for (int g = 0; g < G; g++) t.a[g][0] = __builtin_nontemporal_load(u.pt[g] + (size_t)i * u.pt_stride + k);
        }
    };

    u32 z = blockIdx.y, y = 0;
    Unit cur, nxt;
    if (!find(z, y, cur)) return;
    Term A, B;
    load_term(cur, 0, A);
    for (;;) {
        u32 z2 = z, y2 = y + 1;
        const bool has_next = find(z2, y2, nxt);                 // its descriptor loads overlap this unit's stream
        const DevLevel __attribute__((address_space(4))) *lc = (const DevLevel __attribute__((address_space(4))) *)(uintptr_t)lv;
        const Mod m{ lc->q[cur.j].q, lc->q[cur.j].r0, lc->q[cur.j].r1 };
        const u32 s = lc->mac_shift[cur.j], chunk = KARA ? lc->mac_chunk_k[cur.j] : lc->mac_chunk[cur.j];
        const u32 lomask = (1u << s) - 1;
        u64 s00[G][C][2], sx[G][C][2], s11[G][C][2];
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int p = 0; p < 2; p++) s00[g][c][p] = sx[g][c][p] = s11[g][c][p] = 0;
        auto mac_term = [&](const Term &t) {
            u32 clo[2][C], chi[2][C], csum[2][C];
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int c = 0; c < C; c++) {
                    clo[p][c] = (u32)t.c[p][c] & lomask; chi[p][c] = (u32)(t.c[p][c] >> s);
                    csum[p][c] = clo[p][c] + chi[p][c];
                }
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const u32 alo = (u32)t.a[g][c] & lomask, ahi = (u32)(t.a[g][c] >> s);
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        s00[g][c][p] += (u64)alo * clo[p][c];
                        if (KARA) sx[g][c][p] += (u64)(alo + ahi) * csum[p][c];
                        else {
                            sx[g][c][p] += (u64)alo * chi[p][c];
                            sx[g][c][p] += (u64)ahi * clo[p][c];
                        }
                        s11[g][c][p] += (u64)ahi * chi[p][c];
                    }
                }
        };
        auto fold = [&](bool last) {
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        u128p acc{ s00[g][c][p], 0 };
                        const u64 cross = KARA ? sx[g][c][p] - s00[g][c][p] - s11[g][c][p] : sx[g][c][p];
                        add128(acc, u128p{ cross << s, cross >> (64 - s) });
                        add128(acc, u128p{ s11[g][c][p] << (2 * s), s11[g][c][p] >> (64 - 2 * s) });
                        const u64 r = barrett128(acc, m);
                        if (KARA && !last) { s00[g][c][p] = r & lomask; sx[g][c][p] = (r & lomask) + (r >> s); }
                        else { s00[g][c][p] = r; sx[g][c][p] = 0; }
                        s11[g][c][p] = 0;
                    }
        };
        const u32 cnt = cur.cnt, npairs = cnt >> 1;
        u32 in_chunk = 0;
        for (u32 pr = 0; pr < npairs; pr++) {
            const u32 i = pr * 2;
            load_term(cur, i + 1, B);
            mac_term(A);
            {   // what follows B: this chain's next term, else the NEXT chain's first term (it rides behind this chain's last one),
                // else a re-read of the last term (hits the cache).  Scalar selects on the uniform bases, one load sequence.
                const bool inside = i + 2 < cnt, other = !inside && has_next;
                Unit src;
                src.p0 = other ? nxt.p0 : cur.p0;
                src.pw_stride = other ? nxt.pw_stride : cur.pw_stride; src.pw_poly_stride = other ? nxt.pw_poly_stride : cur.pw_poly_stride;
                src.pt_stride = other ? nxt.pt_stride : cur.pt_stride;
#pragma unroll
                for (int g = 0; g < G; g++) src.pt[g] = other ? nxt.pt[g] : cur.pt[g];
                load_term(src, inside ? i + 2 : (other ? 0u : cnt - 1), A);
            }
            mac_term(B);
            in_chunk += 2;
            if (in_chunk + 3 > chunk) { fold(false); in_chunk = 1; }
        }
        if (cnt & 1) {                                           // A holds the last term; the next chain's first goes to B meanwhile
            if (has_next) load_term(nxt, 0, B);
            mac_term(A);
        }
        fold(true);
#pragma unroll
        for (int g = 0; g < G; g++) {
            if ((u32)g < cur.live) {
                const MacJobC cj = (MacJobC)(uintptr_t)cur.jp;
                u64 *o = cj->out[cur.g0 + g] + (size_t)cur.y * n + k;
                const u32 ops = cj->out_poly_stride;
                if (C == 2) {
                    u64x2 r0, r1;
                    r0[0] = s00[g][0][0]; r0[1] = s00[g][C - 1][0];
                    r1[0] = s00[g][0][1]; r1[1] = s00[g][C - 1][1];
                    *reinterpret_cast<u64x2 *>(o) = r0;
                    *reinterpret_cast<u64x2 *>(o + ops) = r1;
                } else {
                    o[0] = s00[g][0][0];
                    o[ops] = s00[g][0][1];
                }
            }
        }
        if (!has_next) break;
        if (cnt & 1) A = B;
        cur = nxt; z = z2; y = y2;
    }
}


static int g_mac_persist_r = 0;
inline void launch_mac_persist(const DevLevel *lv, const MacJob *jobs, size_t n, int njobs, hipStream_t st, bool kara, int r_req)
{
    constexpr int G = APSU_MAC_G, C = APSU_MAC_C;
    const unsigned gx = (unsigned)((n / C + EW_T - 1) / EW_T), nz = (unsigned)(njobs * (MAC_G / G));
    unsigned r = r_req > 0 ? (unsigned)r_req : std::max(1u, 2u * 256u / gx);
    r = std::min(r, nz);
    if (kara) hipLaunchKernelGGL((k_mac_p<G, C, true>), dim3(gx, r), dim3(EW_T), 0, st, lv, jobs, n, nz);
    else hipLaunchKernelGGL((k_mac_p<G, C, false>), dim3(gx, r), dim3(EW_T), 0, st, lv, jobs, n, nz);
}
} // namespace apsu_he
