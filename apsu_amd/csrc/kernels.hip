// Hand-written gfx950 kernels of the homomorphic query-evaluation engine.
// Each kernel names the seal::Evaluator step it replaces and the reference call sites.
// Conventions: 64-lane waves, 256-thread workgroups for coefficient-parallel kernels (one thread
// per coefficient index, consecutive lanes on consecutive coefficients -> 512 B per wave access),
// level constants read through wave-uniform loads.  No MFMA: the work is modular-integer
// butterflies and dyadic products (BASELINE.json north_star).
#include "device.h"
#include "ntt_wg.h"
#include <cstdlib>
#include <type_traits>

#include <algorithm>

namespace apsu_he {

#define KERNEL_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__); } while (0)
void throw_hip(hipError_t e, const char *file, int line);

// ============================================================================ K1/K2: NTT
// transform_to_ntt_inplace / transform_from_ntt_inplace
// (receiver_osn.cpp:467,475 ; bin_bundle.cpp:154,268,297,321) and every NTT inside
// multiply / relinearize / multiply_plain.  One workgroup per limb polynomial, limb resident in LDS.
// (the workgroup body -- passes + synchronisation -- lives in ntt_wg.h)
// split != 0: the launch transforms the two HALVES of limbs twice this size (poly_modulus_degree 32768: one limb is 256 KiB and
// does not fit a workgroup's LDS).  Block g is half g & 1 of limb g >> 1 and takes table 2 * modulus + half, whose forward
// twiddles are the big transform's for that half (W_h[2^s + b] = W[2^(s+1) + h 2^s + b]); the first Cooley-Tukey stage runs in
// front (k_ntt_first_stage), the inverse's last stage and twist behind (k_ntt_last_stage), so the inverse halves are always RAW.
// C: coefficients per lane -- 16 (T = n / 16 threads per limb: the throughput form) or 8 (T = n / 8: the latency form of round 6 for
// launches that leave CUs idle, ntt_core.h plan_k; chosen by the launch wrappers from the number of limbs, same bits).
// MINW: waves per SIMD the register budget must admit (8: <= 64 VGPRs, two 1024-thread workgroups per CU at n = 8192 -- large forward launches
// over narrow moduli only: -4 ... -7 %, tools/microbench/ntt_forms.hip; slower on 61-bit limbs and on every inverse)
template <int LOGN, bool INV, int T, int C = 16, int MINW = 4>
__global__ __launch_bounds__(T, MINW) void k_ntt(u64 *__restrict__ data, const NttTable *__restrict__ tabs,
                                           const int *__restrict__ modmap, int period, int split)
{
    constexpr int N = 1 << LOGN;
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const int tid = threadIdx.x;
    const size_t g = blockIdx.x;
    const int mv = modmap[(split ? g >> 1 : g) % (size_t)period];
    const NttTable tab = tabs[split ? (((mv & NTT_MAP_MASK) << 1) | (int)(g & 1)) : (mv & NTT_MAP_MASK)];
    u64 *p = data + g * N;
    // (the inverse transform stages its limb into LDS with coalesced loads: SrcStaged, ntt_core.h)
    using SRC = std::conditional_t<INV, SrcStaged, SrcPlain>;
    if (INV && ((mv & NTT_MAP_RAW) || split)) {                                 // wave-uniform branches
        if (tab.narrow) ntt_body<LOGN, INV, NTT_NARROW, T, 0, INV, SRC, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SRC());
        else if (tab.wide_d4) ntt_body<LOGN, INV, NTT_WIDE_NEAR, T, 0, INV, SRC, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SRC());
        else ntt_body<LOGN, INV, NTT_WIDE, T, 0, INV, SRC, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SRC());
        return;
    }
    if (tab.narrow) ntt_body<LOGN, INV, NTT_NARROW, T, 0, false, SRC, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SRC());
    else if (tab.wide_d4) ntt_body<LOGN, INV, NTT_WIDE_NEAR, T, 0, false, SRC, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SRC());
    else ntt_body<LOGN, INV, NTT_WIDE, T, 0, false, SRC, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SRC());
}

constexpr int EW_T = 256;                                         // threads per workgroup of the coefficient-parallel kernels
// ---- poly_modulus_degree 32768: one radix-2 stage over global memory around two half-size LDS-resident transforms.
// Tables: entry 2 m (+ 1) of `tabs`; its ninv / ninv_q fields carry the first stage's twiddle psi^brv(1) (the twist comes
// from the scale table), dit and scale are the full-size tables.
constexpr int SPLIT_LOGN = 15;
// forward, first Cooley-Tukey stage: (x_j, x_{j + n/2}) -> (x + w y, x - w y), canonical; src != nullptr: limb g is gathered from
// src[g] (residues of another modulus, reduced on load: the key switch's decomposition)
__global__ __launch_bounds__(EW_T) void k_ntt_first_stage(const u64 *const *__restrict__ src, u64 *__restrict__ data, size_t n,
                                                          const NttTable *__restrict__ tabs, const int *__restrict__ modmap, int period)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x, h = n >> 1;
    if (k >= h) return;
    const size_t g = blockIdx.y;
    const NttTable tab = tabs[(modmap[g % (size_t)period] & NTT_MAP_MASK) << 1];
    const u64 *in = src ? src[g] : data + g * n;
    u64 x = in[k], y = in[k + h];
    if (src) { x = ntt_reduce_any(x, tab); y = ntt_reduce_any(y, tab); }
    const u64 v = mul_shoup(y, tab.ninv, tab.ninv_q, tab.q);
    u64 *out = data + g * n;
    out[k] = addmod(x, v, tab.q);
    out[k + h] = submod(x, v, tab.q);
}
// inverse, last decimation-in-time stage (E_j + w_j O_j, E_j - w_j O_j with w_j = psi^-2j) on the RAW outputs of the two
// half transforms, then the twist n^-1 psi^-j -- or, for a NTT_MAP_RAW limb, the raw sums for a consumer that applies it
__global__ __launch_bounds__(EW_T) void k_ntt_last_stage(u64 *__restrict__ data, size_t n, const NttTable *__restrict__ tabs,
                                                         const int *__restrict__ modmap, int period)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x, h = n >> 1;
    if (k >= h) return;
    const size_t g = blockIdx.y;
    const int mv = modmap[g % (size_t)period];
    const NttTable tab = tabs[(mv & NTT_MAP_MASK) << 1];
    u64 *p = data + g * n;
    const u64 q = tab.q;
    const u64x2 tw = ldg16(reinterpret_cast<const u64 *>(tab.dit + h + k));
    const u64 e = ntt_reduce_any(p[k], tab);                                    // [0, q)
    const u64 o = mul_shoup_lazy(p[k + h], tw[0], tw[1], q);                    // [0, 2q)
    u64 x0 = e + o, x1 = e + (q << 1) - o;                                      // < 3q, any consumer of RAW values takes them
    if (!(mv & NTT_MAP_RAW)) {
        const u64x2 s0 = ldg16(reinterpret_cast<const u64 *>(tab.scale + k)), s1 = ldg16(reinterpret_cast<const u64 *>(tab.scale + k + h));
        x0 = mul_shoup(x0, s0[0], s0[1], q);
        x1 = mul_shoup(x1, s1[0], s1[1], q);
    }
    p[k] = x0;
    p[k + h] = x1;
}

// Forward NTT of gathered limbs: limb g is read from src[g] (residues of another modulus, reduced on load) and written
// to data + g*N.  Replaces the decompose kernel of the key switch (App. B10): out[I][J] = NTT_I(c2_J mod m_I).
// MINW: waves per SIMD the register budget must admit (8: <= 64 VGPRs, two 1024-thread workgroups per CU at n = 8192 -- the form large
// gathered launches take since round 6: -1 ... -5 %, tools/microbench/ntt_forms.hip)
template <int LOGN, int T, int C = 16, int MINW = 4>
__global__ __launch_bounds__(T, MINW) void k_ntt_gather(const u64 *const *__restrict__ src, u64 *__restrict__ data,
                                                  const NttTable *__restrict__ tabs, const int *__restrict__ modmap, int period, int nored)
{
    constexpr int N = 1 << LOGN;
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const int tid = threadIdx.x;
    const size_t g = blockIdx.x;
    const NttTable tab = tabs[modmap[g % (size_t)period]];
    u64 *p = data + g * N;
    // nored: every source residue fits the lazy range of every (narrow) target as it stands (checked by the host, ntt_gather_nored_ok):
    // the transform is linear and its closing reduction takes any 64-bit value, so the reduction on load is left out
    if (tab.narrow) {
        if (nored) ntt_body<LOGN, false, NTT_NARROW, T, 2, false, SrcPlain, true, false, 0, -1, C>(lds, p, tab, tid, src[g]);
        else ntt_body<LOGN, false, NTT_NARROW, T, 1, false, SrcPlain, true, false, 0, -1, C>(lds, p, tab, tid, src[g]);
    } else if (tab.wide_d4) ntt_body<LOGN, false, NTT_WIDE_NEAR, T, 1, false, SrcPlain, true, false, 0, -1, C>(lds, p, tab, tid, src[g]);
    else ntt_body<LOGN, false, NTT_WIDE, T, 1, false, SrcPlain, true, false, 0, -1, C>(lds, p, tab, tid, src[g]);
}

// Which form of the workgroup does a launch of `count` limbs take?  latency_limbs: 0 = always 16 coefficients per lane; NTT_FORM_AUTO = the
// crossovers measured with tools/microbench/ntt_forms.hip (profiles/r06_ntt_forms_n8192.txt, _n4096.txt): at n = 8192 a limb's
// 1024-thread workgroup wins while a CU gets at most one limb (<= 256 limbs: -3 ... -16 %) and loses above (+2 ... +20 %); at n = 4096
// (512 threads, registers for 7-8 waves per SIMD) the forward, gathered and tensor-on-load transforms win at every size (-2 ... -24 %), the
// plain inverse up to 1 024 limbs; any other value = that threshold for every kind (tests force either form with it).
enum NttKind { NTT_KIND_FORWARD, NTT_KIND_INVERSE, NTT_KIND_GATHER, NTT_KIND_TENSOR };
static bool ntt_use_latency_form(int logn, NttKind kind, size_t count, size_t latency_limbs)
{
    if (!plan_has_latency_form(logn) || latency_limbs == 0) return false;
    if (latency_limbs != NTT_FORM_AUTO) return count <= latency_limbs;
    if (logn == 13) return count <= 256;
    return kind != NTT_KIND_INVERSE || count <= 1024;
}

void launch_ntt_gather(int logn, const u64 *const *src, u64 *data, size_t count, const NttTable *tabs, const int *modmap, int period,
                       hipStream_t st, bool nored, size_t latency_limbs)
{
    const int nr = nored ? 1 : 0;
    if (!count) return;
    if (ntt_use_latency_form(logn, NTT_KIND_GATHER, count, latency_limbs)) {
        if (logn == 13) hipLaunchKernelGGL((k_ntt_gather<13, 1024, 8>), dim3((unsigned)count), dim3(1024), 0, st, src, data, tabs, modmap, period, nr);
        else hipLaunchKernelGGL((k_ntt_gather<12, 512, 8>), dim3((unsigned)count), dim3(512), 0, st, src, data, tabs, modmap, period, nr);
        KERNEL_CHECK();
        return;
    }
    if (latency_limbs == NTT_FORM_AUTO && logn == 13) {          // large gathered launches at n = 8192: 8 coefficients per lane at 8 waves per SIMD
        hipLaunchKernelGGL((k_ntt_gather<13, 1024, 8, 8>), dim3((unsigned)count), dim3(1024), 0, st, src, data, tabs, modmap, period, nr);
        KERNEL_CHECK();
        return;
    }
    if (logn == 15) {                                            // first stage gathers and reduces, the halves are plain transforms
        const size_t n = (size_t)1 << 15;
        hipLaunchKernelGGL(k_ntt_first_stage, dim3((unsigned)((n / 2 + EW_T - 1) / EW_T), (unsigned)count), dim3(EW_T), 0, st, src, data, n, tabs, modmap, period);
        hipLaunchKernelGGL((k_ntt<14, false, 1024>), dim3((unsigned)(count * 2)), dim3(1024), 0, st, data, tabs, modmap, period, 1);
        KERNEL_CHECK();
        return;
    }
#define G_CASE(LN, T) case LN: hipLaunchKernelGGL((k_ntt_gather<LN, T>), dim3((unsigned)count), dim3(T), 0, st, src, data, tabs, modmap, period, nr); break;
    switch (logn) {
    G_CASE(14, 1024) G_CASE(13, 512) G_CASE(12, 256) G_CASE(11, 128) G_CASE(10, 64) G_CASE(8, 64) G_CASE(6, 64)
    default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    }
#undef G_CASE
    KERNEL_CHECK();
}

// Inverse NTT of dyadic tensor products (BEHZ steps 4 + 5 in one launch): workgroup g < n_tensor owns output limb
// (job, poly p of 3, limb e of `limbs`) of product job = g / (3*limbs), forms d_p = a0*b0 | a0*b1 + a1*b0 | a1*b1 in limb e from
// the NTT-form operands while it loads (SrcTensor) and writes the coefficient-form limb to job.d[p][e]: the tensor kernel,
// its 3*E limb writes and the transform's re-read of them are gone.  Workgroups g >= n_tensor transform
// plain[g - n_tensor] in place (limbs that join the same launch).  Operand polys are src_ps words apart.
template <int LOGN, int T, int C = 16>
__global__ __launch_bounds__(T, 4) void k_intt_tensor(const TensorJob *__restrict__ jobs, int limbs, size_t src_ps, size_t n_tensor,
                                                   u64 *__restrict__ plain, const NttTable *__restrict__ tabs,
                                                   const int *__restrict__ modmap, int period)
{
    constexpr int N = 1 << LOGN;
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const int tid = threadIdx.x;
    size_t g = blockIdx.x;
    {
        // XCD-aware order: the three workgroups of one (product, limb) pair -- which read the same operand limbs -- sit 8 apart in
        // the grid (workgroups b and b + 8 share an XCD, hence an L2): pair u = 8 blk + lane, polynomial pl at 24 blk + 8 pl + lane.
        // (One product per block with lane = limb, so that all workgroups of limb e share an XCD and products with a common parent
        //  find it in that L2 too, measured level against this: profiles/r03_ab_xcd.txt.)
        const size_t pairs = n_tensor / 3, padded = (pairs + 7) / 8 * 24;
        if (g < padded) {
            const size_t blk = g / 24, rem = g - blk * 24, pl = rem >> 3, u = blk * 8 + (rem & 7);
            if (u >= pairs) return;                                             // wave-uniform: the padding of the last block of 8 pairs
            const size_t jb = u / (size_t)limbs, e = u - jb * limbs;
            g = jb * 3 * limbs + pl * limbs + e;                                // the logical index every table below is laid out by
        } else {
            g = n_tensor + (g - padded);
        }
    }
    const int mv = modmap[g % (size_t)period];
    const NttTable tab = tabs[mv & NTT_MAP_MASK];
    if (g >= n_tensor) {                                                        // wave-uniform
        u64 *p = plain + (g - n_tensor) * N;
        if (mv & NTT_MAP_RAW) {
            if (tab.narrow) ntt_body<LOGN, true, NTT_NARROW, T, 0, true, SrcStaged, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SrcStaged());
            else if (tab.wide_d4) ntt_body<LOGN, true, NTT_WIDE_NEAR, T, 0, true, SrcStaged, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SrcStaged());
            else ntt_body<LOGN, true, NTT_WIDE, T, 0, true, SrcStaged, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SrcStaged());
        } else {
            if (tab.narrow) ntt_body<LOGN, true, NTT_NARROW, T, 0, false, SrcStaged, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SrcStaged());
            else if (tab.wide_d4) ntt_body<LOGN, true, NTT_WIDE_NEAR, T, 0, false, SrcStaged, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SrcStaged());
            else ntt_body<LOGN, true, NTT_WIDE, T, 0, false, SrcStaged, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, SrcStaged());
        }
        return;
    }
    const size_t per = (size_t)3 * limbs, jb = g / per;
    const int r = (int)(g - jb * per), pl = r / limbs, e = r - pl * limbs;
    const TensorJob job = jobs[jb];
    const u64 *a0 = job.a + (size_t)e * N, *a1 = a0 + src_ps, *b0 = job.b + (size_t)e * N, *b1 = b0 + src_ps;
    // the fold's last word (< 4q) enters the transform as it is where the range discipline takes that (-1.05 % query, profiles/r04_ab_tensor_lazy.txt)
    const bool lazy = ntt_lazy_input_ok(tab, LOGN);
    SrcTensor ops;
    if (pl == 0) ops = SrcTensor{ a0, b0, nullptr, nullptr, lazy };
    else if (pl == 1) ops = SrcTensor{ a0, b1, a1, b0, lazy };
    else ops = SrcTensor{ a1, b1, nullptr, nullptr, lazy };
    u64 *p = job.d + (size_t)r * N;
    if (mv & NTT_MAP_RAW) {
        if (tab.narrow) ntt_body<LOGN, true, NTT_NARROW, T, 0, true, SrcTensor, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, ops);
        else if (tab.wide_d4) ntt_body<LOGN, true, NTT_WIDE_NEAR, T, 0, true, SrcTensor, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, ops);
        else ntt_body<LOGN, true, NTT_WIDE, T, 0, true, SrcTensor, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, ops);
    } else {
        if (tab.narrow) ntt_body<LOGN, true, NTT_NARROW, T, 0, false, SrcTensor, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, ops);
        else if (tab.wide_d4) ntt_body<LOGN, true, NTT_WIDE_NEAR, T, 0, false, SrcTensor, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, ops);
        else ntt_body<LOGN, true, NTT_WIDE, T, 0, false, SrcTensor, true, false, 0, -1, C>(lds, p, tab, tid, nullptr, ops);
    }
}

void launch_intt_tensor(int logn, const TensorJob *jobs, int njobs, int limbs, size_t src_ps, u64 *plain, size_t n_plain,
                        const NttTable *tabs, const int *modmap, int period, hipStream_t st, size_t latency_limbs)
{
    const size_t n_tensor = (size_t)njobs * 3 * limbs;
    const size_t count = (n_tensor / 3 + 7) / 8 * 24 + n_plain;                 // the XCD-aware order pads the pairs to blocks of eight
    if (!(n_tensor + n_plain)) return;
    if (ntt_use_latency_form(logn, NTT_KIND_TENSOR, n_tensor + n_plain, latency_limbs)) {
        if (logn == 13) hipLaunchKernelGGL((k_intt_tensor<13, 1024, 8>), dim3((unsigned)count), dim3(1024), 0, st, jobs, limbs, src_ps, n_tensor, plain, tabs, modmap, period);
        else hipLaunchKernelGGL((k_intt_tensor<12, 512, 8>), dim3((unsigned)count), dim3(512), 0, st, jobs, limbs, src_ps, n_tensor, plain, tabs, modmap, period);
        KERNEL_CHECK();
        return;
    }
#define T_CASE(LN, T) case LN: hipLaunchKernelGGL((k_intt_tensor<LN, T>), dim3((unsigned)count), dim3(T), 0, st, jobs, limbs, src_ps, n_tensor, plain, tabs, modmap, period); break;
    switch (logn) {
    T_CASE(14, 1024) T_CASE(13, 512) T_CASE(12, 256) T_CASE(11, 128) T_CASE(10, 64) T_CASE(8, 64) T_CASE(6, 64)
    default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    }
#undef T_CASE
    KERNEL_CHECK();
}

template <int LOGN, int T>
static void launch_ntt_t(bool inverse, u64 *data, size_t count, const NttTable *tabs, const int *modmap, int period,
                         hipStream_t st, int split = 0)
{
    if (inverse) hipLaunchKernelGGL((k_ntt<LOGN, true, T>), dim3((unsigned)count), dim3(T), 0, st, data, tabs, modmap, period, split);
    else hipLaunchKernelGGL((k_ntt<LOGN, false, T>), dim3((unsigned)count), dim3(T), 0, st, data, tabs, modmap, period, split);
}

static void launch_ntt_split(bool inverse, const u64 *const *src, u64 *data, size_t count, const NttTable *tabs, const int *modmap, int period,
                             hipStream_t st)
{
    const size_t n = (size_t)1 << SPLIT_LOGN;
    const dim3 g((unsigned)((n / 2 + EW_T - 1) / EW_T), (unsigned)count);
    if (!inverse) hipLaunchKernelGGL(k_ntt_first_stage, g, dim3(EW_T), 0, st, src, data, n, tabs, modmap, period);
    launch_ntt_t<SPLIT_LOGN - 1, 1024>(inverse, data, count * 2, tabs, modmap, period, st, 1);
    if (inverse) hipLaunchKernelGGL(k_ntt_last_stage, g, dim3(EW_T), 0, st, data, n, tabs, modmap, period);
}

void launch_ntt(int logn, bool inverse, u64 *data, size_t count, const NttTable *tabs, const int *modmap, int period,
                hipStream_t st, size_t latency_limbs, bool narrow_only)
{
    if (!count) return;
    if (latency_limbs == NTT_FORM_AUTO && logn == 13 && !inverse && narrow_only && count > 256) {
        // large forward launches whose moduli are all narrow (the data primes): 8 coefficients per lane at 8 waves per SIMD
        hipLaunchKernelGGL((k_ntt<13, false, 1024, 8, 8>), dim3((unsigned)count), dim3(1024), 0, st, data, tabs, modmap, period, 0);
        KERNEL_CHECK();
        return;
    }
    if (ntt_use_latency_form(logn, inverse ? NTT_KIND_INVERSE : NTT_KIND_FORWARD, count, latency_limbs)) {   // 8 coefficients per lane
        if (logn == 13) {
            if (inverse) hipLaunchKernelGGL((k_ntt<13, true, 1024, 8>), dim3((unsigned)count), dim3(1024), 0, st, data, tabs, modmap, period, 0);
            else hipLaunchKernelGGL((k_ntt<13, false, 1024, 8>), dim3((unsigned)count), dim3(1024), 0, st, data, tabs, modmap, period, 0);
        } else {
            if (inverse) hipLaunchKernelGGL((k_ntt<12, true, 512, 8>), dim3((unsigned)count), dim3(512), 0, st, data, tabs, modmap, period, 0);
            else hipLaunchKernelGGL((k_ntt<12, false, 512, 8>), dim3((unsigned)count), dim3(512), 0, st, data, tabs, modmap, period, 0);
        }
        KERNEL_CHECK();
        return;
    }
    switch (logn) {
    case SPLIT_LOGN: launch_ntt_split(inverse, nullptr, data, count, tabs, modmap, period, st); break;
    case 14: launch_ntt_t<14, 1024>(inverse, data, count, tabs, modmap, period, st); break;
    case 13: launch_ntt_t<13, 512>(inverse, data, count, tabs, modmap, period, st); break;
    case 12: launch_ntt_t<12, 256>(inverse, data, count, tabs, modmap, period, st); break;
    case 11: launch_ntt_t<11, 128>(inverse, data, count, tabs, modmap, period, st); break;
    case 10: launch_ntt_t<10, 64>(inverse, data, count, tabs, modmap, period, st); break;
    case 8: launch_ntt_t<8, 64>(inverse, data, count, tabs, modmap, period, st); break;
    case 6: launch_ntt_t<6, 64>(inverse, data, count, tabs, modmap, period, st); break;
    default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    }
    KERNEL_CHECK();
}

// ============================================================================ coefficient-parallel helpers
static inline dim3 ew_grid(size_t n, int batch) { return dim3((unsigned)((n + EW_T - 1) / EW_T), (unsigned)batch); }

// K3 (single term): multiply_plain on NTT ct x NTT plaintext (bin_bundle.cpp:147,258,287,320)
__global__ __launch_bounds__(EW_T) void k_dyadic_plain(const DevLevel *__restrict__ lv, const u64 *__restrict__ ct,
                                                       const u64 *__restrict__ pt, u64 *__restrict__ out, int polys,
                                                       size_t n, size_t pt_batch_stride)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const int L = lv->L;
    const size_t b = blockIdx.y;
    const u64 *a = ct + b * polys * L * n;
    const u64 *p = pt + b * pt_batch_stride;
    u64 *o = out + b * polys * L * n;
    for (int j = 0; j < L; j++) {
        const Mod m = lv->q[j];
        const u64 pv = p[j * n + k];
        for (int c = 0; c < polys; c++) o[(c * L + j) * n + k] = mulmod(a[(c * L + j) * n + k], pv, m);
    }
}

void launch_dyadic_plain(const DevLevel *lv, const u64 *ct, const u64 *pt, u64 *out, int polys, size_t n, int batch,
                         size_t pt_batch_stride, hipStream_t st)
{
    hipLaunchKernelGGL(k_dyadic_plain, ew_grid(n, batch), dim3(EW_T), 0, st, lv, ct, pt, out, polys, n, pt_batch_stride);
    KERNEL_CHECK();
}

// K10: add_inplace (bin_bundle.cpp:148,264,273,293,303,323,336)
__global__ __launch_bounds__(EW_T) void k_add(const DevLevel *__restrict__ lv, u64 *__restrict__ acc,
                                              const u64 *__restrict__ x, int polys, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const int L = lv->L;
    const size_t off = (size_t)blockIdx.y * polys * L * n;
    for (int j = 0; j < L; j++) {
        const u64 q = lv->q[j].q;
        for (int c = 0; c < polys; c++) {
            const size_t i = off + (c * L + j) * n + k;
            acc[i] = addmod(acc[i], x[i], q);
        }
    }
}

void launch_add(const DevLevel *lv, u64 *acc, const u64 *x, int polys, size_t n, int batch, hipStream_t st)
{
    hipLaunchKernelGGL(k_add, ew_grid(n, batch), dim3(EW_T), 0, st, lv, acc, x, polys, n);
    KERNEL_CHECK();
}

// acc[b] += sum_i x[b][i]  — exact modular sum of `terms` ciphertexts (order-insensitive)
__global__ __launch_bounds__(EW_T) void k_add_many(const DevLevel *__restrict__ lv, u64 *__restrict__ acc, size_t acc_stride,
                                                   const u64 *__restrict__ x, int terms, int polys, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const int L = lv->L;
    const size_t b = blockIdx.y;
    const size_t ctw = (size_t)polys * L * n;
    for (int j = 0; j < L; j++) {
        const u64 q = lv->q[j].q;
        for (int c = 0; c < polys; c++) {
            const size_t o = (c * L + j) * n + k;
            u64 s = acc[b * acc_stride + o];
            for (int i = 0; i < terms; i++) s = addmod(s, x[(b * terms + i) * ctw + o], q);
            acc[b * acc_stride + o] = s;
        }
    }
}

void launch_add_many(const DevLevel *lv, u64 *acc, size_t acc_stride, const u64 *x, int terms, int polys, size_t n,
                     int batch, hipStream_t st)
{
    hipLaunchKernelGGL(k_add_many, ew_grid(n, batch), dim3(EW_T), 0, st, lv, acc, acc_stride, x, terms, polys, n);
    KERNEL_CHECK();
}

// dst[polys][L][n] = sum over `terms` consecutive ciphertexts src[t][polys][L][n]  (exact modular sums)
__global__ __launch_bounds__(EW_T) void k_sum_jobs(const DevLevel *__restrict__ lv, const SumJob *__restrict__ jobs, int polys, size_t n)
{
    const size_t k = ((size_t)blockIdx.x * EW_T + threadIdx.x) * 2;
    if (k >= n) return;
    const int L = lv->L;
    const SumJob job = jobs[blockIdx.y / (polys * L)];
    const int pl = blockIdx.y % (polys * L);              // (poly, limb) pair
    const u64 q = lv->q[pl % L].q;
    const size_t ctw = (size_t)polys * L * n;
    const u64 *src = job.src + (size_t)pl * n + k;
    u64 s0 = 0, s1 = 0;
    for (int t = 0; t < job.terms; t++) {
        const u64x2 v = ldg16(src + (size_t)t * ctw);
        s0 = addmod(s0, v[0], q);
        s1 = addmod(s1, v[1], q);
    }
    u64x2 r; r[0] = s0; r[1] = s1;
    *reinterpret_cast<u64x2 *>(job.dst + (size_t)pl * n + k) = r;
}

void launch_sum_jobs(const DevLevel *lv, int L, const SumJob *jobs, int polys, size_t n, int njobs, hipStream_t st)
{
    if (!njobs) return;
    hipLaunchKernelGGL(k_sum_jobs, dim3((unsigned)((n / 2 + EW_T - 1) / EW_T), (unsigned)(njobs * polys * L)), dim3(EW_T), 0, st,
                       lv, jobs, polys, n);
    KERNEL_CHECK();
}

// K8: add_plain_inplace (bin_bundle.cpp:159,162,345,346): c0 += round(m*Q/t) in RNS  (App. B7)
__global__ __launch_bounds__(EW_T) void k_add_plain(const DevLevel *__restrict__ lv, const PlainJob *__restrict__ jobs, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const PlainJob job = jobs[blockIdx.y];
    const u64 m = job.pt[k];
    // fix = floor((m * (Q mod t) + floor((t+1)/2)) / t), exact 128-by-64 division (m < t < 2^61)
    u128p num = mul128(m, lv->q_mod_t);
    add128(num, u128p{ lv->threshold, 0 });
    const u64 t = lv->t;
    u64 rem = num.hi % t, lo = num.lo, fix = 0;
    if (num.hi == 0) {
        fix = lo / t;
    } else {
        for (int i = 0; i < 64; i++) {                  // shift-subtract; only for t > 2^32
            rem = (rem << 1) | (lo >> 63);
            lo <<= 1;
            fix <<= 1;
            if (rem >= t) { rem -= t; fix |= 1; }
        }
    }
    u64 *c0 = job.ct;
    for (int j = 0; j < lv->L; j++) {
        const Mod mq = lv->q[j];
        u128p s = mul128(m, lv->coeff_div_plain[j]);
        add128(s, u128p{ fix, 0 });
        const u64 scaled = barrett128(s, mq);
        c0[j * n + k] = addmod(c0[j * n + k], scaled, mq.q);
    }
}

void launch_add_plain(const DevLevel *lv, const PlainJob *jobs, size_t n, int batch, hipStream_t st)
{
    if (!batch) return;
    hipLaunchKernelGGL(k_add_plain, ew_grid(n, batch), dim3(EW_T), 0, st, lv, jobs, n);
    KERNEL_CHECK();
}

// K4 (first half): plaintext lift mod t -> RNS (App. B5); NTT follows as a separate launch.
// no_lift[b] != 0 selects SEAL's monomial shortcut (value copied unlifted).
__global__ __launch_bounds__(EW_T) void k_lift(const DevLevel *__restrict__ lv, const u64 *__restrict__ pt,
                                               u64 *__restrict__ out, size_t n, const unsigned char *__restrict__ no_lift)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const size_t b = blockIdx.y;
    const u64 v = pt[b * n + k];
    const bool lift = (v >= lv->threshold) && !(no_lift && no_lift[b]);
    const int L = lv->L;
    for (int j = 0; j < L; j++) out[(b * L + j) * n + k] = lift ? v + lv->incr[j] : v;
}

void launch_lift(const DevLevel *lv, const u64 *pt, u64 *out, size_t n, int batch, const unsigned char *no_lift,
                 hipStream_t st)
{
    hipLaunchKernelGGL(k_lift, ew_grid(n, batch), dim3(EW_T), 0, st, lv, pt, out, n, no_lift);
    KERNEL_CHECK();
}

// K7: mod_switch_to_next_inplace / mod_switch_to_inplace
// (receiver_osn.cpp:463,471,478 ; bin_bundle.cpp:169,269,298,322,335,355): drop q_last with rounding (B8)
// in: ciphertext c at in + c*in_stride holds `polys` polynomials [L][n]; out: packed [c][polys][L-1][n]
__global__ __launch_bounds__(EW_T) void k_modswitch(const DevLevel *__restrict__ lv, const u64 *__restrict__ in,
                                                    size_t in_stride, int polys, u64 *__restrict__ out, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const int L = lv->L;
    const size_t c = blockIdx.y / polys, p = blockIdx.y % polys;
    const u64 *src = in + c * in_stride + p * (size_t)L * n;
    u64 *dst = out + (size_t)blockIdx.y * (L - 1) * n;
    const u64 ql = lv->q[L - 1].q;
    const u64 last = addmod(src[(size_t)(L - 1) * n + k], lv->half, ql);
    for (int j = 0; j + 1 < L; j++) {
        const Mod m = lv->q[j];
        const u64 tmp = submod(barrett64(last, m), lv->half_mod[j], m.q);
        const u64 v = submod(src[(size_t)j * n + k], tmp, m.q);
        dst[(size_t)j * n + k] = mul_shoup(v, lv->inv_q_last[j].w, lv->inv_q_last[j].wq, m.q);
    }
}

void launch_modswitch(const DevLevel *lv, const u64 *in, size_t in_stride, int polys, u64 *out, size_t n, int cts,
                      hipStream_t st)
{
    if (!cts) return;
    hipLaunchKernelGGL(k_modswitch, ew_grid(n, cts * polys), dim3(EW_T), 0, st, lv, in, in_stride, polys, out, n);
    KERNEL_CHECK();
}

// job-addressed variant used when the source ciphertexts are scattered (PowersDag slot order)
__global__ __launch_bounds__(EW_T) void k_modswitch_jobs(const DevLevel *__restrict__ lv, const CtJob *__restrict__ jobs,
                                                         int polys, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const int L = lv->L;
    const CtJob job = jobs[blockIdx.y / polys];
    const size_t p = blockIdx.y % polys;
    const u64 *src = job.src + p * (size_t)L * n;
    u64 *dst = job.dst + p * (size_t)(L - 1) * n;
    const u64 ql = lv->q[L - 1].q;
    const u64 last = addmod(src[(size_t)(L - 1) * n + k], lv->half, ql);
    for (int j = 0; j + 1 < L; j++) {
        const Mod m = lv->q[j];
        const u64 tmp = submod(barrett64(last, m), lv->half_mod[j], m.q);
        const u64 v = submod(src[(size_t)j * n + k], tmp, m.q);
        dst[(size_t)j * n + k] = mul_shoup(v, lv->inv_q_last[j].w, lv->inv_q_last[j].wq, m.q);
    }
}

void launch_modswitch_jobs(const DevLevel *lv, const CtJob *jobs, int polys, size_t n, int njobs, hipStream_t st)
{
    if (!njobs) return;
    hipLaunchKernelGGL(k_modswitch_jobs, ew_grid(n, njobs * polys), dim3(EW_T), 0, st, lv, jobs, polys, n);
    KERNEL_CHECK();
}

__global__ __launch_bounds__(EW_T) void k_copy_jobs(const CtJob *__restrict__ jobs, size_t words)
{
    const CtJob job = jobs[blockIdx.y];
    for (size_t k = ((size_t)blockIdx.x * EW_T + threadIdx.x) * 2; k < words; k += (size_t)gridDim.x * EW_T * 2)
        *reinterpret_cast<u64x2 *>(job.dst + k) = *reinterpret_cast<const u64x2 *>(job.src + k);
}

// The same for the source ciphertexts of a query ([2][L][n] each): while it copies, every word is held against the prime of its limb
// (seal::is_data_valid_for); a word outside [0, q) writes the query's sequence number `seq` to *bad -- a word of page-locked host memory
// the engine looks at when it next waits for the device (one word per query in flight, so the report names the query: round 6).  The lazy
// transforms take source limbs as they are, so such a word must not pass silently.  src == dst: a check in place (sources that came from
// the host by a plain copy).
__global__ __launch_bounds__(EW_T) void k_copy_sources(const CtJob *__restrict__ jobs, size_t words, const DevLevel *__restrict__ lv, int L, size_t n,
                                                       unsigned *__restrict__ bad, unsigned seq)
{
    const CtJob job = jobs[blockIdx.y];
    bool wrong = false;
    for (size_t k = ((size_t)blockIdx.x * EW_T + threadIdx.x) * 2; k < words; k += (size_t)gridDim.x * EW_T * 2) {
        const u64x2 v = *reinterpret_cast<const u64x2 *>(job.src + k);
        const u64 q = lv->q[(k / n) % (size_t)L].q;                 // (n is even: both words lie in the same limb)
        wrong |= v[0] >= q || v[1] >= q;
        if (job.dst != job.src) *reinterpret_cast<u64x2 *>(job.dst + k) = v;
    }
    if (wrong) atomicMax(bad, seq);
}

void launch_copy_sources(const CtJob *jobs, size_t words, int njobs, const DevLevel *lv, int L, size_t n, unsigned *bad, unsigned seq, hipStream_t st)
{
    if (!njobs) return;
    unsigned gx = (unsigned)std::min<size_t>((words / 2 + EW_T - 1) / EW_T, 64);
    hipLaunchKernelGGL(k_copy_sources, dim3(gx, (unsigned)njobs), dim3(EW_T), 0, st, jobs, words, lv, L, n, bad, seq);
    KERNEL_CHECK();
}

void launch_copy_jobs(const CtJob *jobs, size_t words, int njobs, hipStream_t st)
{
    if (!njobs) return;
    unsigned gx = (unsigned)std::min<size_t>((words / 2 + EW_T - 1) / EW_T, 64);
    hipLaunchKernelGGL(k_copy_jobs, dim3(gx, (unsigned)njobs), dim3(EW_T), 0, st, jobs, words);
    KERNEL_CHECK();
}

// counter-based generator for synthetic DB plaintexts: out[i] = splitmix64(seed + i) % bound
// (documented so tests can regenerate the same values on the host)
__global__ __launch_bounds__(EW_T) void k_fill_random(u64 *__restrict__ out, size_t words, u64 seed, u64 bound)
{
    const size_t i = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (i >= words) return;
    u64 z = seed + (u64)i * 0x9e3779b97f4a7c15ULL + 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z ^= z >> 31;
    out[i] = z % bound;
}

void launch_fill_random(u64 *out, size_t words, u64 seed, u64 bound, hipStream_t st)
{
    if (!words) return;
    hipLaunchKernelGGL(k_fill_random, dim3((unsigned)((words + EW_T - 1) / EW_T)), dim3(EW_T), 0, st, out, words, seed, bound);
    KERNEL_CHECK();
}

// Mask values as the reference draws them (receiver_osn.cpp:221-224,248-251): SEAL's Blake2xb generator under a 64-byte seed,
// out[i] = generate() % bound with the 32-bit generate(), i = the (first + i)-th output of the generator.  One lane per
// 64-byte stream block (16 values): it derives the root hash of its 4096-byte buffer and its own expansion node (3 BLAKE2b
// compressions; the redundant root hashes are ~2 us of work for a whole query's masks).
__global__ __launch_bounds__(EW_T) void k_fill_blake2xb(u64 *__restrict__ out, size_t words, Blake2xbSeed seed, u64 first, u64 bound)
{
    const u64 sb = first / 16 + (u64)blockIdx.x * EW_T + threadIdx.x;       // stream block of this lane
    if (sb * 16 >= first + words) return;
    u64 blk[8];
    blake2xb_stream_block(seed, sb, blk);
#pragma unroll
    for (unsigned j = 0; j < 16; j++) {
        const u64 g = sb * 16 + j;
        if (g >= first && g < first + words) out[g - first] = (u64)blake2xb_stream_u32(blk, j) % bound;
    }
}

void launch_fill_blake2xb(u64 *out, size_t words, const Blake2xbSeed &seed, u64 first, u64 bound, hipStream_t st)
{
    if (!words) return;
    const u64 blocks = (first + words + 15) / 16 - first / 16;
    hipLaunchKernelGGL(k_fill_blake2xb, dim3((unsigned)((blocks + EW_T - 1) / EW_T)), dim3(EW_T), 0, st, out, words, seed, first, bound);
    KERNEL_CHECK();
}

// N3 on the device: util::sample_poly_uniform under SEAL's Blake2xb generator (seal_codec.h) -- the c1 of a seeded ciphertext
// (Encryptor::encrypt_symmetric -> Serializable<Ciphertext>, sender/apsu/plaintext_powers.cpp:41-46) or of a seeded key.
// Pass 1 (k_seed_bulk): one lane per 64-byte stream block fills dst[L][n] with the generator's first L*n words reduced mod
// q_j; a word at or above the largest multiple of q_j below 2^64 - 1 is rejected and its position appended to the job's list.
// Pass 2 (k_seed_fix, one wave per ciphertext): the rejected positions are put in stream order (rank sort in LDS) and
// replaced one after the other by fresh draws that continue behind the bulk fill, exactly in SEAL's order; the wave
// generates 64 stream blocks at a time, every lane follows the same (uniform) control flow and lane 0 writes.
constexpr u32 SEED_REJ_CAP = 8192;                                // rejected positions per ciphertext the fix-up can hold
__global__ __launch_bounds__(EW_T) void k_seed_bulk(const SeedJob *__restrict__ jobs, const DevLevel *__restrict__ lv, const u64 *__restrict__ max_multiple,
                                                    size_t n, u32 *__restrict__ rej)
{
    const u64 sb = (u64)blockIdx.x * EW_T + threadIdx.x;
    const int L = lv->L;
    if (sb * 8 >= (u64)L * n) return;
    const SeedJob job = jobs[blockIdx.y];
    u64 blk[8];
    blake2xb_stream_block(job.seed, sb, blk);
    u32 *list = rej + (size_t)blockIdx.y * (1 + SEED_REJ_CAP);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 w = sb * 8 + i;                                 // n is a multiple of 8: the block lies inside one limb
        const int j = (int)(w / n);
        if (blk[i] >= max_multiple[j]) {
            const u32 slot = atomicAdd(list, 1u);
            if (slot < SEED_REJ_CAP) list[1 + slot] = (u32)w;
        } else job.dst[w] = barrett64(blk[i], lv->q[j]);
    }
}

__global__ __launch_bounds__(64) void k_seed_fix(const SeedJob *__restrict__ jobs, const DevLevel *__restrict__ lv, const u64 *__restrict__ max_multiple,
                                                 size_t n, u32 *__restrict__ rej, int *__restrict__ overflow)
{
    __shared__ u32 sorted[SEED_REJ_CAP];
    __shared__ u64 cache[64 * 8];
    const SeedJob job = jobs[blockIdx.x];
    u32 *list = rej + (size_t)blockIdx.x * (1 + SEED_REJ_CAP);
    const u32 cnt = list[0];
    const u32 lane = threadIdx.x;
    if (cnt > SEED_REJ_CAP) { if (lane == 0) *overflow = 1; return; }
    for (u32 e = lane; e < cnt; e += 64) {                        // rank sort (positions are distinct)
        const u32 v = list[1 + e];
        u32 rank = 0;
        for (u32 k = 0; k < cnt; k++) rank += list[1 + k] < v;
        sorted[rank] = v;
    }
    __syncthreads();
    u64 next = (u64)lv->L * n;                                    // fresh draws continue behind the bulk fill
    u64 cache_block = ~(u64)0;                                    // first stream block held in `cache`
    for (u32 k = 0; k < cnt; k++) {                               // uniform: every lane walks the same list
        const u32 w = sorted[k];
        const int j = (int)(w / n);
        const u64 mm = max_multiple[j];
        u64 r;
        do {
            const u64 b = next >> 3;
            if (cache_block == ~(u64)0 || b < cache_block || b >= cache_block + 64) {
                __syncthreads();
                cache_block = b;
                u64 blk[8];
                blake2xb_stream_block(job.seed, b + lane, blk);
#pragma unroll
                for (int i = 0; i < 8; i++) cache[lane * 8 + i] = blk[i];
                __syncthreads();
            }
            r = cache[(next - (cache_block << 3))];
            next++;
        } while (r >= mm);
        if (lane == 0) job.dst[w] = barrett64(r, lv->q[j]);
    }
    if (lane == 0) list[0] = 0;                                   // ready for the next use of this list
}

void launch_seed_expand(const SeedJob *jobs, int njobs, const DevLevel *lv, int L, const u64 *max_multiple, size_t n, u32 *rej, int *overflow,
                        hipStream_t st)
{
    if (!njobs) return;
    const u64 blocks = ((u64)L * n + 7) / 8;
    hipLaunchKernelGGL(k_seed_bulk, dim3((unsigned)((blocks + EW_T - 1) / EW_T), (unsigned)njobs), dim3(EW_T), 0, st, jobs, lv, max_multiple, n, rej);
    hipLaunchKernelGGL(k_seed_fix, dim3((unsigned)njobs), dim3(64), 0, st, jobs, lv, max_multiple, n, rej, overflow);
    KERNEL_CHECK();
}

// ============================================================================ N1: BinBundle build on the GPU
// polyn_with_roots (common/apsu/util/interpolate.cpp:27-80) for every bin of a BinBundle.  One WAVE per bin: the
// monic polynomial lives in registers, coefficient i in lane i % 64, slot i / 64, so multiplying by (x - a) is one
// multiply-add per held coefficient plus a one-lane shift (P'[i] = P[i-1] - a P[i]); only the slots that can be
// non-zero after r roots are touched.  No LDS, no barriers.  Results are written column-wise ([degree][slot]); the
// buffer is zeroed beforehand, so bins beyond `bins` and degrees beyond a bin's count hold 0
// (BatchedPlaintextPolyn ctor, bin_bundle.cpp:395-405).
template <int SLOTS>
__global__ __launch_bounds__(256) void k_polyn_with_roots(const u64 *__restrict__ roots, const u32 *__restrict__ counts,
                                                          u32 bins, u32 stride, Mod t, u64 *__restrict__ poly, size_t n)
{
    const u32 lane = threadIdx.x & 63;
    const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);           // bin of this wave
    if (s >= bins) return;
    const u32 cnt = counts[s];
    u64 P[SLOTS];
#pragma unroll
    for (int j = 0; j < SLOTS; j++) P[j] = 0;
    if (lane == 0) P[0] = 1;
    const u64 *rt = roots + (size_t)s * stride;
    for (u32 r = 0; r < cnt; r++) {
        const u64 a = rt[r];
        const u64 neg_a = a ? t.q - a : 0;
        const int jmax = (int)((r + 1) >> 6);                    // highest slot holding a coefficient of degree <= r + 1
#pragma unroll
        for (int j = SLOTS - 1; j >= 0; j--) {
            if (j > jmax) continue;                              // wave-uniform
            u64 prev = __shfl_up((unsigned long long)P[j], 1, 64);                       // P[i-1] of the lane below
            const u64 wrap = j > 0 ? (u64)__shfl((unsigned long long)P[j > 0 ? j - 1 : 0], 63, 64) : 0;   // lane 0: last lane of slot j-1
            if (lane == 0) prev = wrap;
            P[j] = addmod(mulmod(P[j], neg_a, t), prev, t.q);
        }
    }
#pragma unroll
    for (int j = 0; j < SLOTS; j++) {
        const u32 d = (u32)j * 64 + lane;
        if (d <= cnt) poly[(size_t)d * n + s] = P[j];
    }
}

// fallback for polynomials beyond the largest wave-per-bin instance (more than 8192 coefficients): one thread per bin,
// coefficients in global memory
__global__ __launch_bounds__(EW_T) void k_polyn_with_roots_serial(const u64 *__restrict__ roots, const u32 *__restrict__ counts,
                                                                  u32 bins, u32 stride, Mod t, u64 *__restrict__ poly, size_t n)
{
    const size_t s = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (s >= bins) return;
    u64 *P = poly + s;                                          // P[d * n], zeroed by the caller
    const u32 cnt = counts[s];
    P[0] = 1;
    for (u32 r = 0; r < cnt; r++) {
        const u64 a = roots[(size_t)s * stride + r];
        const u64 neg_a = a ? t.q - a : 0;
        for (u32 i = r + 1; i > 0; i--)
            P[(size_t)i * n] = addmod(mulmod(P[(size_t)i * n], neg_a, t), P[(size_t)(i - 1) * n], t.q);
        P[0] = mulmod(P[0], neg_a, t);
    }
}

void launch_polyn_with_roots(const u64 *roots, const u32 *counts, u32 bins, u32 stride, u32 max_deg, Mod t, u64 *poly, size_t n,
                             hipStream_t st)
{
    { hipError_t e_ = hipMemsetAsync(poly, 0, (size_t)(max_deg + 1) * n * sizeof(u64), st); if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__); }
    if (!bins) return;
    const u32 slots = max_deg / 64 + 1;
    const dim3 g((bins + 3) / 4), b(256);
#define PW_CASE(S) if (slots <= S) { hipLaunchKernelGGL((k_polyn_with_roots<S>), g, b, 0, st, roots, counts, bins, stride, t, poly, n); KERNEL_CHECK(); return; }
    PW_CASE(1) PW_CASE(2) PW_CASE(4) PW_CASE(8) PW_CASE(16) PW_CASE(24) PW_CASE(32) PW_CASE(48) PW_CASE(64) PW_CASE(96) PW_CASE(128)
#undef PW_CASE
    hipLaunchKernelGGL(k_polyn_with_roots_serial, dim3((bins + EW_T - 1) / EW_T), dim3(EW_T), 0, st, roots, counts, bins, stride, t, poly, n);
    KERNEL_CHECK();
}

// BatchEncoder::encode, first half (App. B4): out[b][slot_map[i]] = in[b][i]; the inverse NTT mod t follows
__global__ __launch_bounds__(EW_T) void k_scatter_slots(const u64 *__restrict__ in, const u32 *__restrict__ slot_map,
                                                        u64 *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (i >= n) return;
    const size_t b = blockIdx.y;
    out[b * n + slot_map[i]] = in[b * n + i];
}

void launch_scatter_slots(const u64 *in, const u32 *slot_map, u64 *out, size_t n, int batch, hipStream_t st)
{
    if (!batch) return;
    hipLaunchKernelGGL(k_scatter_slots, ew_grid(n, batch), dim3(EW_T), 0, st, in, slot_map, out, n);
    KERNEL_CHECK();
}

// BatchEncoder::decode's slot gather (after the forward NTT mod t): out[b][i] = in[b][slot_map[i]]
__global__ __launch_bounds__(EW_T) void k_gather_slots(const u64 *__restrict__ in, const u32 *__restrict__ slot_map,
                                                       u64 *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (i >= n) return;
    const size_t b = blockIdx.y;
    out[b * n + i] = in[b * n + slot_map[i]];
}

void launch_gather_slots(const u64 *in, const u32 *slot_map, u64 *out, size_t n, int batch, hipStream_t st)
{
    if (!batch) return;
    hipLaunchKernelGGL(k_gather_slots, ew_grid(n, batch), dim3(EW_T), 0, st, in, slot_map, out, n);
    KERNEL_CHECK();
}

// N1: algebraize_item (common/apsu/util/db_encoding.cpp:209-256,360-366): felt j of an item = bits [j*bpf, j*bpf + bpf) of its
// first item_bits bits, the item read as a little-endian bit string (bit k = bit k%8 of byte k/8); one thread per felt
__global__ __launch_bounds__(EW_T) void k_algebraize(const unsigned char *__restrict__ items, size_t count, u32 felts, u32 bpf, u32 item_bits,
                                                     u64 *__restrict__ out)
{
    const size_t idx = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (idx >= count * felts) return;
    const size_t it = idx / felts;
    const u32 j = (u32)(idx % felts);
    const unsigned char *p = items + it * 16;
    u64 lo = 0, hi = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) { lo |= (u64)p[b] << (8 * b); hi |= (u64)p[8 + b] << (8 * b); }
    const u32 off = j * bpf;
    const u32 take = off >= item_bits ? 0 : (item_bits - off < bpf ? item_bits - off : bpf);
    u64 v = off >= 64 ? hi >> (off - 64) : (off ? (lo >> off) | (hi << (64 - off)) : lo);
    out[idx] = take ? v & (~(u64)0 >> (64 - take)) : 0;
}

void launch_algebraize(const unsigned char *items, size_t count, u32 felts, u32 bpf, u32 item_bits, u64 *out, hipStream_t st)
{
    if (!count) return;
    hipLaunchKernelGGL(k_algebraize, dim3((unsigned)((count * felts + EW_T - 1) / EW_T)), dim3(EW_T), 0, st, items, count, felts, bpf, item_bits, out);
    KERNEL_CHECK();
}

// N4: vec_to_oc_block (receiver_osn.cpp:53-73): the felts of one item packed into a 128-bit block for the PEQT step,
// out[item] = (lower, higher).  len = bit length of the plain modulus as the reference computes it; the odd-felt
// branch shifts the upper half by len/2 - 1 (not len/2) exactly as the reference does; 64-bit shifts wrap.
__global__ __launch_bounds__(EW_T) void k_pack_blocks(const u64 *__restrict__ values, size_t n, u32 items, u32 felts, u32 len,
                                                      u64 *__restrict__ out)
{
    const size_t it = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (it >= items) return;
    const size_t b = blockIdx.y;
    const u64 *in = values + b * n + it * felts;
    const u64 mask = ((u64)1 << len) - 1, mask_lower = ((u64)1 << (len >> 1)) - 1, mask_higher = mask - mask_lower;
    u64 lower = 0, higher = 0;
    if (felts & 1) {
        lower = in[felts - 1] & mask_lower;
        higher = (in[felts - 1] & mask_higher) >> ((len >> 1) - 1);
    }
    for (u32 p = 0; p + 1 < felts; p += 2) {
        lower = (in[p] & mask) | (lower << len);
        higher = (in[p + 1] & mask) | (higher << len);
    }
    out[(b * items + it) * 2] = lower;
    out[(b * items + it) * 2 + 1] = higher;
}

void launch_pack_blocks(const u64 *values, size_t n, u32 items, u32 felts, u32 len, u64 *out, int batch, hipStream_t st)
{
    if (!batch || !items) return;
    hipLaunchKernelGGL(k_pack_blocks, ew_grid(items, batch), dim3(EW_T), 0, st, values, n, items, felts, len, out);
    KERNEL_CHECK();
}

// N4: the querier's decryption of a result at the last level (result_package.cpp:175-213, Decryptor::decrypt):
// x = c0 + v (v = INTT(NTT(c1) . s), one limb q0), m = round(t x / q0) mod t.
__global__ __launch_bounds__(EW_T) void k_decrypt_round(const u64 *__restrict__ ct, size_t ct_stride, const u64 *__restrict__ v,
                                                        u64 q0, u64 t, u64 *__restrict__ out, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const size_t b = blockIdx.y;
    const u64 x = addmod(ct[b * ct_stride + k], v[b * n + k], q0);
    u128p num = mul128(x, t);
    add128(num, u128p{ q0 >> 1, 0 });
    // floor(num / q0) mod t by restoring division, the quotient folded mod t on the fly (num < 2^124)
    u64 rem = 0, quo = 0;
    for (int i = 127; i >= 0; i--) {
        const u64 bit = i >= 64 ? (num.hi >> (i - 64)) & 1 : (num.lo >> i) & 1;
        rem = (rem << 1) | bit;                                   // rem < q0 < 2^62 before the shift
        quo <<= 1;                                                // quo < t < 2^61
        if (rem >= q0) { rem -= q0; quo |= 1; }
        if (quo >= t) quo -= t;
    }
    out[b * n + k] = quo;
}

void launch_decrypt_round(const u64 *ct, size_t ct_stride, const u64 *v, u64 q0, u64 t, u64 *out, size_t n, int batch, hipStream_t st)
{
    if (!batch) return;
    hipLaunchKernelGGL(k_decrypt_round, ew_grid(n, batch), dim3(EW_T), 0, st, ct, ct_stride, v, q0, t, out, n);
    KERNEL_CHECK();
}

// flag[b] = 1 iff plaintext b has exactly one non-zero coefficient (SEAL's monomial shortcut in multiply_plain)
__global__ __launch_bounds__(EW_T) void k_flag_monomial(const u64 *__restrict__ pt, size_t n, unsigned char *__restrict__ flag)
{
    __shared__ unsigned int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    unsigned int local = 0;
    for (size_t k = threadIdx.x; k < n; k += EW_T) local += pt[(size_t)blockIdx.x * n + k] != 0;
    atomicAdd(&cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) flag[blockIdx.x] = cnt == 1 ? 1 : 0;
}

void launch_flag_monomial(const u64 *pt, size_t n, int batch, unsigned char *flag, hipStream_t st)
{
    if (!batch) return;
    hipLaunchKernelGGL(k_flag_monomial, dim3((unsigned)batch), dim3(EW_T), 0, st, pt, n, flag);
    KERNEL_CHECK();
}

// K9: try_clear_irrelevant_bits (bin_bundle.cpp:67-97)
__global__ __launch_bounds__(EW_T) void k_clear_bits(u64 *__restrict__ ct, size_t words, u64 mask)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k < words) ct[k] &= mask;
}

void launch_clear_bits(u64 *ct, size_t words, int bits, hipStream_t st)
{
    if (bits <= 0) return;
    hipLaunchKernelGGL(k_clear_bits, dim3((unsigned)((words + EW_T - 1) / EW_T)), dim3(EW_T), 0, st, ct, words,
                       ~(((u64)1 << bits) - 1));
    KERNEL_CHECK();
}

// ============================================================================ K5: BFV multiply (BEHZ)
// square / multiply / multiply_inplace (receiver_osn.cpp:422,424 ; bin_bundle.cpp:272,301)
//
// Step (1)+(2): q -> q u Bsk with Montgomery removal of the q-overflow (fastbconv_m_tilde + sm_mrq).
// in: [batch] polynomials [L][n] at stride in_stride ; out: [batch][E][n] (q part copied, Bsk part computed)
// TL / TNB > 0: compile-time limb counts (loops fully unrolled, constants fetched in bulk); 0 = generic.
template <int TL, int TNB>
__global__ __launch_bounds__(EW_T) void k_behz_ext(const DevLevel *__restrict__ lv, const u64 *__restrict__ in,
                                                   size_t in_stride, int polys, u64 *__restrict__ out, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const int L = TL ? TL : lv->L, nBsk = TL ? TNB + 1 : lv->nBsk, E = L + nBsk;
    constexpr int LMAX = TL ? TL : DMAXL;
    constexpr int BMAX = TL ? TNB + 1 : DMAXB;
    const size_t c = blockIdx.y / polys, p = blockIdx.y % polys;
    const u64 *src = in + c * in_stride + p * (size_t)L * n;
    u64 *dst = out + (size_t)blockIdx.y * E * n;
    u64 xs[LMAX];
    u32 mt_acc = 0;
#pragma unroll
    for (int j = 0; j < LMAX; j++) {
        if (TL || j < L) {
            const u64 x = src[(size_t)j * n + k];
            dst[(size_t)j * n + k] = x;
            xs[j] = mul_shoup(x, lv->ext_scale[j].w, lv->ext_scale[j].wq, lv->q[j].q);
            mt_acc += (u32)xs[j] * lv->q_to_mt[j];           // arithmetic mod 2^32 = m_tilde
        }
    }
    const u32 r32 = mt_acc * lv->neg_inv_q_mt;
#pragma unroll
    for (int i = 0; i < BMAX; i++) {
        if (TL || i < nBsk) {
            const Mod m = lv->bsk[i];
            u128p acc{ 0, 0 };
#pragma unroll
            for (int j = 0; j < LMAX; j++)
                if (TL || j < L) mac128(acc, xs[j], lv->q_to_bsk[i][j]);
            const u64 y = barrett128(acc, m);
            // centred lift of r into Z_m (m_tilde is a power of two: ">=")
            u64 r = r32;
            if (r32 >= 0x80000000u) r += m.q - ((u64)1 << 32);
            u128p v = mul128(r, lv->prod_q_bsk[i]);
            add128(v, u128p{ y, 0 });
            const u64 red = barrett128(v, m);
            dst[(size_t)(L + i) * n + k] = mul_shoup(red, lv->inv_mt_bsk[i].w, lv->inv_mt_bsk[i].wq, m.q);
        }
    }
}

HD u64 lazy2(u64 x, const ShoupConst &c, u64 q) { return mul_shoup_lazy(x, c.w, c.wq, q); }      // x*c mod q in [0,2q), any x

// BEHZ steps 1-2 for one coefficient (L = nB = TL <= 3, Shoup-form matrices): xin = the L canonical q-limb residues,
// dst = limb 0 of the extended polynomial ([L q-limbs | nB B-limbs | m_sk][n]) at coefficient k.
template <int TL>
__device__ __forceinline__ void behz_ext2_body(const DevLevel *__restrict__ lv, const u64 *xin, u64 *__restrict__ dst, size_t n, size_t k)
{
    constexpr int L = TL, nBsk = TL + 1;
    u64 xs[L];
    u32 mt_acc = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        const u64 x = xin[j];
        dst[(size_t)j * n + k] = x;
        xs[j] = mul_shoup(x, lv->ext_scale[j].w, lv->ext_scale[j].wq, lv->q[j].q);     // canonical: used as an integer
        mt_acc += (u32)xs[j] * lv->q_to_mt[j];
    }
    const u32 r32 = mt_acc * lv->neg_inv_q_mt;
#pragma unroll
    for (int i = 0; i < nBsk; i++) {
        const u64 m = lv->bsk[i].q;
        u64 r = r32;
        if (r32 >= 0x80000000u) r += m - ((u64)1 << 32);          // centred lift of r into Z_m
        // sm_mrq: (x_i + q r) m_tilde^-1 mod m with x_i = sum_j xs_j (Q/q_j): m_tilde^-1 is folded into both constants (round 4), so
        // the closing product is a reduction of the lazy sum v < (2L + 2) m <= 8 m < 2^64 -- the same residue, one Shoup product less
        u64 v = lazy2(r, lv->s_prod_q_bsk_mt[i], m);
#pragma unroll
        for (int j = 0; j < L; j++) v += lazy2(xs[j], lv->s_q_to_bsk_mt[i][j], m);
        dst[(size_t)(L + i) * n + k] = csub(csub(csub(v, m << 2), m << 1), m);
    }
}

// Fully unrolled variant for L = nB = TL <= 3 with Shoup-form matrices: same values, ~35 % fewer multiplies.
// DROP: the input is one level higher (TL + 1 limbs per polynomial) and is first mod-switched to this level
// (mod_switch_to_next_inplace, bin_bundle.cpp:269,298) — the drop and the extension share one pass over the data.
template <int TL, bool DROP, bool RAW = false>
__global__ __launch_bounds__(EW_T) void k_behz_ext2(const DevLevel *__restrict__ lv, const u64 *__restrict__ in,
                                                    size_t in_stride, int polys, u64 *__restrict__ out, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    constexpr int L = TL, E = 2 * TL + 1, LIN = TL + (DROP ? 1 : 0);
    const size_t c = blockIdx.y / polys, p = blockIdx.y % polys;
    const u64 *src = in + c * in_stride + p * (size_t)LIN * n;
    u64 *dst = out + (size_t)blockIdx.y * E * n;
    u64 xin[L];
    if (DROP) {
        const DevLevel *ld = lv + 1;                              // constants of the level being left
        const u64 ql = ld->q[L].q;
        u64 lastc = src[(size_t)L * n + k];
        if (RAW) {                                                // the inverse transform left its twist to this kernel
            const u64x2 tw = ldg16(reinterpret_cast<const u64 *>(ld->last_tw + k));
            lastc = mul_shoup(lastc, tw[0], tw[1], ql);
        }
        const u64 last = addmod(lastc, ld->half, ql);
#pragma unroll
        for (int j = 0; j < L; j++) {
            const Mod m = ld->q[j];
            const u64 tmp = submod(barrett64(last, m), ld->half_mod[j], m.q);
            if (RAW) {
                // (c_j - tmp) q_last^-1 with c_j = raw_j * twist: two lazy products, the twist folded into the first constant
                const u64x2 tw = ldg16(reinterpret_cast<const u64 *>(ld->drop_tw[j] + k));
                const u64 a = mul_shoup_lazy(src[(size_t)j * n + k], tw[0], tw[1], m.q);                       // [0, 2q)
                const u64 b = mul_shoup_lazy(tmp, ld->inv_q_last[j].w, ld->inv_q_last[j].wq, m.q);           // [0, 2q)
                xin[j] = csub(csub(a + (m.q << 1) - b, m.q << 1), m.q);
            } else xin[j] = mul_shoup(submod(src[(size_t)j * n + k], tmp, m.q), ld->inv_q_last[j].w, ld->inv_q_last[j].wq, m.q);
        }
    } else {
#pragma unroll
        for (int j = 0; j < L; j++) xin[j] = src[(size_t)j * n + k];
    }
    behz_ext2_body<TL>(lv, xin, dst, n, k);
}

// drop one limb, then extend: `in` holds polynomials of L + 1 limbs at level lv + 1.  Only for the unrolled sizes;
// returns false when the caller has to run the two steps separately.
bool launch_drop_behz_ext(const DevLevel *lv, int L, int nB, const u64 *in, size_t in_stride, int polys, u64 *out, size_t n, int cts,
                          hipStream_t st, bool raw)
{
    if (!cts) return true;
    const dim3 g = ew_grid(n, cts * polys), t(EW_T);
#define EXT2_CASE(TL) if (L == TL && nB == TL) { \
        if (raw) hipLaunchKernelGGL((k_behz_ext2<TL, true, true>), g, t, 0, st, lv, in, in_stride, polys, out, n); \
        else hipLaunchKernelGGL((k_behz_ext2<TL, true>), g, t, 0, st, lv, in, in_stride, polys, out, n); \
        KERNEL_CHECK(); return true; }
    EXT2_CASE(1) EXT2_CASE(2) EXT2_CASE(3)
#undef EXT2_CASE
    return false;
}

void launch_behz_ext(const DevLevel *lv, int L, int nB, const u64 *in, size_t in_stride, int polys, u64 *out, size_t n, int cts,
                     hipStream_t st)
{
    if (!cts) return;
    const dim3 g = ew_grid(n, cts * polys), t(EW_T);
#define EXT2_CASE(TL) if (L == TL && nB == TL) { hipLaunchKernelGGL((k_behz_ext2<TL, false>), g, t, 0, st, lv, in, in_stride, polys, out, n); KERNEL_CHECK(); return; }
    EXT2_CASE(1) EXT2_CASE(2) EXT2_CASE(3)
#undef EXT2_CASE
#define EXT_CASE(TL) if (L == TL && nB == TL) { hipLaunchKernelGGL((k_behz_ext<TL, TL>), g, t, 0, st, lv, in, in_stride, polys, out, n); } else
    EXT_CASE(4)
    { hipLaunchKernelGGL((k_behz_ext<0, 0>), g, t, 0, st, lv, in, in_stride, polys, out, n); }
#undef EXT_CASE
    KERNEL_CHECK();
}

// Step (4): tensor product in the NTT domain over all E limbs: d0=a0b0, d1=a0b1+a1b0, d2=a1b1
__global__ __launch_bounds__(EW_T) void k_tensor(const DevLevel *__restrict__ lv, const TensorJob *__restrict__ jobs, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const TensorJob job = jobs[blockIdx.y];
    const int E = lv->E;
    const size_t ps = (size_t)E * n;
    for (int e = 0; e < E; e++) {
        const Mod m = lv->ext[e];
        const size_t o = e * n + k;
        const u64 a0 = job.a[o], a1 = job.a[ps + o], b0 = job.b[o], b1 = job.b[ps + o];
        job.d[o] = mulmod(a0, b0, m);
        u128p mid = mul128(a0, b1);
        mac128(mid, a1, b0);
        job.d[ps + o] = barrett128(mid, m);
        job.d[2 * ps + o] = mulmod(a1, b1, m);
    }
}

void launch_tensor(const DevLevel *lv, const TensorJob *jobs, size_t n, int batch, hipStream_t st)
{
    hipLaunchKernelGGL(k_tensor, ew_grid(n, batch), dim3(EW_T), 0, st, lv, jobs, n);
    KERNEL_CHECK();
}

// Step (4) for ciphertexts of any size (parameter sets without key switching never relinearise, receiver_osn.cpp:430-432, so
// operands keep growing): d_I = sum_{i + j = I} a_i b_j, limb-wise, every product reduced before the modular add
// (Evaluator::bfv_multiply's behz_ciphertext_product).  Not a hot path: no shipped parameter set reaches it.
__global__ __launch_bounds__(EW_T) void k_tensor_conv(const DevLevel *__restrict__ lv, const TensorConvJob *__restrict__ jobs, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const TensorConvJob job = jobs[blockIdx.y];
    const int E = lv->E;
    const size_t ps = (size_t)E * n;
    const int so = job.sa + job.sb - 1;
    for (int e = 0; e < E; e++) {
        const Mod m = lv->ext[e];
        const size_t o = e * n + k;
        for (int I = 0; I < so; I++) {
            const int i0 = I - (job.sb - 1) > 0 ? I - (job.sb - 1) : 0, i1 = I < job.sa - 1 ? I : job.sa - 1;
            u64 acc = 0;
            for (int i = i0; i <= i1; i++) acc = addmod(acc, mulmod(job.a[i * ps + o], job.b[(I - i) * ps + o], m), m.q);
            job.d[I * ps + o] = acc;
        }
    }
}

void launch_tensor_conv(const DevLevel *lv, const TensorConvJob *jobs, size_t n, int njobs, hipStream_t st)
{
    if (!njobs) return;
    hipLaunchKernelGGL(k_tensor_conv, ew_grid(n, njobs), dim3(EW_T), 0, st, lv, jobs, n);
    KERNEL_CHECK();
}

// Step (4) for a SUM of products (eval_patstock's sum over i): the q limbs of every term are kept (their canonical
// coefficient-form residues are needed per term by the finish), the Bsk limbs are summed here in the NTT domain.
// e0 = L: only the Bsk sums (the per-term q limbs are formed by k_intt_tensor on load).
__global__ __launch_bounds__(EW_T) void k_tensor_sum(const DevLevel *__restrict__ lv, const TensorSumJob *__restrict__ jobs, size_t n, int e0)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const TensorSumJob job = jobs[blockIdx.z];
    const int E = lv->E, L = lv->L, e = blockIdx.y + e0;
    const size_t ps = (size_t)E * n, o = (size_t)e * n + k;
    const Mod m = lv->ext[e];
    if (e < L) {
        for (int i = 0; i < job.terms; i++) {
            const u64 *a = job.a + (size_t)i * 2 * ps, *b = job.b + (size_t)i * 2 * ps;
            const u64 a0 = a[o], a1 = a[ps + o], b0 = b[o], b1 = b[ps + o];
            u64 *d = job.dq + (size_t)i * 3 * L * n + o;
            d[0] = mulmod(a0, b0, m);
            u128p mid = mul128(a0, b1);
            mac128(mid, a1, b0);
            d[(size_t)L * n] = barrett128(mid, m);
            d[(size_t)2 * L * n] = mulmod(a1, b1, m);
        }
    } else {
        u128p s0{ 0, 0 }, s1{ 0, 0 }, s2{ 0, 0 };
        u64 r0 = 0, r1 = 0, r2 = 0;
        for (int i = 0; i < job.terms; i++) {
            const u64 *a = job.a + (size_t)i * 2 * ps, *b = job.b + (size_t)i * 2 * ps;
            const u64 a0 = a[o], a1 = a[ps + o], b0 = b[o], b1 = b[ps + o];
            mac128(s0, a0, b0);
            mac128(s1, a0, b1);
            mac128(s1, a1, b0);
            mac128(s2, a1, b1);
            if ((i & 7) == 7 || i + 1 == job.terms) {              // 16 products of < 2^62 bits each fit 128 bits
                r0 = addmod(r0, barrett128(s0, m), m.q);
                r1 = addmod(r1, barrett128(s1, m), m.q);
                r2 = addmod(r2, barrett128(s2, m), m.q);
                s0 = s1 = s2 = u128p{ 0, 0 };
            }
        }
        const int nBsk = E - L;
        u64 *d = job.bs + (size_t)(e - L) * n + k;
        d[0] = r0;
        d[(size_t)nBsk * n] = r1;
        d[(size_t)2 * nBsk * n] = r2;
    }
}

void launch_tensor_sum(const DevLevel *lv, int E, const TensorSumJob *jobs, size_t n, int njobs, int e0, hipStream_t st)
{
    if (!njobs || e0 >= E) return;
    hipLaunchKernelGGL(k_tensor_sum, dim3((unsigned)((n + EW_T - 1) / EW_T), E - e0, njobs), dim3(EW_T), 0, st, lv, jobs, n, e0);
    KERNEL_CHECK();
}

// Steps (6)-(8) for one coefficient of one extended polynomial: multiply by t, fast_floor
// (q u Bsk -> Bsk), fastbconv_sk (Bsk -> q).  d points at limb 0 of the polynomial, stride n.
//
// `terms` > 1 finishes a SUM of products in one go (eval_patstock's sum over i, bin_bundle.cpp:273,303): dq holds the q
// limbs of every term (stride term_stride), dbsk the Bsk limbs of the sum.  The only per-term non-linearity of
// steps (6)-(8) is the canonical residue [t d (Q/q_j)^-1]_{q_j} feeding fastbconv(q -> Bsk); those residues are
// summed as integers, everything after is linear mod Bsk_i and fastbconv_sk is exact, so the result equals the
// sum of the individually finished terms bit for bit (note N1).
template <int TL, int TNB>
__device__ __forceinline__ void behz_finish_coeff(const DevLevel *__restrict__ lv, const u64 *__restrict__ dq, size_t term_stride,
                                                  int terms, const u64 *__restrict__ dbsk, size_t n, u64 *res /* [L] */)
{
    const int L = TL ? TL : lv->L, nB = TL ? TNB : lv->nB, nBsk = nB + 1;
    constexpr int LMAX = TL ? TL : DMAXL;
    constexpr int BMAX = TL ? TNB + 1 : DMAXB;
    u64 xq[LMAX];
#pragma unroll
    for (int j = 0; j < LMAX; j++) xq[j] = 0;
    for (int it = 0; it < terms; it++) {
#pragma unroll
        for (int j = 0; j < LMAX; j++)                           // terms * q_j < 2^64 (host-checked)
            if (TL || j < L) xq[j] += mul_shoup(dq[it * term_stride + (size_t)j * n], lv->t_inv_punct_q[j].w, lv->t_inv_punct_q[j].wq, lv->q[j].q);
    }
    u64 ys[BMAX];
    u64 fl_sk = 0;
#pragma unroll
    for (int i = 0; i < BMAX; i++) {
        if (TL || i < nBsk) {
            const Mod m = lv->bsk[i];
            u128p acc{ 0, 0 };
#pragma unroll
            for (int j = 0; j < LMAX; j++)
                if (TL || j < L) mac128(acc, xq[j], lv->q_to_bsk[i][j]);
            const u64 conv = barrett128(acc, m);
            const u64 xb = mul_shoup(dbsk[(size_t)i * n], lv->t_bsk[i].w, lv->t_bsk[i].wq, m.q);
            const u64 fl = mul_shoup(xb + (m.q - conv), lv->inv_prod_q_bsk[i].w, lv->inv_prod_q_bsk[i].wq, m.q);
            if (i < nB) ys[i] = mul_shoup(fl, lv->inv_punct_B[i].w, lv->inv_punct_B[i].wq, m.q);
            else fl_sk = fl;
        }
    }
    const Mod msk = lv->bsk[nB];
    u128p acc{ 0, 0 };
#pragma unroll
    for (int i = 0; i < BMAX - 1; i++)
        if (TL || i < nB) mac128(acc, ys[i], lv->B_to_msk[i]);
    const u64 z_sk = barrett128(acc, msk);
    const u64 alpha = mul_shoup(z_sk + (msk.q - fl_sk), lv->inv_prod_B_msk.w, lv->inv_prod_B_msk.wq, msk.q);
    const bool neg = alpha > lv->msk_half;                     // alpha represents a negative value
    const u64 a_abs = neg ? msk.q - alpha : alpha;
#pragma unroll
    for (int j = 0; j < LMAX; j++) {
        if (TL || j < L) {
            u128p z{ 0, 0 };
#pragma unroll
            for (int i = 0; i < BMAX - 1; i++)
                if (TL || i < nB) mac128(z, ys[i], lv->B_to_q[j][i]);
            mac128(z, a_abs, neg ? lv->prod_B_q[j] : lv->neg_prod_B_q[j]);
            res[j] = barrett128(z, lv->q[j]);
        }
    }
}

// One job = one output polynomial triple: out[3][L][n] (+)= sum over `terms` consecutive extended
// products d[term][3][E][n] (coefficient form).  The per-term rounding is kept (SURVEY note N1).
template <int TL, int TNB>
__global__ __launch_bounds__(EW_T) void k_behz_finish(const DevLevel *__restrict__ lv, const FinishJob *__restrict__ jobs,
                                                      int accumulate, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const int L = TL ? TL : lv->L, E = TL ? 2 * TL + 1 + (TNB - TL) : lv->E;
    constexpr int LMAX = TL ? TL : DMAXL;
    const FinishJob job = jobs[blockIdx.y / 3];
    const size_t p = blockIdx.y % 3;
    u64 sum[LMAX];
    u64 *o = job.out + p * (size_t)L * n + k;
#pragma unroll
    for (int j = 0; j < LMAX; j++)
        if (TL || j < L) sum[j] = accumulate ? o[(size_t)j * n] : 0;
    for (int i = 0; i < job.terms; i++) {
        const u64 *dp = job.d + (((size_t)i * 3 + p) * (size_t)E) * n + k;
        u64 res[LMAX];
        behz_finish_coeff<TL, TNB>(lv, dp, 0, 1, dp + (size_t)L * n, n, res);
#pragma unroll
        for (int j = 0; j < LMAX; j++)
            if (TL || j < L) sum[j] = addmod(sum[j], res[j], lv->q[j].q);
    }
#pragma unroll
    for (int j = 0; j < LMAX; j++)
        if (TL || j < L) o[(size_t)j * n] = sum[j];
}

// Fully unrolled variant for L = nB = TL <= 3 with Shoup-form matrices (same values as behz_finish_coeff).
// dq / dbsk come from an inverse NTT that left out its twist (NTT_MAP_RAW): raw lazy values, any 64-bit number; the twist
// is part of the per-position constants fin_q / fin_b (kidx = coefficient index).
// The level constants of the unrolled finish (four conversion rows of Shoup pairs, ...) are ~100 scalar registers when the compiler
// hoists every load to the top of the kernel: it then spills scalar registers into vector lanes (v_writelane / v_readlane, 16 % of
// the instructions of k_behz_finish2<3>).  A compiler-level memory fence per output row keeps each row's loads next to their use.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(APSU_FIN_NO_FENCE)
#define FIN_LOAD_FENCE() asm volatile("" ::: "memory")
#else
#define FIN_LOAD_FENCE() do { } while (0)
#endif
template <int TL>
__device__ __forceinline__ void behz_finish_coeff2(const DevLevel *__restrict__ lv, const u64 *__restrict__ dq, size_t term_stride,
                                                   int terms, const u64 *__restrict__ dbsk, size_t n, u64 *res, size_t kidx)
{
    constexpr int L = TL, nB = TL, nBsk = TL + 1;
    u64 xq[L];
#pragma unroll
    for (int j = 0; j < L; j++) xq[j] = 0;
    ShoupConst cq[L];
#pragma unroll
    for (int j = 0; j < L; j++) { const u64x2 v = ldg16(reinterpret_cast<const u64 *>(lv->fin_q[j] + kidx)); cq[j] = ShoupConst{ v[0], v[1] }; }
    for (int it = 0; it < terms; it++) {
#pragma unroll
        for (int j = 0; j < L; j++)                              // canonical: used as integers by the base conversion
            xq[j] += mul_shoup(dq[it * term_stride + (size_t)j * n], cq[j].w, cq[j].wq, lv->q[j].q);
    }
    u64 ys[nB];
    u64 fl_sk = 0;
#pragma unroll
    for (int i = 0; i < nBsk; i++) {
        FIN_LOAD_FENCE();
        const u64 m = lv->bsk[i].q;
        u64 conv = 0;                                            // < 2 L m
#pragma unroll
        for (int j = 0; j < L; j++) conv += lazy2(xq[j], lv->s_q_to_bsk[i][j], m);
        const u64x2 cb = ldg16(reinterpret_cast<const u64 *>(lv->fin_b[i] + kidx));
        const u64 xb = lazy2(dbsk[(size_t)i * n], ShoupConst{ cb[0], cb[1] }, m);
        const u64 diff = xb + ((u64)(2 * L) * m - conv);         // < (2L + 2) m <= 8 m < 2^64
        const u64 f = mul_shoup(diff, lv->s_fl[i].w, lv->s_fl[i].wq, m);      // i < nB: already times (B/b_i)^-1
        if (i < nB) ys[i] = f; else fl_sk = f;
    }
    const u64 msk = lv->bsk[nB].q;
    u64 z_sk = msk - fl_sk;                                      // (z - fl_sk) mod m_sk, lazily
#pragma unroll
    for (int i = 0; i < nB; i++) z_sk += lazy2(ys[i], lv->s_B_to_msk[i], msk);
    const u64 alpha = mul_shoup(z_sk, lv->inv_prod_B_msk.w, lv->inv_prod_B_msk.wq, msk);
    const bool neg = alpha > lv->msk_half;
    const u64 a_abs = neg ? msk - alpha : alpha;
#pragma unroll
    for (int j = 0; j < L; j++) {
        FIN_LOAD_FENCE();
        const u64 q = lv->q[j].q;
        u64 z = lazy2(a_abs, neg ? lv->s_prod_B_q[j] : lv->s_neg_prod_B_q[j], q);       // < (2 nB + 2) q <= 8 q
#pragma unroll
        for (int i = 0; i < nB; i++) z += lazy2(ys[i], lv->s_B_to_q[j][i], q);
        res[j] = csub(csub(csub(z, q << 2), q << 1), q);
    }
}

template <int TL>
__global__ __launch_bounds__(EW_T) void k_behz_finish2(const DevLevel *__restrict__ lv, const FinishJob *__restrict__ jobs,
                                                       int accumulate, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    constexpr int L = TL, E = 2 * TL + 1;
    const FinishJob job = jobs[blockIdx.y / 3];
    const size_t p = blockIdx.y % 3;
    u64 sum[L];
    u64 *o = job.out + p * (size_t)L * n + k;
#pragma unroll
    for (int j = 0; j < L; j++) sum[j] = accumulate ? o[(size_t)j * n] : 0;
    for (int i = 0; i < job.terms; i++) {
        u64 res[L];
        const u64 *dp = job.d + (((size_t)i * 3 + p) * (size_t)E) * n + k;
        behz_finish_coeff2<TL>(lv, dp, 0, 1, dp + (size_t)L * n, n, res, k);
#pragma unroll
        for (int j = 0; j < L; j++) sum[j] = addmod(sum[j], res[j], lv->q[j].q);
    }
#pragma unroll
    for (int j = 0; j < L; j++) o[(size_t)j * n] = sum[j];
}

void launch_behz_finish(const DevLevel *lv, int L, int nB, const FinishJob *jobs, bool accumulate, size_t n, int njobs, hipStream_t st)
{
    if (!njobs) return;
    const dim3 g = ew_grid(n, njobs * 3), t(EW_T);
    const int acc = accumulate ? 1 : 0;
#define FIN2_CASE(TL) if (L == TL && nB == TL) { hipLaunchKernelGGL((k_behz_finish2<TL>), g, t, 0, st, lv, jobs, acc, n); KERNEL_CHECK(); return; }
    FIN2_CASE(1) FIN2_CASE(2) FIN2_CASE(3)
#undef FIN2_CASE
#define FIN_CASE(TL) if (L == TL && nB == TL) { hipLaunchKernelGGL((k_behz_finish<TL, TL>), g, t, 0, st, lv, jobs, acc, n); } else
    FIN_CASE(4)
    { hipLaunchKernelGGL((k_behz_finish<0, 0>), g, t, 0, st, lv, jobs, acc, n); }
#undef FIN_CASE
    KERNEL_CHECK();
}

// Finish of a summed product: one job = one BinBundle; out[3][L][n] = sum over terms of the finished products.
template <int TL, bool LAZY>
__global__ __launch_bounds__(EW_T) void k_behz_finish_sum(const DevLevel *__restrict__ lv, const FinishSumJob *__restrict__ jobs, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const int L = TL ? TL : lv->L, nBsk = TL ? TL + 1 : lv->nBsk;
    constexpr int LMAX = TL ? TL : DMAXL;
    const FinishSumJob job = jobs[blockIdx.y / 3];
    const size_t p = blockIdx.y % 3;
    const u64 *dq = job.dq + p * (size_t)L * n + k;
    const u64 *db = job.bs + p * (size_t)nBsk * n + k;
    u64 res[LMAX];
    if constexpr (LAZY) behz_finish_coeff2<TL>(lv, dq, (size_t)3 * L * n, job.terms, db, n, res, k);
    else behz_finish_coeff<TL, TL>(lv, dq, (size_t)3 * L * n, job.terms, db, n, res);
    u64 *o = job.out + p * (size_t)L * n + k;
#pragma unroll
    for (int j = 0; j < LMAX; j++)
        if (TL || j < L) o[(size_t)j * n] = res[j];
}

void launch_behz_finish_sum(const DevLevel *lv, int L, int nB, const FinishSumJob *jobs, size_t n, int njobs, hipStream_t st)
{
    if (!njobs) return;
    const dim3 g = ew_grid(n, njobs * 3), t(EW_T);
#define FS_CASE(TL) if (L == TL && nB == TL) { hipLaunchKernelGGL((k_behz_finish_sum<TL, true>), g, t, 0, st, lv, jobs, n); KERNEL_CHECK(); return; }
    FS_CASE(1) FS_CASE(2) FS_CASE(3)
#undef FS_CASE
    if (L == 4 && nB == 4) hipLaunchKernelGGL((k_behz_finish_sum<4, false>), g, t, 0, st, lv, jobs, n);
    else hipLaunchKernelGGL((k_behz_finish_sum<0, false>), g, t, 0, st, lv, jobs, n);
    KERNEL_CHECK();
}

// ============================================================================ K6: relinearize (key switch of c2)
// relinearize_inplace (receiver_osn.cpp:431 ; bin_bundle.cpp:309), App. B10.
// The RNS decomposition c2[J] mod m_I is done by k_ntt_gather on load (no separate pass).
// inner product with the key: acc[b][comp][I][k] = sum_J tdec[b][I][J][k] * rk[J][comp][id(I)][k] mod m_I
template <int TL>
__global__ __launch_bounds__(EW_T) void k_ks_inner(const DevKey *__restrict__ key, int Lrt, const u64 *__restrict__ tdec,
                                                   const u64 *__restrict__ rk, u64 *__restrict__ acc, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const size_t b = blockIdx.y;
    const int L = TL ? TL : Lrt;
    const int K = key->K;
    const u64 *td = tdec + b * (size_t)(L + 1) * L * n;
    u64 *o = acc + b * (size_t)2 * (L + 1) * n;
#pragma unroll
    for (int I = 0; I <= (TL ? TL : DMAXL); I++) {
        if (!TL && I > L) continue;
        const int ki = I == L ? K - 1 : I;
        const Mod m = key->q[ki];
        u128p a0{ 0, 0 }, a1{ 0, 0 };
#pragma unroll
        for (int J = 0; J < (TL ? TL : DMAXL); J++) {
            if (!TL && J >= L) continue;
            const u64 tv = td[((size_t)I * L + J) * n + k];
            mac128(a0, tv, rk[(((size_t)J * 2 + 0) * K + ki) * n + k]);
            mac128(a1, tv, rk[(((size_t)J * 2 + 1) * K + ki) * n + k]);
        }
        o[(size_t)I * n + k] = barrett128(a0, m);
        o[((size_t)(L + 1) + I) * n + k] = barrett128(a1, m);
    }
}

void launch_ks_inner(const DevKey *key, int L, const u64 *tdec, const u64 *rk, u64 *acc, size_t n, int batch,
                     hipStream_t st)
{
#define KS_CASE(TL) case TL: hipLaunchKernelGGL((k_ks_inner<TL>), ew_grid(n, batch), dim3(EW_T), 0, st, key, L, tdec, rk, acc, n); break;
    switch (L) { KS_CASE(1) KS_CASE(2) KS_CASE(3) KS_CASE(4) default: hipLaunchKernelGGL((k_ks_inner<0>), ew_grid(n, batch), dim3(EW_T), 0, st, key, L, tdec, rk, acc, n); }
#undef KS_CASE
    KERNEL_CHECK();
}

// mod-down by the special prime with rounding and add into (c0, c1):
// acc: [batch][2][L+1][n] coefficient form ; ct[b]: [>=2][L][n] at stride ct_stride
// EXT: the first n_ext ciphertexts are operands of later products (ComputePowers' parents): their BEHZ extension
// (steps 1-2, behz_ext2_body) is written to ext + (b*2 + comp)*(2L+1)*n while the new (c0, c1) are still in registers --
// the extension kernel of the next DAG level and its re-read of the ciphertexts disappear.
template <int TL, bool EXT = false, bool RAW = false>
__global__ __launch_bounds__(EW_T) void k_ks_moddown(const DevKey *__restrict__ key, int Lrt, const u64 *__restrict__ acc,
                                                     u64 *__restrict__ ct, size_t ct_stride, size_t n,
                                                     const DevLevel *__restrict__ lv = nullptr, u64 *__restrict__ ext = nullptr, int n_ext = 0)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const size_t b = blockIdx.y;
    const int L = TL ? TL : Lrt;
    const Mod pm = key->q[key->K - 1];
#pragma unroll
    for (int comp = 0; comp < 2; comp++) {
        const u64 *a = acc + (b * 2 + comp) * (size_t)(L + 1) * n;
        u64 *c = ct + b * ct_stride + (size_t)comp * L * n;
        u64 ap = a[(size_t)L * n + k];
        if (RAW) {                                                // RAW: the inverse transform left its twist to this kernel
            const u64x2 tw = ldg16(reinterpret_cast<const u64 *>(key->p_tw + k));
            ap = mul_shoup(ap, tw[0], tw[1], pm.q);
        }
        const u64 tl = barrett64(ap + key->p_half, pm);
        u64 x[TL ? TL : 1];
#pragma unroll
        for (int j = 0; j < (TL ? TL : DMAXL); j++) {
            if (!TL && j >= L) continue;
            const Mod m = key->q[j];
            const u64 tk = submod(barrett64(tl, m), key->p_half_mod[j], m.q);
            u64 v;
            if (RAW) {
                const u64x2 tw = ldg16(reinterpret_cast<const u64 *>(key->md_tw[j] + k));
                const u64 u = mul_shoup_lazy(a[(size_t)j * n + k], tw[0], tw[1], m.q);                         // [0, 2q)
                const u64 w = mul_shoup_lazy(tk, key->inv_p[j].w, key->inv_p[j].wq, m.q);                     // [0, 2q)
                v = csub(csub(u + (m.q << 1) - w, m.q << 1), m.q);
            } else v = mul_shoup(submod(a[(size_t)j * n + k], tk, m.q), key->inv_p[j].w, key->inv_p[j].wq, m.q);
            const u64 r = addmod(c[(size_t)j * n + k], v, m.q);
            c[(size_t)j * n + k] = r;
            if (EXT) x[j] = r;
        }
        if constexpr (EXT && TL > 0) {
            if ((int)b < n_ext) behz_ext2_body<TL>(lv, x, ext + ((b * 2 + comp) * (size_t)(2 * TL + 1)) * n, n, k);
        }
    }
}

// lv / ext / n_ext: see k_ks_moddown<TL, true>; only for L == nB <= 3 (the caller checks), ext == nullptr: plain mod-down
void launch_ks_moddown(const DevKey *key, int L, const u64 *acc, u64 *ct, size_t ct_stride, size_t n, int batch,
                       hipStream_t st, const DevLevel *lv, u64 *ext, int n_ext, bool raw)
{
    const dim3 g = ew_grid(n, batch), t(EW_T);
    if (ext && n_ext > 0 && L >= 1 && L <= 3) {
#define KSX_CASE(TL) case TL: if (raw) hipLaunchKernelGGL((k_ks_moddown<TL, true, true>), g, t, 0, st, key, L, acc, ct, ct_stride, n, lv, ext, n_ext); \
                              else hipLaunchKernelGGL((k_ks_moddown<TL, true>), g, t, 0, st, key, L, acc, ct, ct_stride, n, lv, ext, n_ext); break;
        switch (L) { KSX_CASE(1) KSX_CASE(2) KSX_CASE(3) }
#undef KSX_CASE
        KERNEL_CHECK();
        return;
    }
    if (raw && (L < 1 || L > 4)) throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
#define KS_CASE(TL) case TL: if (raw) hipLaunchKernelGGL((k_ks_moddown<TL, false, true>), g, t, 0, st, key, L, acc, ct, ct_stride, n, nullptr, nullptr, 0); \
                             else hipLaunchKernelGGL((k_ks_moddown<TL>), g, t, 0, st, key, L, acc, ct, ct_stride, n, nullptr, nullptr, 0); break;
    switch (L) { KS_CASE(1) KS_CASE(2) KS_CASE(3) KS_CASE(4) default: hipLaunchKernelGGL((k_ks_moddown<0>), g, t, 0, st, key, L, acc, ct, ct_stride, n, nullptr, nullptr, 0); }
#undef KS_CASE
    KERNEL_CHECK();
}

// ============================================================================ K3: dyadic multiply-accumulate
// The inner loops of BatchedPlaintextPolyn::eval / eval_patstock
// (bin_bundle.cpp:140-149, 250-265, 279-294, 314-324, 328-337): out_g = sum_j C^j (.) a_{g,j} in the NTT
// domain for plaintext streams that share the same ciphertext powers (the inner polynomials of the
// BinBundles of one bundle index).  The HBM-resident plaintexts are streamed exactly once (16-byte
// non-temporal loads, two adjacent coefficients per lane); every power load is shared by G streams.
//
// Carry-free accumulation: both operands are < q < 2^(2s) (s = ceil(bits(q)/2)), so each is split into two
// s-bit halves and the three partial sums  S00 += lo*lo,  Sx += lo*hi + hi*lo,  S11 += hi*hi  are plain
// 64-bit v_mad_u64_u32 accumulations (no carries, no compares): 4 multiply-adds per product and nothing
// else.  `chunk` terms (2*chunk*2^(2s) < 2^64) are accumulated before the sums are recombined
// (S00 + Sx*2^s + S11*2^(2s)) and reduced; for the 48..56-bit coefficient primes a whole inner polynomial
// fits in one chunk.
// KARA: three products per (stream, coefficient, polynomial, term) instead of four -- S00 += a0 c0, S11 += a1 c1,
// Smid += (a0 + a1)(c0 + c1), the cross sum recovered at fold time as Smid - S00 - S11; the power-side sums c0 + c1 are formed
// once per term and shared by the G streams.  A middle product has 2s + 2 bits, so a carry-free chunk is half as long
// (lv->mac_chunk_k) and the carried residue r re-enters as the "term" (a0 c0, mid) = (r mod 2^s, r mod 2^s + (r >> s)).
#ifdef APSU_MAC_STAMPS
// diagnostic build of tools/microbench/macbench.hip only: where a k_mac workgroup's lifetime goes (shader-clock stamps of lane 0 of
// wave 0: entry | descriptor and pointers read | first term consumed | last pair consumed | folded | stored; realtime at entry and exit)
__device__ unsigned long long *g_mac_stamps = nullptr;
#define MAC_STAMP(i) do { if (g_mac_stamps && threadIdx.x == 0) { asm volatile("" ::: "memory"); stamp_[i] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } } while (0)
#else
#define MAC_STAMP(i) do { } while (0)
#endif
// PACKED: the plaintexts are stored bit-packed, limb j at lv->mac_bits[j] bits per coefficient (56-bit primes: 7 bytes instead
// of 8, 50-bit: 6.25).  A lane still issues ONE 16-byte load per term and stream: its two coefficients occupy 2 * bits
// consecutive bits from bit 2 * bits * (k / 2) of the row, i.e. inside the 16-byte window that starts at the dword holding that
// bit (the host picks widths for which shift + 2 * bits <= 128 everywhere); the window is shifted down by the lane's bit offset
// with funnel shifts and the operand halves are cut out of it.  Fewer HBM bytes per term, the same number of load instructions.
// Kept sums (round 4): an empty assembly statement on every partial sum of k_mac's inner loop.  Without it the compiler pairs two
// products first (v_mad_u64_u32 with a zero addend, then one with the first product as addend) and adds the pair to the running sum
// with a separate 64-bit add; with it every product is ONE v_mad_u64_u32 whose addend is the sum: 283 instead of 315 VALU instructions
// per two terms, -2.2 % on the 16M-4096 query (the cycles go to the next query's ComputePowers, which runs next to the scan;
// profiles/r04_ab_mac_kept_sums.txt).  APSU_MAC_NO_KEEP restores the compiler's form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(APSU_MAC_NO_KEEP)
#define MAC_KEEP(v) asm("" : "+v"(v))
#else
#define MAC_KEEP(v) do { } while (0)
#endif
#ifndef APSU_MAC_MINWAVES
#define APSU_MAC_MINWAVES 1                                        // waves per SIMD the register allocation must allow (experiment switch)
#endif
template <int G, int C, bool KARA = false, bool PACKED = false>
__global__ __launch_bounds__(EW_T, APSU_MAC_MINWAVES) void k_mac(const DevLevel *__restrict__ lv, const MacJob *__restrict__ jobs, size_t n, int limb_slow)
{
    static_assert(C == 1 || C == 2, "coefficients per lane");
    static_assert(!PACKED || C == 2, "packed rows are read two coefficients per lane");
#ifdef APSU_MAC_STAMPS
    unsigned long long stamp_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
    MAC_STAMP(0);
    // grid order (launch_mac).  Workgroups go to the XCDs round-robin in launch order, so what is FAST in the grid decides which
    // workgroups are resident behind one L2 together, i.e. how much of the shared powers that L2 has to hold:
    //   limb_slow 0: (block, limb, job)  -- an XCD holds blocks x, x + 8 of every limb of ~10 jobs: 2 * limbs * terms * 8 KiB
    //   limb_slow 1: (block, job, limb)  -- ... of ONE limb of ~32 jobs: 2 * terms * 8 KiB  (-2.5 ... -2.9 % on the 256M-4096 query,
    //                level at 16M-4096; one block per XCD measured level with it: profiles/r04_ab_mac_grid_order.txt)
    unsigned b_x = blockIdx.x, b_limb = blockIdx.y, b_job = blockIdx.z;
    if (limb_slow) { b_limb = blockIdx.z; b_job = blockIdx.y; }
    const size_t k = ((size_t)b_x * EW_T + threadIdx.x) * C;
    if (k >= n) return;
    constexpr int SPLIT = MAC_G / G;                            // a job's streams are covered by SPLIT blocks
    const MacJob *__restrict__ jp = jobs + b_job / SPLIT;        // stream pointers are indexed dynamically: read them from memory
    struct { const u64 *pw; u32 cnt, ng, pt_stride, pw_stride, pw_poly_stride, out_poly_stride, limb0; } job =
        { jp->pw, jp->cnt, jp->ng, jp->pt_stride, jp->pw_stride, jp->pw_poly_stride, jp->out_poly_stride, jp->limb0 };
    const int g0 = (b_job % SPLIT) * G;
    if (g0 >= (int)job.ng || b_limb >= jp->nl) return;
    const int j = b_limb + job.limb0;                          // limb
    const Mod m = lv->q[j];
    const u32 s = lv->mac_shift[j], chunk = KARA ? lv->mac_chunk_k[j] : lv->mac_chunk[j];
    const u32 lomask = (1u << s) - 1;                          // s <= 30
    const u64 *p0 = job.pw + (size_t)j * n + k;
    const u64 *p1 = p0 + job.pw_poly_stride;
    const u64 *pt[G];
    const u32 *ptw[G];                                          // PACKED: first dword of this lane's 16-byte window
    u32 psh = 0, kb = 64, himask = 0xffffffffu;
    if constexpr (PACKED) {
        kb = lv->mac_bits[j];
        himask = lv->mac_mask_hi[j];
#ifdef APSU_MAC_TILED_EXPERIMENT
        if (jp->pad == 1) {                                     // tools/microbench/macbench.hip: the G streams' tiles of one (term, limb, block) adjacent in memory
            const u32 bitoff = threadIdx.x * 2 * kb, tile = EW_T * C * kb / 8, nblk = (u32)(n / (EW_T * C));
            psh = bitoff & 31;
#pragma unroll
            for (int g = 0; g < G; g++)
                ptw[g] = reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(jp->pt[0]) + ((size_t)(j * nblk + b_x) * G + g) * tile) + (bitoff >> 5);
        } else if (jp->pad == 2) {                              // macbench: BLOCK-MAJOR rows -- the terms of one (stream, limb, block) contiguous, a tile per term
            const u32 bitoff = threadIdx.x * 2 * kb, tile = EW_T * C * kb / 8, nblk = (u32)(n / (EW_T * C));
            psh = bitoff & 31;
#pragma unroll
            for (int g = 0; g < G; g++)
                ptw[g] = reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(jp->pt[g0 + g < (int)job.ng ? g0 + g : g0]) + ((size_t)(j * nblk + b_x) * job.cnt) * tile) + (bitoff >> 5);
        } else
#endif
        {
        const u32 bitoff = (u32)(k >> 1) * 2 * kb;
        psh = bitoff & 31;
#pragma unroll
        for (int g = 0; g < G; g++)
            ptw[g] = reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(jp->pt[g0 + g < (int)job.ng ? g0 + g : g0]) + lv->mac_row_off[j]) + (bitoff >> 5);
        }
    } else {
#ifdef APSU_MAC_TILED_EXPERIMENT
        if (jp->pad == 1) {
            const u32 nblk = (u32)(n / (EW_T * C));
#pragma unroll
            for (int g = 0; g < G; g++) pt[g] = jp->pt[0] + ((size_t)(j * nblk + b_x) * G + g) * (EW_T * C) + threadIdx.x * C;
        } else
#endif
#pragma unroll
        for (int g = 0; g < G; g++) pt[g] = jp->pt[g0 + g < (int)job.ng ? g0 + g : g0] + (size_t)j * n + k;   // missing streams alias a real one
    }

    // accumulators [stream][coef][poly]
    u64 s00[G][C][2], sx[G][C][2], s11[G][C][2];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int c = 0; c < C; c++)
#pragma unroll
            for (int p = 0; p < 2; p++) s00[g][c][p] = sx[g][c][p] = s11[g][c][p] = 0;

    struct Term { u64 c[2][C]; u64 a[G][C]; };                  // powers (poly, coef) and plaintext values (stream, coef)
#ifdef APSU_MAC_TILED_EXPERIMENT
    // macbench ROTATE (round 6): every workgroup walks its chain from another starting term (the sum is exact in any order), so that the
    // workgroups resident together do not all read offset t x row-stride of their streams at the same time: -0.9 % stand-alone, does
    // not remove the placement effect (profiles/r06_mac_rotate.txt); not in the library
    const u32 rot_ = jp->pad == 3 ? (u32)((b_job * 7u + b_limb * 13u + b_x * 5u) % job.cnt) : 0u;
#endif
    auto load_term = [&](u32 i, Term &t) {
#ifdef APSU_MAC_TILED_EXPERIMENT
        i += rot_; if (i >= job.cnt) i -= job.cnt;
#endif
        if (C == 2) {
            const u64x2 v0 = ldg16(p0 + (size_t)i * job.pw_stride), v1 = ldg16(p1 + (size_t)i * job.pw_stride);
            t.c[0][0] = v0[0]; t.c[0][C - 1] = v0[1]; t.c[1][0] = v1[0]; t.c[1][C - 1] = v1[1];
#pragma unroll
            for (int g = 0; g < G; g++) {
                if constexpr (PACKED) {
                    const u32x4a4 w = ldg16_a4_nt(ptw[g] + (size_t)i * (job.pt_stride >> 2));      // pt_stride in bytes
                    t.a[g][0] = (u64)w[0] | ((u64)w[1] << 32); t.a[g][C - 1] = (u64)w[2] | ((u64)w[3] << 32);
                } else {
                    const u64x2 a = ldg16_nt(pt[g] + (size_t)i * job.pt_stride);
                    t.a[g][0] = a[0]; t.a[g][C - 1] = a[1];
                }
            }
        } else {
            t.c[0][0] = p0[(size_t)i * job.pw_stride]; t.c[1][0] = p1[(size_t)i * job.pw_stride];
#pragma unroll
            for (int g = 0; g < G; g++) t.a[g][0] = __builtin_nontemporal_load(pt[g] + (size_t)i * job.pt_stride);
        }
    };
    auto mac_term = [&](const Term &t) {
        u32 clo[2][C], chi[2][C];
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int c = 0; c < C; c++) { clo[p][c] = (u32)t.c[p][c] & lomask; chi[p][c] = (u32)(t.c[p][c] >> s); }
        u32 csum[2][C];
        if (KARA) {
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int c = 0; c < C; c++) csum[p][c] = clo[p][c] + chi[p][c];
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            u64 av[C];
            if constexpr (PACKED) {
                // window >> psh, then coefficient 0 = bits [0, kb), coefficient 1 = bits [kb, 2 kb)
                const u32 w0 = (u32)t.a[g][0], w1 = (u32)(t.a[g][0] >> 32), w2 = (u32)t.a[g][C - 1], w3 = (u32)(t.a[g][C - 1] >> 32);
                const u32 n0 = __builtin_amdgcn_alignbit(w1, w0, psh), n1 = __builtin_amdgcn_alignbit(w2, w1, psh),
                          n2 = __builtin_amdgcn_alignbit(w3, w2, psh), n3 = w3 >> psh;
                const u64 lo64 = (u64)n0 | ((u64)n1 << 32), hi64 = (u64)n2 | ((u64)n3 << 32);
                av[0] = lo64;
                av[C - 1] = kb == 64 ? hi64 : ((lo64 >> kb) | (hi64 << (64 - kb)));
            } else {
#pragma unroll
                for (int c = 0; c < C; c++) av[c] = t.a[g][c];
            }
#pragma unroll
            for (int c = 0; c < C; c++) {
                const u32 alo = (u32)av[c] & lomask, ahi = PACKED ? ((u32)(av[c] >> s) & himask) : (u32)(av[c] >> s);
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    // (MAC_KEEP: an empty statement on every sum keeps the compiler from pairing two products first and adding the pair
                    //  to the sum with a separate 64-bit add: one v_mad_u64_u32 per product and nothing else)
                    s00[g][c][p] += (u64)alo * clo[p][c]; MAC_KEEP(s00[g][c][p]);
                    if (KARA) { sx[g][c][p] += (u64)(alo + ahi) * csum[p][c]; MAC_KEEP(sx[g][c][p]); }
                    else {
                        sx[g][c][p] += (u64)alo * chi[p][c]; MAC_KEEP(sx[g][c][p]);
                        sx[g][c][p] += (u64)ahi * clo[p][c]; MAC_KEEP(sx[g][c][p]);
                    }
                    s11[g][c][p] += (u64)ahi * chi[p][c]; MAC_KEEP(s11[g][c][p]);
                }
            }
        }
    };
    // recombine S00 + Sx*2^s + S11*2^(2s) (< 2^128) and reduce; the residue re-enters as the next chunk's S00
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

    const u32 cnt = job.cnt;
    Term A, B;                                                   // ping-pong register sets: no copies
    MAC_STAMP(1);
    load_term(0, A);
    u32 in_chunk = 0;
    const u32 npairs = cnt >> 1;
    for (u32 pr = 0; pr < npairs; pr++) {                        // branch-free body: two terms per trip
        const u32 i = pr * 2;
        load_term(i + 1, B);
        mac_term(A);
#ifdef APSU_MAC_STAMPS
        if (pr == 0) MAC_STAMP(2);
#endif
        load_term(i + 2 < cnt ? i + 2 : cnt - 1, A);             // clamped prefetch (a re-read hits the cache)
        mac_term(B);
        in_chunk += 2;
        if (in_chunk + 3 > chunk) { fold(false); in_chunk = 1; } // the folded residue counts as one term
    }
    if (cnt & 1) mac_term(A);                                    // A holds the last term
    MAC_STAMP(3);
    fold(true);
    MAC_STAMP(4);
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g0 + g < (int)job.ng) {
            u64 *o = jp->out[g0 + g] + (size_t)b_limb * n + k;
            if (C == 2) {
                u64x2 r0, r1;
                r0[0] = s00[g][0][0]; r0[1] = s00[g][C - 1][0];
                r1[0] = s00[g][0][1]; r1[1] = s00[g][C - 1][1];
                *reinterpret_cast<u64x2 *>(o) = r0;
                *reinterpret_cast<u64x2 *>(o + job.out_poly_stride) = r1;
            } else {
                o[0] = s00[g][0][0];
                o[job.out_poly_stride] = s00[g][0][1];
            }
        }
    }
#ifdef APSU_MAC_STAMPS
    if (g_mac_stamps && threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);                           // the stores have left
        MAC_STAMP(5);
        const size_t lin = blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
        unsigned long long *o = g_mac_stamps + lin * 8;
        for (int i = 0; i < 6; i++) o[i] = stamp_[i];
        o[6] = rt0_; o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifndef APSU_MAC_G
#define APSU_MAC_G 4
#endif
#ifndef APSU_MAC_C
#define APSU_MAC_C 2
#endif
void launch_mac(const DevLevel *lv, int nlimbs, const MacJob *jobs, size_t n, int njobs, hipStream_t st, bool kara, bool packed)
{
    if (!njobs || !nlimbs) return;
    constexpr int G = APSU_MAC_G, C = APSU_MAC_C;
    // (round 4, measured and not adopted -- tools/microbench/mac_persist.hip, profiles/r04_mac_{units,persist,stagger}.txt: a launch
    //  costs ~0.26 ms more than its chains' length explains, i.e. ~14 us per workgroup; long-lived workgroups that keep the load
    //  pipeline running across chains were 4-9 % SLOWER, starting the first resident generation in phases changed nothing)
    const unsigned gx = (unsigned)((n / C + EW_T - 1) / EW_T), gl = (unsigned)nlimbs, gj = (unsigned)(njobs * (MAC_G / G));
    const int ls = gj <= 65535u ? 1 : 0;                          // limb slowest (k_mac) unless the jobs do not fit grid dimension y
    const dim3 grid = ls ? dim3(gx, gj, gl) : dim3(gx, gl, gj);
    if (packed) {
        if constexpr (C == 2) {
            if (kara) hipLaunchKernelGGL((k_mac<G, C, true, true>), grid, dim3(EW_T), 0, st, lv, jobs, n, ls);
            else hipLaunchKernelGGL((k_mac<G, C, false, true>), grid, dim3(EW_T), 0, st, lv, jobs, n, ls);
        } else throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    } else if (kara) hipLaunchKernelGGL((k_mac<G, C, true>), grid, dim3(EW_T), 0, st, lv, jobs, n, ls);
    else hipLaunchKernelGGL((k_mac<G, C, false>), grid, dim3(EW_T), 0, st, lv, jobs, n, ls);
    KERNEL_CHECK();
}

// ---- single dyadic products on ONE limb (round 4): out[p][k] = a[k] * C_p[k] mod q_limb for p = 0, 1.
// The i = 0 block of eval_patstock (bin_bundle.cpp:314-324) switches every term a_j (.) C^j to the next level on its own, so
// the dropped limb of every term is needed by itself (k_i0_finish): terms x BinBundles chains of length ONE.  As k_mac jobs each
// of them paid a whole workgroup's fixed costs (descriptor and pointer reads, the first term's latency, the 128-bit fold, the
// drain of the stores: ~10 us at two workgroups per CU) for 1.4 us of work -- 9 % of the launch for 2.6 % of its bytes at
// 16M-4096, more at 256M-4096 (31 620 such chains).  Here: one thread per coefficient pair, ~40 registers, full occupancy,
// the same canonical residue (k_mac's fold of a single product IS barrett128 of that product).
template <bool PACKED>
__global__ __launch_bounds__(EW_T) void k_term_product(const DevLevel *__restrict__ lv, const TermJob *__restrict__ jobs, size_t njobs, size_t n, int limb,
                                                       u32 pw_poly_stride, u32 out_poly_stride)
{
    const size_t k = ((size_t)blockIdx.x * EW_T + threadIdx.x) * 2;
    const size_t u = blockIdx.y + (size_t)gridDim.y * blockIdx.z;
    if (k >= n || u >= njobs) return;
    const TermJob job = jobs[u];
    const Mod m = lv->q[limb];
    u64 a0, a1;
    if constexpr (PACKED) {
        const u32 kb = lv->mac_bits[limb];
        const u32 bitoff = (u32)(k >> 1) * 2 * kb, psh = bitoff & 31;
        const u32x4a4 w = ldg16_a4_nt(reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(job.pt) + lv->mac_row_off[limb]) + (bitoff >> 5));
        const u32 n0 = __builtin_amdgcn_alignbit(w[1], w[0], psh), n1 = __builtin_amdgcn_alignbit(w[2], w[1], psh),
                  n2 = __builtin_amdgcn_alignbit(w[3], w[2], psh), n3 = w[3] >> psh;
        const u64 lo64 = (u64)n0 | ((u64)n1 << 32), hi64 = (u64)n2 | ((u64)n3 << 32);
        if (kb == 64) { a0 = lo64; a1 = hi64; }
        else { const u64 mask = ((u64)1 << kb) - 1; a0 = lo64 & mask; a1 = ((lo64 >> kb) | (hi64 << (64 - kb))) & mask; }
    } else {
        const u64x2 a = ldg16_nt(job.pt + (size_t)limb * n + k);
        a0 = a[0]; a1 = a[1];
    }
    const u64 *pw = job.pw + (size_t)limb * n + k;
    const u64x2 c0 = ldg16(pw), c1 = ldg16(pw + pw_poly_stride);
    u64x2 r0, r1;
    r0[0] = barrett128(mul128(a0, c0[0]), m); r0[1] = barrett128(mul128(a1, c0[1]), m);
    r1[0] = barrett128(mul128(a0, c1[0]), m); r1[1] = barrett128(mul128(a1, c1[1]), m);
    *reinterpret_cast<u64x2 *>(job.out + k) = r0;
    *reinterpret_cast<u64x2 *>(job.out + out_poly_stride + k) = r1;
}

void launch_term_product(const DevLevel *lv, const TermJob *jobs, size_t njobs, size_t n, int limb, u32 pw_poly_stride, u32 out_poly_stride,
                         bool packed, hipStream_t st)
{
    if (!njobs) return;
    const unsigned gy = (unsigned)std::min<size_t>(njobs, 32768), gz = (unsigned)((njobs + gy - 1) / gy);
    const dim3 grid((unsigned)((n / 2 + EW_T - 1) / EW_T), gy, gz);
    if (packed) hipLaunchKernelGGL((k_term_product<true>), grid, dim3(EW_T), 0, st, lv, jobs, njobs, n, limb, pw_poly_stride, out_poly_stride);
    else hipLaunchKernelGGL((k_term_product<false>), grid, dim3(EW_T), 0, st, lv, jobs, njobs, n, limb, pw_poly_stride, out_poly_stride);
    KERNEL_CHECK();
}

// ---- bit-packed database rows: dense u64 limbs <-> rows of mac_bits[j] bits per coefficient (DevLevel)
// one thread per OUTPUT dword: bits [32 d, 32 d + 32) of the row come from at most two coefficients (widths are >= 32)
__global__ __launch_bounds__(EW_T) void k_pack_rows(const DevLevel *__restrict__ lv, int L, const u64 *__restrict__ dense, char *__restrict__ packed,
                                                    size_t slot_bytes, size_t n)
{
    const size_t slot = blockIdx.y / L;
    const int j = (int)(blockIdx.y % L);
    const u32 w = lv->mac_bits[j];
    const size_t d = (size_t)blockIdx.x * EW_T + threadIdx.x, ndw = n * w / 32;
    if (d >= ndw) return;
    const u64 *src = dense + (slot * L + j) * n;
    const size_t bit0 = d * 32, c0 = bit0 / w;
    const u32 off = (u32)(bit0 - c0 * w), got = w - off;
    u64 v = src[c0] >> off;
    if (got < 32 && c0 + 1 < n) v |= src[c0 + 1] << got;
    reinterpret_cast<u32 *>(packed + slot * slot_bytes + lv->mac_row_off[j])[d] = (u32)v;
}
// one thread per coefficient (a bit-packed slot is followed by at least 16 readable bytes: the engine pads its buffers)
__global__ __launch_bounds__(EW_T) void k_unpack_rows(const DevLevel *__restrict__ lv, int L, const char *__restrict__ packed, size_t slot_bytes,
                                                      u64 *__restrict__ dense, size_t n)
{
    const size_t slot = blockIdx.y / L;
    const int j = (int)(blockIdx.y % L);
    const size_t c = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (c >= n) return;
    const u32 w = lv->mac_bits[j];
    const u32 *row = reinterpret_cast<const u32 *>(packed + slot * slot_bytes + lv->mac_row_off[j]);
    const size_t bit0 = c * w, d0 = bit0 >> 5;
    const u32 sh = (u32)(bit0 & 31);
    u64 v = ((u64)row[d0] | ((u64)row[d0 + 1] << 32)) >> sh;
    if (sh && w + sh > 64) v |= (u64)row[d0 + 2] << (64 - sh);
    dense[(slot * L + j) * n + c] = w == 64 ? v : (v & (((u64)1 << w) - 1));
}
void launch_pack_rows(const DevLevel *lv, int L, const u64 *dense, void *packed, size_t slot_bytes, size_t n, size_t slots, hipStream_t st)
{
    if (!slots) return;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((n * 2 + EW_T - 1) / EW_T), (unsigned)(slots * L)), dim3(EW_T), 0, st, lv, L, dense,
                       static_cast<char *>(packed), slot_bytes, n);
    KERNEL_CHECK();
}
void launch_unpack_rows(const DevLevel *lv, int L, const void *packed, size_t slot_bytes, u64 *dense, size_t n, size_t slots, hipStream_t st)
{
    if (!slots) return;
    hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)((n + EW_T - 1) / EW_T), (unsigned)(slots * L)), dim3(EW_T), 0, st, lv, L,
                       static_cast<const char *>(packed), slot_bytes, dense, n);
    KERNEL_CHECK();
}

// Fused tail of BatchedPlaintextPolyn::eval / eval_patstock (bin_bundle.cpp:159-171, 345-357):
// add_plain(a_0), add_plain(random_plain) [K8], mod_switch_to_next down to the last level [K7],
// try_clear_irrelevant_bits [K9] — plus up to two exact addends (coefficient-form sums) folded in.
__device__ __forceinline__ u64 scaled_plain(const DevLevel *__restrict__ lv, u64 m, u64 fix, int j)
{
    u128p s = mul128(m, lv->coeff_div_plain[j]);
    add128(s, u128p{ fix, 0 });
    return barrett128(s, lv->q[j]);
}
__device__ __forceinline__ u64 plain_fix(const DevLevel *__restrict__ lv, u64 m)
{
    // floor((m * (Q mod t) + floor((t+1)/2)) / t), exact (m < t < 2^61)
    u128p num = mul128(m, lv->q_mod_t);
    add128(num, u128p{ lv->threshold, 0 });
    const u64 t = lv->t;
    if (num.hi == 0) return num.lo / t;
    u64 rem = num.hi % t, lo = num.lo, fix = 0;
    for (int i = 0; i < 64; i++) {
        rem = (rem << 1) | (lo >> 63);
        lo <<= 1;
        fix <<= 1;
        if (rem >= t) { rem -= t; fix |= 1; }
    }
    return fix;
}

// MD (round 6): the key switch's mod-down (k_ks_moddown<., false, RAW>, App. B10) runs HERE, on the way in: job.ks_acc = the RAW inverse
// transforms of the key-switch sums [2][L+1][n]; (c0, c1) + the rounded quotient never goes to memory and the mod-down launch in front
// of this kernel is gone (one launch and one boundary per query: -6 us on the N = 8 shard's critical path, profiles/r06_ab_fused_tail.txt).
template <bool MD>
__global__ __launch_bounds__(EW_T) void k_eval_epilogue(const DevLevel *__restrict__ levels, int lvl, const EpiJob *__restrict__ jobs,
                                                        size_t ct_poly_stride, u64 clear_mask, size_t n, const DevKey *__restrict__ key)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const EpiJob job = jobs[blockIdx.y];
    const DevLevel *lv = levels + lvl;
    const int L = lvl + 1;
    u64 v[2][DMAXL];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        u64 tl = 0;
        const u64 *a = nullptr;
        if constexpr (MD) {                                      // the special limb of this polynomial's key-switch sum, twisted and rounded
            a = job.ks_acc + (size_t)p * (L + 1) * n;
            const Mod pm = key->q[key->K - 1];
            const u64x2 tw = ldg16(reinterpret_cast<const u64 *>(key->p_tw + k));
            tl = barrett64(mul_shoup(a[(size_t)L * n + k], tw[0], tw[1], pm.q) + key->p_half, pm);
        }
#pragma unroll
        for (int j = 0; j < DMAXL; j++)
            if (j < L) {
                const u64 q = lv->q[j].q;
                u64 x = job.ct[p * ct_poly_stride + (size_t)j * n + k];
                if constexpr (MD) {                              // exactly k_ks_moddown's RAW arithmetic
                    const Mod m = key->q[j];
                    const u64 tk = submod(barrett64(tl, m), key->p_half_mod[j], m.q);
                    const u64x2 tw = ldg16(reinterpret_cast<const u64 *>(key->md_tw[j] + k));
                    const u64 u = mul_shoup_lazy(a[(size_t)j * n + k], tw[0], tw[1], m.q);
                    const u64 w = mul_shoup_lazy(tk, key->inv_p[j].w, key->inv_p[j].wq, m.q);
                    x = addmod(x, csub(csub(u + (m.q << 1) - w, m.q << 1), m.q), m.q);
                }
                if (job.add1) x = addmod(x, job.add1[((size_t)p * L + j) * n + k], q);
                if (job.add2) x = addmod(x, job.add2[((size_t)p * L + j) * n + k], q);
                v[p][j] = x;
            }
    }
    {   // c0 += round(a0 * Q / t) + round(mask * Q / t)
        const u64 m0 = job.a0[k], m1 = job.mask[k];
        const u64 f0 = plain_fix(lv, m0), f1 = plain_fix(lv, m1);
#pragma unroll
        for (int j = 0; j < DMAXL; j++)
            if (j < L) {
                const u64 q = lv->q[j].q;
                v[0][j] = addmod(addmod(v[0][j], scaled_plain(lv, m0, f0, j), q), scaled_plain(lv, m1, f1, j), q);
            }
    }
    for (int l = lvl; l > 0; l--) {                              // drop q_l with rounding (App. B8)
        const DevLevel *ll = levels + l;
        const u64 ql = ll->q[l].q;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            u64 last = 0;
#pragma unroll
            for (int j = 0; j < DMAXL; j++) if (j == l) last = v[p][j];
            last = addmod(last, ll->half, ql);
#pragma unroll
            for (int j = 0; j < DMAXL; j++)
                if (j < l) {
                    const Mod m = ll->q[j];
                    const u64 tmp = submod(barrett64(last, m), ll->half_mod[j], m.q);
                    v[p][j] = mul_shoup(submod(v[p][j], tmp, m.q), ll->inv_q_last[j].w, ll->inv_q_last[j].wq, m.q);
                }
        }
    }
    job.out[k] = v[0][0] & clear_mask;
    job.out[n + k] = v[1][0] & clear_mask;
}

// key != nullptr: every job carries ks_acc and the key switch's mod-down is part of this launch (k_eval_epilogue<true>)
void launch_eval_epilogue(const DevLevel *levels, int lvl, const EpiJob *jobs, size_t ct_poly_stride, int clear_bits, size_t n,
                          int njobs, hipStream_t st, const DevKey *key)
{
    if (!njobs) return;
    const u64 mask = clear_bits > 0 ? ~(((u64)1 << clear_bits) - 1) : ~(u64)0;
    if (key) hipLaunchKernelGGL(k_eval_epilogue<true>, ew_grid(n, njobs), dim3(EW_T), 0, st, levels, lvl, jobs, ct_poly_stride, mask, n, key);
    else hipLaunchKernelGGL(k_eval_epilogue<false>, ew_grid(n, njobs), dim3(EW_T), 0, st, levels, lvl, jobs, ct_poly_stride, mask, n, key);
    KERNEL_CHECK();
}

// Sum of `terms` individually rounded drop-last-limb results, computed from the exact sum S of the kept
// limbs and the per-term last limbs V (all coefficient form):  SURVEY note N1 / DESIGN.md §4.
template <bool RAW>
__global__ __launch_bounds__(EW_T) void k_i0_finish(const DevLevel *__restrict__ lv, const I0Job *__restrict__ jobs, size_t n)
{
    const size_t k = (size_t)blockIdx.x * EW_T + threadIdx.x;
    if (k >= n) return;
    const I0Job job = jobs[blockIdx.y >> 1];
    const int p = blockIdx.y & 1;
    const int L = lv->L;                                        // level BEFORE the drop
    const u64 ql = lv->q[L - 1].q, half = lv->half;
    u64 R = 0;                                                  // integer sum of (v + half) mod q_last; terms*q_last < 2^64 (host-checked)
    const u64 *v = job.v + (size_t)p * n + k;
    u64 tw0 = 0, tw1 = 0;
    if (RAW) { const u64x2 tw = ldg16(reinterpret_cast<const u64 *>(lv->last_tw + k)); tw0 = tw[0]; tw1 = tw[1]; }
    for (int t = 0; t < job.terms; t++) {
        u64 x = v[(size_t)t * 2 * n];
        if (RAW) x = mul_shoup(x, tw0, tw1, ql);                // RAW: the inverse transform left its twist to this kernel
        R += addmod(x, half, ql);
    }
    for (int m = 0; m + 1 < L; m++) {
        const Mod mq = lv->q[m];
        // terms * (half mod q_m) - (R mod q_m)
        const u64 th = barrett128(mul128((u64)job.terms, lv->half_mod[m]), mq);
        const u64 corr = submod(th, barrett64(R, mq), mq.q);
        const size_t o = ((size_t)p * (L - 1) + m) * n + k;
        u64 r;
        if (RAW) {
            // (s + corr) q_last^-1 with s = raw * twist: the twist rides on the first product's constant
            const u64x2 tw = ldg16(reinterpret_cast<const u64 *>(lv->drop_tw[m] + k));
            const u64 a = mul_shoup_lazy(job.s[o], tw[0], tw[1], mq.q);                                         // [0, 2q)
            const u64 b = mul_shoup_lazy(corr, lv->inv_q_last[m].w, lv->inv_q_last[m].wq, mq.q);              // [0, 2q)
            r = csub(csub(a + b, mq.q << 1), mq.q);
        } else {
            const u64 val = addmod(job.s[o], corr, mq.q);
            r = mul_shoup(val, lv->inv_q_last[m].w, lv->inv_q_last[m].wq, mq.q);
        }
        job.acc[o] = job.store ? r : addmod(job.acc[o], r, mq.q);
    }
}

void launch_i0_finish(const DevLevel *lv_low, const I0Job *jobs, size_t n, int njobs, hipStream_t st, bool raw)
{
    if (!njobs) return;
    if (raw) hipLaunchKernelGGL(k_i0_finish<true>, ew_grid(n, njobs * 2), dim3(EW_T), 0, st, lv_low, jobs, n);
    else hipLaunchKernelGGL(k_i0_finish<false>, ew_grid(n, njobs * 2), dim3(EW_T), 0, st, lv_low, jobs, n);
    KERNEL_CHECK();
}

} // namespace apsu_he
