// Round-3 experiment, NOT part of the library: the key switch's inner product with the key (App. B10) formed by the load of the
// inverse transform behind it, the way k_intt_tensor forms the BEHZ tensor product.  Same bits (87 GPU tests), but measured
// with tools/ab_test.py on the whole 16M-4096 query: +0.073 +- 0.018 ms (+2.2 %), N = 8 shard +0.4 % (profiles/r03_ab_fusions.txt):
// six operand streams per output and tdec read twice (once per key component) cost more than k_ks_inner's own pass.
// To rebuild it: SrcKs goes next to SrcTensor in ntt_core.h, k_intt_ks / launch_intt_ks next to k_intt_tensor in kernels.hip, and
// Engine::d_relinearize calls launch_intt_ks(hp_.logn, tdec, rk.data.u(), acc, L, hp_.K, batch, tabs(), amap, st_) in place of
// launch_ks_inner + d_ntt(acc, ...).
#if 0
// The key switch's inner product with the key, computed on load in front of the inverse transform (App. B10):
// value(e) = sum_{J < terms} td[J*td_stride + e] * rk[J*rk_stride + e] mod q, terms <= 4 (one per decomposition limb).
struct SrcKs { const u64 *td, *rk; size_t td_stride, rk_stride; int terms; };
HD u64x2 src_load2(const SrcKs &s, const u64 *, int e, const NttTable &tab)
{
    u128p p0{ 0, 0 }, p1{ 0, 0 };
    for (int J = 0; J < s.terms; J++) {                          // wave-uniform trip count
        const u64x2 x = ldg16(s.td + (size_t)J * s.td_stride + e), y = ldg16(s.rk + (size_t)J * s.rk_stride + e);
        mac128(p0, x[0], y[0]);
        mac128(p1, x[1], y[1]);
    }
    u64x2 r;
    r[0] = ntt_reduce128(p0.hi, p0.lo, tab);
    r[1] = ntt_reduce128(p1.hi, p1.lo, tab);
    return r;
}
HD u64 src_load1(const SrcKs &s, const u64 *, int e, const NttTable &tab)
{
    u128p p{ 0, 0 };
    for (int J = 0; J < s.terms; J++) mac128(p, s.td[(size_t)J * s.td_stride + e], s.rk[(size_t)J * s.rk_stride + e]);
    return ntt_reduce128(p.hi, p.lo, tab);
}

// Inverse NTT of the key switch's inner products (App. B10, the accumulation step and the transform behind it in one
// launch): workgroup g owns acc[b][comp][I] with (b, comp, I) = g / (2 (L+1)), (g / (L+1)) % 2, g % (L+1), forms
// sum_J tdec[b][I][J] (.) rk[J][comp][ki(I)] while it loads (SrcKs) and writes the coefficient-form limb: k_ks_inner, its
// 2 (L+1) limb writes per ciphertext and the transform's re-read of them are gone.  L <= 4.
template <int LOGN, int T>
__global__ __launch_bounds__(T, 4) void k_intt_ks(const u64 *__restrict__ tdec, const u64 *__restrict__ rk, u64 *__restrict__ acc, int L, int K,
                                               const NttTable *__restrict__ tabs, const int *__restrict__ modmap)
{
    constexpr int N = 1 << LOGN;
    __shared__ __attribute__((aligned(16))) u64 lds[lds_slots(N)];
    const int tid = threadIdx.x;
    const size_t g = blockIdx.x, per = (size_t)2 * (L + 1), b = g / per;
    const int r = (int)(g - b * per), comp = r / (L + 1), I = r - comp * (L + 1);
    const int ki = I == L ? K - 1 : I;
    const NttTable tab = tabs[modmap[I] & NTT_MAP_MASK];
    const SrcKs ops{ tdec + ((b * (L + 1) + I) * L) * N, rk + ((size_t)comp * K + ki) * N, (size_t)N, (size_t)2 * K * N, L };
    u64 *p = acc + g * N;
    if (modmap[I] & NTT_MAP_RAW) {                                              // wave-uniform
        if (tab.narrow) ntt_body<LOGN, true, NTT_NARROW, T, false, true, SrcKs>(lds, p, tab, tid, nullptr, ops);
        else if (tab.wide_d4) ntt_body<LOGN, true, NTT_WIDE_NEAR, T, false, true, SrcKs>(lds, p, tab, tid, nullptr, ops);
        else ntt_body<LOGN, true, NTT_WIDE, T, false, true, SrcKs>(lds, p, tab, tid, nullptr, ops);
        return;
    }
    if (tab.narrow) ntt_body<LOGN, true, NTT_NARROW, T, false, false, SrcKs>(lds, p, tab, tid, nullptr, ops);
    else if (tab.wide_d4) ntt_body<LOGN, true, NTT_WIDE_NEAR, T, false, false, SrcKs>(lds, p, tab, tid, nullptr, ops);
    else ntt_body<LOGN, true, NTT_WIDE, T, false, false, SrcKs>(lds, p, tab, tid, nullptr, ops);
}

void launch_intt_ks(int logn, const u64 *tdec, const u64 *rk, u64 *acc, int L, int K, int batch, const NttTable *tabs, const int *modmap,
                    hipStream_t st)
{
    const size_t count = (size_t)batch * 2 * (L + 1);
    if (!count) return;
#define K_CASE(LN, T) case LN: hipLaunchKernelGGL((k_intt_ks<LN, T>), dim3((unsigned)count), dim3(T), 0, st, tdec, rk, acc, L, K, tabs, modmap); break;
    switch (logn) {
    K_CASE(14, 1024) K_CASE(13, 512) K_CASE(12, 256) K_CASE(11, 128) K_CASE(10, 64) K_CASE(8, 64) K_CASE(6, 64)
    default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    }
#undef K_CASE
    KERNEL_CHECK();
}

#endif
