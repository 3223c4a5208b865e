// Negacyclic NTT / INTT over one RNS limb, LDS-resident.
//
// Replaces seal::Evaluator::transform_{to,from}_ntt_inplace
//   (call sites receiver/apsu/receiver_osn.cpp:467,475 ; receiver/apsu/bin_bundle.cpp:154,268,297,321)
// and every NTT hidden inside multiply / relinearize / multiply_plain.
//
// Ordering contract (observable through relin keys and NTT-form DB plaintexts):
//   forward : natural-order input -> bit-reversed output, out[i] = a(psi^(2*brv(i)+1)),
//             psi = minimal primitive 2n-th root;   inverse = exact inverse incl. n^-1.
//
// Structure: one workgroup owns one limb polynomial (n <= 8192 coefficients = 64 KiB of the CU's
// 160 KiB LDS).  The log2(n) butterfly stages are grouped into register-resident passes of K
// stages (radix 2^K); between passes the data is exchanged through LDS.  Butterflies are Harvey
// lazy butterflies: forward values live in [0,4q), inverse values in [0,2q).
// The pass functions are __host__ __device__ so tests can emulate a workgroup on the CPU.
#pragma once
#include "modmath.h"

struct TwPair { u64 w, wq; };          // twiddle and its Shoup quotient, 16 B -> one dwordx4 load

// Per-modulus device tables.  fwd[k] = psi^brv(k), inv[k] = psi^-brv(k)  (k = m + i).
struct NttTable {
    u64 q;
    u64 ninv, ninv_q;                  // n^-1 mod q and its Shoup quotient
    const TwPair *fwd;
    const TwPair *inv;
};

// LDS padding: one 8-byte slot per 16 elements, so that the last pass (16 contiguous
// coefficients per lane, lane stride 128 B) is bank-conflict free for ds_read/write_b64.
HD int lds_slot(int e) { return e + (e >> 4); }
constexpr int lds_slots(int n) { return n + (n >> 4); }

// One forward (Cooley-Tukey) pass over stages s .. s+K-1 for work item w in [0, n >> K).
template <int LOGN, int K>
HD void ntt_fwd_pass(u64 *lds, int w, int s, const TwPair *__restrict__ W, u64 q)
{
    constexpr int R = 1 << K;
    const int lowbits = LOGN - s - K;
    const int block = w >> lowbits;
    const int col = w & ((1 << lowbits) - 1);
    const int base = (block << (LOGN - s)) | col;
    const u64 q2 = q << 1;
    u64 r[R];
#pragma unroll
    for (int j = 0; j < R; j++) r[j] = lds[lds_slot(base | (j << lowbits))];
#pragma unroll
    for (int u = 0; u < K; u++) {
        const int bit = 1 << (K - 1 - u);
        const int tw_base = (1 << (s + u)) + (block << u);
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (j & bit) continue;
            const TwPair t = W[tw_base + (j >> (K - u))];
            u64 x = csub(r[j], q2);
            u64 v = mul_shoup_lazy(r[j | bit], t.w, t.wq, q);
            r[j] = x + v;
            r[j | bit] = x - v + q2;
        }
    }
#pragma unroll
    for (int j = 0; j < R; j++) lds[lds_slot(base | (j << lowbits))] = r[j];
}

// One inverse (Gentleman-Sande) pass over stages s+K-1 .. s (same index algebra, reverse order).
template <int LOGN, int K>
HD void ntt_inv_pass(u64 *lds, int w, int s, const TwPair *__restrict__ W, u64 q)
{
    constexpr int R = 1 << K;
    const int lowbits = LOGN - s - K;
    const int block = w >> lowbits;
    const int col = w & ((1 << lowbits) - 1);
    const int base = (block << (LOGN - s)) | col;
    const u64 q2 = q << 1;
    u64 r[R];
#pragma unroll
    for (int j = 0; j < R; j++) r[j] = lds[lds_slot(base | (j << lowbits))];
#pragma unroll
    for (int u = K - 1; u >= 0; u--) {
        const int bit = 1 << (K - 1 - u);
        const int tw_base = (1 << (s + u)) + (block << u);
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (j & bit) continue;
            const TwPair t = W[tw_base + (j >> (K - u))];
            u64 x = r[j], y = r[j | bit];
            r[j] = csub(x + y, q2);
            r[j | bit] = mul_shoup_lazy(x - y + q2, t.w, t.wq, q);
        }
    }
#pragma unroll
    for (int j = 0; j < R; j++) lds[lds_slot(base | (j << lowbits))] = r[j];
}

// Pass schedule: stage counts per pass (summing to LOGN), resolved at compile time.
constexpr int plan_passes(int logn)
{
    return logn == 13 ? 4 : logn == 12 ? 3 : logn == 11 ? 3 : logn == 10 ? 3 : logn == 8 ? 2 : logn == 6 ? 2 : 0;
}
constexpr int plan_k(int logn, int p)
{
    switch (logn) {
    case 13: return p == 3 ? 4 : 3;               // 3,3,3,4
    case 12: return 4;                            // 4,4,4
    case 11: return p == 2 ? 3 : 4;               // 4,4,3
    case 10: return p == 0 ? 4 : 3;               // 4,3,3   (parity tests only)
    case 8:  return 4;                            // 4,4     (parity tests only)
    case 6:  return 3;                            // 3,3     (parity tests only)
    default: return 0;
    }
}
constexpr int plan_s(int logn, int p)
{
    int s = 0;
    for (int i = 0; i < p; i++) s += plan_k(logn, i);
    return s;
}
constexpr int plan_max_k(int logn)
{
    int m = 0;
    for (int i = 0; i < plan_passes(logn); i++) m = plan_k(logn, i) > m ? plan_k(logn, i) : m;
    return m;
}

// Executes pass number PASS (in execution order) for "thread" tid of a T-thread workgroup.
// The caller separates passes with __syncthreads() (device) or by looping tid (host emulation).
template <int LOGN, bool INV, int PASS>
HD void ntt_pass(u64 *lds, int tid, int T, const NttTable &tab)
{
    constexpr int P = plan_passes(LOGN);
    constexpr int p = INV ? P - 1 - PASS : PASS;      // the inverse walks the passes last-to-first
    constexpr int K = plan_k(LOGN, p);
    constexpr int S = plan_s(LOGN, p);
    for (int w = tid; w < (1 << (LOGN - K)); w += T) {
        if (INV) ntt_inv_pass<LOGN, K>(lds, w, S, tab.inv, tab.q);
        else ntt_fwd_pass<LOGN, K>(lds, w, S, tab.fwd, tab.q);
    }
}

// Final range fix-ups applied when the limb leaves LDS.
HD u64 ntt_fwd_finish(u64 x, u64 q) { return csub(csub(x, q << 1), q); }            // [0,4q) -> [0,q)
HD u64 ntt_inv_finish(u64 x, const NttTable &t) { return mul_shoup(x, t.ninv, t.ninv_q, t.q); }
