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
// 160 KiB LDS).  Every thread keeps 16 coefficients in registers per pass: a pass covers K
// consecutive butterfly stages (radix 2^K) on 16 >> K independent groups (adjacent columns, so one
// twiddle load feeds all groups and global / LDS accesses are 16 B per lane).  Between passes data
// is exchanged through LDS; the first forward pass reads straight from global memory and the last
// inverse pass writes straight to it.
//
// Arithmetic (q < 2^61): Shoup multiplication with an APPROXIMATE high product (three 32x32
// products instead of four; quotient under-estimated by <= 2), so a lazy product lies in [0,4q).
//   narrow moduli ((4*log2(n)+1)*q < 2^64, e.g. the 48..58-bit coefficient primes): the forward
//     transform needs NO conditional subtraction at all (values grow by < 4q per stage) and ends with
//     one Barrett reduction per coefficient;
//   wide moduli (the 61-bit BEHZ primes): values live in [0,2^64), one subtraction of 4q per butterfly, conditional
//   on the top bit alone (csub_top).
// The inverse is computed as a decimation-in-time CYCLIC inverse transform (bit-reversed input, natural
// output, twiddles psi^(-j*n/g)) followed by the twist n^-1 * psi^-j, so it uses the very same butterfly
// (x + w*y, x - w*y) and range discipline as the forward transform.
// All of this is invisible in the results: outputs are canonical residues.
// The pass functions are __host__ __device__ so tests can emulate a workgroup on the CPU.
#pragma once
#include <type_traits>

#include "modmath.h"

struct TwPair { u64 w, wq; };          // twiddle and its Shoup quotient, 16 B -> one dwordx4 load

// Per-modulus device tables.  fwd[k] = psi^brv(k)  (k = m + i).
struct NttTable {
    u64 q;
    u64 ninv, ninv_q;                  // n^-1 mod q and its Shoup quotient
    u64 r1;                            // floor(2^64 / q): single-word Barrett ratio
    const TwPair *fwd;
    // the inverse runs as a decimation-in-time cyclic transform with the psi^-j twist and n^-1 applied at
    // the end: same butterfly as the forward (no conditional subtractions for narrow moduli):
    const TwPair *dit;                 // dit[g + j] = psi^(-j*n/g), j < g, g = 1,2,4,..,n/2
    const TwPair *scale;               // scale[j] = n^-1 * psi^-j
    int narrow;                        // (4*logn+1)*q < 2^64
    // q = 2^k - c with a small c (every prime SEAL's search returns is of this shape: it scans downwards from 2^k in
    // steps of 2n).  Then any 64-bit x reduces with ONE narrow multiply: x = (x >> k) * c + (x mod 2^k)  (mod q), which is
    // below 2q when (2^(64-k) + 2) * c <= 2^k.  fold_k = 0: not available (small or unstructured moduli), Barrett is used.
    u32 fold_k, fold_c;
    u32 wide_d4;                       // 2^63 - 4q when that fits 32 bits (wide moduli next to 2^61), else 0
    u64 r0;                            // low word of floor(2^128 / q) (with r1: the two-word Barrett ratio)
};

HD bool ntt_is_narrow(u64 q, int logn) { return (unsigned __int128)q * (unsigned)(4 * logn + 1) < ((unsigned __int128)1 << 64); }

// May a forward transform modulo q take residues below max_src (of another modulus) WITHOUT reducing them on load?  A narrow
// modulus runs without any range control: a value grows by less than 4q per stage, so max_src + 4 logn q <= 2^64 keeps every
// intermediate inside 64 bits; the transform is linear and its closing reduction takes any 64-bit value.
HD bool ntt_gather_nored_ok(u64 q, u64 max_src, int logn)
{
    return ntt_is_narrow(q, logn) && (unsigned __int128)q * (unsigned)(4 * logn) + max_src <= ((unsigned __int128)1 << 64);
}

// fold parameters of q (k = 0 when the fold reduction does not apply)
HD void ntt_fold_params(u64 q, u32 &k, u32 &c)
{
    int bits = 0;
    while (bits < 64 && (q >> bits)) bits++;
    k = 0; c = 0;
    if (bits < 33 || bits > 62) return;
    const u64 cc = ((u64)1 << bits) - q;
    if (cc >> 32) return;
    if ((unsigned __int128)((((u64)1) << (64 - bits)) + 2) * cc > ((unsigned __int128)1 << bits)) return;
    k = (u32)bits; c = (u32)cc;
}

// 2^63 - 4q if it is a positive 32-bit number (see csub_top_near), else 0
HD u32 ntt_wide_d4(u64 q, bool narrow)
{
    if (narrow || (q >> 61)) return 0;
    const u64 q4 = q << 2;
    if (q4 >= ((u64)1 << 63)) return 0;
    const u64 d = ((u64)1 << 63) - q4;
    return (d >> 32) ? 0 : (u32)d;
}

// canonical residue of ANY 64-bit x
HD u64 ntt_reduce_any(u64 x, const NttTable &tab)
{
    if (tab.fold_k) {                                            // wave-uniform
        const u32 sh = tab.fold_k - 32;
        const u32 hi = (u32)(x >> 32);
        const u64 low = ((u64)(hi & ((1u << sh) - 1)) << 32) | (u32)x;
        const u64 v = (u64)(hi >> sh) * tab.fold_c + low;        // < 2q
        return v >= tab.q ? v - tab.q : v;
    }
    const u64 v = x - mulhi64(x, tab.r1) * tab.q;
    return v >= tab.q ? v - tab.q : v;
}

// Canonical residue of a 128-bit P = hi*2^64 + lo < 2^(2k+2) (a sum of up to FOUR products of residues) for q = 2^k - c with
// 44 <= k <= 61 and c < 2^24 (ntt_fold128_ok):
// P = Ph*2^k + Pl = Ph*c + Pl with Ph < 2^(k+2); Ph*c is taken in two 32-bit halves of Ph, the upper product T (< 2^(k-6)) is
// folded once more at its own bit k-32, and every partial sum stays below 2^64:  Pl + A + Th*c + (Tl << 32) < 2^61 + 2^56 + 2^50 + 2^61.
// Five narrow multiplies and no quotient word, against ~12 for the two-word Barrett step.
HD bool ntt_fold128_ok(u32 fold_k, u32 fold_c) { return fold_k >= 44 && fold_k <= 61 && fold_c < (1u << 24); }
HD u64 ntt_reduce128_fold(u64 hi, u64 lo, const NttTable &tab)
{
    const u32 k = tab.fold_k, c = tab.fold_c, s = k - 32;
    const u64 pl = lo & (((u64)1 << k) - 1);
    const u64 ph = (hi << (64 - k)) | (lo >> k);                 // < 2^(k+1)
    const u64 a = (u64)(u32)ph * c;
    const u64 t = (u64)(u32)(ph >> 32) * c;
    const u64 w = pl + a + (u64)(u32)(t >> s) * c + ((t & (((u64)1 << s) - 1)) << 32);
    return ntt_reduce_any(w, tab);
}

// canonical residue of a 128-bit sum of at most four products of canonical residues: the fold where the modulus admits it,
// the two-word Barrett step otherwise (wave-uniform choice)
HD u64 ntt_reduce128(u64 hi, u64 lo, const NttTable &tab)
{
    if (ntt_fold128_ok(tab.fold_k, tab.fold_c)) return ntt_reduce128_fold(hi, lo, tab);
    return barrett128(u128p{ lo, hi }, Mod{ tab.q, tab.r0, tab.r1 });
}
// The same value as the INPUT of an inverse transform, which does not need a canonical residue: the fold's last word w
// (< 2^(k+1) + 2^32 c + 4 c^2: below 4q for the 56-bit primes, up to ~8q for a 50-bit prime with a large c -- ntt_lazy_bound_q) enters as it is -- the closing ntt_reduce_any (a multiply and a conditional
// subtraction per coefficient) is left out -- when the transform's range discipline takes it: a wide modulus takes any 64-bit
// value (csub_top), a narrow one runs without range control and needs 4q + 4 logn q < 2^64 (round 4).
// Upper bound, in units of q, of ntt_reduce128_lazy's result: w = pl + a + t' c + (t'' << 32) < 2^(k+1) + 2^32 c + 4 c^2 (a = low word of
// ph times c, up to 2^32 c: for a 50-bit prime with c ~ 2^20 that alone is 4q, so "below 4q" holds for k >= 56 only -- the 256M-4096
// workload caught this in round 6).  With q > 2^(k-1): w / q < 4 + ((c 2^32 + 4 c^2) >> (k - 1)) + 1.  Shifts only: evaluated per workgroup.
HD u64 ntt_lazy_bound_q(const NttTable &tab)
{
    const u64 c = tab.fold_c;
    return 5 + ((((c << 32) + 4 * c * c)) >> (tab.fold_k - 1));
}
// (round 6: a narrow modulus' first inverse pass runs its psi^0 butterflies without a product, which doubles the bound per stage there:
//  2^K b0 q after the K stages of that pass + 4q for each of the others; K = the larger of the two forms' first inverse passes)
constexpr int plan_passes(int logn, int c);
constexpr int plan_k(int logn, int p, int c);
HD bool ntt_lazy_input_ok(const NttTable &tab, int logn)
{
    if (!ntt_fold128_ok(tab.fold_k, tab.fold_c)) return false;
    if (!tab.narrow) return true;
    const int k16 = plan_k(logn, plan_passes(logn, 16) - 1, 16);
    const u64 mult = (ntt_lazy_bound_q(tab) << k16) + (u64)(4 * (logn - k16));
    return (unsigned __int128)tab.q * mult < ((unsigned __int128)1 << 64);
}
HD u64 ntt_reduce128_lazy(u64 hi, u64 lo, const NttTable &tab)
{
    const u32 k = tab.fold_k, c = tab.fold_c, s = k - 32;
    const u64 pl = lo & (((u64)1 << k) - 1);
    const u64 ph = (hi << (64 - k)) | (lo >> k);
    const u64 a = (u64)(u32)ph * c;
    const u64 t = (u64)(u32)(ph >> 32) * c;
    return pl + a + (u64)(u32)(t >> s) * c + ((t & (((u64)1 << s) - 1)) << 32);
}

// Where a pass that reads global memory takes its coefficients from.
struct SrcPlain {};                                               // the limb itself
// The limb itself, STAGED (round 5): the inverse transform's first pass would read 16 contiguous coefficients per lane straight from
// global memory -- 128 bytes per lane, 64 different lines per load instruction, every line touched by eight consecutive instructions of
// the wave (512 vector-L1 look-ups per wave instead of 64; with 16 waves per CU the 8 KiB a wave keeps live do not fit the L1).  With
// this source type the workgroup body takes the path of the tensor-on-load transform: coalesced 16-byte loads into the LDS image, the
// first pass from there.  Same bits; -3.5 ... -6 % on launches of 6 840 limbs and more, level below 700 (tools/microbench/ntt_variants.hip,
// profiles/r05_ntt_staged_inverse.txt).
struct SrcStaged {};
// The dyadic tensor product of two NTT-form ciphertexts, computed on load in front of the inverse transform
// (BEHZ step 4, d0 = a0*b0, d1 = a0*b1 + a1*b0, d2 = a1*b1): value(e) = x0[e]*y0[e] (+ x1[e]*y1[e]) mod q.
struct SrcTensor { const u64 *x0, *y0, *x1, *y1; bool lazy; };   // x1 == nullptr: one product; lazy: ntt_lazy_input_ok for this limb
HD u64x2 src_load2(const SrcPlain &, const u64 *glob, int e, const NttTable &) { return *reinterpret_cast<const u64x2 *>(glob + e); }
HD u64 src_load1(const SrcPlain &, const u64 *glob, int e, const NttTable &) { return glob[e]; }
HD u64x2 src_load2(const SrcStaged &, const u64 *glob, int e, const NttTable &) { return *reinterpret_cast<const u64x2 *>(glob + e); }
HD u64 src_load1(const SrcStaged &, const u64 *glob, int e, const NttTable &) { return glob[e]; }
HD u64x2 src_load2(const SrcTensor &s, const u64 *, int e, const NttTable &tab)
{
    const u64x2 x = ldg16(s.x0 + e), y = ldg16(s.y0 + e);
    u128p p0 = mul128(x[0], y[0]), p1 = mul128(x[1], y[1]);
    if (s.x1) {                                                   // wave-uniform
        const u64x2 u = ldg16(s.x1 + e), v = ldg16(s.y1 + e);
        mac128(p0, u[0], v[0]);
        mac128(p1, u[1], v[1]);
    }
    u64x2 r;
    if (s.lazy) {                                                 // wave-uniform
        r[0] = ntt_reduce128_lazy(p0.hi, p0.lo, tab);
        r[1] = ntt_reduce128_lazy(p1.hi, p1.lo, tab);
    } else {
        r[0] = ntt_reduce128(p0.hi, p0.lo, tab);
        r[1] = ntt_reduce128(p1.hi, p1.lo, tab);
    }
    return r;
}
HD u64 src_load1(const SrcTensor &s, const u64 *, int e, const NttTable &tab)
{
    u128p p = mul128(s.x0[e], s.y0[e]);
    if (s.x1) mac128(p, s.x1[e], s.y1[e]);
    return s.lazy ? ntt_reduce128_lazy(p.hi, p.lo, tab) : ntt_reduce128(p.hi, p.lo, tab);
}
// Upper bound, in units of q, of what a source hands to the inverse transform's first pass: canonical residues (1) or the tensor fold's
// last word (ntt_lazy_bound_q) -- the constant the multiplication-free butterflies of that pass add (ntt_pass16, round 6).
HD u64 src_in_bound(const SrcPlain &, const NttTable &) { return 1; }
HD u64 src_in_bound(const SrcStaged &, const NttTable &) { return 1; }
HD u64 src_in_bound(const SrcTensor &s, const NttTable &tab) { return s.lazy ? ntt_lazy_bound_q(tab) : 1; }
// LDS padding: 16 bytes per 16 coefficients.  Keeps coefficient pairs 16-B aligned (ds_*_b128) and
// makes the 128-B-per-lane stride of the contiguous pass conflict free (lane stride 144 B = 36 banks).
HD int lds_slot(int e) { return e + ((e >> 4) << 1); }
constexpr int lds_slots(int n) { return n + (n >> 3); }

#if defined(__HIP_DEVICE_COMPILE__)
#define UNIFORM_INT(x) __builtin_amdgcn_readfirstlane(x)
#else
#define UNIFORM_INT(x) (x)
#endif

// x * w mod q, lazy in [0,4q), for any 64-bit x.  nq = 2^64 - q.
HD u64 mul_lazy4(u64 x, u64 w, u64 wq, u64 nq)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), a0 = (u32)wq, a1 = (u32)(wq >> 32);
    const u64 t1 = (u64)x1 * a0, t2 = (u64)x0 * a1;
    const u64 h = (u64)x1 * a1 + (t1 >> 32) + (t2 >> 32);          // floor(x*wq / 2^64) - {0,1,2}
    const u32 h0 = (u32)h, h1 = (u32)(h >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    const u64 lo = (u64)x0 * w0 + (u64)h0 * n0;
    const u64 mid = (u64)x0 * w1 + (u64)x1 * w0 + (u64)h0 * n1 + (u64)h1 * n0;
    return lo + (mid << 32);                                       // x*w - h*q  (mod 2^64)
}

// Lazy butterfly (x, y) -> (x + v, x - v + 4q) with v = y*w mod q in [0,4q).  The sum rides on the multiply-add
// chain of the low product (its 64-bit addend is free), the difference is (2x + 4q) - (x + v): three 64-bit
// add-class instructions fewer per butterfly pair than add / sub / add (64-bit adds cost as much as a multiply here).
// Round 4 (second half): the four cross products y0 w1 + y1 w0 + h0 n1 + h1 n0 (mod 2^32) as ONE chain of four v_mad_u64_u32
// whose 64-bit addend starts as the high word of the low product -- so the chain ends with that word already summed -- instead
// of four v_mul_lo_u32, two adds and the 64-bit shift-add that joined them: 16.1 instead of 17.8 VALU instructions per butterfly,
// the register-only butterfly loop 10 % faster (tools/microbench/bfly.hip, profiles/r04_bfly_mad_chain.txt).  The compiler
// narrows the same chain written in C back to 32-bit multiplies, hence the instruction by name.  UNI: the twiddle is
// wave-uniform (scalar registers; one scalar source per instruction is what gfx9 allows); nq always is.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(APSU_NTT_NO_MAD_CHAIN)
template <bool SB> __device__ __forceinline__ u64 ntt_mad64(u32 a, u32 b, u64 c)
{
    u64 d;
    unsigned long long carry;
    if constexpr (SB) asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "s"(b), "v"(c));
    else asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return d;
}
#define NTT_MAD_CHAIN 1
#endif
// where the chain pays (measured per direction, profiles/r04_ntt_mad_chain.txt); APSU_NTT_MAD_CHAIN_MODE: 1 forward passes
// (default), 2 forward passes with wave-uniform twiddles only, 3 every pass of both directions, 4 forward as 1 + inverse with the
// chain started from zero, 5 both directions from zero, 6 forward as 1 + the inverse's contiguous pass (constant twiddle indices) only,
// 7 both directions with the chain in C behind opaque sums, 8 forward as 1 + inverse as 7
#ifndef APSU_NTT_MAD_CHAIN_MODE
#define APSU_NTT_MAD_CHAIN_MODE 1
#endif
#if defined(NTT_MAD_CHAIN)
template <bool SB> __device__ __forceinline__ u64 ntt_mul64(u32 a, u32 b)         // a * b as the head of a chain (addend 0)
{
    u64 d;
    unsigned long long carry;
    if constexpr (SB) asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "s"(b));
    else asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
    return d;
}
#endif
// CHAIN: 0 the compiler's form; 1 chain on top of the low product's high word; 2 chain from zero, joined by one 64-bit shift-add
template <bool UNI = false, int CHAIN = 0>
HD void bfly_lazy4(u64 &x, u64 &y, u64 w, u64 wq, u64 nq, u64 q4)
{
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), a0 = (u32)wq, a1 = (u32)(wq >> 32);
    const u64 t1 = (u64)y1 * a0, t2 = (u64)y0 * a1;
    const u64 h = (u64)y1 * a1 + (t1 >> 32) + (t2 >> 32);          // floor(y*wq / 2^64) - {0,1,2}
    const u32 h0 = (u32)h, h1 = (u32)(h >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    const u64 lo = (u64)h0 * n0 + ((u64)y0 * w0 + x);
    u64 s;
#if defined(NTT_MAD_CHAIN)
    if constexpr (CHAIN == 2) {
        u64 acc = ntt_mul64<UNI>(y0, w1);
        acc = ntt_mad64<UNI>(y1, w0, acc);
        acc = ntt_mad64<true>(h0, n1, acc);
        acc = ntt_mad64<true>(h1, n0, acc);
        s = lo + (acc << 32);
    } else if constexpr (CHAIN == 3) {                           // the chain in C; an empty statement keeps every sum whole (64 bits), so the
        u64 acc = lo >> 32;                                        // compiler takes v_mad_u64_u32 but picks registers and operand kinds itself
        acc = (u64)y0 * w1 + acc; asm("" : "+v"(acc));
        acc = (u64)y1 * w0 + acc; asm("" : "+v"(acc));
        acc = (u64)h0 * n1 + acc; asm("" : "+v"(acc));
        acc = (u64)h1 * n0 + acc; asm("" : "+v"(acc));
        s = (u64)(u32)lo | (acc << 32);
    } else if constexpr (CHAIN == 1) {
        u64 acc = lo >> 32;
        acc = ntt_mad64<UNI>(y0, w1, acc);
        acc = ntt_mad64<UNI>(y1, w0, acc);
        acc = ntt_mad64<true>(h0, n1, acc);
        acc = ntt_mad64<true>(h1, n0, acc);
        s = (u64)(u32)lo | (acc << 32);                            // x + v  (mod 2^64)
    } else
#endif
    {
        const u32 mid = y0 * w1 + y1 * w0 + h0 * n1 + h1 * n0;
        s = lo + ((u64)mid << 32);                                 // x + v  (mod 2^64)
    }
    y = ((x << 1) + q4) - s;                                       // x - v + 4q
    x = s;
}

// Range control for wide moduli (q < 2^61, 4q < 2^63): subtract 4q exactly when the top bit is set.  The result is
// below max(2^63, 2^64 - 4q), so a following x + v (v < 4q) and x - v + 4q still fit 64 bits; no comparison with 4q
// is needed.  n4 = 2^64 - 4q.
HD u64 csub_top(u64 x, u64 n4) { return x + ((u64)((int64_t)x >> 63) & n4); }
// The same when 2^63 - 4q fits 32 bits (every 61-bit prime SEAL's search returns: q = 2^61 - c, 4c < 2^24):
// x - 4q = (x with bit 63 cleared) + (2^63 - 4q), one shift, one mask and one multiply-add instead of a 64-bit masked add.
HD u64 csub_top_near(u64 x, u32 d4)
{
    const u32 hi = (u32)(x >> 32);
    const u64 low = ((u64)(hi & 0x7fffffffu) << 32) | (u32)x;
    return (u64)(hi >> 31) * d4 + low;
}

enum PassIo { IO_LDS = 0, IO_GLOBAL = 1 };
// narrow: no range control at all; wide: subtract 4q when the top bit is set; wide-near: the same for q next to 2^61
enum NttMode { NTT_NARROW = 0, NTT_WIDE = 1, NTT_WIDE_NEAR = 2 };
HD int ntt_mode(const NttTable &tab) { return tab.narrow ? NTT_NARROW : (tab.wide_d4 ? NTT_WIDE_NEAR : NTT_WIDE); }

// ---- the twiddles of one pass for one work item, as a register array -------------------------------------------------
// Forward (Cooley-Tukey, twiddle per block): stage u of the pass uses 2^u distinct table entries per group of columns
// (column groups share them; groups that are adjacent BLOCKS have their own).  Inverse (decimation in time, twiddle per
// position inside the block): stage with row gap `bit` uses `bit` row positions, times the G columns when the groups are
// columns.  Either way (2^K - 1) * GM entries; slot numbering below.  A pass either loads them where it uses them (TwInline)
// or takes a set that was loaded EARLIER -- by the pass in front of it, just before that pass stored its results, so that
// the table loads (L2 latency) overlap the LDS turnaround instead of following it (round 4; ntt_wg.h).
template <int LOGN, int S, int K, bool INV, int C = 16> struct PassShape {
    static constexpr int R = 1 << K, G = C >> K, LOWBITS = LOGN - S - K;
    static constexpr bool COLS = (1 << LOWBITS) >= G;
    static constexpr int CG = COLS ? ((1 << LOWBITS) / G) : 1;
    static constexpr int GM = INV ? (COLS ? G : 1) : (COLS ? 1 : G);
    static constexpr int NT = (R - 1) * GM;
    // forward: (stage u, row j, group g) -> slot;  inverse: (row gap bit, row j, group gg) -> slot
    static constexpr int fwd_slot(int u, int j, int g) { return ((1 << u) - 1) * GM + (j >> (K - u)) * GM + (COLS ? 0 : g); }
    static constexpr int inv_slot(int bit, int j, int gg) { return (bit - 1) * GM + (j & (bit - 1)) * GM + (COLS ? gg : 0); }
};
struct TwInline {};                                               // marker: the pass loads its twiddles itself
template <int NT> struct TwRegs { u64x2 t[NT > 0 ? NT : 1]; };
struct NoHook { HD void operator()() const {} };

// table entries of pass (S, K) for work item w into tw (same index algebra as ntt_pass16 below)
template <int LOGN, int S, int K, bool INV, int C = 16>
HD void ntt_load_twiddles(TwRegs<PassShape<LOGN, S, K, INV, C>::NT> &tw, int w, const NttTable &tab)
{
    using PS = PassShape<LOGN, S, K, INV, C>;
    constexpr int R = PS::R, G = PS::G, LOWBITS = PS::LOWBITS, CG = PS::CG;
    constexpr bool COLS = PS::COLS;
    constexpr bool UNIFORM_TW = COLS && (CG % 64 == 0);
    int block, c0;
    if (COLS) { block = w / CG; c0 = (w % CG) * G; }
    else { block = w * G; c0 = 0; }
#pragma unroll
    for (int uu = 0; uu < K; uu++) {
        const int u = INV ? K - 1 - uu : uu;
        const int bit = 1 << (K - 1 - u);
        if (INV) {
#pragma unroll
            for (int jl = 0; jl < bit; jl++)
#pragma unroll
                for (int gg = 0; gg < PS::GM; gg++) {
                    const int jj = COLS ? ((jl << LOWBITS) | (c0 + gg)) : jl;
                    const int gap = COLS ? (bit << LOWBITS) : bit;
                    tw.t[PS::inv_slot(bit, jl, gg)] = ldg16(reinterpret_cast<const u64 *>(tab.dit + gap + jj));
                }
        } else {
#pragma unroll
            for (int jh = 0; jh < (1 << u); jh++)
#pragma unroll
                for (int g = 0; g < PS::GM; g++) {
                    int ti = (1 << (S + u)) + ((COLS ? block : block + g) << u) + jh;
                    if (UNIFORM_TW) ti = UNIFORM_INT(ti);
                    tw.t[PS::fwd_slot(u, jh << (K - u), g)] = ldg16(reinterpret_cast<const u64 *>(tab.fwd + ti));
                }
        }
    }
    (void)R;
}

// One pass over stages S .. S+K-1 for work item w in [0, n/C): C coefficients per work item (16: the throughput form, one limb per
// 512-thread workgroup at n = 8192; 8: the latency form of round 6, twice the waves per limb -- see plan_k below).
//   forward: Cooley-Tukey, stages ascending; inverse: decimation-in-time cyclic inverse, stages descending (gap 1 first).
//   IN / OUT: where the 16 coefficients come from / go to (LDS image or the limb in global memory).
// MODE: range discipline of the modulus (wave-uniform per limb): see NTT_NARROW / NTT_WIDE / NTT_WIDE_NEAR
//   TW: TwInline, or the pass's twiddles loaded earlier (TwRegs);  HOOK: called once between the butterflies and the stores
//   (ntt_wg.h uses it to issue the NEXT pass's twiddle loads).
template <int LOGN, int S, int K, bool INV, int MODE, int IN, int OUT, int RED = 0, bool RAW = false, class SRC = SrcPlain,
          class TW = TwInline, class HOOK = NoHook, int C = 16>
HD void ntt_pass16(u64 *lds, u64 *__restrict__ glob, int w, const NttTable &tab, const SRC &src = SRC(), const TW &tw = TW(),
                   const HOOK &hook = HOOK())
{
    using PS = PassShape<LOGN, S, K, INV, C>;
    static_assert((C == 16 || C == 8 || C == 4) && (1 << K) <= C, "radix 2^K on C coefficients per work item");
    constexpr bool PRE = !std::is_same<TW, TwInline>::value;
    constexpr int R = 1 << K;                  // radix
    constexpr int G = C >> K;                  // independent groups per thread
    constexpr int LOWBITS = LOGN - S - K;      // bits of the column index
    constexpr bool COLS = (1 << LOWBITS) >= G; // groups = adjacent columns (else adjacent blocks)
    constexpr int CG = COLS ? ((1 << LOWBITS) / G) : 1;      // column groups per block
    constexpr bool UNIFORM_TW = COLS && (CG % 64 == 0);      // every lane of a wave shares the twiddles
    const TwPair *__restrict__ W = tab.fwd;
    const u64 q = tab.q, nq = (u64)0 - q, q4 = q << 2, n4 = (u64)0 - q4;
    const u32 d4 = tab.wide_d4;                // wave-uniform
    // wave-uniform: b0 q of the product-free butterflies below (narrow moduli: every psi^0 butterfly of the first inverse pass; wide moduli:
    // only stage 1, and only behind the tensor loader, whose output range is known)
    constexpr bool TRIV_WIDE1 = INV && MODE != NTT_NARROW && LOWBITS == 0 && std::is_same<SRC, SrcTensor>::value;
    const u64 qb0 = ((INV && MODE == NTT_NARROW && LOWBITS == 0) || TRIV_WIDE1) ? q * src_in_bound(src, tab) : 0;

    int block, c0;
    if (COLS) { block = w / CG; c0 = (w % CG) * G; }
    else { block = w * G; c0 = 0; }
    // element index of group g, row j
    auto idx = [&](int g, int j) -> int {
        return COLS ? ((block << (LOGN - S)) | (j << LOWBITS) | (c0 + g)) : (((block + g) << K) | j);
    };

    // access modes: PAIR_G = adjacent columns (g, g+1) form a 16-byte pair; PAIR_J = the 16 coefficients
    // are contiguous so rows (j, j+1) pair up; otherwise single 8-byte accesses.
    constexpr bool PAIR_G = COLS && G >= 2;
    constexpr bool PAIR_J = !PAIR_G && LOWBITS == 0;
    u64 r[G][R];
    if (PAIR_G) {
#pragma unroll
        for (int j = 0; j < R; j++)
#pragma unroll
            for (int g = 0; g < G; g += 2) {
                const int e = idx(g, j);
                const u64x2 v = (IN == IO_GLOBAL) ? src_load2(src, glob, e, tab)
                                                   : *reinterpret_cast<const u64x2 *>(lds + lds_slot(e));
                r[g][j] = v[0]; r[g + 1][j] = v[1];
            }
    } else if (PAIR_J) {
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int j = 0; j < R; j += 2) {
                const int e = idx(g, j);
                const u64x2 v = (IN == IO_GLOBAL) ? src_load2(src, glob, e, tab)
                                                   : *reinterpret_cast<const u64x2 *>(lds + lds_slot(e));
                r[g][j] = v[0]; r[g][j + 1] = v[1];
            }
    } else {
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int j = 0; j < R; j++) {
                const int e = idx(g, j);
                r[g][j] = (IN == IO_GLOBAL) ? src_load1(src, glob, e, tab) : lds[lds_slot(e)];
            }
    }

    if (RED == 1 && IN == IO_GLOBAL) {         // gathered input holds residues of ANOTHER modulus: reduce on load (RED == 2: the host
                                               // has checked that they fit the lazy range as they are, ntt_wg.h)
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int j = 0; j < R; j++) r[g][j] = ntt_reduce_any(r[g][j], tab);
    }

#pragma unroll
    for (int uu = 0; uu < K; uu++) {
        const int u = INV ? K - 1 - uu : uu;
        const int bit = 1 << (K - 1 - u);
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (j & bit) continue;
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (COLS && g > 0) continue;   // column groups share the twiddle: handled below
                if (INV) {
                    // decimation-in-time butterfly of the cyclic inverse: twiddle depends on the position inside
                    // the block (row low bits and column), not on the block
#pragma unroll
                    for (int gg = 0; gg < G; gg++) {
                        if (!COLS && gg != g) continue;
                        const int jj = COLS ? (((j & (bit - 1)) << LOWBITS) | (c0 + gg)) : (j & (bit - 1));
                        const int gap = COLS ? (bit << LOWBITS) : bit;
                        u64x2 tv;
                        if constexpr (PRE) tv = tw.t[PS::inv_slot(bit, j, gg)];
                        else tv = ldg16(reinterpret_cast<const u64 *>(tab.dit + gap + jj));
                        u64 &x = r[gg][j], &y = r[gg][j | bit];
                        // Round 6: the pass executed first (LOWBITS == 0: 2^K contiguous coefficients per group) meets the twiddle psi^0 = 1
                        // wherever the row's low bits are zero -- all of stage 1, half of stage 2, a quarter of stage 3 ... (15 of the 32
                        // butterflies of a radix-16 pass): (x, y) -> (x + y, x - y + M) needs no product.  M is the multiple of q that bounds
                        // y there: both operands are sums of `bit` inputs, each below b0 q (b0 = 1 for canonical input,
                        // ntt_lazy_bound_q for the tensor fold's last word).  Narrow moduli only (no range control to keep: the values stay below (2^K b0 + 4 (log n - K)) q,
                        // inside the narrow criterion for b0 = 1 and checked by ntt_lazy_input_ok for b0 = 4); -3 ... -7 % on inverse
                        // launches over the data primes (tools/microbench/ntt_forms.hip, profiles/r06_ntt_trivial_twiddles.txt).
                        // Wide moduli (61-bit BEHZ primes, values anywhere in 64 bits): only stage 1 behind the tensor loader -- both operands are
                        // its outputs, below b0 q <= 5q < 2^63.4 (canonical: q), so x + y < 2^63.1 and x - y + b0 q < 2^64 fit, and stage 2 is a
                        // regular butterfly again, whose csub_top takes any 64-bit value.
                        if (((MODE == NTT_NARROW && LOWBITS == 0) || (TRIV_WIDE1 && bit == 1)) && (j & (bit - 1)) == 0) {
                            const u64 M = qb0 * (u64)bit, sum = x + y;                    // (bit is a power of two and a constant here: a shift)
                            y = x + M - y;
                            x = sum;
                            continue;
                        }
                        if (MODE == NTT_WIDE) x = csub_top(x, n4);
                    if (MODE == NTT_WIDE_NEAR) x = csub_top_near(x, d4);
                        bfly_lazy4<(!COLS && !PRE), (APSU_NTT_MAD_CHAIN_MODE == 3 ? 1 : (APSU_NTT_MAD_CHAIN_MODE == 4 || APSU_NTT_MAD_CHAIN_MODE == 5) ? 2 : (APSU_NTT_MAD_CHAIN_MODE == 6 && !COLS && !PRE) ? 1 : (APSU_NTT_MAD_CHAIN_MODE == 7 || APSU_NTT_MAD_CHAIN_MODE == 8) ? 3 : 0)>(x, y, tv[0], tv[1], nq, q4);   // !COLS: the twiddle index is a compile-time constant
                    }
                    continue;
                }
                int ti = (1 << (S + u)) + (((COLS ? block : block + g)) << u) + (j >> (K - u));
                if (UNIFORM_TW) ti = UNIFORM_INT(ti);
                u64x2 tv;
                if constexpr (PRE) tv = tw.t[PS::fwd_slot(u, j, g)];
                else tv = ldg16(reinterpret_cast<const u64 *>(W + ti));
                const TwPair t{ tv[0], tv[1] };
#pragma unroll
                for (int gg = 0; gg < G; gg++) {
                    if (!COLS && gg != g) continue;
                    u64 &x = r[gg][j], &y = r[gg][j | bit];
                    if (MODE == NTT_WIDE) x = csub_top(x, n4);
                    if (MODE == NTT_WIDE_NEAR) x = csub_top_near(x, d4);
                    bfly_lazy4<UNIFORM_TW && !PRE, (APSU_NTT_MAD_CHAIN_MODE == 2 ? ((UNIFORM_TW && !PRE) ? 1 : 0) : APSU_NTT_MAD_CHAIN_MODE == 5 ? 2 : APSU_NTT_MAD_CHAIN_MODE == 7 ? 3 : 1)>(x, y, t.w, t.wq, nq, q4);
                }
            }
        }
    }

    hook();
    if (OUT == IO_GLOBAL && !(INV && RAW)) {   // leaving the transform: canonical residues (RAW: the consumer applies the twist)
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int j = 0; j < R; j++) {
                u64 v = r[g][j];
                if (INV) {
                    const u64x2 sv = ldg16(reinterpret_cast<const u64 *>(tab.scale + idx(g, j)));
                    v = mul_shoup(v, sv[0], sv[1], q);      // (a 2^k - c fold product here measured 7 % slower)
                } else v = ntt_reduce_any(v, tab);
                r[g][j] = v;
            }
    }
    if (PAIR_G) {
#pragma unroll
        for (int j = 0; j < R; j++)
#pragma unroll
            for (int g = 0; g < G; g += 2) {
                const int e = idx(g, j);
                u64x2 v; v[0] = r[g][j]; v[1] = r[g + 1][j];
                if (OUT == IO_GLOBAL) *reinterpret_cast<u64x2 *>(glob + e) = v;
                else *reinterpret_cast<u64x2 *>(lds + lds_slot(e)) = v;
            }
    } else if (PAIR_J) {
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int j = 0; j < R; j += 2) {
                const int e = idx(g, j);
                u64x2 v; v[0] = r[g][j]; v[1] = r[g][j + 1];
                if (OUT == IO_GLOBAL) *reinterpret_cast<u64x2 *>(glob + e) = v;
                else *reinterpret_cast<u64x2 *>(lds + lds_slot(e)) = v;
            }
    } else {
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int j = 0; j < R; j++) {
                const int e = idx(g, j);
                if (OUT == IO_GLOBAL) glob[e] = r[g][j];
                else lds[lds_slot(e)] = r[g][j];
            }
    }
}

// Final range fix-up when the last pass leaves its result in LDS (forward transform).
template <int MODE>
HD u64 ntt_fwd_finish(u64 v, const NttTable &tab)
{
    return ntt_reduce_any(v, tab);                                // any 64-bit v (wide moduli leave values up to 2^64 - 1)
}

// ---- pass schedule (stage counts per pass, summing to LOGN), resolved at compile time ----
// c = coefficients per work item.  c = 16: the throughput form (rounds 1-5).  c = 8 (round 6): the LATENCY form for launches that
// cannot fill the chip -- a limb's workgroup has twice the waves (n = 8192: 1 024 threads = 4 waves per SIMD of its CU instead of 2), so a
// lone limb's LDS turnarounds, twiddle loads and global round trips are covered by three other waves per SIMD instead of one; the
// price is one more pass (radix <= 8) and one more workgroup barrier.  Same twiddle tables, same lazy ranges, same bits.
//   n = 8192: 2,3,3,3,2 -- the first pass (forward: reads global memory; inverse: writes it) keeps column PAIRS, i.e. 16-byte
//             coalesced global accesses;  n = 4096: 3,3,3,3 (one barrier, 8-byte global accesses).
constexpr int plan_passes(int logn, int c = 16)
{
    if (c == 8) return logn == 13 ? 5 : logn == 12 ? 4 : 0;
    return logn == 14 ? 4 : logn == 13 ? 4 : logn == 12 ? 4 : logn == 11 ? 4 : logn == 10 ? 3 : logn == 8 ? 3 : logn == 6 ? 2 : 0;
}
constexpr int plan_k(int logn, int p, int c = 16)
{
    if (c == 8) {
        switch (logn) {
        case 13: return (p == 0 || p == 4) ? 2 : 3;   // 2,3,3,3,2
        case 12: return 3;                            // 3,3,3,3
        default: return 0;
        }
    }
    switch (logn) {
    case 14: return p >= 2 ? 4 : 3;               // 3,3,4,4   (n = 16384: 144 KiB of LDS, one 1024-thread workgroup per CU)
    case 13: return p == 3 ? 4 : 3;               // 3,3,3,4
    case 12: return 3;                            // 3,3,3,3
    case 11: return p == 3 ? 2 : 3;               // 3,3,3,2
    case 10: return p == 2 ? 4 : 3;               // 3,3,4   (parity tests only)
    case 8:  return p == 2 ? 2 : 3;               // 3,3,2   (parity tests only)
    case 6:  return 3;                            // 3,3     (parity tests only)
    default: return 0;
    }
}
constexpr int plan_s(int logn, int p, int c = 16)
{
    int s = 0;
    for (int i = 0; i < p; i++) s += plan_k(logn, i, c);
    return s;
}
constexpr bool plan_has_latency_form(int logn) { return plan_passes(logn, 8) > 0; }

// shape of the pass executed PASS-th (for the twiddle sets above)
template <int LOGN, bool INV, int PASS, int C = 16> struct ExecPass {
    static constexpr int P = plan_passes(LOGN, C), p = INV ? P - 1 - PASS : PASS, K = plan_k(LOGN, p, C), S = plan_s(LOGN, p, C);
    using Shape = PassShape<LOGN, S, K, INV, C>;
    using Tw = TwRegs<Shape::NT>;
};
template <int LOGN, bool INV, int PASS, int C = 16>
HD void ntt_pass_twiddles(typename ExecPass<LOGN, INV, PASS, C>::Tw &tw, int tid, const NttTable &tab)
{
    using E = ExecPass<LOGN, INV, PASS, C>;
    ntt_load_twiddles<LOGN, E::S, E::K, INV, C>(tw, tid, tab);
}

// Executes pass number PASS (in execution order) for "thread" tid of a T-thread workgroup.
// The caller separates passes with __syncthreads() (device) or by looping tid (host emulation).
//   forward: pass 0 reads the limb from global memory, the last pass leaves data in LDS (the caller
//            then stores it coalesced);   inverse: pass 0 reads global memory too (C contiguous coefficients per lane),
//            the last pass writes the scaled result straight to global memory.
// STAGED: the caller has already put the input into the LDS image (k_intt_tensor forms its products with coalesced loads).
template <int LOGN, bool INV, int MODE, int PASS, int RED = 0, bool RAW = false, class SRC = SrcPlain, bool STAGED = false,
          class TW = TwInline, class HOOK = NoHook, int C = 16>
HD void ntt_pass(u64 *lds, u64 *glob, int tid, int T, const NttTable &tab, const SRC &src = SRC(), const TW &tw = TW(),
                 const HOOK &hook = HOOK())
{
    constexpr int P = plan_passes(LOGN, C);
    constexpr int p = INV ? P - 1 - PASS : PASS;      // the inverse walks the passes last-to-first
    constexpr int K = plan_k(LOGN, p, C);
    constexpr int S = plan_s(LOGN, p, C);
    constexpr int IN = (PASS == 0 && !STAGED) ? IO_GLOBAL : IO_LDS;   // both directions read the limb straight from global memory
    constexpr int OUT = (INV && PASS == P - 1) ? IO_GLOBAL : IO_LDS;
    constexpr int LOGC = C == 16 ? 4 : C == 8 ? 3 : 2;
    for (int w = tid; w < (1 << (LOGN - LOGC)); w += T)
        ntt_pass16<LOGN, S, K, INV, MODE, IN, OUT, RED, RAW, SRC, TW, HOOK, C>(lds, glob, w, tab, src, tw, hook);
}
