// 64-bit modular arithmetic primitives for gfx950 (and a host path used by the CPU-side
// emulation tests).  All functions are exact integer arithmetic; "lazy" variants state their
// output range.  No SEAL code: the reduction identities are the textbook Barrett / Shoup ones
// (SURVEY.md App. B2), which any correct implementation must agree with bit-for-bit because
// every public result is the canonical residue in [0, q).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

typedef uint64_t u64;
typedef uint32_t u32;

// 16-byte pair of coefficients for 128-bit global / LDS accesses.  may_alias: the pair is always a
// view of two adjacent u64 array elements (no strict-aliasing assumptions for the compiler to exploit).
#if defined(__clang__)
typedef u64 u64x2 __attribute__((ext_vector_type(2), __may_alias__));
#else
typedef u64 u64x2 __attribute__((vector_size(16), __may_alias__));
#endif

// 16-byte loads that are known to hit global memory (pointers read from job descriptors lose their
// address space and would otherwise compile to flat_load).  _nt = non-temporal (streamed-once data).
HD u64x2 ldg16(const u64 *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const u64x2 __attribute__((address_space(1))) *)(const void *)p;
#else
    return *reinterpret_cast<const u64x2 *>(p);
#endif
}
HD u64x2 ldg16_nt(const u64 *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load((const u64x2 __attribute__((address_space(1))) *)(const void *)p);
#else
    return *reinterpret_cast<const u64x2 *>(p);
#endif
}

// 16 bytes from a 4-byte aligned address (bit-packed database rows): one global_load_dwordx4
#if defined(__clang__)
typedef u32 u32x4a4 __attribute__((ext_vector_type(4), __may_alias__, aligned(4)));
#else
typedef u32 u32x4a4 __attribute__((vector_size(16), __may_alias__, aligned(4)));
#endif
HD u32x4a4 ldg16_a4_nt(const u32 *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load((const u32x4a4 __attribute__((address_space(1))) *)(const void *)p);
#else
    return *reinterpret_cast<const u32x4a4 *>(p);
#endif
}

struct u128p { u64 lo, hi; };   // 128-bit value as a pair

HD u64 mulhi64(u64 a, u64 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

HD u128p mul128(u64 a, u64 b) { return { a * b, mulhi64(a, b) }; }

HD void add128(u128p &acc, u128p x)
{
    u64 lo = acc.lo + x.lo;
    acc.hi += x.hi + (lo < acc.lo);
    acc.lo = lo;
}

HD void mac128(u128p &acc, u64 a, u64 b) { add128(acc, mul128(a, b)); }

// Modulus with the two-word Barrett ratio floor(2^128 / q); valid for any x < 2^128, q < 2^63.
struct Mod {
    u64 q;
    u64 r0, r1;
};

// x mod q for a full 128-bit x.  Quotient estimate floor(x * ratio / 2^128) is off by at most 1.
HD u64 barrett128(u128p x, const Mod &m)
{
    u64 carry = mulhi64(x.lo, m.r0);
    u128p t2 = mul128(x.lo, m.r1);
    u64 tmp1 = t2.lo + carry;
    u64 tmp3 = t2.hi + (tmp1 < carry);
    t2 = mul128(x.hi, m.r0);
    u64 s = tmp1 + t2.lo;
    carry = t2.hi + (s < tmp1);
    u64 qhat = x.hi * m.r1 + tmp3 + carry;
    u64 r = x.lo - qhat * m.q;
    return r >= m.q ? r - m.q : r;
}

HD u64 barrett64(u64 x, const Mod &m)
{
    u64 r = x - mulhi64(x, m.r1) * m.q;
    return r >= m.q ? r - m.q : r;
}

HD u64 mulmod(u64 a, u64 b, const Mod &m) { return barrett128(mul128(a, b), m); }

HD u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
HD u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

// Shoup multiplication by a fixed operand w with wq = floor(w * 2^64 / q):
// returns x*w mod q in [0, 2q) for ANY 64-bit x (q < 2^63).
HD u64 mul_shoup_lazy(u64 x, u64 w, u64 wq, u64 q) { return x * w - mulhi64(x, wq) * q; }

HD u64 mul_shoup(u64 x, u64 w, u64 wq, u64 q)
{
    u64 r = mul_shoup_lazy(x, w, wq, q);
    return r >= q ? r - q : r;
}

HD u64 csub(u64 x, u64 q) { return x >= q ? x - q : x; }
