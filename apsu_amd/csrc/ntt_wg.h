// The workgroup body of the LDS-resident transform (device code only): the passes of ntt_core.h in execution order with the
// synchronisation between them.  Shared by kernels.hip and tools/microbench/ntt_variants.hip.
//
// Synchronisation (round 4).  A pass whose butterfly blocks are at most 1024 coefficients long (LOGN - S <= 10) touches, in
// wave w, exactly the coefficients [1024 w, 1024 w + 1024): 64 lanes x 16 coefficients.  Two consecutive passes of that kind
// exchange data only INSIDE a wave, through that wave's own 8 KiB of the LDS image, so no workgroup barrier is needed between
// them -- LDS instructions of one wave execute in order, the compiler only has to keep the stores in front of the loads
// (wave-scope fence).  At n = 8192 (passes 3,3,3,4) that leaves ONE s_barrier per transform (behind the forward's first pass /
// in front of the inverse's last one) instead of four / three: the eight waves of a workgroup drift apart, and one wave's LDS
// round trip and twiddle loads are covered by the others' butterflies instead of all eight waiting at the same barrier.
// The forward's closing store and the tensor loader's staging follow the same wave-private ranges.  WS = false restores the
// barrier-per-pass schedule of rounds 1-3 (kept for the A/B in tools/microbench/ntt_variants.hip).
#pragma once
#include <type_traits>

#include "ntt_core.h"

#if defined(__HIPCC__)

// does pass p (forward numbering) of a T-thread workgroup stay inside each wave's own range of 64 C coefficients (C per lane: 1024 at C = 16)?
template <int LOGN, int T, int C = 16> constexpr bool ntt_wave_private(int p) { return T == 64 || (LOGN - plan_s(LOGN, p, C) <= (C == 16 ? 10 : C == 8 ? 9 : 8)); }

template <bool WAVE> __device__ __forceinline__ void ntt_sync()
{
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else __syncthreads();
}

// RED: 0 = the limb is transformed in place; 1 = forward transform of a limb gathered from `src`, residues of another modulus
// reduced on load (the key switch's decomposition); 2 = gathered, NOT reduced: the host has checked that the source residues fit
// the lazy range of a narrow modulus (max source modulus + 4 LOGN q < 2^64), the transform is linear and its closing reduction
// takes any 64-bit value.
// PF: every pass issues the NEXT pass's twiddle loads between its butterflies and its stores (ntt_core.h, TwRegs), so the table
// reads (L2 latency, 7 to 15 x 16 bytes per lane and pass) are in flight during the LDS turnaround instead of behind it.
// Measured level to slightly slower than WS alone (tools/microbench/ntt_variants.hip, profiles/r04_ntt_variants.txt: the other
// waves already cover that latency, and the held twiddles cost the compiler its scheduling room): off by default.
// NEXT (round 4, second half): the workgroup that will occupy this workgroup's slot next -- in steady state the one `slots` further
// on in the grid, which the round-robin dispatch puts on the same XCD and therefore behind the same L2 -- reads its limb from HBM
// with nothing else to do while it waits.  `next` != nullptr: this thread touches one 128-byte line of that limb (T threads x
// 128 B = the whole limb at n = 16 T) right before the pass executed LAST, so that the successor's first pass finds its input in
// the L2 instead of waiting for HBM.  The load has no consumer: its destination register is kept reserved until the end of the body.
// STAGGER: waves 4..7 (the SIMD partners of waves 0..3) sleep 64 x STAGGER cycles once, so that the two waves a workgroup has
// on each SIMD do not reach their memory phases together (experiment switch of the microbenchmark).
__device__ __forceinline__ void ntt_touch_line(const void *line, unsigned &sink)
{
    asm volatile("global_load_dword %0, %1, off" : "=v"(sink) : "v"(line));
}
__device__ __forceinline__ void ntt_touch_done(unsigned &sink) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)); }

// C (round 6): coefficients per lane.  16 = the throughput form (T = n / 16 threads per limb); 8 = the latency form for launches that leave
// CUs idle (T = n / 8: twice the waves per limb, five passes 2,3,3,3,2 at n = 8192 with two workgroup barriers -- ntt_core.h, plan_k).
template <int LOGN, bool INV, int MODE, int T, int RED = 0, bool RAW = false, class SRC = SrcPlain, bool WS = true, bool PF = false, int STAGGER = 0,
          int NEXTPASS = -1, int C = 16>
__device__ __forceinline__ void ntt_body(u64 *lds, u64 *__restrict__ p, const NttTable &tab, int tid, const u64 *src = nullptr,
                                         const SRC &operands = SRC(), const void *next = nullptr)
{
    constexpr int N = 1 << LOGN;
    constexpr int P = plan_passes(LOGN, C);
    static_assert(T % 64 == 0 && T * C >= N, "one work item per thread");
    static_assert(P > 0, "no pass schedule for this ring size and coefficients per lane");
    static_assert(C == 16 || (!PF && STAGGER == 0 && NEXTPASS < 0), "the experiment switches exist for the 16-coefficient form only");
    constexpr int WCO = 64 * C / 128;                // 16-byte-per-lane sweeps over a wave's own range of 64 C coefficients
    // forward numbering of the pass executed k-th
    auto fp = [](int k) constexpr { return INV ? P - 1 - k : k; };
    // the sync in front of the pass executed k-th (k >= 1) may be wave-level iff that pass and the one before it are wave-private
    // (S grows with the forward numbering, so the lower-numbered of the two decides)
    constexpr bool W1 = WS && P > 1 && ntt_wave_private<LOGN, T, C>(fp(0) < fp(1) ? fp(0) : fp(1));
    constexpr bool W2 = WS && P > 2 && ntt_wave_private<LOGN, T, C>(fp(1) < fp(2) ? fp(1) : fp(2));
    constexpr bool W3 = WS && P > 3 && ntt_wave_private<LOGN, T, C>(fp(2) < fp(3) ? fp(2) : fp(3));
    constexpr bool W4 = WS && P > 4 && ntt_wave_private<LOGN, T, C>(fp(3) < fp(4) ? fp(3) : fp(4));
    // the wave's own range for the coalesced loops (tensor staging, forward store): 16-byte pieces, 1 KiB per wave instruction
    const int wbase = (tid & ~63) * C, lane = tid & 63;
    unsigned sink = 0;
    if constexpr (NEXTPASS == 0) { if (next) ntt_touch_line(next, sink); }
    if constexpr (STAGGER > 0 && INV) { if (tid & 256) __builtin_amdgcn_s_sleep(STAGGER); }
    // (with SRC = SrcPlain the first inverse pass reads its 16 contiguous coefficients per lane from global memory: 128 B per lane, every
    //  line is consumed by the wave's eight consecutive loads.  Round 2 measured staging the limb through LDS with coalesced loads for
    //  SMALL launches only -- no gain, profiles/r02_ntt_latency.txt -- and left it; on launches of thousands of limbs the direct reads
    //  thrash the vector L1, and since round 5 the library's inverse transforms pass SrcStaged and take the staged path below:
    //  -3.5 ... -6 %, profiles/r05_ntt_staged_inverse.txt)
    if constexpr (PF && P > 1) {
        constexpr bool COMPUTED = !std::is_same<SRC, SrcPlain>::value;
        // (the pass that leaves a non-RAW inverse transform also loads its 16 twist constants: with a prefetched set on top the
        //  kernel spills, so that pass keeps loading its twiddles itself)
        constexpr bool PF_LAST = !(INV && !RAW);
        constexpr bool ON1 = P > 2 || PF_LAST, ON2 = P > 2 && (P > 3 || PF_LAST), ON3 = P > 3 && PF_LAST;
        std::conditional_t<ON1, typename ExecPass<LOGN, INV, 1>::Tw, TwInline> t1;
        std::conditional_t<ON2, typename ExecPass<LOGN, INV, (P > 2 ? 2 : 1)>::Tw, TwInline> t2;
        std::conditional_t<ON3, typename ExecPass<LOGN, INV, (P > 3 ? 3 : 1)>::Tw, TwInline> t3;
        auto h1 = [&]() { if constexpr (ON1) ntt_pass_twiddles<LOGN, INV, 1>(t1, tid, tab); };
        auto h2 = [&]() { if constexpr (ON2) ntt_pass_twiddles<LOGN, INV, 2>(t2, tid, tab); };
        auto h3 = [&]() { if constexpr (ON3) ntt_pass_twiddles<LOGN, INV, 3>(t3, tid, tab); };
        if constexpr (RED != 0) ntt_pass<LOGN, INV, MODE, 0, RED, false, SrcPlain, false, TwInline, decltype(h1)>(lds, const_cast<u64 *>(src), tid, T, tab, SrcPlain(), TwInline(), h1);
        else if constexpr (COMPUTED) {
            constexpr bool WL = WS && ntt_wave_private<LOGN, T>(fp(0));
            if constexpr (WL) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int e = wbase + 128 * i + 2 * lane;
                    if (e < N) *reinterpret_cast<u64x2 *>(lds + lds_slot(e)) = src_load2(operands, p, e, tab);
                }
            } else {
                for (int e = 2 * tid; e < N; e += 2 * T) *reinterpret_cast<u64x2 *>(lds + lds_slot(e)) = src_load2(operands, p, e, tab);
            }
            ntt_sync<WL>();
            ntt_pass<LOGN, INV, MODE, 0, 0, false, SRC, true, TwInline, decltype(h1)>(lds, p, tid, T, tab, operands, TwInline(), h1);
        } else ntt_pass<LOGN, INV, MODE, 0, 0, false, SrcPlain, false, TwInline, decltype(h1)>(lds, p, tid, T, tab, SrcPlain(), TwInline(), h1);
        ntt_sync<W1>();
        ntt_pass<LOGN, INV, MODE, 1, 0, RAW, SrcPlain, false, decltype(t1), decltype(h2)>(lds, p, tid, T, tab, SrcPlain(), t1, h2);
        if constexpr (P > 2) { ntt_sync<W2>(); ntt_pass<LOGN, INV, MODE, 2, 0, RAW, SrcPlain, false, decltype(t2), decltype(h3)>(lds, p, tid, T, tab, SrcPlain(), t2, h3); }
        if constexpr (P > 3) { ntt_sync<W3>(); ntt_pass<LOGN, INV, MODE, 3, 0, RAW, SrcPlain, false, decltype(t3)>(lds, p, tid, T, tab, SrcPlain(), t3); }
    } else {
    if constexpr (RED != 0) ntt_pass<LOGN, INV, MODE, 0, RED, false, SrcPlain, false, TwInline, NoHook, C>(lds, const_cast<u64 *>(src), tid, T, tab);   // forward only: pass 0 just reads
    else if constexpr (!std::is_same<SRC, SrcPlain>::value) {
        // computed input (tensor product on load): four operand streams read with the first pass's 128-byte lane stride
        // thrash the vector L1 (measured: the fused launch 65 % slower than tensor + transform apart), so the products are
        // formed with coalesced 16-byte loads into the LDS image and the first pass starts from there
        constexpr bool WL = WS && ntt_wave_private<LOGN, T, C>(fp(0));
        if constexpr (WL) {
#pragma unroll
            for (int i = 0; i < WCO; i++) {
                const int e = wbase + 128 * i + 2 * lane;
                if (e < N) *reinterpret_cast<u64x2 *>(lds + lds_slot(e)) = src_load2(operands, p, e, tab);
            }
        } else {
            for (int e = 2 * tid; e < N; e += 2 * T) *reinterpret_cast<u64x2 *>(lds + lds_slot(e)) = src_load2(operands, p, e, tab);
        }
        ntt_sync<WL>();
        // (the source rides along although the pass reads the LDS image: its input bound sets the constant of the first pass's
        //  multiplication-free butterflies, ntt_core.h src_in_bound)
        ntt_pass<LOGN, INV, MODE, 0, 0, false, SRC, true, TwInline, NoHook, C>(lds, p, tid, T, tab, operands);
    } else ntt_pass<LOGN, INV, MODE, 0, 0, false, SrcPlain, false, TwInline, NoHook, C>(lds, p, tid, T, tab);
    // (RAW only concerns the pass that leaves the inverse transform, the last one)
    if constexpr (P > 1) {
        ntt_sync<W1>();
        if constexpr (STAGGER > 0 && !INV) { if (tid & 256) __builtin_amdgcn_s_sleep(STAGGER); }
        if constexpr (NEXTPASS == 1) { if (next) ntt_touch_line(next, sink); }
        ntt_pass<LOGN, INV, MODE, 1, 0, RAW, SrcPlain, false, TwInline, NoHook, C>(lds, p, tid, T, tab);
    }
    if constexpr (P > 2) { ntt_sync<W2>(); if constexpr (NEXTPASS == 2) { if (next) ntt_touch_line(next, sink); } ntt_pass<LOGN, INV, MODE, 2, 0, RAW, SrcPlain, false, TwInline, NoHook, C>(lds, p, tid, T, tab); }
    if constexpr (P > 3) { ntt_sync<W3>(); if constexpr (NEXTPASS == 3) { if (next) ntt_touch_line(next, sink); } ntt_pass<LOGN, INV, MODE, 3, 0, RAW, SrcPlain, false, TwInline, NoHook, C>(lds, p, tid, T, tab); }
    if constexpr (P > 4) { ntt_sync<W4>(); ntt_pass<LOGN, INV, MODE, 4, 0, RAW, SrcPlain, false, TwInline, NoHook, C>(lds, p, tid, T, tab); }
    }
    if constexpr (!INV) {                        // forward: the last pass left 16 contiguous coefficients per lane in LDS
        constexpr bool WF = WS && ntt_wave_private<LOGN, T, C>(P - 1);
        ntt_sync<WF>();
        if constexpr (WF) {
#pragma unroll
            for (int i = 0; i < WCO; i++) {
                const int e = wbase + 128 * i + 2 * lane;
                if (e < N) {
                    u64x2 v = *reinterpret_cast<const u64x2 *>(lds + lds_slot(e));
                    v[0] = ntt_fwd_finish<MODE>(v[0], tab);
                    v[1] = ntt_fwd_finish<MODE>(v[1], tab);
                    *reinterpret_cast<u64x2 *>(p + e) = v;
                }
            }
        } else {
            for (int e = 2 * tid; e < N; e += 2 * T) {
                u64x2 v = *reinterpret_cast<const u64x2 *>(lds + lds_slot(e));
                v[0] = ntt_fwd_finish<MODE>(v[0], tab);
                v[1] = ntt_fwd_finish<MODE>(v[1], tab);
                *reinterpret_cast<u64x2 *>(p + e) = v;
            }
        }
    }
    if constexpr (NEXTPASS >= 0) { if (next) ntt_touch_done(sink); }
}

#endif // __HIPCC__
