// Shared pieces of the 4096-point radix-16 x 16 x 16 kernels (welch4096.hip, csd4096.hip):
// the register-resident 16-point DFT, the LDS image geometry and the wave-level reduction.
// Index conventions and the bank-conflict argument are in welch4096.hip / DESIGN.md.
#pragma once
#include "fft_lds.hip.h"
#include "oth_internal.h"

static_assert(oth::kPilotProbes == 8, "load_pilot() (fft_lds.hip.h) adds eight probe means");

namespace oth {
namespace {

constexpr int T4 = 256;
constexpr int RS = 272;                    // float2 per k0 region (16 x 17)
constexpr int LDS_X = 16 * RS;             // exchange image
constexpr int LDS_RED = 16;                // per-wave half-segment sums (2 x 4) + chunk ticket
constexpr size_t LDS_BYTES = (LDS_X + LDS_RED) * sizeof(float2);

constexpr float C1 = 0.92387953251128674f;   // cos(pi/8)
constexpr float S1 = 0.38268343236508977f;   // sin(pi/8)
constexpr float RH = 0.70710678118654752f;   // sqrt(1/2)

__device__ __forceinline__ float2 mul_w1(float2 a) { return make_float2(fmaf(a.x, C1, a.y * S1), fmaf(a.y, C1, -a.x * S1)); }
__device__ __forceinline__ float2 mul_w2(float2 a) { return make_float2((a.x + a.y) * RH, (a.y - a.x) * RH); }
__device__ __forceinline__ float2 mul_w3(float2 a) { return make_float2(fmaf(a.x, S1, a.y * C1), fmaf(a.y, S1, -a.x * C1)); }
__device__ __forceinline__ float2 mul_w4(float2 a) { return make_float2(a.y, -a.x); }
__device__ __forceinline__ float2 mul_w6(float2 a) { return make_float2((a.y - a.x) * RH, -(a.x + a.y) * RH); }
__device__ __forceinline__ float2 mul_w9(float2 a) { return make_float2(-fmaf(a.x, C1, a.y * S1), fmaf(a.x, S1, -a.y * C1)); }

// position of output k of dft16() inside v[]
__host__ __device__ constexpr int r16(int k) { return 4 * (k & 3) + (k >> 2); }

constexpr float T1 = 0.41421356237309503f;   // tan(pi/8)

// Second butterfly layer of the 16-point DFT with the inner twiddles W16^(kl j) folded into the butterflies.
// Each twiddled input is written as (real scale) x (cheap combination p): W16^2 u = sqrt(1/2) (x + y, y - x),
// W16^1 u = cos(pi/8) (x + t y, y - t x), W16^3 u = cos(pi/8) (t x + y, t y - x), t = tan(pi/8), and so on;
// inside one 4-point butterfly the two odd inputs share their scale, so the scale rides on the fused
// multiply-adds that replace the butterfly's additions.  144 operations per 16-point DFT instead of 160
// (64 + 16 + 20 + 22 + 22 against 64 + 16 + 24 + 28 + 28).  `OTH_DFT16_PLAIN` restores the multiply-then-add form.
__device__ __forceinline__ void dft4_tail(float2 t0, float2 t1, float2 p1, float2 p3, float S, float2 &a0, float2 &a1,
                                          float2 &a2, float2 &a3) {
    const float2 q = cadd(p1, p3), r = csub(p1, p3);
    a0 = make_float2(fmaf(S, q.x, t0.x), fmaf(S, q.y, t0.y));
    a2 = make_float2(fmaf(-S, q.x, t0.x), fmaf(-S, q.y, t0.y));
    a1 = make_float2(fmaf(S, r.y, t1.x), fmaf(-S, r.x, t1.y));       // t1 + (-i) S r
    a3 = make_float2(fmaf(-S, r.y, t1.x), fmaf(S, r.x, t1.y));
}
#ifdef OTH_DFT16_PLAIN
constexpr bool kDft16Plain = true;
#else
constexpr bool kDft16Plain = false;
#endif
// PLAIN: multiply-then-add (160 operations; every inner twiddle product rounded on its own) - the periodogram-chain
// builds take it, their single rows are compared bin by bin and the chain is HBM-bound; the Welch averages take the
// folded form
template <bool PLAIN = kDft16Plain>
__device__ __forceinline__ void dft16_layer2(float2 (&v)[16]) {
  if constexpr (PLAIN) {
    v[5] = mul_w1(v[5]);
    v[9] = mul_w2(v[9]);
    v[13] = mul_w3(v[13]);
    v[6] = mul_w2(v[6]);
    v[10] = mul_w4(v[10]);
    v[14] = mul_w6(v[14]);
    v[7] = mul_w3(v[7]);
    v[11] = mul_w6(v[11]);
    v[15] = mul_w9(v[15]);
#pragma unroll
    for (int kl = 0; kl < 4; ++kl) dft4<false>(v[4 * kl], v[4 * kl + 1], v[4 * kl + 2], v[4 * kl + 3]);
  } else {
    dft4<false>(v[0], v[1], v[2], v[3]);
    {   // kl = 1: W^1, W^2, W^3 on v[5], v[6], v[7]
        const float2 u1 = v[5], u2 = v[6], u3 = v[7];
        const float2 p1 = make_float2(fmaf(T1, u1.y, u1.x), fmaf(-T1, u1.x, u1.y));       // W^1 u = C1 p1
        const float2 p2 = make_float2(u2.x + u2.y, u2.y - u2.x);                           // W^2 u = RH p2
        const float2 p3 = make_float2(fmaf(T1, u3.x, u3.y), fmaf(T1, u3.y, -u3.x));        // W^3 u = C1 p3
        const float2 t0 = make_float2(fmaf(RH, p2.x, v[4].x), fmaf(RH, p2.y, v[4].y));
        const float2 t1 = make_float2(fmaf(-RH, p2.x, v[4].x), fmaf(-RH, p2.y, v[4].y));
        dft4_tail(t0, t1, p1, p3, C1, v[4], v[5], v[6], v[7]);
    }
    {   // kl = 2: W^2, W^4 = -i, W^6 on v[9], v[10], v[11]
        const float2 u1 = v[9], u2 = v[10], u3 = v[11];
        const float2 p1 = make_float2(u1.x + u1.y, u1.y - u1.x);                           // W^2 u = RH p1
        const float2 p3 = make_float2(u3.y - u3.x, -u3.x - u3.y);                          // W^6 u = RH p3
        const float2 t0 = make_float2(v[8].x + u2.y, v[8].y - u2.x);
        const float2 t1 = make_float2(v[8].x - u2.y, v[8].y + u2.x);
        dft4_tail(t0, t1, p1, p3, RH, v[8], v[9], v[10], v[11]);
    }
    {   // kl = 3: W^3, W^6, W^9 = -W^1 on v[13], v[14], v[15]
        const float2 u1 = v[13], u2 = v[14], u3 = v[15];
        const float2 p1 = make_float2(fmaf(T1, u1.x, u1.y), fmaf(T1, u1.y, -u1.x));        // W^3 u = C1 p1
        const float2 p2 = make_float2(u2.y - u2.x, -u2.x - u2.y);                          // W^6 u = RH p2
        const float2 p3 = make_float2(fmaf(-T1, u3.y, -u3.x), fmaf(T1, u3.x, -u3.y));      // W^9 u = C1 p3
        const float2 t0 = make_float2(fmaf(RH, p2.x, v[12].x), fmaf(RH, p2.y, v[12].y));
        const float2 t1 = make_float2(fmaf(-RH, p2.x, v[12].x), fmaf(-RH, p2.y, v[12].y));
        dft4_tail(t0, t1, p1, p3, C1, v[12], v[13], v[14], v[15]);
    }
  }
}

// Forward 16-point DFT in place: in v[a], a = 0..15; out y[k] at v[r16(k)].
template <bool PLAIN = kDft16Plain>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
#pragma unroll
    for (int a0 = 0; a0 < 4; ++a0) dft4<false>(v[a0], v[a0 + 4], v[a0 + 8], v[a0 + 12]);
    dft16_layer2<PLAIN>(v);
}

// 8-point DFT, natural order in and out; the sqrt(1/2) of W8^1 and W8^3 rides on the last layer's additions
// (PLAIN: multiplied out first, as dft16_layer2<true>)
template <bool PLAIN = kDft16Plain>
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    dft4<false>(v[0], v[2], v[4], v[6]);
    dft4<false>(v[1], v[3], v[5], v[7]);
    const float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    const float2 o0 = v[1], o2 = mul_w4(v[5]);                                   // W8^k = W16^(2k)
    if constexpr (PLAIN) {
        const float2 o1 = mul_w2(v[3]), o3 = mul_w6(v[7]);
        v[0] = cadd(e0, o0);
        v[4] = csub(e0, o0);
        v[1] = cadd(e1, o1);
        v[5] = csub(e1, o1);
        v[2] = cadd(e2, o2);
        v[6] = csub(e2, o2);
        v[3] = cadd(e3, o3);
        v[7] = csub(e3, o3);
        return;
    }
    const float2 p1 = make_float2(v[3].x + v[3].y, v[3].y - v[3].x);             // W8^1 o = RH p1
    const float2 p3 = make_float2(v[7].y - v[7].x, -v[7].x - v[7].y);            // W8^3 o = RH p3
    v[0] = cadd(e0, o0);
    v[4] = csub(e0, o0);
    v[1] = make_float2(fmaf(RH, p1.x, e1.x), fmaf(RH, p1.y, e1.y));
    v[5] = make_float2(fmaf(-RH, p1.x, e1.x), fmaf(-RH, p1.y, e1.y));
    v[2] = cadd(e2, o2);
    v[6] = csub(e2, o2);
    v[3] = make_float2(fmaf(RH, p3.x, e3.x), fmaf(RH, p3.y, e3.y));
    v[7] = make_float2(fmaf(-RH, p3.x, e3.x), fmaf(-RH, p3.y, e3.y));
}

// dft16() of sixteen float2 read from LDS at base[STRIDE * i]: the reads are issued as plain ds_read_b64 in the
// order the first butterfly layer takes them (hipcc pairs them into ds_read2_b64 - half the LDS rate on gfx950 -
// and waits for all sixteen), each butterfly waits only for its own four.  The reads must be the wave's most
// recent LDS operations when this is called; older LDS/scalar-memory operations only make the waits longer.
// `issued()` runs once the reads are out (e.g. to drop the wave's priority for the butterflies).
// `mid()` runs between the two butterfly layers, when every read of this call has returned: LDS reads issued there
// (a twiddle table, say) have the second layer to hide behind and do not disturb the counted waits above them.
template <int STRIDE, class F, class M>
__device__ __forceinline__ void dft16_from_lds(float2 (&v)[16], const float2 *base, F issued, M mid) {
#ifdef OTH_PLAIN_LDS_READS      // A/B switch: compiler-scheduled reads
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = base[STRIDE * i];
    issued();
    mid();
    dft16(v);
    return;
#endif
    const unsigned addr = (unsigned)(unsigned long long)base;      // LDS byte address = low half of the flat address
    double r[16];
#define OTH_LDS_READ(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(8 * STRIDE * (i)))
    OTH_LDS_READ(0); OTH_LDS_READ(4); OTH_LDS_READ(8); OTH_LDS_READ(12);
    OTH_LDS_READ(1); OTH_LDS_READ(5); OTH_LDS_READ(9); OTH_LDS_READ(13);
    OTH_LDS_READ(2); OTH_LDS_READ(6); OTH_LDS_READ(10); OTH_LDS_READ(14);
    OTH_LDS_READ(3); OTH_LDS_READ(7); OTH_LDS_READ(11); OTH_LDS_READ(15);
#undef OTH_LDS_READ
    issued();
    // each wait also names an output of the butterfly before it, which keeps that butterfly in front of the wait
#define OTH_LDS_WAIT(n, a, dep) \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r[a]), "+v"(r[a + 4]), "+v"(r[a + 8]), "+v"(r[a + 12]), "+v"(dep))
    float dep = 0.f;
#pragma unroll
    for (int a0 = 0; a0 < 4; ++a0) {
        if (a0 == 0) OTH_LDS_WAIT(12, 0, dep);
        else if (a0 == 1) OTH_LDS_WAIT(8, 1, v[0].x);
        else if (a0 == 2) OTH_LDS_WAIT(4, 2, v[1].x);
        else OTH_LDS_WAIT(0, 3, v[2].x);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[a0 + 4 * j] = __builtin_bit_cast(float2, r[a0 + 4 * j]);
        dft4<false>(v[a0], v[a0 + 4], v[a0 + 8], v[a0 + 12]);
    }
#undef OTH_LDS_WAIT
    mid();
    dft16_layer2(v);
}
template <int STRIDE, class F>
__device__ __forceinline__ void dft16_from_lds(float2 (&v)[16], const float2 *base, F issued) {
    dft16_from_lds<STRIDE>(v, base, issued, [] {});
}

// Non-temporal load of a sample that is read once: it does not displace the tables and partial sums in L2.
__device__ __forceinline__ float2 load_once(const float2 *p) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 r = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(p));
    return make_float2(r.x, r.y);
}

// Chunk c of the segment schedule: the first `nbig` chunks have `chunk` segments, the rest `tail_chunk`
// (smaller chunks for the last round even out the finish of the dynamic schedule).
// 4-byte non-temporal-free load of a window value at (uniform row base) + (lane offset), pinned like load_row_nt: the
// compiler would hoist sixteen loop-invariant window loads into sixteen registers the two segments in flight need
__device__ __forceinline__ void load_win(float &dst, unsigned lane_off, const char *row) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(row) : "memory");
}

__device__ __forceinline__ long long chunk_count(const WelchArgs &p) {
    const long long rest = p.nseg - p.nbig * p.chunk;
    return p.nbig + (rest + p.tail_chunk - 1) / p.tail_chunk;
}
__device__ __forceinline__ void chunk_range(const WelchArgs &p, long long c, long long &sb, long long &se) {
    if (c < p.nbig) {
        sb = c * p.chunk;
        se = sb + p.chunk;
    } else {
        sb = p.nbig * p.chunk + (c - p.nbig) * p.tail_chunk;
        se = sb + p.tail_chunk < p.nseg ? sb + p.tail_chunk : p.nseg;
    }
}

// the same schedule for kernels that do not take a WelchArgs (segfft.hip); I = int where the launcher has checked that
// the segment count fits (welch16k1x.hip: ten 64-bit loop variables cost the kernel its last scalar registers)
template <class I> __device__ __forceinline__ I chunk_count_of(I nseg, I nbig, int chunk, int tail_chunk) {
    const I rest = nseg - nbig * chunk;
    return nbig + (rest + tail_chunk - 1) / tail_chunk;
}
template <class I> __device__ __forceinline__ void chunk_range_of(I nseg, I nbig, int chunk, int tail_chunk, I c, I &sb, I &se) {
    if (c < nbig) {
        sb = c * chunk;
        se = sb + chunk;
    } else {
        sb = nbig * chunk + (c - nbig) * tail_chunk;
        se = sb + tail_chunk < nseg ? sb + tail_chunk : nseg;
    }
}

// Wave priority.  Everything that starts long-latency work or releases other waves (global loads, the
// segment sum, LDS exchanges, barriers) runs at raised priority so that it is issued as early as
// possible; the three 16-point butterflies, pure VALU, run at base priority and fill the gaps of the
// other waves on the SIMD.  Measured on welch4096: 0.676 -> 0.655 ms.
__device__ __forceinline__ void prio_latency() { __builtin_amdgcn_s_setprio(2); }
__device__ __forceinline__ void prio_compute() { __builtin_amdgcn_s_setprio(0); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would
// wait for the global prefetch of the next half-segment right after it was issued; here outstanding
// global loads stay in flight across the barrier (the compiler still waits for them at first use).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// out[STRIDE * k] = v[r16(k)] * W^k for k = 0..15, where W^k is rebuilt from p1 = W and p4 = W^4 as
// (W^4)^i * W^j, k = 4i + j: 13 complex products per call instead of 15 stored (30 VGPRs) or loaded
// values.  Loading them from an LDS table costs more than the arithmetic: hipcc issues one table read,
// waits lgkmcnt(0), multiplies and writes, fifteen exposed LDS round trips per exchange.  The empty asm
// keeps the loop-invariant powers from being hoisted back into registers.
template <int STRIDE>
__device__ __forceinline__ void scatter_pow16(const float2 (&v)[16], float2 *out, float2 p1, float2 p4) {
    float2 wj[4], wi[4];
    wj[1] = p1;
    wi[1] = p4;
    asm volatile("" : "+v"(wj[1].x), "+v"(wj[1].y), "+v"(wi[1].x), "+v"(wi[1].y));
    wj[2] = cmul(wj[1], wj[1]);
    wj[3] = cmul(wj[2], wj[1]);
    wi[2] = cmul(wi[1], wi[1]);
    wi[3] = cmul(wi[2], wi[1]);
    out[0] = v[0];                              // r16(0) == 0
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const int i = k >> 2, j = k & 3;
        const float2 w = (i == 0) ? wj[j] : ((j == 0) ? wi[i] : cmul(wi[i], wj[j]));
        out[STRIDE * k] = cmul(v[r16(k)], w);
    }
}

// The same scatter with six stored powers W, W^2, W^3, W^4, W^8, W^12: nine products instead of thirteen (the four
// squarings / cubings of scatter_pow16 are gone), at eight more registers for the caller.
template <int STRIDE>
__device__ __forceinline__ void scatter_pow16_six(const float2 (&v)[16], float2 *out, float2 p1, float2 p2, float2 p3, float2 p4,
                                                  float2 p8, float2 p12) {
    float2 wj[4], wi[4];
    wj[1] = p1, wj[2] = p2, wj[3] = p3;
    wi[1] = p4, wi[2] = p8, wi[3] = p12;
    asm volatile("" : "+v"(wj[1].x), "+v"(wj[1].y), "+v"(wj[2].x), "+v"(wj[2].y), "+v"(wj[3].x), "+v"(wj[3].y));
    out[0] = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const int i = k >> 2, j = k & 3;
        const float2 w = (i == 0) ? wj[j] : ((j == 0) ? wi[i] : cmul(wi[i], wj[j]));
        out[STRIDE * k] = cmul(v[r16(k)], w);
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
    return v + __int_as_float(x);
}

// Sum over the 64 lanes of a wave, same value returned in every lane.
__device__ __forceinline__ float wave_total(float v) {
    // DPP row operations + v_readlane: no LDS round trips (six dependent ds_bpermute otherwise)
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror: every lane now holds its row-of-16 sum
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48));
}

// Sum over the 64 lanes of a wave, valid in lane 63 only (rows 1, 3 take in the row before them, then rows 2, 3
// the first half): two DPP adds instead of four v_readlane + their hazard slots when one lane stores the result.
__device__ __forceinline__ float wave_total_lane63(float v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    // row_bcast:15 into rows 1, 3 and row_bcast:31 into rows 2, 3 as ONE masked add each: rows the mask leaves out
    // keep their value.  Through __builtin_amdgcn_update_dpp(0, ...) + add the compiler emits v_mov 0, v_mov_dpp, v_add
    // per step (the builtin's semantics give the other rows 0 first): 8 VALU instructions more per complex sum.  The
    // s_nop covers the VALU-write -> DPP-read hazard, which the compiler does not see inside the asm.
#ifdef OTH_WAVE_TOTAL_BUILTIN      // A/B switch: the form of rounds 1-2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));   // row_bcast:31
#else
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
#endif
    return v;
}

// WelchArgs.pilot_inline: a 256-thread producer team's share of forming the pilot inside the launch.  Probe k (of
// kPilotProbes = 8) is 256 consecutive samples - thread t takes sample t - out of segment (nseg - 1) k / 7, the k-th
// eighth of it, so the probes cover the launch in time like pilot_mean_kernel's and each is one coalesced 2 KiB read
// (all eight in flight at once; after the first workgroup of an XCD they are L2 hits).  inline_pilot_load() only issues
// the loads (the caller puts its first sample loads in front of them: one round trip for both); inline_pilot_store()
// leaves every probe's per-wave total in slot[4 k + wave] (lane 63); the CALLER places a workgroup barrier behind it
// and then reads inline_pilot_value() = the average of the eight probe means, as load_pilot().
struct PilotProbes {
    float2 v[kPilotProbes];
};
__device__ __forceinline__ PilotProbes inline_pilot_load(const float2 *xb, long long nseg, int step, int t) {
    PilotProbes r;
#pragma unroll
    for (int k = 0; k < kPilotProbes; ++k) {
        const long long seg = ((nseg - 1) * k) / (kPilotProbes - 1);
        const float2 *src = xb + (size_t)__builtin_amdgcn_readfirstlane((int)seg) * (size_t)step + 512 * k;
        r.v[k] = src[(unsigned)t];
    }
    return r;
}
__device__ __forceinline__ void inline_pilot_store(const PilotProbes &r, int t, float2 *slot) {
#pragma unroll
    for (int k = 0; k < kPilotProbes; ++k) {
        const float2 s = make_float2(wave_total_lane63(r.v[k].x), wave_total_lane63(r.v[k].y));
        if ((t & 63) == 63) slot[4 * k + (t >> 6)] = s;
    }
}
__device__ __forceinline__ float2 inline_pilot_value(const float2 *slot) {
    float2 q[kPilotProbes];
#pragma unroll
    for (int k = 0; k < kPilotProbes; ++k) {
        const float2 a = cadd(slot[4 * k], slot[4 * k + 1]), b = cadd(slot[4 * k + 2], slot[4 * k + 3]);
        q[k] = make_float2((a.x + b.x) * (1.0f / 256.0f), (a.y + b.y) * (1.0f / 256.0f));
    }
    return pilot_of_probes(q);
}


}  // namespace
}  // namespace oth
