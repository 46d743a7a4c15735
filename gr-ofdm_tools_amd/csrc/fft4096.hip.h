// Shared pieces of the 4096-point radix-16 x 16 x 16 kernels (welch4096.hip, csd4096.hip):
// the register-resident 16-point DFT, the LDS image geometry and the wave-level reduction.
// Index conventions and the bank-conflict argument are in welch4096.hip / DESIGN.md.
#pragma once
#include "fft_lds.hip.h"
#include "oth_internal.h"

#ifndef OTH_W4096_DPP
#define OTH_W4096_DPP 1      // segment-sum wave reduction with DPP row ops + v_readlane (no LDS round trips)
#endif

namespace oth {
namespace {

constexpr int T4 = 256;
constexpr int RS = 272;                    // float2 per k0 region (16 x 17)
constexpr int LDS_X = 16 * RS;             // exchange image
constexpr int LDS_TW2 = 256;               // W256^(k1 c) as [k1][c]
constexpr int LDS_RED = 16;                // per-wave half-segment sums (2 x 4) + chunk ticket
constexpr size_t LDS_BYTES = (LDS_X + LDS_TW2 + LDS_RED) * sizeof(float2);

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

// Forward 16-point DFT in place: in v[a], a = 0..15; out y[k] at v[r16(k)].
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
#pragma unroll
    for (int a0 = 0; a0 < 4; ++a0) dft4<false>(v[a0], v[a0 + 4], v[a0 + 8], v[a0 + 12]);
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
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would
// wait for the global prefetch of the next half-segment right after it was issued; here outstanding
// global loads stay in flight across the barrier (the compiler still waits for them at first use).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16 x ds_read_b64 at byte offsets i * STRIDE_BYTES from one LDS address, issued as plain b64 reads
// (hipcc pairs them into ds_read2_b64, which moves half the bytes per LDS cycle) and waited for in
// the same statement, so the compiler never sees a register whose data is still in flight.
typedef float oth_v2f __attribute__((ext_vector_type(2)));
template <int STRIDE_BYTES>
__device__ __forceinline__ void lds_read16_b64(float2 (&v)[16], const float2 *lds_ptr) {
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const float2 *)lds_ptr;
    oth_v2f r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
    asm volatile(
        "ds_read_b64 %0, %16 offset:%17\n\tds_read_b64 %1, %16 offset:%18\n\tds_read_b64 %2, %16 offset:%19\n\t"
        "ds_read_b64 %3, %16 offset:%20\n\tds_read_b64 %4, %16 offset:%21\n\tds_read_b64 %5, %16 offset:%22\n\t"
        "ds_read_b64 %6, %16 offset:%23\n\tds_read_b64 %7, %16 offset:%24\n\tds_read_b64 %8, %16 offset:%25\n\t"
        "ds_read_b64 %9, %16 offset:%26\n\tds_read_b64 %10, %16 offset:%27\n\tds_read_b64 %11, %16 offset:%28\n\t"
        "ds_read_b64 %12, %16 offset:%29\n\tds_read_b64 %13, %16 offset:%30\n\tds_read_b64 %14, %16 offset:%31\n\t"
        "ds_read_b64 %15, %16 offset:%32\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8),
          "=&v"(r9), "=&v"(r10), "=&v"(r11), "=&v"(r12), "=&v"(r13), "=&v"(r14), "=&v"(r15)
        : "v"(a), "n"(0 * STRIDE_BYTES), "n"(1 * STRIDE_BYTES), "n"(2 * STRIDE_BYTES), "n"(3 * STRIDE_BYTES),
          "n"(4 * STRIDE_BYTES), "n"(5 * STRIDE_BYTES), "n"(6 * STRIDE_BYTES), "n"(7 * STRIDE_BYTES),
          "n"(8 * STRIDE_BYTES), "n"(9 * STRIDE_BYTES), "n"(10 * STRIDE_BYTES), "n"(11 * STRIDE_BYTES),
          "n"(12 * STRIDE_BYTES), "n"(13 * STRIDE_BYTES), "n"(14 * STRIDE_BYTES), "n"(15 * STRIDE_BYTES)
        : "memory");
    __builtin_amdgcn_sched_barrier(0);
    const oth_v2f r[16] = {r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15};
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = make_float2(r[i].x, r[i].y);
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
#if OTH_W4096_DPP
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror: every lane now holds its row-of-16 sum
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48));
#else
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
#endif
}


}  // namespace
}  // namespace oth
