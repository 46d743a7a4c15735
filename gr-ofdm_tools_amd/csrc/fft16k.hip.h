// Pieces of the 16384-point scheme of welch16k1x.hip (one 1024-thread workgroup, N = 16 x 16 x 16 x 4 with one cross-wave
// exchange) that welch32k.hip shares: the LDS region geometry, the in-place twiddle of pass 3 and the radix-4 butterfly over
// the four lanes of a quad.
#pragma once
#include "fft4096.hip.h"

namespace oth {
namespace {

constexpr int XROW = 68;                       // float2 per exchange-B row (64 + 4 pad)
constexpr int XREG = 16 * XROW;                // float2 per wave region (1088: exchange A uses [0, 1024))

// v[r16(k)] *= W^k, k = 1..15, W^k rebuilt from p1 = W and p4 = W^4 as in scatter_pow16
__device__ __forceinline__ void twiddle_pow16_inplace(float2 (&v)[16], float2 p1, float2 p4) {
    float2 wj[4], wi[4];
    wj[1] = p1;
    wi[1] = p4;
    asm volatile("" : "+v"(wj[1].x), "+v"(wj[1].y), "+v"(wi[1].x), "+v"(wi[1].y));
    wj[2] = cmul(wj[1], wj[1]);
    wj[3] = cmul(wj[2], wj[1]);
    wi[2] = cmul(wi[1], wi[1]);
    wi[3] = cmul(wi[2], wi[1]);
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const int i = k >> 2, j = k & 3;
        const float2 w = (i == 0) ? wj[j] : ((j == 0) ? wi[i] : cmul(wi[i], wj[j]));
        v[r16(k)] = cmul(v[r16(k)], w);
    }
}

// Radix-4 butterfly over the four lanes of a quad (lane q holds input q of every one of its sixteen registers), in
// place; lane q ends with output k3 = bit-reversed q, up to a factor of -1 or -i (the caller takes |.|^2):
//   stage 1 (partner q ^ 2)  x <- x + s1 x'        s1 = +1, +1, -1, -1:  e0, e1, -(d0), -(d1)
//   stage 2 (partner q ^ 1)  x <- x + c x'          c = 1, -1, -i, -i:   X0, -X2, -X1, -(d1 - i d0) = i X3 / ... |.| equal
// c = al - i be:  re += al re' + be im',  im += al im' - be re'.
__device__ __forceinline__ void quad_dft4_dpp(float2 (&v)[16], float s1, float al, float be, float nbe) {
    // Each asm block is ONE statement: the compiler cannot put a VALU write of a register between the wait states and
    // the DPP read of it (VALU write -> DPP read needs two wait states, which its hazard recognizer does not see inside
    // inline asm).  Stage 1 blocks open with s_nop 1; in the stage 2 blocks the four copies come first - four VALU
    // instructions between the block's start and its first DPP read, and between each copy and the DPP read of it.
#define OTH_Q1 "v_fmac_f32_dpp %0, %0, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
               "v_fmac_f32_dpp %1, %1, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
               "v_fmac_f32_dpp %2, %2, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
               "v_fmac_f32_dpp %3, %3, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
               "v_fmac_f32_dpp %4, %4, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
               "v_fmac_f32_dpp %5, %5, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
               "v_fmac_f32_dpp %6, %6, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
               "v_fmac_f32_dpp %7, %7, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
#pragma unroll
    for (int i = 0; i < 16; i += 4)
        asm volatile("s_nop 1\n\t" OTH_Q1
                     : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i + 1].x), "+v"(v[i + 1].y), "+v"(v[i + 2].x), "+v"(v[i + 2].y),
                       "+v"(v[i + 3].x), "+v"(v[i + 3].y)
                     : "v"(s1));
#undef OTH_Q1
    // x = %0/%2/%4/%6, y = %1/%3/%5/%7, old-x copies %8..%11, al %12, be %13, -be %14
#define OTH_Q2(x, y, t)                                                                            \
    "v_fmac_f32_dpp " x ", " x ", %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"         \
    "v_fmac_f32_dpp " x ", " y ", %13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"         \
    "v_fmac_f32_dpp " y ", " y ", %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"         \
    "v_fmac_f32_dpp " y ", " t ", %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        float t0, t1, t2, t3;
        asm volatile("v_mov_b32 %8, %0\n\tv_mov_b32 %9, %2\n\tv_mov_b32 %10, %4\n\tv_mov_b32 %11, %6\n\t"      // the partner reads the OLD real part last
                     OTH_Q2("%0", "%1", "%8") OTH_Q2("%2", "%3", "%9") OTH_Q2("%4", "%5", "%10") OTH_Q2("%6", "%7", "%11")
                     : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i + 1].x), "+v"(v[i + 1].y), "+v"(v[i + 2].x), "+v"(v[i + 2].y),
                       "+v"(v[i + 3].x), "+v"(v[i + 3].y), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                     : "v"(al), "v"(be), "v"(nbe));
    }
#undef OTH_Q2
}

}  // namespace
}  // namespace oth
