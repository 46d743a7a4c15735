// Workgroup-cooperative FFT in LDS for any power-of-two size (16 .. 16384).
//
// Stockham autosort, radix 4 (one radix-2 pass first when log2 N is odd), run
// in place through registers: every thread pulls the inputs of all its
// butterflies into VGPRs, the workgroup meets at a barrier, then the outputs go
// back to the same LDS array in autosorted positions.  Twiddles W_N^k come from
// an N-entry table in global memory (built in double precision on the host;
// it lives in L1/L2 after the first segment).
//
// This is the coverage kernel: every size and every chain the path needs runs
// through it.  The headline 4096-point Welch uses the register-resident
// radix-16 kernel in welch4096.hip instead.
#pragma once
#include <hip/hip_runtime.h>

namespace oth {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// WelchArgs.pilot / SegArgs.pilot of one stream as a wave-uniform value (two scalar registers; zero without a table).
// The pilot of a stream: the average of its eight probe means (oth_internal.h: kPilotProbes), added in one fixed order so
// that every wave of the launch holds the same bits.  (Round 5 tried the MEDIAN of the probe means, so that one probe
// inside a transient would not pull the pilot away from the other segments: over four seeds of
// test_pilot_under_a_transient_and_a_drifting_offset the mean read 4.7-6.3e-5 on the 3000-sigma opening transient and the
// median 4.8e-5 ... 1.3e-4 - what decides there is the float32 rounding of the one or two segments that hold the step,
// which no constant removes; the exact time-domain builds read 3.9e-5 ... 1.1e-4 on the same inputs.)
__device__ __forceinline__ float2 pilot_of_probes(const float2 (&q)[8]) {
    const float2 a = cadd(cadd(q[0], q[1]), cadd(q[2], q[3])), b = cadd(cadd(q[4], q[5]), cadd(q[6], q[7]));
    float2 pv = make_float2((a.x + b.x) * 0.125f, (a.y + b.y) * 0.125f);
    pv.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pv.x)));
    pv.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pv.y)));
    return pv;
}
__device__ __forceinline__ float2 load_pilot(const float2 *pilot, int idx) {
    if (!pilot) return make_float2(0.f, 0.f);
    const float2 *src = pilot + 8 * idx;
    const float2 q[8] = {src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]};
    return pilot_of_probes(q);
}
// multiply by -i (forward) / +i (inverse)
template <bool INV> __device__ __forceinline__ float2 rot90(float2 a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <bool INV> __device__ __forceinline__ void dft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3) {
    float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = rot90<INV>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }

// One Stockham pass of radix R over sub-transform length NS.
template <int N, int T, int R, int NS, bool INV>
__device__ __forceinline__ void stockham_pass(float2 *buf, const float2 *__restrict__ tw, int tid) {
    constexpr int NBF = N / R;                 // butterflies in the pass
    constexpr int NB = (NBF + T - 1) / T;      // per thread
    float2 v[NB][R];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int j = tid + q * T;
        if (j < NBF) {
#pragma unroll
            for (int m = 0; m < R; ++m) v[q][m] = buf[j + m * NBF];
            if (NS > 1) {
                const int k = (j & (NS - 1)) * (N / (NS * R));
#pragma unroll
                for (int m = 1; m < R; ++m) {
                    float2 w = tw[m * k];
                    if (INV) w.y = -w.y;
                    v[q][m] = cmul(v[q][m], w);
                }
            }
            if (R == 4) {
                dft4<INV>(v[q][0], v[q][1], v[q][2], v[q][3]);
            } else {
                float2 a = v[q][0], b = v[q][1];
                v[q][0] = cadd(a, b);
                v[q][1] = csub(a, b);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int j = tid + q * T;
        if (j < NBF) {
            const int j0 = (j / NS) * NS * R + (j & (NS - 1));
#pragma unroll
            for (int m = 0; m < R; ++m) buf[j0 + m * NS] = v[q][m];
        }
    }
    __syncthreads();
}

template <int N, int T, int NS, bool INV> struct StockhamChain {
    static __device__ __forceinline__ void run(float2 *buf, const float2 *__restrict__ tw, int tid) {
        if constexpr (NS < N) {
            stockham_pass<N, T, 4, NS, INV>(buf, tw, tid);
            StockhamChain<N, T, NS * 4, INV>::run(buf, tw, tid);
        }
    }
};

// buf[0..N) natural order in, natural order out.  Caller has synchronised the
// workgroup after filling buf; on return every thread may read any bin.
template <int N, int T, bool INV = false>
__device__ __forceinline__ void fft_lds(float2 *buf, const float2 *__restrict__ tw, int tid) {
    if constexpr (ilog2c(N) & 1) {
        stockham_pass<N, T, 2, 1, INV>(buf, tw, tid);
        StockhamChain<N, T, 2, INV>::run(buf, tw, tid);
    } else {
        StockhamChain<N, T, 1, INV>::run(buf, tw, tid);
    }
}

// threads per workgroup used by the generic kernels for size N
constexpr int generic_threads(int n) { return n >= 16384 ? 1024 : (n >= 8192 ? 512 : (n >= 1024 ? 256 : 64)); }

}  // namespace oth
