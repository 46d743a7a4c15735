// welch16k: segment-averaged |FFT_16384(detrend(x) * w)|^2 for nperseg = nfft = 16384
// (BASELINE config 5: multichannel_scanner, 64 channel streams x 16384-point PSD,
// python/multichannel_scanner.py:78-86 chain averaged over the kept vectors; also any
// scipy.signal.welch call with nperseg = nfft = 16384).
//
// 16384 = 4 x 4096, decimation in frequency.  One 1024-thread workgroup per segment, 16 points
// per thread:
//
//   pass 0  n = 4096 a' + m: thread tid holds a' = 0..3 for m = tid + 1024 j (j = 0..3), radix-4 over
//           a' -> k' = k mod 4, times W16384^(k' m); scattered to the LDS region of sub-FFT k'
//   then    the 256 threads tid >> 8 == k' run the radix-16 x 16 x 16 scheme of welch4096.hip on
//           their 4096 points (same LDS image, inside region k') -> bins k = k' + 4 q.
//
// LDS: 4 regions of 16 x 272 float2 (139 KiB) + the shared W256 table; one workgroup per CU,
// 16 waves = 4 per SIMD at <= 128 VGPRs.  Four workgroup barriers per segment.
#include "fft4096.hip.h"

#ifndef OTH_16K_NT
#define OTH_16K_NT 1           // segments do not overlap: every sample is read once
#endif
#if OTH_16K_NT
#define OTH_16K_LOAD(p) load_once(p)
#else
#define OTH_16K_LOAD(p) (*(p))
#endif

namespace oth {
namespace {

constexpr int T16 = 1024;
constexpr int REGION = 16 * RS;                     // float2 per sub-FFT image
constexpr int LDS16_RED = 32;                       // 16 wave sums + ticket
constexpr size_t LDS16_BYTES = (4 * REGION + LDS16_RED) * sizeof(float2);

// multiply by exp(-2 pi i q / 16), q a compile-time constant 0..9 (the products j * k' that occur)
template <int Q> __device__ __forceinline__ float2 mul_w16(float2 a) {
    if constexpr (Q == 0) return a;
    else if constexpr (Q == 1) return mul_w1(a);
    else if constexpr (Q == 2) return mul_w2(a);
    else if constexpr (Q == 3) return mul_w3(a);
    else if constexpr (Q == 4) return mul_w4(a);
    else if constexpr (Q == 6) return mul_w6(a);
    else return mul_w9(a);
}

template <int J> __device__ __forceinline__ void pass0_scatter(float2 (&v)[16], const float2 (&wt)[4], float2 *l0) {
    // v[4 J + a'] -> radix-4 over a' -> k' = 0..3, times W16384^(k' (1024 J + tid)) = wt[k'] * W16^(J k')
    dft4<false>(v[4 * J], v[4 * J + 1], v[4 * J + 2], v[4 * J + 3]);
    l0[0 * REGION + 1024 * J] = v[4 * J];
    l0[1 * REGION + 1024 * J] = mul_w16<J>(cmul(v[4 * J + 1], wt[1]));
    l0[2 * REGION + 1024 * J] = mul_w16<2 * J>(cmul(v[4 * J + 2], wt[2]));
    l0[3 * REGION + 1024 * J] = mul_w16<3 * J>(cmul(v[4 * J + 3], wt[3]));
}

template <bool DETREND>
__global__ __launch_bounds__(T16) void welch16k_kernel(WelchArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    float2 *red = lds + 4 * REGION;

    const int tid = threadIdx.x;
    const int kp = tid >> 8, t = tid & 255;          // sub-FFT k', thread inside it
    const int hi = t >> 4, lo = t & 15;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const long long s0 = (p.nseg * wg) / W, s1 = (p.nseg * (wg + 1)) / W;
    const float2 *xb = p.x + (size_t)stream * p.stream_stride;

    // thread-constant tables: window for n = 4096 a' + 1024 j + tid (stored at [4 j + a']),
    // W16384^(k' tid) for k' = 1..3, and for the sub-FFT W4096^t, W4096^(4t) = W16384^(4t), W16384^(16t)
    float win[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < 4; ++a) win[4 * j + a] = p.win[4096 * a + 1024 * j + tid];
    float2 wt[4];
#pragma unroll
    for (int k = 1; k < 4; ++k) wt[k] = p.tw[k * tid];
    const float2 b1 = p.tw[4 * t], b4 = p.tw[16 * t];
    const float2 c1 = p.tw[64 * lo], c4 = p.tw[256 * lo];   // W256^c = W16384^(64 c), W256^(4c)

    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;

    float2 *l0 = lds + tid;                           // pass-0 scatter base: element (k', m) at k' REGION + m
    float2 *lx = lds + kp * REGION;                   // this sub-FFT's image
    const int w1 = hi * 17 + lo, r1 = hi * RS + lo, w2 = hi * RS + lo, r2 = hi * RS + lo * 17;

    const int sched = p.sched;
    const long long nchunks = sched ? chunk_count(p) : 1;
    int *lnext = reinterpret_cast<int *>(red + 16);
    unsigned ticket = 0;
    for (long long cur = sched ? wg : 0; cur < nchunks;) {
        long long sb = s0, se = s1;
        if (sched) chunk_range(p, cur, sb, se);
        for (long long s = sb; s < se; ++s) {
            float2 v[16];
            prio_latency();
            const float2 *xs = xb + s * p.step + tid;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < 4; ++a) v[4 * j + a] = OTH_16K_LOAD(xs + 4096 * a + 1024 * j);
            float2 mean = make_float2(0.f, 0.f);
            if (DETREND) {
                float2 sum = v[0];
#pragma unroll
                for (int a = 1; a < 16; ++a) sum = cadd(sum, v[a]);
                sum.x = wave_total(sum.x);
                sum.y = wave_total(sum.y);
                if ((tid & 63) == 0) red[tid >> 6] = sum;
            }
            lds_barrier();   // A0: previous segment's reads are done; red[] visible
            prio_compute();
            if (sched == 2 && tid == 0) {
                if (s == sb) ticket = atomicAdd(p.queue + stream, 1u);
                if (s == se - 1) *lnext = (int)ticket;
            }
            if (DETREND) {
                float2 tot = red[0];
#pragma unroll
                for (int w = 1; w < 16; ++w) tot = cadd(tot, red[w]);
                mean = make_float2(tot.x * (1.0f / 16384.0f), tot.y * (1.0f / 16384.0f));
            }
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] = make_float2((v[a].x - mean.x) * win[a], (v[a].y - mean.y) * win[a]);
            prio_latency();
            pass0_scatter<0>(v, wt, l0);
            pass0_scatter<1>(v, wt, l0);
            pass0_scatter<2>(v, wt, l0);
            pass0_scatter<3>(v, wt, l0);
            lds_barrier();   // B0
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] = lx[256 * a + t];
            lds_barrier();   // A: every thread holds its 16 points, the image may be overwritten
            prio_compute();

            // 4096-point transform of sub-FFT k' (welch4096.hip passes 1..3)
            dft16(v);
            prio_latency();
            scatter_pow16<RS>(v, lx + w1, b1, b4);
            lds_barrier();   // B
            dft16_from_lds<17>(v, lx + r1, [] { prio_compute(); });      // ordered reads, counted waits
            prio_latency();
            wave_lds_sync();
            scatter_pow16<17>(v, lx + w2, c1, c4);
            wave_lds_sync();
            dft16_from_lds<1>(v, lx + r2, [] { prio_compute(); });
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                const float2 X = v[r16(k2)];
                acc[k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[k2]));
            }
        }
        if (sched == 0) break;
        cur = (sched == 1) ? cur + W : (long long)W + *lnext;
    }

    // bin k' + 4 (k0 + 16 k1 + 256 k2) sits at 4096 k' + 16 k0 + k1 + 256 k2 (finalize_kernel layout 2)
    float *dst = p.partial + ((size_t)stream * W + wg) * 16384 + 4096 * kp;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) dst[256 * k2 + t] = acc[k2];
}

}  // namespace

hipError_t launch_welch_tuned16k(const WelchArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    hipError_t e = hipSuccess;
    if (a.detrend) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(welch16k_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS16_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((welch16k_kernel<true>), grid, dim3(T16), LDS16_BYTES, s, a);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(welch16k_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS16_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((welch16k_kernel<false>), grid, dim3(T16), LDS16_BYTES, s, a);
    }
    return hipGetLastError();
}

}  // namespace oth
