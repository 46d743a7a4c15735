// csd4096: two-channel Welch cross spectrum for nperseg = nfft = 4096 (BASELINE config 3).
//
// Semantics of scipy.signal.csd / coherence with the Welch parameters of
// ofdm_cr_tools.py:322,342 (the producer of coherence_detector's first
// input, coherence_detector.py:45).  Per segment the workgroup transforms x, keeps its 16 bins per
// thread in registers, transforms y through the same LDS image, and accumulates
//     Pxx += |X|^2   Pyy += |Y|^2   Pxy += conj(X) Y
// in registers over its chunks of segments.  The FFT is the radix-16 x 16 x 16 scheme of
// welch4096.hip (same LDS image, same twiddle handling as its pipelined build: W^(k0 t) rebuilt
// from W^t and W^(4t)).  Register budget: 64 accumulators + 32 (X) + 32 (data) -> the window
// lives in LDS (16 KiB, conflict-free b32 reads) instead of 16 VGPRs, which keeps the kernel at
// 3 workgroups per CU on both the VGPR and the LDS side (53 KiB each).
#include "fft4096.hip.h"

namespace oth {
namespace {

constexpr int LDS_WIN = 4096;   // floats
constexpr size_t CSD_LDS_BYTES = LDS_BYTES + LDS_WIN * sizeof(float);

template <bool DETREND, bool PILOT = false>
__global__ __launch_bounds__(T4, 3) void csd4096_kernel(WelchArgs p) {
    static_assert(DETREND || !PILOT, "the pilot belongs to the detrend");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lx = reinterpret_cast<float2 *>(smem);
    float2 *red = lx + LDS_X;
    float *lwin = reinterpret_cast<float *>(red + LDS_RED);

    const int t = threadIdx.x;
    const int hi = t >> 4, lo = t & 15;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const long long s0 = (p.nseg * wg) / W, s1 = (p.nseg * (wg + 1)) / W;
    const float2 *xb = p.x + (size_t)stream * p.stream_stride;
    const float2 *yb = p.y + (size_t)stream * p.stream_stride;

#pragma unroll
    for (int a = 0; a < 16; ++a) lwin[256 * a + t] = p.win[256 * a + t];
    const float2 b1 = p.tw[t], b4 = p.tw[4 * t];
    const float2 c1 = p.tw[16 * lo], c4 = p.tw[64 * lo];

    float axx[16], ayy[16], are[16], aim[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) axx[k] = ayy[k] = are[k] = aim[k] = 0.f;

    const int w1 = hi * 17 + lo, r1 = hi * RS + lo, w2 = hi * RS + lo, r2 = hi * RS + lo * 17;
    // PILOT (every detrending plan but OTH_DETREND_CONSTANT_FAST): WelchArgs.pilot of either channel comes off every sample as it arrives
    const float2 px = load_pilot(PILOT ? p.pilot : nullptr, stream);
    const float2 py = load_pilot(PILOT ? p.pilot : nullptr, p.nstreams + stream);

    // One 4096-point transform of the segment at xs; result bins k0 + 16 k1 + 256 k2 in v[r16(k2)].
    auto transform = [&](const float2 *xs, float2 pv, float2(&v)[16]) {      // pv: WelchArgs.pilot of the channel
        prio_latency();
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = PILOT ? csub(xs[256 * a], pv) : xs[256 * a];
        float2 mean = make_float2(0.f, 0.f);
        if (DETREND) {
            // pairwise, like NumPy's float32 mean (section 2 of DESIGN.md: with a DC line far above the signal the order
            // of these adds shows in bins 0, +-1 of a few-segment result)
            float2 t8[8], t4[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) t8[i] = cadd(v[2 * i], v[2 * i + 1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) t4[i] = cadd(t8[2 * i], t8[2 * i + 1]);
            float2 sum = cadd(cadd(t4[0], t4[1]), cadd(t4[2], t4[3]));
            sum.x = wave_total(sum.x);
            sum.y = wave_total(sum.y);
            if ((t & 63) == 0) red[t >> 6] = sum;
        }
        __syncthreads();   // A
        prio_compute();
        if (DETREND) {
            const float2 s01 = cadd(red[0], red[1]), s23 = cadd(red[2], red[3]);
            mean = make_float2((s01.x + s23.x) * (1.0f / 4096.0f), (s01.y + s23.y) * (1.0f / 4096.0f));
        }
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const float w = lwin[256 * a + t];
            v[a] = make_float2((v[a].x - mean.x) * w, (v[a].y - mean.y) * w);
        }
        dft16(v);
        prio_latency();
        scatter_pow16<RS>(v, lx + w1, b1, b4);
        __syncthreads();   // B
        dft16_from_lds<17>(v, lx + r1, [] { prio_compute(); });      // ordered reads, counted waits
        prio_latency();
        wave_lds_sync();
        scatter_pow16<17>(v, lx + w2, c1, c4);
        wave_lds_sync();
        dft16_from_lds<1>(v, lx + r2, [] { prio_compute(); });
    };

    const int sched = p.sched;
    const long long nchunks = sched ? chunk_count(p) : 1;
    int *lnext = reinterpret_cast<int *>(red + 8);
    unsigned ticket = 0;
    for (long long cur = sched ? wg : 0; cur < nchunks;) {
        long long sb = s0, se = s1;
        if (sched) chunk_range(p, cur, sb, se);
        for (long long s = sb; s < se; ++s) {
            float2 X[16], v[16];
            transform(xb + s * p.step + t, px, X);
            if (sched == 2 && t == 0) {   // after barrier B of the x transform, before barrier A of the y one
                if (s == sb) ticket = atomicAdd(p.queue + stream, 1u);
                if (s == se - 1) *lnext = (int)ticket;
            }
            transform(yb + s * p.step + t, py, v);
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                const float2 x = X[r16(k2)], y = v[r16(k2)];
                axx[k2] = fmaf(x.x, x.x, fmaf(x.y, x.y, axx[k2]));
                ayy[k2] = fmaf(y.x, y.x, fmaf(y.y, y.y, ayy[k2]));
                are[k2] = fmaf(x.x, y.x, fmaf(x.y, y.y, are[k2]));      // conj(X) Y
                aim[k2] = fmaf(x.x, y.y, fmaf(-x.y, y.x, aim[k2]));
            }
        }
        if (sched == 0) break;
        cur = (sched == 1) ? cur + W : (long long)W + *lnext;
    }

    // channels xx, yy, re, im; bin k0 + 16 k1 + 256 k2 at t + 256 k2 (finalize_kernel layout 1)
    float *dst = p.partial + ((size_t)stream * W + wg) * 4 * 4096;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        dst[256 * k2 + t] = axx[k2];
        dst[4096 + 256 * k2 + t] = ayy[k2];
        dst[8192 + 256 * k2 + t] = are[k2];
        dst[12288 + 256 * k2 + t] = aim[k2];
    }
}

}  // namespace

int csd4096_blocks_per_cu() {
    static int cached = 0;
    if (cached) return cached;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, csd4096_kernel<true>, T4, CSD_LDS_BYTES) != hipSuccess || n < 1)
        n = 2;
    return cached = n;
}

hipError_t launch_csd_tuned4096(const WelchArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    if (a.detrend && a.pilot)
        hipLaunchKernelGGL((csd4096_kernel<true, true>), grid, dim3(T4), CSD_LDS_BYTES, s, a);
    else if (a.detrend)
        hipLaunchKernelGGL((csd4096_kernel<true>), grid, dim3(T4), CSD_LDS_BYTES, s, a);
    else
        hipLaunchKernelGGL((csd4096_kernel<false>), grid, dim3(T4), CSD_LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace oth
