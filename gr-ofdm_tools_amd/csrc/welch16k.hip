// welch16k: segment-averaged |FFT_N(detrend(x) * w)|^2 for nperseg = nfft = N = 4096 F, F = 4 (16384) or 2 (8192)
// (BASELINE config 5: multichannel_scanner, 64 channel streams x 16384-point PSD,
// python/multichannel_scanner.py:78-86 chain averaged over the kept vectors; also any
// scipy.signal.welch call with nperseg = nfft = 8192 / 16384).
//
// N = F x 4096, decimation in frequency.  One workgroup of T = 256 F threads per segment, 16 points
// per thread:
//
//   pass 0  n = 4096 a' + m: thread tid holds a' = 0..F-1 for m = tid + T j (j = 0..16/F-1), radix-F over
//           a' -> k' = k mod F, times W_N^(k' m); scattered to the LDS region of sub-FFT k'
//   then    the 256 threads tid >> 8 == k' run the radix-16 x 16 x 16 scheme of welch4096.hip on
//           their 4096 points (same LDS image, inside region k') -> bins k = k' + F q.
//
// LDS: F regions of 16 x 272 float2 (139 KiB at F = 4: one workgroup per CU; 70 KiB at F = 2: two);
// 16 waves per CU = 4 per SIMD at <= 128 VGPRs.  Four workgroup barriers per segment.
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

constexpr int REGION = 16 * RS;                     // float2 per sub-FFT image
constexpr int LDS16_RED = 48;                       // up to 16 wave sums of the new half, 16 of a chunk's first half, ticket [32],
                                                    // segment totals by step parity [40..41] (frequency-domain detrend)
template <int F> constexpr size_t lds16_bytes() { return (F * REGION + LDS16_RED) * sizeof(float2); }

// multiply by exp(-2 pi i q / 16), q a compile-time constant (the products j * k' that occur: 0..7 and 9)
template <int Q> __device__ __forceinline__ float2 mul_w16(float2 a) {
    if constexpr (Q == 0) return a;
    else if constexpr (Q == 1) return mul_w1(a);
    else if constexpr (Q == 2) return mul_w2(a);
    else if constexpr (Q == 3) return mul_w3(a);
    else if constexpr (Q == 4) return mul_w4(a);
    else if constexpr (Q == 5) return mul_w4(mul_w1(a));
    else if constexpr (Q == 6) return mul_w6(a);
    else if constexpr (Q == 7) return mul_w4(mul_w3(a));
    else {
        static_assert(Q == 9, "twiddle power not provided");
        return mul_w9(a);
    }
}

// radix-F butterfly over a' of the points v[F J + a'] (m = T J + tid), times W_N^(k' m) = wt[k'] * W16^(J k')
// (T / N = 1 / 16 for both sizes), scattered to element m of region k'
template <int J, int F> __device__ __forceinline__ void pass0_scatter(float2 (&v)[16], const float2 (&wt)[4], float2 *l0) {
    constexpr int T = 256 * F;
    if constexpr (F == 4) {
        dft4<false>(v[4 * J], v[4 * J + 1], v[4 * J + 2], v[4 * J + 3]);
        l0[0 * REGION + T * J] = v[4 * J];
        l0[1 * REGION + T * J] = mul_w16<J>(cmul(v[4 * J + 1], wt[1]));
        l0[2 * REGION + T * J] = mul_w16<2 * J>(cmul(v[4 * J + 2], wt[2]));
        l0[3 * REGION + T * J] = mul_w16<3 * J>(cmul(v[4 * J + 3], wt[3]));
    } else {
        const float2 u0 = v[2 * J], u1 = v[2 * J + 1];
        l0[0 * REGION + T * J] = cadd(u0, u1);
        l0[1 * REGION + T * J] = mul_w16<J>(cmul(csub(u0, u1), wt[1]));
    }
}

// Transform of one segment whose windowed points sit in v[F j + a'] (n = 4096 a' + T j + tid): pass 0 (radix F, scatter to
// the F sub-FFT images), then the 4096-point scheme of welch4096.hip on this thread's sub-FFT k'.  On return
// v[r16(k2)] = X[k' + F (k0 + 16 k1 + 256 k2)] with (k0, k1) = (hi, lo) of t = tid & 255.  The caller has passed
// barrier A0 (the previous segment's exchange reads are done everywhere).
template <int F>
__device__ __forceinline__ void transform16k(float2 (&v)[16], const float2 (&wt)[4], float2 b1, float2 b4, float2 c1, float2 c4,
                                             float2 *l0, float2 *lx, int t, int w1, int r1, int w2, int r2) {
    prio_latency();
    pass0_scatter<0, F>(v, wt, l0);
    pass0_scatter<1, F>(v, wt, l0);
    pass0_scatter<2, F>(v, wt, l0);
    pass0_scatter<3, F>(v, wt, l0);
    if constexpr (F == 2) {
        pass0_scatter<4, F>(v, wt, l0);
        pass0_scatter<5, F>(v, wt, l0);
        pass0_scatter<6, F>(v, wt, l0);
        pass0_scatter<7, F>(v, wt, l0);
    }
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
}

// HALF: step = N / 2 - the second half of a segment is the first half of the next one at the same (j, tid), so it
// is kept (raw) in registers and every sample is read once; otherwise segments are loaded whole (any step).
// DET: 0 none, 1 in the time domain (the mean is subtracted before the window), 2 in the frequency domain
// (X -= mean FFT(w) on the two bins of each thread the window's spectrum reaches, WelchArgs.fd - as welch4096ws.hip;
// the mean is then needed only at the end of the step, which is what lets the 16384-point build keep the
// overlapped half in registers: with DET = 1 it spilled 18 registers and loaded every sample twice).
// PAD: nperseg = N / 4 zero-padded to N (the sweeper's call at these sizes, spectrum_sweeper.py:263): samples only at
// a' = 0, j < 4 - four loads per thread, the rest compile-time zeros (the window array is zero-extended, so the
// detrended, windowed padding is exactly 0 as well); the mean is over nperseg.  Whole-segment loads, any step - or
// PADHALF: step = nperseg / 2 (that call's own overlap), the two values of the overlapped half stay in registers and
// every sample is read once (round 4: HBM traffic 1.9 x -> 1.0 x the algorithmic bytes).
template <int DET, int F, bool HALF, bool PAD = false, bool PILOT = false, bool PADHALF = false>
__global__ __launch_bounds__(256 * F) void welch16k_kernel(WelchArgs p) {
    static_assert(!PAD || (!HALF && DET != 2), "zero-padded build: time-domain detrend");
    static_assert(!PADHALF || PAD, "PADHALF: the zero-padded build at step = nperseg / 2");
    constexpr bool DETREND = DET != 0;
    constexpr int T16 = 256 * F, N = 4096 * F, NJ = 16 / F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    float2 *red = lds + F * REGION;

    const int tid = threadIdx.x;
    const int kp = tid >> 8, t = tid & 255;          // sub-FFT k', thread inside it
    const int hi = t >> 4, lo = t & 15;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const long long s0 = (p.nseg * wg) / W, s1 = (p.nseg * (wg + 1)) / W;
    const float2 *xb = p.x + (size_t)stream * p.stream_stride;

    // thread-constant tables: window for n = 4096 a' + T j + tid (stored at [F j + a']),
    // W_N^(k' tid) for k' = 1..F-1, and for the sub-FFT W4096^t = W_N^(F t), W4096^(4t)
    float win[16];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int a = 0; a < F; ++a) win[F * j + a] = p.win[4096 * a + T16 * j + tid];
    float2 wt[4] = {};
#pragma unroll
    for (int k = 1; k < F; ++k) wt[k] = p.tw[k * tid];
    const float2 b1 = p.tw[F * t], b4 = p.tw[4 * F * t];
    const float2 c1 = p.tw[16 * F * lo], c4 = p.tw[64 * F * lo];   // W256^c = W_N^(16 F c), W256^(4c)

    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;

    float2 *l0 = lds + tid;                           // pass-0 scatter base: element (k', m) at k' REGION + m
    float2 *lx = lds + kp * REGION;                   // this sub-FFT's image
    const int w1 = hi * 17 + lo, r1 = hi * RS + lo, w2 = hi * RS + lo, r2 = hi * RS + lo * 17;

    const int sched = p.sched;
    const long long nchunks = sched ? chunk_count(p) : 1;
    int *lnext = reinterpret_cast<int *>(red + 32);
    unsigned ticket = 0;
    float2 keep[HALF ? 8 : (PADHALF ? 2 : 1)];
    float2 prev_tot = make_float2(0.f, 0.f);
    static_assert(DET != 0 || !PILOT, "the pilot belongs to the detrend");
    // PILOT (every detrending plan but OTH_DETREND_CONSTANT_FAST): WelchArgs.pilot comes off every sample as it arrives
    const float2 pv = load_pilot(PILOT ? p.pilot : nullptr, stream);
    for (long long cur = sched ? wg : 0; cur < nchunks;) {
        long long sb = s0, se = s1;
        if (sched) chunk_range(p, cur, sb, se);
        for (long long s = sb; s < se; ++s) {
            float2 v[16];
            prio_latency();
            const float2 *xs = xb + s * p.step + tid;
            float2 sum = make_float2(0.f, 0.f), sumf = make_float2(0.f, 0.f);
            if constexpr (HALF) {
                constexpr int H = F / 2;
                if (s == sb) {      // the chunk's first segment brings its own first half
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int a = 0; a < H; ++a) {
                            keep[H * j + a] = OTH_16K_LOAD(xs + 4096 * a + T16 * j);
                            if (PILOT) keep[H * j + a] = csub(keep[H * j + a], pv);
                            sumf = cadd(sumf, keep[H * j + a]);
                        }
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int a = 0; a < H; ++a) {
                        float2 r = OTH_16K_LOAD(xs + 4096 * (H + a) + T16 * j);
                        if (PILOT) r = csub(r, pv);
                        v[F * j + a] = keep[H * j + a];
                        v[F * j + H + a] = r;
                        keep[H * j + a] = r;
                        sum = cadd(sum, r);
                    }
            } else {
                if constexpr (PADHALF) {      // rows j = 0, 1 are the previous segment's rows 2, 3
                    if (s == sb) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            keep[j] = OTH_16K_LOAD(xs + T16 * j);
                            if (PILOT) keep[j] = csub(keep[j], pv);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int a = 0; a < F; ++a) {
                        if (PAD && (a != 0 || j >= 4)) v[F * j + a] = make_float2(0.f, 0.f);
                        else if (PADHALF && j < 2) v[F * j + a] = keep[j];
                        else {
                            v[F * j + a] = OTH_16K_LOAD(xs + 4096 * a + T16 * j);
                            if (PILOT) v[F * j + a] = csub(v[F * j + a], pv);      // (the padding zeros stay zeros)
                        }
                    }
                if constexpr (PADHALF) {
                    keep[0] = v[F * 2];
                    keep[1] = v[F * 3];
                }
                if (DETREND) {
                    // pairwise, like NumPy's float32 mean: with a DC line far above the signal the ORDER of the adds is
                    // what separates 1e-4 from 4e-4 in bins 0, +-1 of a few-segment result (the padding zeros add exactly)
                    float2 t8[8], t4[4];
#pragma unroll
                    for (int i = 0; i < 8; ++i) t8[i] = cadd(v[2 * i], v[2 * i + 1]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) t4[i] = cadd(t8[2 * i], t8[2 * i + 1]);
                    sum = cadd(cadd(t4[0], t4[1]), cadd(t4[2], t4[3]));
                }
            }
            float2 mean = make_float2(0.f, 0.f);
            if (DET == 2) {      // windowed before the barrier: the mean comes off in the frequency domain
#pragma unroll
                for (int a = 0; a < 16; ++a) v[a] = make_float2(v[a].x * win[a], v[a].y * win[a]);
            }
            // DET = 2 (HALF builds): the per-wave sums of the new half go to the slot of this step's parity, the other
            // slot still holds the half before it (a chunk's first step fills both); after the barrier wave 0 adds the
            // 32 values up and leaves the segment total in LDS, where every thread picks it up at the END of the step.
            // Nothing of the detrend stays in registers across the transform.
            const int par = (int)(s & 1);
            if (DET == 2) {
                sum.x = wave_total_lane63(sum.x);
                sum.y = wave_total_lane63(sum.y);
                if (s == sb) {
                    sumf.x = wave_total_lane63(sumf.x);
                    sumf.y = wave_total_lane63(sumf.y);
                }
                if ((tid & 63) == 63) {
                    red[16 * par + (tid >> 6)] = sum;
                    if (s == sb) red[16 * (par ^ 1) + (tid >> 6)] = sumf;
                }
            } else if (DETREND) {
                sum.x = wave_total(sum.x);
                sum.y = wave_total(sum.y);
                if (HALF && s == sb) {
                    sumf.x = wave_total(sumf.x);
                    sumf.y = wave_total(sumf.y);
                }
                if ((tid & 63) == 0) {
                    red[tid >> 6] = sum;
                    if (HALF && s == sb) red[16 + (tid >> 6)] = sumf;
                }
            }
            lds_barrier();   // A0: previous segment's reads are done; red[] visible
            prio_compute();
            if (sched == 2 && tid == 0) {
                if (s == sb) ticket = atomicAdd(p.queue + stream, 1u);
                if (s == se - 1) *lnext = (int)ticket;
            }
            if (DET == 2 && tid < 64) {      // wave 0: slots 0 .. T16/64-1 and 16 .. 16+T16/64-1 hold this segment's halves
                float2 part = make_float2(0.f, 0.f);
                if ((tid & 15) < T16 / 64 && tid < 32) part = red[tid];
                part.x = wave_total_lane63(part.x);
                part.y = wave_total_lane63(part.y);
                if (tid == 63) red[40 + par] = part;      // read behind the barriers of the transform
            }
            if (DET == 1) {
                float2 tot;      // the T16 / 64 wave sums, pairwise as well
                {
                    constexpr int NW = T16 / 64;
                    float2 t[NW];
#pragma unroll
                    for (int w = 0; w < NW; ++w) t[w] = red[w];
#pragma unroll
                    for (int n = NW / 2; n >= 1; n /= 2)
#pragma unroll
                        for (int w = 0; w < n; ++w) t[w] = cadd(t[2 * w], t[2 * w + 1]);
                    tot = t[0];
                }
                if (HALF) {
                    if (s == sb) {
                        float2 ft = red[16];
#pragma unroll
                        for (int w = 1; w < T16 / 64; ++w) ft = cadd(ft, red[16 + w]);
                        prev_tot = ft;
                    }
                    const float2 both = cadd(prev_tot, tot);
                    prev_tot = tot;
                    tot = both;
                }
                constexpr float inv = 1.0f / (PAD ? N / 4 : N);      // mean over nperseg
                mean = make_float2(tot.x * inv, tot.y * inv);
            }
            if (DET != 2) {
#pragma unroll
                for (int a = 0; a < 16; ++a) v[a] = make_float2((v[a].x - mean.x) * win[a], (v[a].y - mean.y) * win[a]);
            }
            transform16k<F>(v, wt, b1, b4, c1, c4, l0, lx, t, w1, r1, w2, r2);
            if (DET == 2) {
                const float4 fw = p.fd[tid];      // FFT(w) at this thread's k2 = 0 and k2 = 15 bins (zero where out of reach)
                const float2 tot = red[40 + par];
                mean = make_float2(tot.x * (1.0f / N), tot.y * (1.0f / N));
                v[r16(0)] = make_float2(v[r16(0)].x - (mean.x * fw.x - mean.y * fw.y), v[r16(0)].y - (mean.x * fw.y + mean.y * fw.x));
                v[r16(15)] = make_float2(v[r16(15)].x - (mean.x * fw.z - mean.y * fw.w),
                                         v[r16(15)].y - (mean.x * fw.w + mean.y * fw.z));
            }
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                const float2 X = v[r16(k2)];
                acc[k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[k2]));
            }
        }
        if (sched == 0) break;
        cur = (sched == 1) ? cur + W : (long long)W + *lnext;
    }

    // bin k' + F (k0 + 16 k1 + 256 k2) sits at 4096 k' + 16 k0 + k1 + 256 k2 (finalize_kernel layout 2 / 3)
    float *dst = p.partial + ((size_t)stream * W + wg) * N + 4096 * kp;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) dst[256 * k2 + t] = acc[k2];
}

// ---------------------------------------------------------------------------------------------------------
// chain16k: the fused periodogram chain (segfft.hip's CHAIN build) at 8192 / 16384 points - multichannel_scanner /
// spectrum_sensor_v2 as a streaming block at BASELINE config 5's size (python/multichannel_scanner.py:78-86 takes any
// fft_len).  Kept vectors are segments with step = keep_n N; window, shift, |X| or |X|^2 [x 1/N^2], IIR weighted sum /
// peak max accumulated per workgroup (closed by chain_reduce / chain_state, kernels_misc.hip; partial rows in this
// kernel's own bin order, finalize layout 2 / 3), rows stored only for
// s >= store_from.  The next segment is prefetched while this one is transformed (one workgroup per CU at 16384
// points: without it all sixteen waves wait for their loads at the same barrier).
// WINDOW = false: rectangular (the v2 / scanner chain passes `()` as its window, spectrum_sensor_v2.py:90) - no window
// registers, which is what lets the prefetch fit under 128 VGPRs at 16384 points.  PREFETCH = false: the windowed
// 16384-point build (psd_logger / local_worker at that size) loads at the top of the step instead.
enum { C16_WSUM = 1, C16_MAX = 2 };
template <int F> constexpr size_t chain16k_lds_bytes() { return lds16_bytes<F>() + (256 + 16) * sizeof(float4); }
template <int F, bool WINDOW, bool PREFETCH>
__global__ __launch_bounds__(256 * F, 4) void chain16k_kernel(SegArgs p) {
    constexpr int T16 = 256 * F, N = 4096 * F, NJ = 16 / F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    const int kp = tid >> 8, t = tid & 255;
    const int hi = t >> 4, lo = t & 15;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const float2 *xb = p.x + (size_t)stream * p.stream_stride + p.first + tid;

    float win[WINDOW ? 16 : 1];
    if (WINDOW) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int a = 0; a < F; ++a) win[F * j + a] = p.win[4096 * a + T16 * j + tid];
    }
    float2 wt[4] = {};
#pragma unroll
    for (int k = 1; k < F; ++k) wt[k] = p.tw[k * tid];
    // the twiddle seeds of the 4096-point passes live in a small LDS table, read per segment: eight registers less,
    // which is what keeps the prefetching build out of scratch memory
    float4 *tb = reinterpret_cast<float4 *>(lds + F * REGION + LDS16_RED);      // [256]: W^(F t), W^(4 F t); [16]: W256^c, W256^(4c)
    if (tid < 256) {
        const float2 u = p.tw[F * tid], w = p.tw[4 * F * tid];
        tb[tid] = make_float4(u.x, u.y, w.x, w.y);
    }
    if (tid < 16) {
        const float2 u = p.tw[16 * F * tid], w = p.tw[64 * F * tid];
        tb[256 + tid] = make_float4(u.x, u.y, w.x, w.y);
    }
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    float2 *l0 = lds + tid, *lx = lds + kp * REGION;
    const int w1 = hi * 17 + lo, r1 = hi * RS + lo, w2 = hi * RS + lo, r2 = hi * RS + lo * 17;
    const int kbase = kp + F * (hi + 16 * lo);      // bin of v[r16(k2)]: kbase + 256 F k2

    const int sched = p.sched;
    const long long nchunks = sched ? chunk_count_of(p.nseg, p.nbig, p.chunk, p.tail_chunk) : 1;
    const long long s0 = (p.nseg * wg) / W, s1 = (p.nseg * (wg + 1)) / W;
    auto load_segment = [&](float2(&d)[16], long long s) {
        const float2 *xs = xb + s * p.step;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int a = 0; a < F; ++a) d[F * j + a] = OTH_16K_LOAD(xs + 4096 * a + T16 * j);
    };
    float2 nxt[PREFETCH ? 16 : 1];
    bool primed = false;
    for (long long cur = sched ? wg : 0; cur < nchunks;) {
        long long sb = s0, se = s1;
        if (sched) chunk_range_of(p.nseg, p.nbig, p.chunk, p.tail_chunk, cur, sb, se);
        long long sb_next = -1;
        if (PREFETCH && sched && cur + W < nchunks) {
            long long se_next;
            chunk_range_of(p.nseg, p.nbig, p.chunk, p.tail_chunk, cur + W, sb_next, se_next);
        }
        if constexpr (PREFETCH) {
            if (sb < se && !primed) load_segment(nxt, sb);
        }
        primed = sb_next >= 0;
        for (long long s = sb; s < se; ++s) {
            float2 v[16];
            prio_latency();
            if constexpr (PREFETCH) {
#pragma unroll
                for (int a = 0; a < 16; ++a) v[a] = nxt[a];
                // unconditional prefetch (a conditional one turns nxt into a phi: 32 copies per segment): the chunk's
                // last segment fetches the next chunk's first one, or itself again (valid, L2-resident)
                load_segment(nxt, s + 1 < se ? s + 1 : (sb_next >= 0 ? sb_next : s));
            } else {
                load_segment(v, s);
            }
            if (WINDOW) {
#pragma unroll
                for (int a = 0; a < 16; ++a) v[a] = make_float2(v[a].x * win[a], v[a].y * win[a]);
            }
            lds_barrier();   // A0: the previous segment's exchange reads are done everywhere
            const float4 bb = tb[t], cc = tb[256 + lo];
            transform16k<F>(v, wt, make_float2(bb.x, bb.y), make_float2(bb.z, bb.w), make_float2(cc.x, cc.y),
                            make_float2(cc.z, cc.w), l0, lx, t, w1, r1, w2, r2);
            const bool st = s >= p.store_from, ac = s < p.acc_end;
            float val[16];
            if (p.epilogue == 0) {
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) {
                    const float2 X = v[r16(k2)];
                    val[k2] = __builtin_amdgcn_sqrtf(fmaf(X.x, X.x, X.y * X.y));
                }
            } else {
                const float sc = p.scale;
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) {
                    const float2 X = v[r16(k2)];
                    val[k2] = fmaf(X.x, X.x, X.y * X.y) * sc;
                }
            }
            if (st) {
                float *row = p.rows + ((size_t)stream * (p.nseg - p.store_from) + (size_t)(s - p.store_from)) * N;
                const int sh = p.fftshift ? N / 2 : 0;
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) row[(kbase + 256 * F * k2 + sh) & (N - 1)] = val[k2];
            }
            if (ac) {
                if (p.acc_mode == C16_WSUM) {
                    const long long kk = p.acc_end - 1 - s;
                    const float w = kk == 0 ? 1.0f : exp2f(p.l2 * (float)kk);
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[k] = fmaf(w, val[k], acc[k]);
                } else if (p.acc_mode == C16_MAX) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm("v_max_f32 %0, %1, %2" : "=v"(acc[k]) : "v"(acc[k]), "v"(val[k]));
                }
            }
        }
        if (sched == 0) break;
        cur += W;
    }
    if (p.partial) {
        // the kernel's own order, contiguous per wave (bin k' + F (k0 + 16 k1 + 256 k2) at 4096 k' + 16 k0 + k1 + 256 k2:
        // layout 2 / 3 of bin_pos, un-permuted by chain_state_kernel).  Natural-order 4-byte stores at a stride of 64
        // floats cost 4 x their bytes in HBM writes at 16384 points (profiles: 70 MB against 16.8).
        float *dst = p.partial + ((size_t)stream * W + wg) * N + 4096 * kp;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) dst[256 * k2 + t] = acc[k2];
    }
}

}  // namespace

template <int F, bool WINDOW, bool PREFETCH> hipError_t launch_chain16k_f(const SegArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    constexpr size_t lds = chain16k_lds_bytes<F>();
    const void *fn = reinterpret_cast<const void *>(chain16k_kernel<F, WINDOW, PREFETCH>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((chain16k_kernel<F, WINDOW, PREFETCH>), grid, dim3(256 * F), lds, s, a);
    return hipGetLastError();
}

// rect: the chain's window is all ones (a.win is still valid)
hipError_t launch_chain16k(int nfft, const SegArgs &a, bool rect, hipStream_t s) {
    if (nfft == 16384) return rect ? launch_chain16k_f<4, false, true>(a, s) : launch_chain16k_f<4, true, false>(a, s);
    if (nfft == 8192) return rect ? launch_chain16k_f<2, false, true>(a, s) : launch_chain16k_f<2, true, false>(a, s);
    return hipErrorInvalidValue;
}

template <int DET, int F, bool HALF, bool PAD, bool PILOT, bool PADHALF> hipError_t launch16k_p(const WelchArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    constexpr size_t lds = lds16_bytes<F>();
    const void *fn = reinterpret_cast<const void *>(welch16k_kernel<DET, F, HALF, PAD, PILOT, PADHALF>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((welch16k_kernel<DET, F, HALF, PAD, PILOT, PADHALF>), grid, dim3(256 * F), lds, s, a);
    return hipGetLastError();
}
template <int DET, int F, bool HALF, bool PAD = false, bool PADHALF = false> hipError_t launch16k(const WelchArgs &a, hipStream_t s) {
    if constexpr (DET != 0)
        if (a.pilot) return launch16k_p<DET, F, HALF, PAD, true, PADHALF>(a, s);
    return launch16k_p<DET, F, HALF, PAD, false, PADHALF>(a, s);
}

template <int F> hipError_t launch16k_f(const WelchArgs &a, hipStream_t s) {
    if (a.nperseg * 4 == 4096 * F) {      // zero-padded segments; at that call's own 50 % overlap the shared half is kept
        if (a.step * 2 == a.nperseg)
            return a.detrend ? launch16k<1, F, false, true, true>(a, s) : launch16k<0, F, false, true, true>(a, s);
        return a.detrend ? launch16k<1, F, false, true>(a, s) : launch16k<0, F, false, true>(a, s);
    }
    // 50 % overlap: the overlapped half stays in registers.  With a detrend that needs the frequency-domain form at
    // 16384 points (a.fd: the window's spectrum is confined); a window without the table loads whole segments there.
    if (a.step == 2048 * F) {
        if (!a.detrend) return launch16k<0, F, true>(a, s);
        if (a.fd) return launch16k<2, F, true>(a, s);
        if constexpr (F == 2) return launch16k<1, F, true>(a, s);      // (constexpr: the 16384-point build of this form spilled 18 registers and was never launched)
    }
    return a.detrend ? launch16k<1, F, false>(a, s) : launch16k<0, F, false>(a, s);
}

hipError_t launch_welch_tuned16k(int nfft, const WelchArgs &a, hipStream_t s) {
    if (nfft == 16384) return launch16k_f<4>(a, s);
    if (nfft == 8192) return launch16k_f<2>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace oth
