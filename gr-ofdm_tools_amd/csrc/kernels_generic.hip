// Coverage kernels: segment-averaged |X|^2 (Welch), cross spectrum (CSD),
// per-vector periodogram rows (the GNU Radio chains) and xcorr for every
// power-of-two size 64..16384, built on the LDS Stockham FFT of fft_lds.hip.h.
//
// Reference arithmetic these replace:
//   scipy.signal.welch    ofdm_cr_tools.py:214,322,342 ; spectrum_sweeper.py:263
//   fft_vcc + c2mag[2]    spectrum_sensor_v2.py:90-93 ; psd_logger.py:48-53 ; local_worker.py:63-65
//   xcorr / fac           ofdm_cr_tools.py:155-166
#include "fft_lds.hip.h"
#include "oth_internal.h"

namespace oth {

__device__ __forceinline__ float2 wave_sum(float2 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v.x += __shfl_xor(v.x, off, 64);
        v.y += __shfl_xor(v.y, off, 64);
    }
    return v;
}

// Sum of v over the workgroup; `red` has T/64 slots and is free for reuse after
// the caller's next barrier.
template <int T> __device__ __forceinline__ float2 block_sum(float2 v, float2 *red, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float2 s = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < T / 64; ++w) s = cadd(s, red[w]);
    return s;
}

// Load one segment (zero-padded past nperseg), remove its mean if asked,
// apply the window and leave it in buf.  Ends with a barrier.
template <int N, int T>
__device__ __forceinline__ void stage_segment(const float2 *__restrict__ xs, const float *__restrict__ win,
                                              int nperseg, int detrend, float2 pilot, float2 *buf, float2 *red, int tid) {
    constexpr int NQ = N / T;
    float2 v[NQ];
    float2 sum = make_float2(0.f, 0.f);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int n = tid + q * T;
        v[q] = (n < nperseg) ? csub(xs[n], pilot) : make_float2(0.f, 0.f);      // pilot: WelchArgs.pilot, or zero (x - 0 is exact)
        sum = cadd(sum, v[q]);
    }
    float2 mean = make_float2(0.f, 0.f);
    if (detrend) {
        const float2 tot = block_sum<T>(sum, red, tid);
        const float inv = 1.0f / (float)nperseg;
        mean = make_float2(tot.x * inv, tot.y * inv);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int n = tid + q * T;
        float2 o = make_float2(0.f, 0.f);
        if (n < nperseg) {
            const float w = win[n];
            o = make_float2((v[q].x - mean.x) * w, (v[q].y - mean.y) * w);
        }
        buf[n] = o;
    }
    __syncthreads();
}

template <int N, int T, bool CSD>
__global__ __launch_bounds__(T) void welch_generic_kernel(WelchArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *buf = reinterpret_cast<float2 *>(smem);
    float2 *red = buf + N;
    constexpr int NQ = N / T;
    constexpr int NCH = CSD ? 4 : 1;
    const int tid = threadIdx.x;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const long long s0 = (p.nseg * wg) / W, s1 = (p.nseg * (wg + 1)) / W;
    const float2 *xb = p.x + (size_t)stream * p.stream_stride;
    const float2 *yb = CSD ? p.y + (size_t)stream * p.stream_stride : nullptr;

    float acc[NCH][NQ];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[c][q] = 0.f;

    const float2 px = load_pilot(p.detrend ? p.pilot : nullptr, stream);
    const float2 py = load_pilot(CSD && p.detrend ? p.pilot : nullptr, p.nstreams + stream);
    for (long long s = s0; s < s1; ++s) {
        stage_segment<N, T>(xb + s * p.step, p.win, p.nperseg, p.detrend, px, buf, red, tid);
        fft_lds<N, T>(buf, p.tw, tid);
        if constexpr (!CSD) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float2 X = buf[tid + q * T];
                acc[0][q] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[0][q]));
            }
            __syncthreads();
        } else {
            float2 X[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) X[q] = buf[tid + q * T];
            __syncthreads();
            stage_segment<N, T>(yb + s * p.step, p.win, p.nperseg, p.detrend, py, buf, red, tid);
            fft_lds<N, T>(buf, p.tw, tid);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float2 Y = buf[tid + q * T];
                acc[0][q] = fmaf(X[q].x, X[q].x, fmaf(X[q].y, X[q].y, acc[0][q]));
                acc[1][q] = fmaf(Y.x, Y.x, fmaf(Y.y, Y.y, acc[1][q]));
                // conj(X) * Y
                acc[2][q] = fmaf(X[q].x, Y.x, fmaf(X[q].y, Y.y, acc[2][q]));
                acc[3][q] = fmaf(X[q].x, Y.y, fmaf(-X[q].y, Y.x, acc[3][q]));
            }
            __syncthreads();
        }
    }
    float *dst = p.partial + ((size_t)stream * W + wg) * NCH * N;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < NQ; ++q) dst[c * N + tid + q * T] = acc[c][q];
}

template <int N, int T> __global__ __launch_bounds__(T) void pgram_kernel(PgramArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *buf = reinterpret_cast<float2 *>(smem);
    float2 *red = buf + N;
    constexpr int NQ = N / T;
    const int tid = threadIdx.x;
    for (long long r = blockIdx.x; r < p.nrows; r += gridDim.x) {
        const float2 *xs = p.x + (size_t)(p.first_vec + r * p.keep_n) * N;
        stage_segment<N, T>(xs, p.win, N, 0, make_float2(0.f, 0.f), buf, red, tid);
        fft_lds<N, T>(buf, p.tw, tid);
        float *row = p.rows + (size_t)r * N;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int k = tid + q * T;
            const float2 X = buf[k];
            const float m2 = fmaf(X.x, X.x, X.y * X.y);
            const float v = (p.epilogue == 0) ? sqrtf(m2) : m2 * p.scale;
            row[p.fftshift ? ((k + N / 2) & (N - 1)) : k] = v;
        }
        __syncthreads();
    }
}

// mode 0: xcorr  out[i] = |ifft(fft(b) * conj(fft(a)))[i]|,           i < L/2
// mode 1: fac    out[i] = |fft(|fft(a)|)[i]|,                          i < L/2
// (the reference's fftshift(...)[L/2:] of an even-length array is the first half)
template <int N, int T> __global__ __launch_bounds__(T) void xcorr_kernel(const float2 *a, const float2 *b,
                                                                           const float2 *tw, float *out, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *buf = reinterpret_cast<float2 *>(smem);
    constexpr int NQ = N / T;
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NQ; ++q) buf[tid + q * T] = a[tid + q * T];
    __syncthreads();
    fft_lds<N, T>(buf, tw, tid);
    float2 E[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) E[q] = buf[tid + q * T];
    __syncthreads();
    if (mode == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) buf[tid + q * T] = b[tid + q * T];
        __syncthreads();
        fft_lds<N, T>(buf, tw, tid);
        float2 G[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float2 F = buf[tid + q * T];
            G[q] = make_float2(fmaf(F.x, E[q].x, F.y * E[q].y), fmaf(F.y, E[q].x, -F.x * E[q].y));
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) buf[tid + q * T] = G[q];
        __syncthreads();
        fft_lds<N, T, true>(buf, tw, tid);
        const float inv = 1.0f / (float)N;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int k = tid + q * T;
            if (k < N / 2) out[k] = sqrtf(fmaf(buf[k].x, buf[k].x, buf[k].y * buf[k].y)) * inv;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            buf[tid + q * T] = make_float2(sqrtf(fmaf(E[q].x, E[q].x, E[q].y * E[q].y)), 0.f);
        __syncthreads();
        fft_lds<N, T>(buf, tw, tid);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int k = tid + q * T;
            if (k < N / 2) out[k] = sqrtf(fmaf(buf[k].x, buf[k].x, buf[k].y * buf[k].y));
        }
    }
}

bool generic_supported(int nfft) { return nfft >= 64 && nfft <= 16384 && (nfft & (nfft - 1)) == 0; }
int generic_threads_for(int nfft) { return generic_threads(nfft); }
size_t generic_lds_bytes(int nfft) { return (size_t)nfft * sizeof(float2) + 16 * sizeof(float2); }

template <typename K> static hipError_t allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

#define OTH_FOR_EACH_N(X) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096) X(8192) X(16384)

hipError_t launch_welch_generic(int nfft, const WelchArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    const size_t lds = generic_lds_bytes(nfft);
    hipError_t e;
    switch (nfft) {
#define X(N)                                                                                       \
    case N: {                                                                                      \
        constexpr int T = generic_threads(N);                                                      \
        if (a.y) {                                                                                 \
            if ((e = allow_lds(welch_generic_kernel<N, T, true>, lds)) != hipSuccess) return e;   \
            hipLaunchKernelGGL((welch_generic_kernel<N, T, true>), grid, dim3(T), lds, s, a);     \
        } else {                                                                                   \
            if ((e = allow_lds(welch_generic_kernel<N, T, false>, lds)) != hipSuccess) return e;  \
            hipLaunchKernelGGL((welch_generic_kernel<N, T, false>), grid, dim3(T), lds, s, a);    \
        }                                                                                          \
        break;                                                                                     \
    }
        OTH_FOR_EACH_N(X)
#undef X
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_pgram(int nfft, const PgramArgs &a, hipStream_t s) {
    const size_t lds = generic_lds_bytes(nfft);
    const int grid = (int)(a.nrows < 4096 ? a.nrows : 4096);
    hipError_t e;
    switch (nfft) {
#define X(N)                                                                          \
    case N: {                                                                         \
        constexpr int T = generic_threads(N);                                         \
        if ((e = allow_lds(pgram_kernel<N, T>, lds)) != hipSuccess) return e;        \
        hipLaunchKernelGGL((pgram_kernel<N, T>), dim3(grid), dim3(T), lds, s, a);    \
        break;                                                                        \
    }
        OTH_FOR_EACH_N(X)
#undef X
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_xcorr(int L, const float2 *a, const float2 *b, const float2 *tw, float *out, int mode,
                        hipStream_t s) {
    const size_t lds = generic_lds_bytes(L);
    hipError_t e;
    switch (L) {
#define X(N)                                                                                         \
    case N: {                                                                                        \
        constexpr int T = generic_threads(N);                                                        \
        if ((e = allow_lds(xcorr_kernel<N, T>, lds)) != hipSuccess) return e;                       \
        hipLaunchKernelGGL((xcorr_kernel<N, T>), dim3(1), dim3(T), lds, s, a, b, tw, out, mode);    \
        break;                                                                                       \
    }
        OTH_FOR_EACH_N(X)
#undef X
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace oth
