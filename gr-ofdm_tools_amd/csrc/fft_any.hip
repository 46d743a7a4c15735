// Transforms of ANY length (round 6): the coverage engine behind every size the tuned and the radix-4 coverage kernels do
// not take - lengths that are not a power of two, and powers of two above 16384 (or below 64).
//
// The reference puts no limit on the transform length: fft.fft_vcc(self.fft_len, ...) (psd_logger.py:48,
// spectrum_sensor_v2.py:90, local_worker.py:62-63), sg.welch(..., nperseg=nFFT, nfft=nFFT) (ofdm_cr_tools.py:214,322,342),
// np.fft.fft(..., nFFT) (ofdm_cr_tools.py:177, 157-160), and fast_spectrum_scan chooses nFFT = 2^ceil(log2(npts)) itself
// (ofdm_cr_tools.py:474-475); the web gateway takes --nfft as a free integer (sdr_webserver/local_hw_gateway.py:284-285).
//
// One kernel, any_fft_kernel, is a workgroup-cooperative mixed-radix (16, 8, 4, 2, 3, 5, 7) Stockham autosort FFT in LDS
// over a TILE of C adjacent transforms of length n (n C <= 16384 points: 128 KiB of LDS), with the radix list, the
// sub-transform lengths and the twiddle stride as launch arguments; what it does before, between and after the passes is
// chosen per launch:
//
//   load    plain (a tile of the workspace) | stage (samples of a segment: detrend, window, Bluestein chirp, zero padding)
//   mid     none | Bluestein (v = conj(v B[k]), then the same transform again)
//   store   plain (+ four-step twiddle W_L^(a b)) | accumulate |X|^2 (or the four two-channel sums) over the workgroup's
//           segments | periodogram rows (the GNU Radio chains' epilogues)
//
// From these the host builds four routes (any_describe()):
//   direct      n = nfft (2, 3, 5, 7-smooth, <= 16384): one launch, a workgroup per segment, nothing leaves LDS.
//   two-level   L = L1 L2 (powers of two 32768 ... 1048576, L2 = 256): the four-step form.  K1 stages tiles of C columns
//               (stride L2), transforms along L1, multiplies by W_L^(k1 n2) and writes the workspace once - each segment
//               passes through L2 / HBM exactly once between the halves (a 65536-point segment is 512 KiB: it cannot
//               live in one CU's LDS); K2 transforms row k1 (L2 contiguous points) and accumulates bins k1 + L1 k2.  The
//               workspace holds a chunk of segments sized to stay in the 256 MiB Infinity Cache.
//   bluestein   any other length N through a power of two M >= 2 N - 1: a[n] = x[n] w[n] c[n], c[n] = exp(-i pi n^2 / N);
//               X[k] = c[k] IFFT_M(FFT_M(a) FFT_M(conj c))[k].  M <= 16384: one launch (load-stage, transform, mid,
//               transform, store); above: K1, K2 (row transform, x B, row transform, twiddle), K3 (column transform to
//               natural order).  |c| = 1, so |X|^2 and conj(X) Y need neither the final chirp nor the conjugations.
// Every transform is a forward one: the inverse of the Bluestein product is conj(FFT(conj(.))) / M, with the 1 / M folded
// into the B table and the conjugations into the elementwise steps.
//
// Detrend (scipy.signal.welch detrend='constant'): any_mean_kernel adds each segment's samples in double and hands the
// mean over as a float pair hi + lo; the stage subtracts hi (exact for samples near the mean), then lo.
#include "fft4096.hip.h"
#include "oth_internal.h"

namespace oth {

namespace {

// ---- small DFTs, natural order in and out ------------------------------------------------------------------------------
__device__ __forceinline__ void dft2(float2 &a, float2 &b) {
    const float2 s = cadd(a, b), d = csub(a, b);
    a = s;
    b = d;
}

// odd prime P: X[k] = x0 + sum_m cos(2 pi k m / P) (x_m + x_(P-m)) - i sum_m sin(2 pi k m / P) (x_m - x_(P-m))
template <int P> struct PrimeTab;
template <> struct PrimeTab<3> {
    static constexpr float c[1] = {-0.5f};
    static constexpr float s[1] = {0.86602540378443865f};
};
template <> struct PrimeTab<5> {
    static constexpr float c[2] = {0.30901699437494742f, -0.80901699437494742f};
    static constexpr float s[2] = {0.95105651629515357f, 0.58778525229247313f};
};
template <> struct PrimeTab<7> {
    static constexpr float c[3] = {0.62348980185873353f, -0.22252093395631440f, -0.90096886790241913f};
    static constexpr float s[3] = {0.78183148246802981f, 0.97492791218182361f, 0.43388373911755812f};
};
template <int P> __device__ __forceinline__ void dft_prime(float2 (&v)[P]) {
    constexpr int H = (P - 1) / 2;
    float2 a[H], b[H];
#pragma unroll
    for (int m = 1; m <= H; ++m) {
        a[m - 1] = cadd(v[m], v[P - m]);
        b[m - 1] = csub(v[m], v[P - m]);
    }
    float2 x0 = v[0], sum = v[0];
#pragma unroll
    for (int m = 0; m < H; ++m) sum = cadd(sum, a[m]);
    v[0] = sum;
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        float2 re = x0, im = make_float2(0.f, 0.f);
#pragma unroll
        for (int m = 1; m <= H; ++m) {
            const int j = (k * m) % P;                   // angle 2 pi j / P
            const int jj = j <= H ? j : P - j;           // cos is even, sin odd about P / 2
            const float cc = PrimeTab<P>::c[jj - 1], ss = (j <= H ? 1.f : -1.f) * PrimeTab<P>::s[jj - 1];
            re = make_float2(fmaf(cc, a[m - 1].x, re.x), fmaf(cc, a[m - 1].y, re.y));
            im = make_float2(fmaf(ss, b[m - 1].x, im.x), fmaf(ss, b[m - 1].y, im.y));
        }
        // X[k] = re - i im, X[P - k] = re + i im
        v[k] = make_float2(re.x + im.y, re.y - im.x);
        v[P - k] = make_float2(re.x - im.y, re.y + im.x);
    }
}

template <int R> __device__ __forceinline__ void dft_small(float2 (&v)[R]) {
    if constexpr (R == 2) dft2(v[0], v[1]);
    else if constexpr (R == 4) dft4<false>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 8) dft8<true>(v);
    else if constexpr (R == 16) {
        dft16<true>(v);
        float2 t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = v[r16(k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = t[k];
    } else dft_prime<R>(v);
}

}  // namespace

// j / d and j % d for 0 <= j < 2^22 without an integer division (inv = 1 / d in float; one correction step either way)
__device__ __forceinline__ void any_divmod(int j, int d, float inv, int &q, int &r) {
    q = (int)((float)j * inv);
    r = j - q * d;
    if (r < 0) {
        r += d;
        --q;
    } else if (r >= d) {
        r -= d;
        ++q;
    }
}

// Point i of column c of a tile sits at LDS index (i << logC) + ((c + i) & (C - 1)): the rotation makes the transposed
// fill of the row tiles (lanes walk i, stride C points) as free of bank conflicts as the passes (lanes walk c).
__device__ __forceinline__ int any_lds(int i, int c, int logC, int cmask) { return (i << logC) + ((c + i) & cmask); }

// The q-th point of thread tid: (i, c) and its LDS index.  Column tiles (cs == 1: the C columns are adjacent in memory)
// walk the LDS linearly; row tiles (cs != 1: each of the C transforms is contiguous) walk i first.
template <int T, bool C1> __device__ __forceinline__ bool any_point(const AnyArgs &a, int tid, int q, int E, int &i, int &c, int &lds) {
    const int e = tid + q * T;
    if (e >= E) return false;
    if constexpr (C1) {      // one transform per tile (the direct and one-launch Bluestein routes): no column arithmetic at all
        i = e;
        c = 0;
        lds = e;
        return true;
    }
    const int logC = a.f.logC, cmask = (1 << logC) - 1;
    if (a.cs == 1) {
        i = e >> logC;
        c = ((e & cmask) - i) & cmask;
        lds = e;
    } else {
        any_divmod(e, a.f.n, a.inv_n, c, i);
        lds = any_lds(i, c, logC, cmask);
    }
    return true;
}

// One Stockham pass of radix R on the tile: its C columns are transformed together.
template <int T, int R, bool TWLDS, bool C1>
__device__ __forceinline__ void any_pass(float2 *buf, const float2 *twl, const AnyFftDesc &f, const AnyPass &p, int tid) {
    constexpr int Q = (16 + R - 1) / R;       // butterflies per thread: the tile has at most 16 T points
    const int logC = C1 ? 0 : f.logC, cmask = C1 ? 0 : (1 << logC) - 1;      // C1: one column - the index arithmetic folds away
    const int nbf = p.nbf;                    // butterflies per column = n / R
    const int NBF = nbf << logC;
    float2 v[Q][R];
    int dst[Q];      // LDS row of output 0 of butterfly q (kept from the read half: the quotient costs ten instructions)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int bf = tid + q * T;
        dst[q] = 0;
        if (bf < NBF) {
            const int c = bf & cmask, j = bf >> logC;
#pragma unroll
            for (int m = 0; m < R; ++m) v[q][m] = buf[any_lds(j + m * nbf, c, logC, cmask)];
            int jq = 0, jn = j;
            if (p.NS > 1) {
                any_divmod(j, p.NS, p.inv_ns, jq, jn);
                const int k = jn * p.inner;              // twiddle W_(NS R)^(jn m) = W_n^(m k), m k < n
                if constexpr (TWLDS) {                   // the n twiddles of this transform staged in LDS (any_fft_kernel)
#pragma unroll
                    for (int m = 1; m < R; ++m) v[q][m] = cmul(v[q][m], twl[m * k]);
                } else {                                 // gathered from the order-L table in L2
                    const int kg = k * f.tws;
#pragma unroll
                    for (int m = 1; m < R; ++m) v[q][m] = cmul(v[q][m], f.tw[m * kg]);
                }
            } else {
                jq = j;      // NS == 1: j / 1, j % 1
                jn = 0;
            }
            dst[q] = jq * p.NS * R + jn;
            dft_small<R>(v[q]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int bf = tid + q * T;
        if (bf < NBF) {
            const int c = bf & cmask;
#pragma unroll
            for (int m = 0; m < R; ++m) buf[any_lds(dst[q] + m * p.NS, c, logC, cmask)] = v[q][m];
        }
    }
    __syncthreads();
}

// all passes of the descriptor; the caller has synchronised the workgroup after filling buf, and may read any point after
template <int T, bool TWLDS, bool C1> __device__ __forceinline__ void any_fft_lds(float2 *buf, const float2 *twl, const AnyFftDesc &f, int tid) {
    for (int ip = 0; ip < f.npass; ++ip) {
        const AnyPass &p = f.pass[ip];
        switch (p.R) {
            case 16: any_pass<T, 16, TWLDS, C1>(buf, twl, f, p, tid); break;
            case 8: any_pass<T, 8, TWLDS, C1>(buf, twl, f, p, tid); break;
            case 4: any_pass<T, 4, TWLDS, C1>(buf, twl, f, p, tid); break;
            case 2: any_pass<T, 2, TWLDS, C1>(buf, twl, f, p, tid); break;
            case 3: any_pass<T, 3, TWLDS, C1>(buf, twl, f, p, tid); break;
            case 5: any_pass<T, 5, TWLDS, C1>(buf, twl, f, p, tid); break;
            default: any_pass<T, 7, TWLDS, C1>(buf, twl, f, p, tid); break;
        }
    }
}

// STORE: 0 plain (workspace, optional four-step twiddle), 1 accumulate one channel, 2 accumulate two channels,
// 3 periodogram rows
template <int T, int STORE, bool TWLDS, bool C1> __global__ __launch_bounds__(T) void any_fft_kernel(AnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char any_smem[];
    float2 *buf = reinterpret_cast<float2 *>(any_smem);
    const int tid = threadIdx.x, t = blockIdx.x;
    const int E = C1 ? a.f.n : a.f.n << a.f.logC;
    // The n twiddles W_n^j of the tile's transform, staged once per workgroup behind the tile where LDS allows (the host
    // decides: AnyFftDesc.tw_lds): a pass then gathers them from LDS instead of the order-L table in L2 - R - 1 dependent
    // L2 round trips per butterfly and pass were most of a small transform's time (w1000: 2.57 ms -> see DESIGN 4.6).
    const float2 *twl = nullptr;
    if constexpr (TWLDS) {
        float2 *stage = buf + E;
        for (int j = tid; j < a.f.n; j += T) stage[j] = a.f.tw[j * a.f.tws];
        twl = stage;      // (the first __syncthreads() of the segment loop orders it in front of the first pass)
    }
    constexpr int NACC = STORE == 2 ? 4 : (STORE == 1 ? 1 : 0);
    constexpr int kStoreUnroll = (STORE == 1 || STORE == 2) ? 16 : 4;
    float acc[NACC > 0 ? NACC : 1][16];
    float2 X0[STORE == 2 ? 16 : 1];
    if constexpr (NACC > 0) {
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    }
    const int ch_lo = STORE == 0 ? (int)blockIdx.z : 0, ch_hi = STORE == 0 ? ch_lo + 1 : (STORE == 2 ? 2 : 1);
    for (long long s = blockIdx.y; s < a.nseg; s += gridDim.y) {
        for (int ch = ch_lo; ch < ch_hi; ++ch) {
            // ---- load
            float2 mhi = make_float2(0.f, 0.f), mlo = make_float2(0.f, 0.f);
            const float2 *src;
            if (a.load_op == 1) {
                src = (ch ? a.y : a.x) + a.first + s * a.seg_step;
                if (a.mean) {
                    const float4 m = a.mean[(size_t)ch * a.mean_ch_stride + s];
                    mhi = make_float2(m.x, m.y);
                    mlo = make_float2(m.z, m.w);
                }
            } else {
                src = a.ws + (size_t)ch * a.ws_ch_stride + (size_t)s * a.ws_seg_stride;
            }
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                int i, c, lds;
                if (any_point<T, C1>(a, tid, q, E, i, c, lds)) {
                    const int n = i * a.es + t * a.tile_stride + c * a.cs;
                    float2 v;
                    if (a.load_op == 1) {
                        v = make_float2(0.f, 0.f);
                        if (n < a.nperseg) {
                            v = src[n];
                            v = csub(csub(v, mhi), mlo);
                            const float w = a.win[n];
                            v = make_float2(v.x * w, v.y * w);
                            if (a.chirp) v = cmul(v, a.chirp[n]);
                        }
                    } else {
                        v = src[n];
                    }
                    buf[lds] = v;
                }
            }
            __syncthreads();
            any_fft_lds<T, TWLDS, C1>(buf, twl, a.f, tid);
            if (a.mid_op) {      // Bluestein: multiply by B / M, conjugate, transform again
#pragma unroll 4
                for (int q = 0; q < 16; ++q) {
                    int i, c, lds;
                    if (any_point<T, C1>(a, tid, q, E, i, c, lds)) {
                        const int nat = i * a.nat_i + t * a.nat_t + c * a.nat_c;
                        const float2 v = cmul(buf[lds], a.midtab[nat]);
                        buf[lds] = make_float2(v.x, -v.y);
                    }
                }
                __syncthreads();
                any_fft_lds<T, TWLDS, C1>(buf, twl, a.f, tid);
            }
            // ---- store (the accumulating forms index registers by q: fully unrolled; the others four points at a time)
#pragma unroll kStoreUnroll
            for (int q = 0; q < 16; ++q) {
                int i, c, lds;
                if (any_point<T, C1>(a, tid, q, E, i, c, lds)) {
                    float2 v = buf[lds];
                    if constexpr (STORE == 0) {
                        if (a.twbig) v = cmul(v, a.twbig[i * (t * a.tw_t + c * a.tw_c)]);
                        const int n = i * a.es + t * a.tile_stride + c * a.cs;
                        a.ws[(size_t)ch * a.ws_ch_stride + (size_t)s * a.ws_seg_stride + n] = v;
                    } else if constexpr (STORE == 1) {
                        acc[0][q] = fmaf(v.x, v.x, fmaf(v.y, v.y, acc[0][q]));
                    } else if constexpr (STORE == 2) {
                        if (ch == 0) {
                            X0[q] = v;
                        } else {
                            float2 Xv = X0[q], Yv = v;
                            if (a.conj_out) {      // Bluestein leaves conj(X) / conj(Y)
                                Xv.y = -Xv.y;
                                Yv.y = -Yv.y;
                            }
                            acc[0][q] = fmaf(Xv.x, Xv.x, fmaf(Xv.y, Xv.y, acc[0][q]));
                            acc[1][q] = fmaf(Yv.x, Yv.x, fmaf(Yv.y, Yv.y, acc[1][q]));
                            acc[2][q] = fmaf(Xv.x, Yv.x, fmaf(Xv.y, Yv.y, acc[2][q]));       // conj(X) Y
                            acc[3][q] = fmaf(Xv.x, Yv.y, fmaf(-Xv.y, Yv.x, acc[3][q]));
                        }
                    } else {
                        const int nat = i * a.nat_i + t * a.nat_t + c * a.nat_c;
                        if (nat < a.nbins) {
                            const float m2 = fmaf(v.x, v.x, v.y * v.y);
                            const float o = (a.epilogue == 0) ? sqrtf(m2) : m2 * a.scale;
                            int pos = nat;
                            if (a.fftshift) {
                                pos = nat + a.nbins / 2;
                                if (pos >= a.nbins) pos -= a.nbins;
                            }
                            a.rows[(size_t)s * a.nbins + pos] = o;
                        }
                    }
                }
            }
            __syncthreads();      // buf is rewritten by the next load
        }
    }
    if constexpr (NACC > 0) {
        // the workgroup's sums: partial[row g = blockIdx.y][channel][position], added to what earlier chunks left.  The sums
        // pass through the (now free) LDS tile two channels at a time, each thread through its own slots, so that the
        // read-modify-write loop need not index registers and runs four points at a time (sixteen unrolled updates with
        // their addresses held the kernel at 146-226 registers)
        float *dst = a.partial + (size_t)blockIdx.y * NACC * a.nbins;
#pragma unroll
        for (int half = 0; half < (NACC + 1) / 2; ++half) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int i, c, lds;
                if (any_point<T, C1>(a, tid, q, E, i, c, lds)) buf[lds] = make_float2(acc[2 * half][q], NACC > 1 ? acc[NACC > 1 ? 2 * half + 1 : 0][q] : 0.f);
            }
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                int i, c, lds;
                if (any_point<T, C1>(a, tid, q, E, i, c, lds)) {
                    const int nat = i * a.nat_i + t * a.nat_t + c * a.nat_c;      // the bin (filter: Bluestein's M > N outputs)
                    const int pp = i * a.pp_i + t * a.pp_t + c * a.pp_c;          // where the partial row keeps it
                    if (nat < a.nbins) {
                        const float2 sums = buf[lds];
                        float *d0 = dst + (size_t)(2 * half) * a.nbins + pp;
                        *d0 = a.first_chunk ? sums.x : *d0 + sums.x;
                        if (NACC > 1) {
                            float *d1 = d0 + a.nbins;
                            *d1 = a.first_chunk ? sums.y : *d1 + sums.y;
                        }
                    }
                }
            }
        }
    }
}

// per segment (and channel: blockIdx.y) the mean of its nperseg samples, added in double, as hi + lo floats
__global__ __launch_bounds__(64) void any_mean_kernel(const float2 *x, const float2 *y, long long first, long long seg_step,
                                                      int nperseg, long long nseg, float4 *out, size_t ch_stride) {
    const long long s = blockIdx.x;
    const float2 *src = (blockIdx.y ? y : x) + first + s * seg_step;
    double sr = 0.0, si = 0.0;
    int n = threadIdx.x;
    for (; n + 448 < nperseg; n += 512) {      // eight independent loads per trip (one at a time a 1000-point segment took
        float2 v[8];                           // sixteen dependent memory round trips: 13 % of the whole call)
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[n + 64 * u];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            sr += (double)v[u].x + (double)v[u + 1].x;
            si += (double)v[u].y + (double)v[u + 1].y;
        }
    }
    for (; n < nperseg; n += 64) {
        const float2 v = src[n];
        sr += (double)v.x;
        si += (double)v.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_xor(sr, off, 64);
        si += __shfl_xor(si, off, 64);
    }
    if (threadIdx.x == 0) {
        const double mr = sr / nperseg, mi = si / nperseg;
        const float hr = (float)mr, hi = (float)mi;
        out[(size_t)blockIdx.y * ch_stride + s] = make_float4(hr, hi, (float)(mr - (double)hr), (float)(mi - (double)hi));
    }
}

// ---- elementwise helpers of the plain natural -> natural transform (xcorr / fac at any length) -------------------------
// op 0: dst[k] = (k < nsrc ? src[k] : 0) * (tab ? tab[k] : 1)                 (zero padding + chirp)
// op 1: dst[k] = conj(src[k] * tab[k])                                         (Bluestein product)
// op 2: dst[k] = conj(src[k]) * tab[k]                                         (Bluestein output, k < n)
// op 3: dst[k1 + L1 k2] = src[k1 L2 + k2]                                      (two-level order -> natural)
// op 4: dst[k] = conj(src2[k]) * src[k]                                        (xcorr: conj(f conj(e)) = conj(f) e)
// op 5: dst[k] = (|src[k]|, 0)                                                 (fac)
__global__ void any_ew_kernel(int op, float2 *dst, const float2 *src, const float2 *src2, const float2 *tab, int n, int nsrc,
                              int L1, int L2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    switch (op) {
        case 0: {
            float2 v = k < nsrc ? src[k] : make_float2(0.f, 0.f);
            if (tab && k < nsrc) v = cmul(v, tab[k]);
            dst[k] = v;
            break;
        }
        case 1: {
            const float2 v = cmul(src[k], tab[k]);
            dst[k] = make_float2(v.x, -v.y);
            break;
        }
        case 2: dst[k] = cmul(make_float2(src[k].x, -src[k].y), tab[k]); break;
        case 3: {
            const int k1 = k / L2, k2 = k - k1 * L2;
            dst[k1 + L1 * k2] = src[k];
            break;
        }
        case 4: dst[k] = cmul(make_float2(src2[k].x, -src2[k].y), src[k]); break;
        default: dst[k] = make_float2(sqrtf(fmaf(src[k].x, src[k].x, src[k].y * src[k].y)), 0.f); break;
    }
}
// out[i] = |src[i]| * scale, i < n
__global__ void any_abs_kernel(float *out, const float2 *src, int n, float scale) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = sqrtf(fmaf(src[k].x, src[k].x, src[k].y * src[k].y)) * scale;
}

// ---- host side ---------------------------------------------------------------------------------------------------------
bool any_smooth(int n) {
    if (n < 1) return false;
    for (int p : {2, 3, 5, 7})
        while (n % p == 0) n /= p;
    return n == 1;
}

int any_describe(int nfft, AnyShape *out) {
    AnyShape s{};
    s.nfft = nfft;
    if (nfft < 1 || nfft > kAnyMaxFft) return -1;
    const bool pow2 = (nfft & (nfft - 1)) == 0;
    if (nfft <= kAnyMaxTile && any_smooth(nfft)) {
        s.kind = ANY_DIRECT;
        s.L = nfft;
    } else if (pow2) {
        s.kind = ANY_TWOLEVEL;
        s.L = nfft;
    } else {
        int M = 1;
        while (M < 2 * nfft - 1) M <<= 1;
        if (M > kAnyMaxFft) return -1;
        s.L = M;
        s.kind = M <= kAnyMaxTile ? ANY_BLUESTEIN : ANY_BLUESTEIN2;
    }
    if (s.kind == ANY_TWOLEVEL || s.kind == ANY_BLUESTEIN2) {
        s.L2 = s.L == 32768 ? 128 : 256;     // 32768 = 256 x 128: the split fft_tl.hip's kernels take
        s.L1 = s.L / s.L2;
        int C = kAnyMaxTile / 2 / s.L1;      // tiles of at most 8192 points: two workgroups of 64 KiB per CU
        if (C > 32) C = 32;
        if (C < 4) C = kAnyMaxTile / s.L1;   // 4096 rows: the full 16384-point tile
        s.C = C;
    }
    *out = s;
    return 0;
}

int any_threads_for(int points) { return points <= 1024 ? 64 : (points <= 4096 ? 256 : 1024); }

// radix list (16 / 8 / 4 / 2 for the power of two, then the odd primes) and the per-pass constants of a length-n transform
// whose twiddles come from a table of order `order` (a multiple of n)
void any_make_desc(int n, int C, const float2 *tw, int order, AnyFftDesc *d) {
    *d = AnyFftDesc{};
    d->n = n;
    int logC = 0;
    while ((1 << logC) < C) ++logC;
    d->logC = logC;
    d->tw = tw;
    int rad[kAnyMaxPasses], np = 0, m = n;
    for (int p : {7, 5, 3})
        while (m % p == 0) {
            rad[np++] = p;
            m /= p;
        }
    while (m % 16 == 0) {
        rad[np++] = 16;
        m /= 16;
    }
    if (m % 8 == 0) {
        rad[np++] = 8;
        m /= 8;
    }
    if (m % 4 == 0) {
        rad[np++] = 4;
        m /= 4;
    }
    if (m % 2 == 0) {
        rad[np++] = 2;
        m /= 2;
    }
    d->npass = np;
    int NS = 1;
    d->tws = order / n;
    // twiddles in LDS when tile + table stay inside 64 KiB (no opt-in, two or more workgroups per CU) or the tile alone
    // already needs the opt-in and both fit the CU's 160 KiB
    const size_t tile = (size_t)n * C * sizeof(float2), both = tile + (size_t)n * sizeof(float2);
    d->tw_lds = (both <= 64 * 1024 || (tile > 64 * 1024 && both <= 150 * 1024)) ? 1 : 0;
    for (int i = 0; i < np; ++i) {
        AnyPass &p = d->pass[i];
        p.R = rad[i];
        p.NS = NS;
        p.nbf = n / rad[i];
        p.inner = n / (NS * rad[i]);
        p.inv_ns = 1.0f / (float)NS;
        NS *= rad[i];
    }
}

template <typename K> static hipError_t any_allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// grid = (tiles, rows of segments, channels for store 0); store: 0 plain, 1 accumulate, 2 two-channel accumulate, 3 rows
hipError_t launch_any_fft(const AnyArgs &a, int tiles, int gy, int gz, int store, hipStream_t s) {
    const int points = a.f.n << a.f.logC;
    if (points > kAnyMaxTile || tiles < 1 || gy < 1 || gz < 1) return hipErrorInvalidValue;
    const int T = any_threads_for(points);
    const size_t lds = ((size_t)points + (a.f.tw_lds ? (size_t)a.f.n : 0)) * sizeof(float2);
    const dim3 grid(tiles, gy, gz);
    hipError_t e;
#define OTH_ANY_LAUNCH1(TT, ST, TW, CC)                                                                     \
    do {                                                                                                    \
        if ((e = any_allow_lds(any_fft_kernel<TT, ST, TW, CC>, lds)) != hipSuccess) return e;               \
        hipLaunchKernelGGL((any_fft_kernel<TT, ST, TW, CC>), grid, dim3(TT), lds, s, a);                    \
    } while (0)
    // one transform per tile (logC == 0, column mode): the builds without column arithmetic
#define OTH_ANY_LAUNCH(TT, ST)                                                                              \
    do {                                                                                                    \
        const bool c1 = a.f.logC == 0 && a.cs == 1;                                                         \
        if (a.f.tw_lds) {                                                                                   \
            if (c1) OTH_ANY_LAUNCH1(TT, ST, true, true);                                                    \
            else OTH_ANY_LAUNCH1(TT, ST, true, false);                                                      \
        } else {                                                                                            \
            if (c1) OTH_ANY_LAUNCH1(TT, ST, false, true);                                                   \
            else OTH_ANY_LAUNCH1(TT, ST, false, false);                                                     \
        }                                                                                                   \
    } while (0)
#define OTH_ANY_T(ST)                                     \
    do {                                                  \
        if (T == 64) OTH_ANY_LAUNCH(64, ST);              \
        else if (T == 256) OTH_ANY_LAUNCH(256, ST);       \
        else OTH_ANY_LAUNCH(1024, ST);                    \
    } while (0)
    switch (store) {
        case 0: OTH_ANY_T(0); break;
        case 1: OTH_ANY_T(1); break;
        case 2: OTH_ANY_T(2); break;
        case 3: OTH_ANY_T(3); break;
        default: return hipErrorInvalidValue;
    }
#undef OTH_ANY_T
#undef OTH_ANY_LAUNCH
#undef OTH_ANY_LAUNCH1
    return hipGetLastError();
}

hipError_t launch_any_mean(const float2 *x, const float2 *y, long long first, long long seg_step, int nperseg, long long nseg,
                           float4 *out, size_t ch_stride, hipStream_t s) {
    hipLaunchKernelGGL(any_mean_kernel, dim3((unsigned)nseg, y ? 2 : 1), dim3(64), 0, s, x, y, first, seg_step, nperseg, nseg, out,
                       ch_stride);
    return hipGetLastError();
}

hipError_t launch_any_ew(int op, float2 *dst, const float2 *src, const float2 *src2, const float2 *tab, int n, int nsrc, int L1,
                         int L2, hipStream_t s) {
    hipLaunchKernelGGL(any_ew_kernel, dim3((n + 255) / 256), dim3(256), 0, s, op, dst, src, src2, tab, n, nsrc, L1, L2);
    return hipGetLastError();
}

hipError_t launch_any_abs(float *out, const float2 *src, int n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(any_abs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, out, src, n, scale);
    return hipGetLastError();
}

}  // namespace oth
