// The four-step route at 32768 and 65536 points with register-resident radix-16 butterflies (round 6): the two lengths above
// the tuned kernels that fast_spectrum_scan picks for every block of 16 Ki ... 64 Ki samples (ofdm_cr_tools.py:474-475) and a
// flowgraph reaches with --nfft (sdr_webserver/local_hw_gateway.py:284-285).  fft_any.hip's coverage kernel runs the same
// four steps from launch arguments (runtime radix list, 128 KiB tiles, table twiddles); here every size is a template
// argument and a transform of 256 points is sixteen threads x sixteen registers, as in segfft.hip:
//
//   L = L1 L2, L1 = 256, L2 = 256 (65536) or 128 (32768); sample n = n1 L2 + n2, bin k = k1 + L1 k2.
//   tl_k1_kernel   one workgroup = 16 adjacent columns n2 (128 contiguous bytes per row n1) x 16 threads: thread (c, a) loads
//                  n1 = a + 16 b (window, detrend, zero padding), dft16 over b, x W256^(a kb), exchange through LDS across
//                  the sixteen a, dft16 over a -> k1 = kb + 16 ka, x W_L^(k1 n2), workspace.  The segment leaves the chip
//                  once: workspace [segment][column tile][k1][16 columns] - a workgroup's output is one 32 KiB block.
//   tl_k2_kernel   row k1 of the workspace (L2 contiguous points) per team of 16 (L2 = 256) or 8 (128) lanes of one wave:
//                  dft16 over b, twiddle, exchange inside the wave, dft16 (two dft8) -> k2; |X|^2 added over the
//                  workgroup's segments; partial rows in [k1][k2] order (finalize layout 6).
//   two channels   (oth_csd_exec / coherence): K1 per channel on blockIdx.z, K2<CSD> transforms row k1 of x and of y back
//                  to back and adds |X|^2, |Y|^2, conj(X) Y: partial rows [W][4][L].
// The workspace holds a chunk of segments (any_run, api.hip) small enough to stay in the Infinity Cache between the two.
#include "fft4096.hip.h"
#include "oth_internal.h"

namespace oth {

namespace {
constexpr int TL_L1 = 256;
constexpr int TL_RS = 272;      // float2 per kb region of the K1 exchange image: 16 a x 16 c + 16 (bank rotation)
}

template <int L2> __global__ __launch_bounds__(256, 4) void tl_k1_kernel(TlArgs p) {
    constexpr int L = TL_L1 * L2;
    __shared__ float2 lds[16 * TL_RS];
    const int tid = threadIdx.x, c = tid & 15, a = tid >> 4;
    const int n2 = blockIdx.x * 16 + c;
    // W256^a and W256^(4a): the inner twiddles of the 256-point column transform
    const float2 w1 = p.tw[a * (L / 256)], w4 = p.tw[4 * a * (L / 256)];
    // four-step twiddles W_L^((a + 16 ka) n2) = base pw^ka
    float2 base = p.tw[a * n2], pw = p.tw[16 * n2];
    const int ch = blockIdx.z;      // two-channel plans: channel 1 = y, its sub-block sums / means / workspace one stride further
    for (long long s = blockIdx.y; s < p.nseg; s += gridDim.y) {
        const float2 *src = (ch ? p.y : p.x) + p.first + s * p.seg_step;
        float2 mhi = make_float2(0.f, 0.f), mlo = make_float2(0.f, 0.f);
        if (p.bsum) {      // the segment's mean from the sums of its sub-blocks (tl_blocksum_kernel): one read of the chunk's
            const double2 *bs = p.bsum + (size_t)ch * p.aux_ch_stride + s * p.sub_step;      // samples for all the overlapping segments
            double mr = 0.0, mi = 0.0;
            for (int j = 0; j < p.nsub; ++j) {
                mr += bs[j].x;
                mi += bs[j].y;
            }
            mr /= p.nperseg;
            mi /= p.nperseg;
            mhi = make_float2((float)mr, (float)mi);
            mlo = make_float2((float)(mr - (double)mhi.x), (float)(mi - (double)mhi.y));
        } else if (p.mean) {
            const float4 m = p.mean[(size_t)ch * p.aux_ch_stride + s];
            mhi = make_float2(m.x, m.y);
            mlo = make_float2(m.z, m.w);
        }
        // The window table holds L values, zero behind nperseg: rows of the zero padding read the segment's last sample
        // (a valid address) and multiply it by 0 - sixteen unconditional loads, no branch per row.
        float2 v[16];
        float w[16];
        const int last = p.nperseg - 1;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const int n = (a + 16 * b) * L2 + n2;
            v[b] = src[n < last ? n : last];      // (plain loads: the overlapped half is read again from L2 / MALL)
            w[b] = p.win[n];
        }
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const float2 d = csub(csub(v[b], mhi), mlo);
            v[b] = make_float2(d.x * w[b], d.y * w[b]);
        }
        dft16(v);                                             // Z_a[kb] at v[r16(kb)]
        scatter_pow16<TL_RS>(v, lds + a * 16 + c, w1, w4);    // lds[kb][a][c] = Z_a[kb] W256^(a kb)
        __syncthreads();
        float2 u[16];                                         // this thread now is (c, kb = a)
#pragma unroll
        for (int a2 = 0; a2 < 16; ++a2) u[a2] = lds[a * TL_RS + a2 * 16 + c];
        __syncthreads();
        dft16(u);                                             // X1[kb + 16 ka] at u[r16(ka)]
        // workspace [segment][column tile][k1][16 columns]: this workgroup's 256 rows x 128 B are ONE contiguous 32 KiB block
        // (row-major [k1][n2] made every store instruction four 128-byte pieces 2 KiB apart, and the sixteen tiles of a row
        // arrived at different times), and tl_k2's loads of rows k1 .. k1 + 15 of a tile are 2 KiB contiguous
        float2 *dst = p.ws + (size_t)ch * p.ws_ch_stride + (size_t)s * p.ws_seg_stride + ((size_t)blockIdx.x * TL_L1 + a) * 16 + c;
        // the sixteen four-step twiddles are multiplied out of the two seeds HERE, behind the second butterfly: held from
        // the kernel's prologue they were fourteen registers of a kernel at the 128-register line (nine products per segment)
        asm volatile("" : "+v"(base.x), "+v"(base.y), "+v"(pw.x), "+v"(pw.y));
        float2 bj[4], wi[4];
        {
            const float2 p2 = cmul(pw, pw), p3 = cmul(p2, pw);
            bj[0] = base;
            bj[1] = cmul(base, pw);
            bj[2] = cmul(base, p2);
            bj[3] = cmul(base, p3);
            wi[1] = cmul(p2, p2);
            wi[2] = cmul(wi[1], wi[1]);
            wi[3] = cmul(wi[2], wi[1]);
        }
#pragma unroll
        for (int ka = 0; ka < 16; ++ka) {
            const int i = ka >> 2, j = ka & 3;
            const float2 w = i == 0 ? bj[j] : cmul(wi[i], bj[j]);
            dst[16 * 16 * ka] = cmul(u[r16(ka)], w);
        }
    }
}

// One row transform of tl_k2: row k1 of the workspace at `row` -> this thread's sixteen outputs X2[kb + 16 ka] in out[]
// (TEAM 16: kb = a, out[ka]; TEAM 8: kb = a -> out[0..7], kb = a + 8 -> out[8..15], ka = 0..7)
template <int L2> __device__ __forceinline__ void tl_row_fft(const float2 *row, float2 *team, int a, float2 w1, float2 w4, float2 (&out)[16]) {
    constexpr int TEAM = L2 / 16, TS = TEAM + 1;
    float2 v[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int pt = a + TEAM * b;      // point p of row k1 sits in column tile p / 16 at column p % 16 (tl_k1's workspace layout)
        v[b] = row[(size_t)(pt >> 4) * (TL_L1 * 16) + (pt & 15)];
    }
    dft16(v);                                            // Z_a[kb], kb < 16, at v[r16(kb)]
    scatter_pow16<TS>(v, team + a, w1, w4);              // team[kb][a] = Z_a[kb] W_L2^(a kb)
    wave_lds_sync();
    if constexpr (TEAM == 16) {
        float2 u[16];
#pragma unroll
        for (int a2 = 0; a2 < 16; ++a2) u[a2] = team[a * TS + a2];       // kb = a
        wave_lds_sync();
        dft16(u);                                        // X2[kb + 16 ka] at u[r16(ka)]
#pragma unroll
        for (int ka = 0; ka < 16; ++ka) out[ka] = u[r16(ka)];
    } else {
        float2 u0[8], u1[8];                             // kb = a and kb = a + 8
#pragma unroll
        for (int a2 = 0; a2 < 8; ++a2) {
            u0[a2] = team[a * TS + a2];
            u1[a2] = team[(a + 8) * TS + a2];
        }
        wave_lds_sync();
        dft8(u0);
        dft8(u1);
#pragma unroll
        for (int ka = 0; ka < 8; ++ka) {
            out[ka] = u0[ka];
            out[8 + ka] = u1[ka];
        }
    }
}

// CSD: both channels' rows per segment, the four two-channel sums (partial rows [W][4][L]: xx, yy, re, im)
template <int L2, bool CSD> __global__ __launch_bounds__(256) void tl_k2_kernel(TlArgs p) {
    constexpr int L = TL_L1 * L2;
    constexpr int TEAM = L2 / 16;                 // lanes per row: 16 (256 points) or 8 (128)
    constexpr int ROWS = 256 / TEAM;              // rows per workgroup
    constexpr int TS = TEAM + 1;                  // float2 per kb line of a team's exchange image
    constexpr int NACC = CSD ? 4 : 1;
    __shared__ float2 lds[ROWS * 16 * TS];
    const int tid = threadIdx.x, a = tid & (TEAM - 1), r = tid / TEAM;
    const int k1 = blockIdx.x * ROWS + r;
    float2 *team = lds + r * 16 * TS;
    const float2 w1 = p.tw[a * (L / L2)], w4 = p.tw[4 * a * (L / L2)];      // W_L2^a, W_L2^(4a)
    float acc[NACC][16];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    for (long long s = blockIdx.y; s < p.nseg; s += gridDim.y) {
        const float2 *row = p.ws + (size_t)s * p.ws_seg_stride + (size_t)k1 * 16;
        float2 X[16];
        tl_row_fft<L2>(row, team, a, w1, w4, X);
        if constexpr (!CSD) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[0][q] = fmaf(X[q].x, X[q].x, fmaf(X[q].y, X[q].y, acc[0][q]));
        } else {
            float2 Y[16];
            tl_row_fft<L2>(row + p.ws_ch_stride, team, a, w1, w4, Y);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc[0][q] = fmaf(X[q].x, X[q].x, fmaf(X[q].y, X[q].y, acc[0][q]));
                acc[1][q] = fmaf(Y[q].x, Y[q].x, fmaf(Y[q].y, Y[q].y, acc[1][q]));
                acc[2][q] = fmaf(X[q].x, Y[q].x, fmaf(X[q].y, Y[q].y, acc[2][q]));       // conj(X) Y
                acc[3][q] = fmaf(X[q].x, Y[q].y, fmaf(-X[q].y, Y[q].x, acc[3][q]));
            }
        }
    }
    // partial row g = blockIdx.y: [channel][position k1 L2 + k2]; out index q <-> k2 = kb + 16 ka as in tl_row_fft
    float *dst = p.partial + (size_t)blockIdx.y * NACC * L + (size_t)k1 * L2;
#pragma unroll
    for (int c = 0; c < NACC; ++c) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k2 = TEAM == 16 ? a + 16 * q : (q < 8 ? a + 16 * q : a + 8 + 16 * (q - 8));
            float *d = dst + (size_t)c * L + k2;
            *d = p.first_chunk ? acc[c][q] : *d + acc[c][q];
        }
    }
}

// Segment means for the detrend, wide: 256 threads per segment, four independent loads per trip, sums in double
// (any_mean_kernel's one wave per segment walks a 32768-point segment in 512 dependent trips).
__global__ __launch_bounds__(256) void tl_mean_kernel(const float2 *x, long long first, long long seg_step, int nperseg, float4 *out) {
    __shared__ double red[2][4];
    const float2 *src = x + first + (long long)blockIdx.x * seg_step;
    double sr = 0.0, si = 0.0;
    int n = threadIdx.x;
    for (; n + 768 < nperseg; n += 1024) {
        const float2 v0 = src[n], v1 = src[n + 256], v2 = src[n + 512], v3 = src[n + 768];
        sr += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        si += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; n < nperseg; n += 256) {
        const float2 v = src[n];
        sr += (double)v.x;
        si += (double)v.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_xor(sr, off, 64);
        si += __shfl_xor(si, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sr;
        red[1][threadIdx.x >> 6] = si;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double mr = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / nperseg;
        const double mi = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / nperseg;
        const float hr = (float)mr, hi = (float)mi;
        out[blockIdx.x] = make_float4(hr, hi, (float)(mr - (double)hr), (float)(mi - (double)hi));
    }
}

// Sums of sub-blocks of kTlSub samples, in double: when the segment length and the step are multiples of kTlSub every
// segment's mean is the sum of nperseg / kTlSub consecutive sub-block sums, and the samples of overlapping segments are
// read once for all of them.
__global__ __launch_bounds__(256) void tl_blocksum_kernel(const float2 *x, long long first, double2 *out) {
    __shared__ double red[2][4];
    const float2 *src = x + first + (long long)blockIdx.x * kTlSub;
    float2 v[kTlSub / 256];
#pragma unroll
    for (int q = 0; q < kTlSub / 256; ++q) v[q] = src[threadIdx.x + 256 * q];
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int q = 0; q < kTlSub / 256; ++q) {
        sr += (double)v[q].x;
        si += (double)v[q].y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_xor(sr, off, 64);
        si += __shfl_xor(si, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sr;
        red[1][threadIdx.x >> 6] = si;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        out[blockIdx.x] = make_double2((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
}

hipError_t launch_tl_blocksum(const float2 *x, long long first, long long nblocks, double2 *out, hipStream_t s) {
    hipLaunchKernelGGL(tl_blocksum_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, x, first, out);
    return hipGetLastError();
}

bool tl_supported(int L) { return L == 32768 || L == 65536; }

hipError_t launch_tl_mean(const float2 *x, long long first, long long seg_step, int nperseg, long long nseg, float4 *out, hipStream_t s) {
    hipLaunchKernelGGL(tl_mean_kernel, dim3((unsigned)nseg), dim3(256), 0, s, x, first, seg_step, nperseg, out);
    return hipGetLastError();
}

hipError_t launch_tl_k1(int L, const TlArgs &a, hipStream_t s) {
    const int gy = (int)(a.nseg < 65535 ? a.nseg : 65535), gz = a.y ? 2 : 1;
    if (L == 65536) hipLaunchKernelGGL(tl_k1_kernel<256>, dim3(16, gy, gz), dim3(256), 0, s, a);
    else if (L == 32768) hipLaunchKernelGGL(tl_k1_kernel<128>, dim3(8, gy, gz), dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// W partial rows: workgroup (row tile, g) adds the segments g, g + W, ...
hipError_t launch_tl_k2(int L, const TlArgs &a, int W, hipStream_t s) {
    if (L == 65536) {
        if (a.y) hipLaunchKernelGGL((tl_k2_kernel<256, true>), dim3(16, W), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((tl_k2_kernel<256, false>), dim3(16, W), dim3(256), 0, s, a);
    } else if (L == 32768) {
        if (a.y) hipLaunchKernelGGL((tl_k2_kernel<128, true>), dim3(8, W), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((tl_k2_kernel<128, false>), dim3(8, W), dim3(256), 0, s, a);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace oth
