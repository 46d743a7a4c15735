// welch32k: segment-averaged |FFT_32768((x - mean) w)|^2 with the WHOLE segment inside one workgroup - the first length
// above the tuned kernels (fast_spectrum_scan picks it for every block of 32 Ki ... 64 Ki samples, ofdm_cr_tools.py:474-475;
// --nfft of sdr_webserver/local_hw_gateway.py:284-285), any overlap, scipy.signal.welch semantics (ofdm_cr_tools.py:322,342).
//
// The four-step route (fft_tl.hip) moves a segment through a workspace between its two halves: 6.3 x the algorithmic
// traffic and 11 % of the HBM roofline, whatever the kernels do.  A 32768-point segment is 256 KiB - more
// than a CU's LDS (160 KiB) but HALF of its register file: 1024 threads x 32 points.  One radix-2 step in registers
// (decimation in frequency) turns it into two 16384-point transforms that run one after the other through the scheme of
// welch16k1x.hip (N = 16 x 16 x 16 x 4, ONE cross-wave exchange through 136 KiB of LDS, the last radix-4 over the lanes of a
// quad by DPP), and nothing but the partial sums leaves the CU:
//
//   n = tid + 1024 r, r = 0..15          y[n] = (x[n] - mean) w[n],  y[n + M] likewise (M = 16384)
//   a[n] = y[n] + y[n + M]               X[2 k]     = FFT_M(a)[k]
//   b[n] = (y[n] - y[n + M]) W_N^n       X[2 k + 1] = FFT_M(b)[k]          W_N^n = W_N^tid W_32^r (constants)
//
// Thread (wave k0, lane 4 k1 + q) ends a transform with k = k0 + 16 k1 + 256 k2 + 4096 bitrev2(q) in register k2 (up to a
// factor -1 or -i, as in welch16k1x.hip - only |.|^2 leaves); partial row [A | B]: position 1024 k2 + tid of half A holds
// bin 2 k, of half B bin 2 k + 1 (finalize layout 7).
//
// Detrend in the time domain with the exact mean (scipy's detrend='constant'; no pilot needed): every thread adds its 32
// samples in double, the waves' totals meet in LDS behind one workgroup barrier, the mean is subtracted as a float pair
// (hi + lo).  That barrier also covers the exchange-A hand-over of the first transform: four workgroup barriers per segment.
// The twiddle seeds live in LDS (148.8 of 160 KiB with the exchange regions).
//
// Samples are read with ordinary (cached) loads: at 50 % overlap every sample is wanted by two segments, and the segments of
// one round are dealt out so that neighbours run on the same XCD (workgroup b runs on XCD b % 8: it takes slot
// (b % 8) (W / 8) + b / 8) - the second reader finds the half in that XCD's L2.
#include "fft16k.hip.h"

namespace oth {
namespace {

constexpr int W32_M = 16384, W32_N = 32768;
#ifndef W32_TABLES
#define W32_TABLES 0      // A/B: 1 = passes 2 and 3 read all fifteen twiddles from the LDS tables instead of rebuilding them from rows 1 and 4
#endif
// LDS behind the sixteen exchange regions: the waves' sample totals, the seed W_N^tid of every thread (radix-2 step and, squared,
// pass 1) and the twiddle tables of passes 2 and 3, [k][lane]: W_1024^(k l) and W_64^(k q), k < 16 - no vector-memory load
// inside the transforms.  The passes take rows 1 and 4 as seeds and rebuild the other powers (thirteen products, 52
// instructions per pass); reading all fifteen from the table instead (W32_TABLES=1) is 8 % fewer VALU instructions and 3 %
// SLOWER same-box (0.562 against 0.545 ms per 2^27 samples: thirty more LDS round trips per transform).
// Also measured and dropped: requests for the next segment's cache lines issued under the transforms (two or four dwords per
// thread whose values are never used): 22.7 % of the roofline against 24.7 % without them.
constexpr size_t W32_LDS_BYTES = 16 * XREG * sizeof(float2) + 16 * sizeof(double2) + 1024 * sizeof(float2) + 16 * 64 * sizeof(float2) +
                                 16 * 4 * sizeof(float2);

// exp(-2 pi i r / 32), r = 0..15
constexpr float W32_RE[16] = {1.0f, 0.98078528040323044f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                              0.55557023301960222f, 0.38268343236508977f, 0.19509032201612827f, 0.0f, -0.19509032201612827f,
                              -0.38268343236508977f, -0.55557023301960222f, -0.70710678118654752f, -0.83146961230254524f,
                              -0.92387953251128674f, -0.98078528040323044f};
constexpr float W32_IM[16] = {-0.0f, -0.19509032201612827f, -0.38268343236508977f, -0.55557023301960222f, -0.70710678118654752f,
                              -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323044f, -1.0f, -0.98078528040323044f,
                              -0.92387953251128674f, -0.83146961230254524f, -0.70710678118654752f, -0.55557023301960222f,
                              -0.38268343236508977f, -0.19509032201612827f};

// Loads at (uniform row base in scalar registers) + (the lane's 32-bit byte offset): one offset register serves every row
// of a segment.  Written as inline asm because the compiler, left to itself, forms thirty-two 64-bit row addresses per
// lane, keeps them across the loop and spills them.  A register written here is NOT valid until the s_waitcnt that
// covers it (the waits below name the registers they make valid; tools/isa_async_hazard.py checks the code object).
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load_row8(f2v &dst, unsigned lane_off, const void *row) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(row) : "memory");
}
__device__ __forceinline__ void load_row4(float &dst, unsigned lane_off, const void *row) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(row) : "memory");
}
#define W32_IO8(r) "+v"((r)[0]), "+v"((r)[1]), "+v"((r)[2]), "+v"((r)[3]), "+v"((r)[4]), "+v"((r)[5]), "+v"((r)[6]), "+v"((r)[7])
#define W32_IO16(r) W32_IO8(r), "+v"((r)[8]), "+v"((r)[9]), "+v"((r)[10]), "+v"((r)[11]), "+v"((r)[12]), "+v"((r)[13]), "+v"((r)[14]), "+v"((r)[15])
// every vector-memory load of this wave has landed: r[0..15] valid behind this statement
__device__ __forceinline__ void vm_arrived16(f2v (&r)[16]) { asm volatile("s_waitcnt vmcnt(0)" : W32_IO16(r) : : "memory"); }
// all but the youngest eight loads have landed (the caller has issued the NEXT batch of eight behind this one)
__device__ __forceinline__ void vm_arrived8_keep8(float (&w)[8]) { asm volatile("s_waitcnt vmcnt(8)" : W32_IO8(w) : : "memory"); }
__device__ __forceinline__ void vm_arrived8(float (&w)[8]) { asm volatile("s_waitcnt vmcnt(0)" : W32_IO8(w) : : "memory"); }

// scatter_pow16 (fft4096.hip.h) with the rows k >= 8 addressed from a second base: 8 XREG float2 is past the 64 KiB a
// ds_write offset reaches, and the compiler otherwise keeps eight more address registers across the loop
__device__ __forceinline__ void scatter_pow16_exa(const float2 (&v)[16], float2 *lo, float2 *hi, float2 p1, float2 p4) {
    float2 wj[4], wi[4];
    wj[1] = p1;
    wi[1] = p4;
    asm volatile("" : "+v"(wj[1].x), "+v"(wj[1].y), "+v"(wi[1].x), "+v"(wi[1].y));
    wj[2] = cmul(wj[1], wj[1]);
    wj[3] = cmul(wj[2], wj[1]);
    wi[2] = cmul(wi[1], wi[1]);
    wi[3] = cmul(wi[2], wi[1]);
    lo[0] = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const int i = k >> 2, j = k & 3;
        const float2 w = (i == 0) ? wj[j] : ((j == 0) ? wi[i] : cmul(wi[i], wj[j]));
        (k < 8 ? lo : hi)[XREG * (k & 7)] = cmul(v[r16(k)], w);
    }
}

// out[STRIDE k] = v[r16(k)] tab[TS k], k = 0..15 (tab[0] = 1); STRIDE 0: the products stay in v.  Four table values at a time:
// the scheduler may not gather all fifteen reads in front (thirty registers the first transform does not have)
template <int STRIDE, int TS> __device__ __forceinline__ void scatter_tab16(float2 (&v)[16], float2 *out, const float2 *tab) {
    if (STRIDE) out[0] = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const float2 r = cmul(v[r16(k)], tab[TS * k]);
        if (STRIDE) out[STRIDE * k] = r;
        else v[r16(k)] = r;
        if ((k & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    }
}

// Sum over the 64 lanes of a wave, the same value in every lane: DPP row operations on the two halves of the double +
// v_readlane of the four row totals (wave_total of fft4096.hip.h in double; __shfl_xor would be twelve ds_bpermute and six
// address registers kept across the loop)
template <int CTRL> __device__ __forceinline__ double dpp_add_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v = dpp_add_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add_f64<0x141>(v);   // row_half_mirror
    v = dpp_add_f64<0x140>(v);   // row_mirror: every lane now holds its row-of-16 sum
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = (int)b, hi = (int)(b >> 32);
    double t = 0.0;
#pragma unroll
    for (int row = 0; row < 4; ++row)
        t += __builtin_bit_cast(double, ((long long)__builtin_amdgcn_readlane(hi, 16 * row) << 32) |
                                            (unsigned)__builtin_amdgcn_readlane(lo, 16 * row));
    return t;
}

template <bool DETREND>
__global__ __launch_bounds__(1024) void welch32k_kernel(W32kArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    double2 *msum = reinterpret_cast<double2 *>(lds + 16 * XREG);
    float2 *dtab = reinterpret_cast<float2 *>(msum + 16);
    float2 *btab = dtab + 1024;          // [k1][l] = W_1024^(k1 l)
    float2 *ctab = btab + 16 * 64;       // [k2][q] = W_64^(k2 q)

    const int tid = threadIdx.x;
    const int W = gridDim.x, b = blockIdx.x;
    const int slot = (W & 7) ? b : (b & 7) * (W >> 3) + (b >> 3);

    // Nothing but the thread index and the thirty-two sums is carried around the segment loop in vector registers: the lane's
    // LDS addresses, the twiddle seeds (p.tw[k] = W_N^k: W, W^4 of the three twiddled passes of a 16384-point transform,
    // W_N^tid of the radix-2 step - L1 / L2 hits) and the window rows are formed again where they are used, from values the
    // compiler cannot see through (`opaque`) - it would otherwise hoist them out of the loop and spill them.
    auto opaque = [](int v) {
        asm volatile("" : "+v"(v));
        return v;
    };

    {
        const int t = threadIdx.x;
        dtab[t] = p.tw[t];
        btab[t] = p.tw[(32 * (t >> 6) * (t & 63)) & (W32_N - 1)];
        if (t < 64) ctab[t] = p.tw[(512 * (t >> 2) * (t & 3)) & (W32_N - 1)];
        __syncthreads();
    }
    float accA[16], accB[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) accA[k] = accB[k] = 0.f;

    // one 16384-point transform of v (in place: thread (w, l) holds n = tid + 1024 r) added to acc; `handover`: a workgroup
    // barrier in front of the exchange-A writes (every wave through with the reads of the transform before)
    auto transform = [&](float2 (&v)[16], float (&acc)[16], bool handover) {
        const int t = opaque(tid), wv = t >> 6, l = t & 63, g = l >> 2, q = l & 3;
        float2 *wa = lds + t;                                // exchange A write: + XREG k0 (k0 < 8), wa8 + XREG (k0 - 8)
        float2 *wa8 = wa + opaque(8 * XREG);                 // (a VALUE the compiler cannot fold; the pointer stays an LDS pointer)
        const float2 *ra = lds + XREG * wv + l;              // exchange A read:  + 64 w
        float2 *wb = lds + XREG * wv + l;                    // exchange B write: + XROW k1
        const float2 *rb = lds + XREG * wv + XROW * g + q;   // exchange B read:  + 4 j
        prio_compute();
        dft16(v);                                              // pass 1: r -> k0
        prio_latency();
        if (handover) lds_barrier();
        {
            const float2 d = dtab[t], a1 = cmul(d, d), a2 = cmul(a1, a1);      // W_M^tid = (W_N^tid)^2 and its fourth power
            scatter_pow16_exa(v, wa, wa8, a1, cmul(a2, a2));                   // x W_M^(k0 tid) -> [k0][w][l]
        }
        lds_barrier();
        dft16_from_lds<64>(v, ra, [] { prio_compute(); });     // pass 2: w -> k1
        prio_latency();
        wave_lds_sync();
#if W32_TABLES
        scatter_tab16<XROW, 64>(v, wb, btab + l);              // x W_1024^(k1 l) -> row k1, column l of this wave's region
#else
        scatter_pow16<XROW>(v, wb, btab[64 + l], btab[256 + l]);
#endif
        wave_lds_sync();
        dft16_from_lds<4>(v, rb, [] { prio_compute(); });      // pass 3: g -> k2
        {
#if W32_TABLES
            scatter_tab16<0, 4>(v, nullptr, ctab + q);                 // x W_64^(k2 q), in place
#else
            twiddle_pow16_inplace(v, ctab[4 + q], ctab[16 + q]);
#endif
            const float qs1 = q < 2 ? 1.0f : -1.0f;
            const float qal = q == 0 ? 1.0f : (q == 1 ? -1.0f : 0.0f);
            const float qbe = q >= 2 ? 1.0f : 0.0f;
            quad_dft4_dpp(v, qs1, qal, qbe, -qbe);                     // pass 4: q -> k3
        }
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const float2 X = v[r16(k2)];
            acc[k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[k2]));
        }
    };

    for (long long s = slot; s < p.nseg; s += W) {
        const long long off = p.first + s * p.step;
        const float2 *xs = p.x + (((long long)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)off));
        const unsigned ut8 = ((unsigned)opaque(tid) & 1023u) * 8u, ut4 = ut8 >> 1;
        const float *wn = p.win;
        asm volatile("" : "+s"(wn));      // (or thirty-two row bases stay in scalar registers across the loop)
        float2 va[16], vb[16];
        prio_latency();
        {
            f2v la[16], lb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                load_row8(la[r], ut8, xs + 1024 * r);
                load_row8(lb[r], ut8, xs + W32_M + 1024 * r);
            }
            vm_arrived16(la);
            vm_arrived16(lb);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                va[r] = make_float2(la[r].x, la[r].y);
                vb[r] = make_float2(lb[r].x, lb[r].y);
            }
        }
        // window values in batches of four rows (w[n] in wq[.][0..3], w[n + M] in wq[.][4..7]), two batches in flight: all
        // thirty-two at once do not fit beside the 64 registers of samples and the 32 of sums.  The first two go out in
        // front of the mean's barrier.
        float wq[2][8];
        auto win_issue = [&](int bt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                load_row4(wq[bt & 1][i], ut4, wn + 1024 * (4 * bt + i));
                load_row4(wq[bt & 1][4 + i], ut4, wn + W32_M + 1024 * (4 * bt + i));
            }
        };
        float2 mhi = make_float2(0.f, 0.f), mlo = mhi;
        if (DETREND) {
            double sx = 0.0, sy = 0.0;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // four samples at a time in float, the groups in double: the float rounding of a group is random from group to
                // group (8192 of them per segment), the mean's error from it ~1e-9 of the offset
                sx += (double)((va[r].x + vb[r].x) + (va[r + 1].x + vb[r + 1].x));
                sy += (double)((va[r].y + vb[r].y) + (va[r + 1].y + vb[r + 1].y));
            }
            win_issue(0);
            win_issue(1);
            sx = wave_sum_f64(sx);
            sy = wave_sum_f64(sy);
            if ((opaque(tid) & 63) == 0) msum[opaque(tid) >> 6] = make_double2(sx, sy);
            lds_barrier();      // (also: every wave is through with the exchanges of the segment before)
            double tx = 0.0, ty = 0.0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const double2 t = msum[w];
                tx += t.x;
                ty += t.y;
            }
            tx *= 1.0 / W32_N;
            ty *= 1.0 / W32_N;
            mhi = make_float2((float)tx, (float)ty);
            mlo = make_float2((float)(tx - (double)mhi.x), (float)(ty - (double)mhi.y));
        } else {
            win_issue(0);
            win_issue(1);
        }
        float2 d1 = dtab[opaque(tid)];
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) {
            // batch bt + 1 is in flight behind this one (bt < 3): eight younger loads
            if (bt < 3) vm_arrived8_keep8(wq[bt & 1]);
            else vm_arrived8(wq[bt & 1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * bt + i;
                asm volatile("" : "+v"(d1.x), "+v"(d1.y));
                float2 y0 = va[r], y1 = vb[r];
                if (DETREND) {
                    y0 = make_float2((y0.x - mhi.x) - mlo.x, (y0.y - mhi.y) - mlo.y);
                    y1 = make_float2((y1.x - mhi.x) - mlo.x, (y1.y - mhi.y) - mlo.y);
                }
                y0 = make_float2(y0.x * wq[bt & 1][i], y0.y * wq[bt & 1][i]);
                y1 = make_float2(y1.x * wq[bt & 1][4 + i], y1.y * wq[bt & 1][4 + i]);
                va[r] = cadd(y0, y1);
                const float2 tw = cmul(d1, make_float2(W32_RE[r], W32_IM[r]));
                vb[r] = cmul(csub(y0, y1), tw);
            }
            if (bt < 2) win_issue(bt + 2);
        }
        transform(va, accA, !DETREND);
        transform(vb, accB, true);
    }

    float *dst = p.partial + (size_t)b * W32_N + tid;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        dst[1024 * k2] = accA[k2];
        dst[W32_M + 1024 * k2] = accB[k2];
    }
}

}  // namespace

int welch32k_rows(long long nseg, int cus) {
    if (cus < 1) cus = 256;
    return (int)(nseg < cus ? (nseg < 1 ? 1 : nseg) : cus);
}

hipError_t launch_welch32k(const W32kArgs &a, int W, hipStream_t s) {
    static bool armed[64] = {};        // 136 KiB of dynamic LDS needs the opt-in, once per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!armed[dev]) {
        hipError_t e = hipSuccess;
        for (const void *fn : {reinterpret_cast<const void *>(welch32k_kernel<true>), reinterpret_cast<const void *>(welch32k_kernel<false>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W32_LDS_BYTES);
        if (e != hipSuccess) return e;
        armed[dev] = true;
    }
    if (a.detrend) hipLaunchKernelGGL(welch32k_kernel<true>, dim3(W), dim3(1024), W32_LDS_BYTES, s, a);
    else hipLaunchKernelGGL(welch32k_kernel<false>, dim3(W), dim3(1024), W32_LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace oth
