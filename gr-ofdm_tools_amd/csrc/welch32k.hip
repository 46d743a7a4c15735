// welch32k: segment-averaged |FFT_32768((x - mean) w)|^2 with the WHOLE segment inside one workgroup - and the 65536-point
// form on top of it (a pair of workgroups per segment; see `FRONT` at the kernel) - the first lengths
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
// samples (groups of four in float, the groups in double), the waves' totals meet in LDS behind one workgroup barrier, the mean is subtracted as a float pair
// (hi + lo).  That barrier also covers the exchange-A hand-over of the first transform: four workgroup barriers per segment.
// The twiddle seeds live in LDS (153.4 of 160 KiB with the exchange regions).
//
// Measured and NOT kept: the detrend after the transform (pilot off every sample, the waves' sums to LDS without a barrier, the
// residual mean taken off the 32 bins a cosine-sum window's spectrum reaches, as the tuned kernels do) - parity green, 1 %
// SLOWER same-box (0.505-0.509 against 0.499-0.502 ms): the time the early waves spend at the mean's barrier moves to the next
// barrier, it is not idle time of the SIMDs.
//
// Wave priorities are the 16384-point kernels' (raised around memory / LDS issue, lowered for the butterflies).  A FIXED priority
// per wave by its rank on the SIMD - to even out who reaches a barrier first - is 3.6 x SLOWER (1.77 against 0.49 ms per 2^27
// samples): the low ranks starve and every barrier waits for them.
//
// Samples are read with ordinary (cached) loads: at 50 % overlap every sample is wanted by two segments, and the segments of
// one round are dealt out so that neighbours run on the same XCD (workgroup b runs on XCD b % 8: it takes slot
// (b % 8) (W / 8) + b / 8) - the second reader finds the half in that XCD's L2.
#include <type_traits>
#include "fft16k.hip.h"

namespace oth {
namespace {

constexpr int W32_M = 16384, W32_N = 32768;
// -DW32_DIAG=1 (make EXP=1 EXTRA=-DW32_DIAG=1; tools/w32_phases.py): cycle counts of eight phases of a step per wave, written
// behind the partial rows (1 KiB per workgroup)
#ifndef W32_DIAG
#define W32_DIAG 0
#endif
#if W32_DIAG
#define W32_STAMP(i)                                                     \
    do {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        __builtin_amdgcn_s_waitcnt(0xC07F);                              \
        phase[i] += now_ - last_;                                        \
        last_ = now_;                                                    \
        __builtin_amdgcn_sched_barrier(0);                               \
    } while (0)
#else
#define W32_STAMP(i) do { } while (0)
#endif
// LDS behind the sixteen exchange regions: the waves' sample totals and the twiddle seeds - W_32768^tid of every thread
// (radix-2 step and, squared, pass 1), W_65536^tid (the front step of the 65536-point form), W and W^4 of passes 2 and 3 per
// lane (W_1024^l, W_64^q) - so that the transforms wait for no vector-memory load.  Each pass rebuilds its other powers from
// W and W^4 (thirteen products, 52 instructions).  Measured and dropped: all fifteen twiddles of passes 2 and 3 from LDS
// tables - 8 % fewer VALU instructions, 3 % SLOWER same-box (0.562 against 0.545 ms per 2^27 samples: thirty more LDS round
// trips per transform); requests for the next segment's cache lines issued under the transforms (two or four dwords per thread
// whose values are never used) - 22.7 % of the roofline against 24.7 % without them.
constexpr size_t W32_LDS_BYTES = 16 * XREG * sizeof(float2) + 16 * sizeof(double2) + 2 * 1024 * sizeof(float2) + 64 * sizeof(float4) +
                                 4 * sizeof(float4);
static_assert(W32_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

// exp(-2 pi i r / 32), r = 0..15
constexpr float W32_RE[16] = {1.0f, 0.98078528040323044f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                              0.55557023301960222f, 0.38268343236508977f, 0.19509032201612827f, 0.0f, -0.19509032201612827f,
                              -0.38268343236508977f, -0.55557023301960222f, -0.70710678118654752f, -0.83146961230254524f,
                              -0.92387953251128674f, -0.98078528040323044f};
constexpr float W32_IM[16] = {-0.0f, -0.19509032201612827f, -0.38268343236508977f, -0.55557023301960222f, -0.70710678118654752f,
                              -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323044f, -1.0f, -0.98078528040323044f,
                              -0.92387953251128674f, -0.83146961230254524f, -0.70710678118654752f, -0.55557023301960222f,
                              -0.38268343236508977f, -0.19509032201612827f};

// exp(-2 pi i j / 64), j = 0..31 (the front step of the 65536-point form)
constexpr float W64_RE[32] = {1.0f, 0.99518472667219693f, 0.98078528040323044f, 0.95694033573220882f, 0.92387953251128674f,
                              0.88192126434835503f, 0.83146961230254524f, 0.77301045336273699f, 0.70710678118654752f,
                              0.63439328416364549f, 0.55557023301960222f, 0.47139673682599764f, 0.38268343236508977f,
                              0.29028467725446233f, 0.19509032201612827f, 0.09801714032956060f, 0.0f, -0.09801714032956060f,
                              -0.19509032201612827f, -0.29028467725446233f, -0.38268343236508977f, -0.47139673682599764f,
                              -0.55557023301960222f, -0.63439328416364549f, -0.70710678118654752f, -0.77301045336273699f,
                              -0.83146961230254524f, -0.88192126434835503f, -0.92387953251128674f, -0.95694033573220882f,
                              -0.98078528040323044f, -0.99518472667219693f};
constexpr float W64_IM[32] = {-0.0f, -0.09801714032956060f, -0.19509032201612827f, -0.29028467725446233f, -0.38268343236508977f,
                              -0.47139673682599764f, -0.55557023301960222f, -0.63439328416364549f, -0.70710678118654752f,
                              -0.77301045336273699f, -0.83146961230254524f, -0.88192126434835503f, -0.92387953251128674f,
                              -0.95694033573220882f, -0.98078528040323044f, -0.99518472667219693f, -1.0f, -0.99518472667219693f,
                              -0.98078528040323044f, -0.95694033573220882f, -0.92387953251128674f, -0.88192126434835503f,
                              -0.83146961230254524f, -0.77301045336273699f, -0.70710678118654752f, -0.63439328416364549f,
                              -0.55557023301960222f, -0.47139673682599764f, -0.38268343236508977f, -0.29028467725446233f,
                              -0.19509032201612827f, -0.09801714032956060f};

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
// the front step's batch of four rows: x[n] (in place in z), x[n + 32768], w[n], w[n + 32768] - sixteen loads; KEEP = the loads
// of the next batch already issued behind it (16) or none
template <int KEEP>
__device__ __forceinline__ void vm_arrived_front(f2v &z0, f2v &z1, f2v &z2, f2v &z3, f2v (&x2)[4], float (&w1)[4], float (&w2)[4]) {
    static_assert(KEEP == 0 || KEEP == 16 || KEEP == 32 || KEEP == 48, "whole batches of sixteen loads");
#define W32_FRONT_WAIT(n)                                                                                                                  \
    asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3), "+v"(x2[0]), "+v"(x2[1]), "+v"(x2[2]), "+v"(x2[3]), \
                 "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]), "+v"(w2[0]), "+v"(w2[1]), "+v"(w2[2]), "+v"(w2[3]) : : "memory")
    if (KEEP == 48) W32_FRONT_WAIT(48);
    else if (KEEP == 32) W32_FRONT_WAIT(32);
    else if (KEEP == 16) W32_FRONT_WAIT(16);
    else W32_FRONT_WAIT(0);
#undef W32_FRONT_WAIT
}

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

// FRONT: the 65536-point form.  A 512 KiB segment does not fit a CU, but one more radix-2 step (decimation in frequency) splits
// it into two INDEPENDENT 32768-point problems - the even bins are FFT_32768(y[n] + y[n + 32768]), the odd bins
// FFT_32768((y[n] - y[n + 32768]) W_65536^n) - and a PAIR of workgroups takes one each (h = 0 / 1): both read the whole
// segment (the second reader out of L2: the pair sits on one XCD), form their 32768 points in registers and go on as the
// 32768-point kernel does; nothing is exchanged between them.  The mean is only known when all 65536 samples have passed, so
// the front step forms z0 = x w +- x' w' first and takes m (w +- w') off behind the barrier (p.wpm: that table, built by the
// plan in double) - the rounding of the two forms differs by ~1e-7 of the OFFSET per sample, incoherent from sample to sample.
template <bool DETREND, bool FRONT = false>
__global__ __launch_bounds__(1024) void welch32k_kernel(W32kArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    double2 *msum = reinterpret_cast<double2 *>(lds + 16 * XREG);
    float2 *dtab = reinterpret_cast<float2 *>(msum + 16);      // W_32768^tid
    float2 *etab = dtab + 1024;                                // W_65536^tid (FRONT)
    float4 *btab = reinterpret_cast<float4 *>(etab + 1024);    // (W_1024^l, W_1024^(4 l))
    float4 *ctab = btab + 64;                                  // (W_64^q, W_64^(4 q))
    constexpr int TWS = FRONT ? 2 : 1;                         // p.tw[k] = W_L^k, L = 32768 TWS

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    // FRONT: workgroups (slot, h = 0 / 1) of a pair on one XCD (workgroup b runs on XCD b % 8) when the grid allows
    const int G = gridDim.x, W = FRONT ? G >> 1 : G;           // W: segments in flight = partial rows
    int slot, h = 0;
    if (FRONT) {
        if (G & 15) slot = b >> 1, h = b & 1;
        else slot = (b & 7) * (G >> 4) + (b >> 4), h = (b >> 3) & 1;
    } else {
        slot = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
    }

    // Nothing but the thread index, the thirty-two sums and (32768 points) the sixteen rows requested ahead is carried around the
    // segment loop in vector registers: the lane's LDS addresses, the twiddle seeds (from the LDS tables filled below) and the
    // window rows are formed again where they are used, from values the compiler cannot see through (`opaque`) - it would
    // otherwise hoist them out of the loop and spill them.
    auto opaque = [](int v) {
        asm volatile("" : "+v"(v));
        return v;
    };

    {
        const int t = threadIdx.x;
        dtab[t] = p.tw[TWS * t];
        etab[t] = p.tw[t];
        if (t < 64) {
            const float2 e = p.tw[TWS * 32 * t], f = p.tw[TWS * 128 * t];
            btab[t] = make_float4(e.x, e.y, f.x, f.y);
        }
        if (t < 4) {
            const float2 e = p.tw[TWS * 512 * t], f = p.tw[TWS * 2048 * t];
            ctab[t] = make_float4(e.x, e.y, f.x, f.y);
        }
        __syncthreads();
    }
    float accA[16], accB[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) accA[k] = accB[k] = 0.f;
#if W32_DIAG
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif

    // one 16384-point transform of v (in place: thread (w, l) holds n = tid + 1024 r) added to acc; `handover`: a workgroup
    // barrier in front of the exchange-A writes (every wave through with the reads of the transform before)
    auto transform = [&](float2 (&v)[16], float (&acc)[16], bool handover, int ph) {
        const int t = opaque(tid), wv = t >> 6, l = t & 63, g = l >> 2, q = l & 3;
        float2 *wa = lds + t;                                // exchange A write: + XREG k0 (k0 < 8), wa8 + XREG (k0 - 8)
        float2 *wa8 = wa + opaque(8 * XREG);                 // (a VALUE the compiler cannot fold; the pointer stays an LDS pointer)
        const float2 *ra = lds + XREG * wv + l;              // exchange A read:  + 64 w
        float2 *wb = lds + XREG * wv + l;                    // exchange B write: + XROW k1
        const float2 *rb = lds + XREG * wv + XROW * g + q;   // exchange B read:  + 4 j
        prio_compute();
        dft16(v);                                              // pass 1: r -> k0
        prio_latency();
        W32_STAMP(ph);
        if (handover) lds_barrier();
        {
            const float2 d = dtab[t], a1 = cmul(d, d), a2 = cmul(a1, a1);      // W_M^tid = (W_N^tid)^2 and its fourth power
            scatter_pow16_exa(v, wa, wa8, a1, cmul(a2, a2));                   // x W_M^(k0 tid) -> [k0][w][l]
        }
        lds_barrier();
        W32_STAMP(ph + 1);
        dft16_from_lds<64>(v, ra, [] { prio_compute(); });     // pass 2: w -> k1
        prio_latency();
        wave_lds_sync();
        {
            const float4 e = btab[l];
            scatter_pow16<XROW>(v, wb, make_float2(e.x, e.y), make_float2(e.z, e.w));      // x W_1024^(k1 l) -> row k1, column l of this wave's region
        }
        wave_lds_sync();
        dft16_from_lds<4>(v, rb, [] { prio_compute(); });      // pass 3: g -> k2
        {
            const float4 e = ctab[q];
            twiddle_pow16_inplace(v, make_float2(e.x, e.y), make_float2(e.z, e.w));      // x W_64^(k2 q)
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
        W32_STAMP(ph + 2);
    };

    // The segment loop, specialised on the half a FRONT workgroup takes (a run-time branch around the odd half's twiddle inside the
    // loop made the compiler park 25 registers in scratch in front of it)
    auto run = [&](auto half_) {
    constexpr int H = decltype(half_)::value;
    // (!FRONT) The second half of a segment - rows x[n + M] - is requested one transform ahead: while the second transform of
    // the step before runs, the registers of its first transform's data are free.  The phase stamps (tools/w32_phases.py)
    // showed 40 % of a step between the first load and the mean's barrier with all 32 rows requested at the top.
    f2v pre[16];
    auto prefetch = [&](long long s2) {
        const long long o2 = p.first + s2 * p.step;
        const float2 *x2 = p.x + (((long long)__builtin_amdgcn_readfirstlane((int)(o2 >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)o2));
        const unsigned u8 = ((unsigned)opaque(tid) & 1023u) * 8u;
#pragma unroll
        for (int r = 0; r < 16; ++r) load_row8(pre[r], u8, x2 + W32_M + 1024 * r);
    };
    if (!FRONT && slot < p.nseg) prefetch(slot);
    for (long long s = slot; s < p.nseg; s += W) {
        const long long off = p.first + s * p.step;
        const float2 *xs = p.x + (((long long)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)off));
        const unsigned ut8 = ((unsigned)opaque(tid) & 1023u) * 8u, ut4 = ut8 >> 1;
        const float *wn = p.win;
        asm volatile("" : "+s"(wn));      // (or thirty-two row bases stay in scalar registers across the loop)
        float2 va[16], vb[16];
        prio_latency();
        if constexpr (!FRONT) {
        {
            f2v la[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) load_row8(la[r], ut8, xs + 1024 * r);
            vm_arrived16(la);
            vm_arrived16(pre);
            W32_STAMP(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                va[r] = make_float2(la[r].x, la[r].y);
                vb[r] = make_float2(pre[r].x, pre[r].y);
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
            W32_STAMP(1);
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
        } else {
            // ---- front step: z[j] at n = tid + 1024 j, j < 32.  Batches of four rows, the next one in flight while one is
            // used as long as registers allow (the finished z stay: 8 registers per batch)
            // The thirty-two sums wait in LDS while the front step needs their registers: the wave's OWN exchange region (8.5 KiB;
            // nobody else touches it between this wave's last pass-3 read and the exchange-A writes of the next transform,
            // which every wave issues behind a barrier this wave reaches after `unpark`).  Left to the compiler they went to
            // scratch: 1.7 GB of writes per 2^27 samples.
            float4 *park = reinterpret_cast<float4 *>(lds + XREG * (opaque(tid) >> 6)) + (opaque(tid) & 63);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                park[64 * i] = make_float4(accA[4 * i], accA[4 * i + 1], accA[4 * i + 2], accA[4 * i + 3]);
                park[64 * (4 + i)] = make_float4(accB[4 * i], accB[4 * i + 1], accB[4 * i + 2], accB[4 * i + 3]);
            }
            asm volatile("" ::: "memory");
            f2v z[32], x2[4][4];
            float w1[4][4], w2[4][4];
            constexpr float sgn = H ? -1.0f : 1.0f;
            auto issue = [&](int bt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = 4 * bt + i;
                    load_row8(z[j], ut8, xs + 1024 * j);
                    load_row8(x2[bt & 3][i], ut8, xs + W32_N + 1024 * j);
                    load_row4(w1[bt & 3][i], ut4, wn + 1024 * j);
                    load_row4(w2[bt & 3][i], ut4, wn + W32_N + 1024 * j);
                }
            };
            // The sums are parked, so early in the front step almost every register is free: four batches go out at once, and a
            // batch is issued as soon as the finished z (8 registers per batch) and the batches in flight (24 each) leave room:
            //   used up:    0      1      2      3      4      5      6      7
            //   in flight:  1-3    2-4    3,4    4,5    5,6    6      7      -
            double sx = 0.0, sy = 0.0;
            issue(0);
            issue(1);
            issue(2);
            issue(3);
            // DETREND: a pilot - the segment's first sample, a scalar load - comes off every sample as it arrives, so that the
            // products z0 carry no offset whose float rounding (1e-7 of the OFFSET per sample) would stay behind when the
            // mean is taken off them afterwards; the mean below is then the small residual mean(x - pilot).  Every workgroup
            // that touches the segment reads the same sample: same bits.
            float2 pv = make_float2(0.f, 0.f);
            if (DETREND) pv = xs[0];
#pragma unroll
            for (int bt = 0; bt < 8; ++bt) {
                {
                    f2v &z0 = z[4 * bt], &z1 = z[4 * bt + 1], &z2 = z[4 * bt + 2], &z3 = z[4 * bt + 3];
                    if (bt <= 1) vm_arrived_front<48>(z0, z1, z2, z3, x2[bt & 3], w1[bt & 3], w2[bt & 3]);
                    else if (bt <= 4) vm_arrived_front<32>(z0, z1, z2, z3, x2[bt & 3], w1[bt & 3], w2[bt & 3]);
                    else if (bt <= 6) vm_arrived_front<16>(z0, z1, z2, z3, x2[bt & 3], w1[bt & 3], w2[bt & 3]);
                    else vm_arrived_front<0>(z0, z1, z2, z3, x2[bt & 3], w1[bt & 3], w2[bt & 3]);
                }
                if (DETREND) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        z[4 * bt + i].x -= pv.x, z[4 * bt + i].y -= pv.y;
                        x2[bt & 3][i].x -= pv.x, x2[bt & 3][i].y -= pv.y;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const int j = 4 * bt + i;
                    if (DETREND) {
                        sx += (double)((z[j].x + x2[bt & 3][i].x) + (z[j + 1].x + x2[bt & 3][i + 1].x));
                        sy += (double)((z[j].y + x2[bt & 3][i].y) + (z[j + 1].y + x2[bt & 3][i + 1].y));
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = 4 * bt + i;
                    const float a = w1[bt & 3][i], c = w2[bt & 3][i] * sgn;
                    z[j].x = fmaf(x2[bt & 3][i].x, c, z[j].x * a);
                    z[j].y = fmaf(x2[bt & 3][i].y, c, z[j].y * a);
                    asm volatile("" : "+v"(z[j]));      // the batch is used up HERE: arithmetic may sink below the next loads otherwise,
                }                                       // and the batch waits for it in scratch
                if (DETREND) asm volatile("" : "+v"(sx), "+v"(sy));
                if (bt == 0) issue(4);
                if (bt == 2 || bt == 3) issue(bt + 3);
                if (bt == 5) issue(7);
            }
            {      // unpark
                float4 *back = reinterpret_cast<float4 *>(lds + XREG * (opaque(tid) >> 6)) + (opaque(tid) & 63);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 a = back[64 * i], c = back[64 * (4 + i)];
                    accA[4 * i] = a.x, accA[4 * i + 1] = a.y, accA[4 * i + 2] = a.z, accA[4 * i + 3] = a.w;
                    accB[4 * i] = c.x, accB[4 * i + 1] = c.y, accB[4 * i + 2] = c.z, accB[4 * i + 3] = c.w;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(accA[i]), "+v"(accB[i]));
            }
            if (DETREND) {
                // m (w[n] +- w[n + 32768]) off every z: the table in batches of four, two in flight
                float cq[2][4];
                const float *wpm = p.wpm + (H ? W32_N : 0);
                asm volatile("" : "+s"(wpm));
                auto cissue = [&](int bt) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) load_row4(cq[bt & 1][i], ut4, wpm + 1024 * (4 * bt + i));
                };
                cissue(0);
                cissue(1);
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
                tx *= 1.0 / (2 * W32_N);
                ty *= 1.0 / (2 * W32_N);
                const float2 mhi = make_float2((float)tx, (float)ty);
                const float2 mlo = make_float2((float)(tx - (double)mhi.x), (float)(ty - (double)mhi.y));
#pragma unroll
                for (int bt = 0; bt < 8; ++bt) {
                    if (bt < 7) asm volatile("s_waitcnt vmcnt(4)" : "+v"(cq[bt & 1][0]), "+v"(cq[bt & 1][1]), "+v"(cq[bt & 1][2]), "+v"(cq[bt & 1][3]) : : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(cq[bt & 1][0]), "+v"(cq[bt & 1][1]), "+v"(cq[bt & 1][2]), "+v"(cq[bt & 1][3]) : : "memory");
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 4 * bt + i;
                        const float c = cq[bt & 1][i];
                        z[j].x = fmaf(-mlo.x, c, fmaf(-mhi.x, c, z[j].x));
                        z[j].y = fmaf(-mlo.y, c, fmaf(-mhi.y, c, z[j].y));
                        asm volatile("" : "+v"(z[j]));
                    }
                    if (bt < 6) cissue(bt + 2);
                }
            }
            if constexpr (H) {      // the odd bins' half: x W_65536^n, n = tid + 1024 j
                float2 e1 = etab[opaque(tid)];
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    asm volatile("" : "+v"(e1.x), "+v"(e1.y));
                    const float2 r = cmul(make_float2(z[j].x, z[j].y), cmul(e1, make_float2(W64_RE[j], W64_IM[j])));
                    z[j].x = r.x, z[j].y = r.y;
                }
            }
            // the 32768-point radix-2 step on z: rows r and 16 + r
            float2 d1 = dtab[opaque(tid)];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                asm volatile("" : "+v"(d1.x), "+v"(d1.y));
                const float2 y0 = make_float2(z[r].x, z[r].y), y1 = make_float2(z[16 + r].x, z[16 + r].y);
                va[r] = cadd(y0, y1);
                vb[r] = cmul(csub(y0, y1), cmul(d1, make_float2(W32_RE[r], W32_IM[r])));
            }
        }
        W32_STAMP(2);
        transform(va, accA, !DETREND, 3);
        if (!FRONT) prefetch(s + W < p.nseg ? s + W : s);      // (no branch: the last step asks for its own rows again)
        transform(vb, accB, true, 5);
    }
    };
    if (FRONT && h) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});

#if W32_DIAG
    if ((threadIdx.x & 63) == 0) {      // phases: 0 loads, 1 sums + mean barrier, 2 window + radix 2, 3 pass 1 a, 4 exch A + barrier a, 5 rest of a,
                                        // (5 also: pass 1 b), 6 handover barrier + exch A + barrier b, 7 rest of b
        unsigned long long *st = reinterpret_cast<unsigned long long *>(p.partial + (size_t)W * (FRONT ? 2 * W32_N : W32_N)) +
                                 128 * (size_t)b + 8 * (threadIdx.x >> 6);
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = phase[i];
    }
#endif
    // FRONT: row `slot` of [W][65536], half h; position p of that half holds bin 2 (layout-7 bin of p) + h (finalize layout 8)
    float *dst = p.partial + (FRONT ? (size_t)slot * (2 * W32_N) + (size_t)h * W32_N : (size_t)b * W32_N) + tid;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        dst[1024 * k2] = accA[k2];
        dst[W32_M + 1024 * k2] = accB[k2];
    }
}

}  // namespace

int welch32k_rows(long long nseg, int cus, bool front) {
    if (cus < 1) cus = 256;
    if (front) cus >>= 1;      // a pair of workgroups per 65536-point segment
    return (int)(nseg < cus ? (nseg < 1 ? 1 : nseg) : cus);
}

hipError_t launch_welch32k(const W32kArgs &a, int W, hipStream_t s) {
    static bool armed[64] = {};        // 153 KiB of dynamic LDS needs the opt-in, once per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!armed[dev]) {
        hipError_t e = hipSuccess;
        for (const void *fn : {reinterpret_cast<const void *>(welch32k_kernel<true, false>), reinterpret_cast<const void *>(welch32k_kernel<false, false>),
                               reinterpret_cast<const void *>(welch32k_kernel<true, true>), reinterpret_cast<const void *>(welch32k_kernel<false, true>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W32_LDS_BYTES);
        if (e != hipSuccess) return e;
        armed[dev] = true;
    }
    if (a.front) {
        if (!a.wpm && a.detrend) return hipErrorInvalidValue;
        if (a.detrend) hipLaunchKernelGGL((welch32k_kernel<true, true>), dim3(2 * W), dim3(1024), W32_LDS_BYTES, s, a);
        else hipLaunchKernelGGL((welch32k_kernel<false, true>), dim3(2 * W), dim3(1024), W32_LDS_BYTES, s, a);
    } else {
        if (a.detrend) hipLaunchKernelGGL((welch32k_kernel<true, false>), dim3(W), dim3(1024), W32_LDS_BYTES, s, a);
        else hipLaunchKernelGGL((welch32k_kernel<false, false>), dim3(W), dim3(1024), W32_LDS_BYTES, s, a);
    }
    return hipGetLastError();
}

}  // namespace oth
