// welch16k1x: segment-averaged |FFT_16384(x * w)|^2 for non-overlapping 16384-point vectors - BASELINE config 5, the
// multichannel_scanner chain (python/multichannel_scanner.py:78-86: stream_to_vector -> fft_vcc(rect, shift) ->
// |.|^2 -> 1/N^2) averaged over the kept vectors of each channel stream; also any scipy.signal.welch call with
// nperseg = nfft = 16384, detrend=False and no overlap.
//
// One 1024-thread workgroup (16 waves) per segment, 16 points per thread, N = 16 x 16 x 16 x 4 decimation in
// frequency with the index bits placed so that only ONE of the three exchanges crosses waves:
//
//   n = tid + 1024 r           tid = 64 w + l (wave w, lane l),  l = 4 g + q
//   k = k0 + 16 k1 + 256 k2 + 4096 k3
//   pass 1  thread (w, l) holds r = 0..15          -> k0,  x W_N^(k0 tid)
//   exch A  LDS [k0][w][l]: thread (w, l) -> thread (wave k0, lane l)       the only cross-wave exchange
//   pass 2  thread (k0, l) holds w = 0..15         -> k1,  x W_1024^(k1 l)
//   exch B  inside wave k0's own 8.5 KiB region: lane (g, q) -> lane (k1, q)   wave-level ordering only
//   pass 3  thread (k0, k1, q) holds g = 0..15     -> k2,  x W_64^(k2 q)
//   pass 4  radix 4 over q, the four lanes of a quad, through DPP inside the multiply-adds -> k3 (no LDS)
//
// welch16k_kernel (4 x 4096: radix-4 pass, scatter to four sub-FFT images, then the 4096 scheme) moves every point
// through LDS three times, two of them across waves, behind four workgroup barriers per segment; with one
// workgroup per CU (139 KiB image) a barrier idles the whole CU, and its counters showed that (profiles/r03_C5.txt:
// VALU issue 42 %, waves waiting 38 % of their time at 56 % of the HBM roofline).  Here a point crosses LDS twice, two
// workgroup barriers per segment (around the exchange-A writes), and between them the waves run unsynchronised from
// the exchange-A reads to the next segment's pass 1, so their LDS, VALU and global-load phases interleave.
//
// LDS: 16 wave regions of 16 rows x 68 float2 (exchange B pads its rows by 4: lanes (k1, q) of a 32-lane read group
// then cover 32 distinct 8-byte slots; exchange A uses the first 1024 slots of a region, conflict-free as it is:
// consecutive lanes, consecutive slots) = 136 KiB, one workgroup per CU.
// Only |X|^2 leaves pass 4, so a bin may come out multiplied by -1 or -i: lanes 2, 3 of a quad carry the negated odd
// half, which saves the lane-dependent add / subtract.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "fft16k.hip.h"

#ifndef OTH_X1_PPLACES
#define OTH_X1_PPLACES 0x22222222u     // the same for the pipelined kernel (places: see its `spread`)
#endif
#ifndef OTH_X1H_FAKEWIN
#define OTH_X1H_FAKEWIN 0
#endif
#ifndef OTH_X1H_WIN_EARLY
#define OTH_X1H_WIN_EARLY 0  // 50 %-overlap kernel, A/B: 1 = the window values of a step are requested at the end of the step before (no gain: 16384 points 0.3874-0.3907 against 0.3855-0.3876 ms, 8192 points 0.3694-0.3753 against 0.3789 ms on the builds without a pilot, and the PILOT builds then spill 16 registers)
#endif
#ifndef OTH_X1_DIAG
#define OTH_X1_DIAG 0        // 1: per-wave phase cycle counters behind the partial sums (tools/archive/diag_x1.py)
#endif
#if OTH_X1_DIAG
#define X1_STAMP(i)                                                      \
    do {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        __builtin_amdgcn_s_waitcnt(0xC07F);                              \
        phase[i] += now_ - last_;                                        \
        last_ = now_;                                                    \
        __builtin_amdgcn_sched_barrier(0);                               \
    } while (0)
#define X1_DRAIN_VM() __builtin_amdgcn_s_waitcnt(0x0F70)
#define X1_DRAIN_LGKM() __builtin_amdgcn_s_waitcnt(0xC07F)
#else
#define X1_STAMP(i) do { } while (0)
#define X1_DRAIN_VM() do { } while (0)
#define X1_DRAIN_LGKM() do { } while (0)
#endif

namespace oth {
namespace {

constexpr int X1_RED = 8;                      // chunk tickets (two slots, by chunk parity)
// NW = waves per workgroup = N / 1024: 16 (N = 16384, one workgroup per CU) or 8 (N = 8192, two per CU; round 5)
template <int NW = 16> constexpr size_t x1_lds_bytes() { return (NW * XREG + X1_RED) * sizeof(float2); }
// the pipelined kernel adds its twiddle tables (passes 2 and 3) and the quad-butterfly constants
template <int NW = 16> constexpr size_t x1p_lds_bytes() {
    return x1_lds_bytes<NW>() + (16 * 64 + 16 * 4) * sizeof(float2) + 4 * sizeof(float4);
}

// The 8-wave form (N = 8192 = 16 x 8 x 16 x 4): pass 1 still leaves sixteen k0 per thread, but there are only eight
// waves to take them, so wave k0' takes k0 = 2 k0' and 2 k0' + 1 - sixteen exchange-A rows [k0 & 1][w] of 64 lanes in its
// region - and pass 2 is TWO radix-8 butterflies over w (register 8 h + k1, natural order) instead of one radix-16; the
// exchange-B row of a register is c = 8 h + k1, lane (c, q) then holds the sixteen g of (k0 = 2 k0' + h, k1, q) and passes
// 3 and 4 run as at 16384 points.  Bin k0 + 16 k1 + 128 k2 + 2048 bitrev2(q) sits at 512 k2 + tid (finalize layout 5).
//   exa_off: float2 offset of pass-1 output k0 of thread tid relative to lds + tid
//   p2reg:   register that holds pass-2 output (= table row) k
template <int NW> __host__ __device__ constexpr int exa_off(int k) { return NW == 16 ? XREG * k : XREG * (k >> 1) + 512 * (k & 1); }
template <int NW> __host__ __device__ constexpr int p2reg(int k) { return NW == 16 ? r16(k) : k; }

// Non-temporal 8-byte load at (uniform row base in scalar registers) + (lane offset): one lane-offset register serves
// all sixteen rows of a segment, and the compiler can neither move the load nor gather it with its neighbours - the
// next segment's loads are spread over the step on purpose.  The caller waits (s_waitcnt vmcnt) before the first use.
typedef float f2v __attribute__((ext_vector_type(2)));      // a register pair the inline asm below can name as one operand
__device__ __forceinline__ void load_row_nt(f2v &dst, unsigned lane_off, const char *row) {
    asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(dst) : "v"(lane_off), "s"(row) : "memory");
}

// ---- waiting for pinned loads ------------------------------------------------------------------------------------------
// A register written by an inline-asm load is NOT valid until the s_waitcnt that covers it, and the compiler does not
// know: to it the value exists from the asm on.  Two things follow (tools/isa_async_hazard.py checks the shipped code
// objects for both, tests/test_abi_cpu.py runs it):
//   * a wait has to NAME the registers it makes valid, or nothing keeps an instruction that reads them behind it (the
//     bare `s_waitcnt lgkmcnt(0)` in front of the last segment's tail was overtaken by the sixty instructions of its
//     first butterfly layer - late round 5; it only worked because a vmcnt wait happened to stand in between);
//   * naming them as in-out operands ("+v") is not enough when the value has to end up somewhere else (the kept half of
//     the 50 %-overlap builds): the compiler then copies the INPUT of the wait into the register it wants, i.e. reads
//     the loading register in front of the wait.  arrive8 / arrive16 take the loading registers as inputs only and hand
//     out fresh ones, wait and copies in one statement.
#define OTH_IO8(r) "+v"((r)[0]), "+v"((r)[1]), "+v"((r)[2]), "+v"((r)[3]), "+v"((r)[4]), "+v"((r)[5]), "+v"((r)[6]), "+v"((r)[7])
#define OTH_IO16(r) OTH_IO8(r), "+v"((r)[8]), "+v"((r)[9]), "+v"((r)[10]), "+v"((r)[11]), "+v"((r)[12]), "+v"((r)[13]), "+v"((r)[14]), "+v"((r)[15])
// every vector-memory load of this wave has landed; r[0..15] are valid behind this statement (in place)
__device__ __forceinline__ void vm_arrived16(f2v (&r)[16]) { asm volatile("s_waitcnt vmcnt(0)" : OTH_IO16(r) : : "memory"); }
// every LDS read of this wave has landed (with the workgroup barrier: lds_barrier() for a wave that holds reads in flight)
__device__ __forceinline__ void lds_arrived16(f2v (&r)[16]) { asm volatile("s_waitcnt lgkmcnt(0)" : OTH_IO16(r) : : "memory"); }
__device__ __forceinline__ void lds_barrier_arrived16(f2v (&r)[16]) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : OTH_IO16(r) : : "memory");
}
// wait, then out[i] = in[i] in fresh registers (eight pairs)
__device__ __forceinline__ void vm_arrive8(float2 (&out)[8], const f2v (&in)[8]) {
    f2v o[8];
    asm volatile("s_waitcnt vmcnt(0)\n\t"
                 "v_mov_b64 %0, %8\n\tv_mov_b64 %1, %9\n\tv_mov_b64 %2, %10\n\tv_mov_b64 %3, %11\n\t"
                 "v_mov_b64 %4, %12\n\tv_mov_b64 %5, %13\n\tv_mov_b64 %6, %14\n\tv_mov_b64 %7, %15"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                 : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]), "v"(in[4]), "v"(in[5]), "v"(in[6]), "v"(in[7])
                 : "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = make_float2(o[i].x, o[i].y);
}
// the sixteen window values of a step (load_win): valid in place behind this statement; they are used up inside the step,
// nothing of them lives across a loop edge
__device__ __forceinline__ void vm_arrived_win16(float (&w)[16]) { asm volatile("s_waitcnt vmcnt(0)" : OTH_IO16(w) : : "memory"); }

// The same two operations with six stored powers W, W^2, W^3, W^4, W^8, W^12 (nine products instead of thirteen)
struct Pow6x {
    float2 j1, j2, j3, i1, i2, i3;
};
__device__ __forceinline__ Pow6x pow6_load(const float2 *tw, int e) {
    return Pow6x{tw[e], tw[2 * e], tw[3 * e], tw[4 * e], tw[8 * e], tw[12 * e]};
}
template <int STRIDE, bool STORE>
__device__ __forceinline__ void twiddle6(float2 (&v)[16], float2 *out, const Pow6x &w) {
    float2 wj[4], wi[4];
    wj[1] = w.j1, wj[2] = w.j2, wj[3] = w.j3;
    wi[1] = w.i1, wi[2] = w.i2, wi[3] = w.i3;
    // every product below has a wj factor: making those opaque keeps the nine products out of loop-invariant registers
    asm volatile("" : "+v"(wj[1].x), "+v"(wj[1].y), "+v"(wj[2].x), "+v"(wj[2].y), "+v"(wj[3].x), "+v"(wj[3].y));
    if (STORE) out[0] = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const int i = k >> 2, j = k & 3;
        const float2 tw = (i == 0) ? wj[j] : ((j == 0) ? wi[i] : cmul(wi[i], wj[j]));
        const float2 r = cmul(v[r16(k)], tw);
        if (STORE) out[STRIDE * k] = r;
        else v[r16(k)] = r;
    }
}

// exchange-A writes of either form: out[exa_off<NW>(k)] = v[r16(k)] W^k from the six stored powers
template <int NW> __device__ __forceinline__ void twiddle6_exa(float2 (&v)[16], float2 *out, const Pow6x &w) {
    float2 wj[4], wi[4];
    wj[1] = w.j1, wj[2] = w.j2, wj[3] = w.j3;
    wi[1] = w.i1, wi[2] = w.i2, wi[3] = w.i3;
    asm volatile("" : "+v"(wj[1].x), "+v"(wj[1].y), "+v"(wj[2].x), "+v"(wj[2].y), "+v"(wj[3].x), "+v"(wj[3].y));
    out[0] = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const int i = k >> 2, j = k & 3;
        const float2 tw = (i == 0) ? wj[j] : ((j == 0) ? wi[i] : cmul(wi[i], wj[j]));
        out[exa_off<NW>(k)] = cmul(v[r16(k)], tw);
    }
}
// the same from two seeds (the 50 %-overlap kernel: no room for six)
template <int NW> __device__ __forceinline__ void scatter_pow16_exa(const float2 (&v)[16], float2 *out, float2 p1, float2 p4) {
    float2 wj[4], wi[4];
    wj[1] = p1;
    wi[1] = p4;
    asm volatile("" : "+v"(wj[1].x), "+v"(wj[1].y), "+v"(wi[1].x), "+v"(wi[1].y));
    wj[2] = cmul(wj[1], wj[1]);
    wj[3] = cmul(wj[2], wj[1]);
    wi[2] = cmul(wi[1], wi[1]);
    wi[3] = cmul(wi[2], wi[1]);
    out[0] = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const int i = k >> 2, j = k & 3;
        const float2 w = (i == 0) ? wj[j] : ((j == 0) ? wi[i] : cmul(wi[i], wj[j]));
        out[exa_off<NW>(k)] = cmul(v[r16(k)], w);
    }
}

// The plain loop: every phase of a segment in program order, its sixteen loads at the top.  The pipelined kernel below
// is the default for the scanner's rectangular window; this one takes windowed plans (sixteen window values in
// registers do not fit next to two segments in flight) and serves as the A/B reference ("plain").
template <bool WINDOW>
__global__ __launch_bounds__(1024) void welch16k1x_kernel(WelchArgs p) {
    constexpr int N = 16384;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    int *lnext = reinterpret_cast<int *>(lds + 16 * XREG);

    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63, g = l >> 2, q = l & 3;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const long long s0 = (p.nseg * wg) / W, s1 = (p.nseg * (wg + 1)) / W;
    const float2 *xb = p.x + (size_t)stream * p.stream_stride;

    // twiddle seeds W, W^4 of the three twiddled passes (p.tw[k] = W_N^k)
    const float2 a1 = p.tw[tid], a4 = p.tw[4 * tid];             // W_N^tid
    const float2 b1 = p.tw[16 * l], b4 = p.tw[64 * l];           // W_1024^l
    const float2 c1 = p.tw[256 * q], c4 = p.tw[1024 * q];        // W_64^q
    float win[WINDOW ? 16 : 1];
    if constexpr (WINDOW) {
#pragma unroll
        for (int r = 0; r < 16; ++r) win[r] = p.win[tid + 1024 * r];
    }
    // quad butterfly constants by lane
    const float qs1 = q < 2 ? 1.0f : -1.0f;
    const float qal = q == 0 ? 1.0f : (q == 1 ? -1.0f : 0.0f);
    const float qbe = q >= 2 ? 1.0f : 0.0f, qnbe = -qbe;

    float2 *wa = lds + tid;                           // exchange A write: + XREG k0
    const float2 *ra = lds + XREG * wv + l;           // exchange A read:  + 64 w
    float2 *wb = lds + XREG * wv + l;                 // exchange B write: + XROW k1
    const float2 *rb = lds + XREG * wv + XROW * g + q;   // exchange B read:  + 4 j   (this lane is (k1 = g, q))

    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;

    const int sched = p.sched;
    const long long nchunks = sched ? chunk_count(p) : 1;
    int par = 0;
#if OTH_X1_DIAG
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
    for (long long cur = sched ? wg : 0; cur < nchunks;) {
        long long sb = s0, se = s1;
        if (sched) chunk_range(p, cur, sb, se);
        for (long long s = sb; s < se; ++s) {
            float2 v[16];
            prio_latency();
            const float2 *xs = xb + s * p.step + tid;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = load_once(xs + 1024 * r);
            if constexpr (WINDOW) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = make_float2(v[r].x * win[r], v[r].y * win[r]);
            }
            X1_STAMP(7);
            X1_DRAIN_VM();
            X1_STAMP(0);
            prio_compute();
            dft16(v);                                              // pass 1: r -> k0
            prio_latency();
            X1_STAMP(1);
            lds_barrier();      // 1: every wave is through with the previous segment's exchanges (A reads, B in its region)
            X1_STAMP(2);
            if (sched == 2 && s == sb && tid == 0) lnext[par] = (int)atomicAdd(p.queue + stream, 1u);
            scatter_pow16<XREG>(v, wa, a1, a4);                    // x W_N^(k0 tid) -> [k0][w][l]
            X1_DRAIN_LGKM();
            X1_STAMP(3);
            lds_barrier();      // 2
            X1_STAMP(4);
            dft16_from_lds<64>(v, ra, [] { prio_compute(); });     // pass 2: w -> k1
            prio_latency();
            wave_lds_sync();
            scatter_pow16<XROW>(v, wb, b1, b4);                    // x W_1024^(k1 l) -> row k1, column l of this wave's region
            X1_DRAIN_LGKM();
            X1_STAMP(5);
            wave_lds_sync();
            dft16_from_lds<4>(v, rb, [] { prio_compute(); });      // pass 3: g -> k2
            twiddle_pow16_inplace(v, c1, c4);                      // x W_64^(k2 q)
            quad_dft4_dpp(v, qs1, qal, qbe, qnbe);                 // pass 4: q -> k3
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                const float2 X = v[r16(k2)];
                acc[k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[k2]));
            }
            X1_STAMP(6);
        }
        if (sched == 0) break;
        cur = (sched == 1) ? cur + W : (long long)W + lnext[par];
        par ^= 1;
    }

    // bin k0 + 16 k1 + 256 k2 + 4096 bitrev2(q) sits at 1024 k2 + 64 k0 + 4 k1 + q (finalize layout 4)
    float *dst = p.partial + ((size_t)stream * W + wg) * N + tid;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) dst[1024 * k2] = acc[k2];
#if OTH_X1_DIAG
    if (l == 0) {
        unsigned long long *st = reinterpret_cast<unsigned long long *>(p.partial + (size_t)p.nstreams * W * N) +
                                 128 * ((size_t)stream * W + wg) + 8 * wv;
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = phase[i];
    }
#endif
}

}  // namespace

// Pass twiddles from an LDS table instead of rebuilding them from two seeds (13 complex products = 52 VALU
// instructions per pass): tab[TS * k] = W^k for this lane, k = 1..15.  Two batches of reads, 8 + 7 values, so that at
// most sixteen registers are in flight: TW_READ_A is issued by the caller early enough to be there (behind a
// butterfly layer), the second batch costs one exposed LDS round trip - the other three waves of the SIMD fill it.
struct TwBatch {
    f2v w[8];
};
#define OTH_TW_RD(b, addr, TSB, k, i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"((b).w[i]) : "v"(addr), "n"((TSB) * (k)))
template <int TS> __device__ __forceinline__ void tw_read_a(TwBatch &b, const float2 *tab) {
    const unsigned addr = (unsigned)(unsigned long long)tab;
    OTH_TW_RD(b, addr, 8 * TS, 1, 0); OTH_TW_RD(b, addr, 8 * TS, 2, 1); OTH_TW_RD(b, addr, 8 * TS, 3, 2); OTH_TW_RD(b, addr, 8 * TS, 4, 3);
    OTH_TW_RD(b, addr, 8 * TS, 5, 4); OTH_TW_RD(b, addr, 8 * TS, 6, 5); OTH_TW_RD(b, addr, 8 * TS, 7, 6); OTH_TW_RD(b, addr, 8 * TS, 8, 7);
}
template <int TS> __device__ __forceinline__ void tw_read_b(TwBatch &b, const float2 *tab) {
    const unsigned addr = (unsigned)(unsigned long long)tab;
    OTH_TW_RD(b, addr, 8 * TS, 9, 0); OTH_TW_RD(b, addr, 8 * TS, 10, 1); OTH_TW_RD(b, addr, 8 * TS, 11, 2); OTH_TW_RD(b, addr, 8 * TS, 12, 3);
    OTH_TW_RD(b, addr, 8 * TS, 13, 4); OTH_TW_RD(b, addr, 8 * TS, 14, 5); OTH_TW_RD(b, addr, 8 * TS, 15, 6);
}
#undef OTH_TW_RD
// every LDS operation of this wave so far has completed; names the batch so that no use of it moves above the wait
__device__ __forceinline__ void tw_wait(TwBatch &b) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b.w[0]), "+v"(b.w[1]), "+v"(b.w[2]), "+v"(b.w[3]), "+v"(b.w[4]), "+v"(b.w[5]), "+v"(b.w[6]), "+v"(b.w[7])
                 :
                 : "memory");
}
// v[r16(k)] *= W^k from the table (batch A already issued into `a`); STORE: out[STRIDE * k] = the product instead
template <int TS, int STRIDE, bool STORE, int NW = 16>      // NW: which register holds row k (p2reg; pass 3 always r16)
__device__ __forceinline__ void twiddle_table16(float2 (&v)[16], float2 *out, const float2 *tab, TwBatch &a) {
    TwBatch b;
    tw_wait(a);
    tw_read_b<TS>(b, tab);
    if (STORE) out[0] = v[0];
#pragma unroll
    for (int k = 1; k <= 8; ++k) {
        const float2 r = cmul(v[p2reg<NW>(k)], make_float2(a.w[k - 1].x, a.w[k - 1].y));
        if (STORE) out[STRIDE * k] = r;
        else v[p2reg<NW>(k)] = r;
    }
    tw_wait(b);
#pragma unroll
    for (int k = 9; k < 16; ++k) {
        const float2 r = cmul(v[p2reg<NW>(k)], make_float2(b.w[k - 9].x, b.w[k - 9].y));
        if (STORE) out[STRIDE * k] = r;
        else v[p2reg<NW>(k)] = r;
    }
}

// Pass 2 on the sixteen exchange-A rows of this wave's region (base[64 i]): one radix-16 butterfly over w at 16 waves,
// two radix-8 butterflies (rows 0..7: k0 even, rows 8..15: k0 odd; natural order in and out) at 8 waves.
template <int NW, class F, class M>
__device__ __forceinline__ void pass2_from_lds(float2 (&v)[16], const float2 *base, F issued, M mid) {
    if constexpr (NW == 16) {
        dft16_from_lds<64>(v, base, issued, mid);
    } else {
        const unsigned addr = (unsigned)(unsigned long long)base;
        double r[16];
#define OTH_LDS_READ(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(8 * 64 * (i)))
        OTH_LDS_READ(0); OTH_LDS_READ(2); OTH_LDS_READ(4); OTH_LDS_READ(6);      // the first butterfly layer of dft8 takes 0, 2, 4, 6
        OTH_LDS_READ(1); OTH_LDS_READ(3); OTH_LDS_READ(5); OTH_LDS_READ(7);
        OTH_LDS_READ(8); OTH_LDS_READ(10); OTH_LDS_READ(12); OTH_LDS_READ(14);
        OTH_LDS_READ(9); OTH_LDS_READ(11); OTH_LDS_READ(13); OTH_LDS_READ(15);
#undef OTH_LDS_READ
        issued();
        float2 h0[8], h1[8];
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
        for (int i = 0; i < 8; ++i) h0[i] = __builtin_bit_cast(float2, r[i]);
        dft8(h0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(h0[0].x));
#pragma unroll
        for (int i = 0; i < 8; ++i) h1[i] = __builtin_bit_cast(float2, r[8 + i]);
        mid();
        dft8(h1);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = h0[i], v[8 + i] = h1[i];
    }
}

// Sixteen ds_read_b64 at base[STRIDE * i], issued and NOT waited for: the caller's next lds_barrier() (lgkmcnt(0)) or
// an explicit wait makes r[] valid.  Nothing may read r[] before that.
template <int STRIDE> __device__ __forceinline__ void lds_issue16(f2v (&r)[16], const float2 *base) {
    const unsigned addr = (unsigned)(unsigned long long)base;
#define OTH_LDS_READ(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(8 * STRIDE * (i)))
    OTH_LDS_READ(0); OTH_LDS_READ(4); OTH_LDS_READ(8); OTH_LDS_READ(12);
    OTH_LDS_READ(1); OTH_LDS_READ(5); OTH_LDS_READ(9); OTH_LDS_READ(13);
    OTH_LDS_READ(2); OTH_LDS_READ(6); OTH_LDS_READ(10); OTH_LDS_READ(14);
    OTH_LDS_READ(3); OTH_LDS_READ(7); OTH_LDS_READ(11); OTH_LDS_READ(15);
#undef OTH_LDS_READ
}

// The software-pipelined form (the default).  Per-wave phase stamps of the plain loop above (tools/archive/diag_x1.py,
// profiles/r04_x1_phases.txt) showed the two halves of a segment badly matched: between barrier 1 and barrier 2 every
// wave only multiplies by the pass-1 twiddles and writes exchange A - the CU's LDS store path is the limit (16 waves x
// 16 ds_write_b64) and the SIMDs idle - while everything else (passes 2, 3, 4, the accumulation, the next pass 1)
// queues up between barrier 2 and barrier 1, where the four waves of a SIMD take turns oldest first (they reach
// barrier 1 about 1650 cycles apart).  Here the tail of segment s - 1 (pass 3, its twiddles, pass 4, |X|^2: 456 VALU
// instructions that touch registers only) runs AFTER barrier 1 of segment s, beside the exchange-A writes, and the
// other half keeps exchange A read, pass 2, exchange B and the next pass 1:
//
//   P1   pass 1 of segment s (its samples were loaded one phase earlier)
//   ---- barrier 1: every wave holds its exchange-B values of segment s - 1 in registers
//   WA   pass-1 twiddles, exchange A writes of segment s
//   T3   pass 3, twiddles, pass 4, accumulation of segment s - 1
//   ---- barrier 2
//   LD   loads of segment s + 1
//   RA   exchange A reads, pass 2, twiddles, exchange B writes and reads (reads issued, waited for at barrier 1)
//
// 624 + 456 VALU instructions per wave on the two sides instead of 168 + 912; registers: the samples / pass-1 outputs
// of one segment and the exchange-B values of the one before, 32 + 32.
// The segment schedule as the body below takes it (WelchArgs and SegArgs both carry these fields)
struct X1Sched {
    int nseg, nbig;       // 32-bit: launch1x_* refuse a launch of 2^31 segments (2^44 samples) or more
    int chunk, tail_chunk, sched;
    unsigned *queue;      // this stream's ticket word (dynamic schedule) or nullptr
};

// The pipelined transform loop.  `epi.take(v, s)` receives segment s after pass 4 (v[r16(k2)] = bin k0 + 16 k1 + 256 k2
// + 4096 bitrev2(q) of thread (wave k0, lane 4 k1 + q), times 1, -1, -1 or i); the first call has s = -1 and zeros.
template <int NW, bool WINDOW, class Epi>
__device__ __forceinline__ void x1_pipe_body(const float2 *xb, long long step, const X1Sched &sc, int wg, int W, const float *win,
                                             const float2 *tw, float2 *lds, Epi &epi, unsigned long long *diag_out) {
    int *lnext = reinterpret_cast<int *>(lds + NW * XREG);
    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63, g = l >> 2, q = l & 3;
    // segment indices are 32-bit in this loop (X1Sched): its ten loop variables as 64-bit pairs took the chain build past
    // the scalar register file (11 SGPRs parked in a VGPR, round 4)
    const int s0 = (int)(((long long)sc.nseg * wg) / W), s1 = (int)(((long long)sc.nseg * (wg + 1)) / W);

    // Twiddles: pass 1 (W_N^(k0 tid), a different set per thread) is rebuilt from six register-resident powers; passes
    // 2 and 3 depend on the lane only and are read from LDS tables where they are used (tabB[k1][l] = W_1024^(k1 l),
    // 8 KiB; tabC[k2][q] = W_64^(k2 q), 512 B) - no products to rebuild them, none of their seeds in registers; the
    // quad-butterfly constants sit there too.
    float2 *tabB = lds + NW * XREG + X1_RED;                                   // [16][64]
    float2 *tabC = tabB + 16 * 64;                                              // [16][4]
    float4 *quadK = reinterpret_cast<float4 *>(tabC + 16 * 4);                  // [4]: s1, alpha, beta, -beta
    {
        // row i of tabB multiplies pass-2 output i: k1 = i at 16 waves, k1 = i & 7 (register 8 h + k1) at 8; in both
        // W_(N/16)^(k1 l) = W_N^(16 k1 l)
        for (int i = tid; i < 1024; i += 64 * NW) tabB[i] = tw[16 * (((i >> 6) & (NW - 1)) * (i & 63))];
        if (tid < 64) tabC[tid] = tw[16 * NW * ((tid >> 2) * (tid & 3))];       // k2 = tid >> 2, q = tid & 3: W_64 = W_N^(N / 64)
        if (tid < 4) {
            const float be = tid >= 2 ? 1.0f : 0.0f;
            quadK[tid] = make_float4(tid < 2 ? 1.0f : -1.0f, tid == 0 ? 1.0f : (tid == 1 ? -1.0f : 0.0f), be, -be);
        }
    }
    const Pow6x a6 = pow6_load(tw, tid);
    __syncthreads();

    float2 *wa = lds + tid;
    const float2 *ra = lds + XREG * wv + l;
    float2 *wb = lds + XREG * wv + l;
    const float2 *rb = lds + XREG * wv + XROW * g + q;

    const int sched = sc.sched;
    const int nchunks = sched ? chunk_count_of(sc.nseg, sc.nbig, sc.chunk, sc.tail_chunk) : 1;
    int cur = sched ? wg : 0, sb = s0, se = s1;
    if (sched && cur < nchunks) chunk_range_of(sc.nseg, sc.nbig, sc.chunk, sc.tail_chunk, cur, sb, se);
    bool live = sched ? cur < nchunks : s0 < s1;
    int s = sb, sp = -1;            // this segment, the one before (whose tail runs in this step)
    int par = 0;
#if OTH_X1_DIAG
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
    f2v pfr[16];      // the next segment's samples as the pinned loads deliver them (valid behind the vmcnt wait)
    f2v rB[16];       // exchange-B values of the segment before; zeros in front of the first one
#pragma unroll
    for (int i = 0; i < 16; ++i) rB[i] = f2v{0.f, 0.f};
    bool have_prev = false;
    if (live) {
        const char *x0 = reinterpret_cast<const char *>(xb + (long long)s * step);
#pragma unroll
        for (int r = 0; r < 16; ++r) load_row_nt(pfr[r], 8u * tid, x0 + 512 * NW * r);
        vm_arrived16(pfr);
    }
    auto tail = [&](auto hook1, auto hook2) {      // pass 3, twiddles, pass 4 of the segment whose exchange-B values sit in rB
        float2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = make_float2(rB[i].x, rB[i].y);
        TwBatch ta;
        tw_read_a<4>(ta, tabC + q);
        const float4 qk = quadK[q];
        dft16(v);
        hook1();
        twiddle_table16<4, 1, false>(v, nullptr, tabC + q, ta);
        hook2();
        quad_dft4_dpp(v, qk.x, qk.y, qk.z, qk.w);
        epi.take(v, sp);
    };
    using std::integral_constant;
    while (live) {
        float2 pf[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pf[r] = make_float2(pfr[r].x, pfr[r].y);
        if constexpr (WINDOW) {      // sixteen L2-resident window values, loaded where they are used (see load_win)
            float wv16[16];
            const char *wbase = reinterpret_cast<const char *>(win);
#pragma unroll
            for (int r = 0; r < 16; ++r) load_win(wv16[r], 4u * tid, wbase + 256 * NW * r);
            vm_arrived_win16(wv16);
#pragma unroll
            for (int r = 0; r < 16; ++r) pf[r] = make_float2(pf[r].x * wv16[r], pf[r].y * wv16[r]);
        }
        X1_STAMP(7);
        X1_DRAIN_VM();
        X1_STAMP(0);
        prio_compute();
        dft16(pf);                                             // P1
        prio_latency();
        X1_STAMP(1);
        lds_barrier_arrived16(rB);      // 1 (also waits for this wave's exchange-B reads of the segment before: rB valid)
        X1_STAMP(2);
        const bool first_of_chunk = s == sb;
        if (sched == 2 && first_of_chunk && tid == 0) lnext[par] = (int)atomicAdd(sc.queue, 1u);
        twiddle6_exa<NW>(pf, wa, a6);                          // WA
        X1_STAMP(3);
        // The segment after this one: its loads go out two at a time over the whole step.  A ticket (dynamic schedule)
        // is drawn with a chunk's FIRST segment and published before barrier 2 of that segment, so at the chunk's last
        // segment it is known here - the launcher never makes one-segment chunks except the very last one of a stream,
        // behind which no ticket can name another chunk.
        int ns = s + 1;
        bool more = true;
        if (ns >= se) {
            if (sched == 0 || (sched == 2 && first_of_chunk)) {
                more = false;
            } else {
                cur = (sched == 1) ? cur + W : W + __builtin_amdgcn_readfirstlane(lnext[par]);
                par ^= 1;
                more = cur < nchunks;
                if (more) {
                    chunk_range_of(sc.nseg, sc.nbig, sc.chunk, sc.tail_chunk, cur, sb, se);
                    ns = sb;
                }
            }
        }
        // uniform row base (scalar registers) + one lane offset: no per-load address registers.  Behind the last
        // segment the same loads run once more on the segment just done: an unconditional definition keeps the sixteen
        // registers free between pass 1 and the first group of loads (a conditional one keeps their OLD values alive).
        const char *xn = reinterpret_cast<const char *>(xb + (long long)(more ? ns : s) * step);
        const unsigned voff = 8u * tid;
        auto spread = [&](auto gc) {
            // place grp of the step (0: after the exchange-A writes, 1: after the tail's butterflies, 2: after its
            // twiddles, 3: end of the tail, 4: behind barrier 2, 5: between pass 2's layers, 6: after pass 2, 7: after
            // the exchange-B writes) issues nibble grp of OTH_X1_PPLACES of the next segment's sixteen loads
            constexpr int grp = decltype(gc)::value;
            constexpr unsigned plan = OTH_X1_PPLACES;
            static_assert(((plan >> 0) & 15) + ((plan >> 4) & 15) + ((plan >> 8) & 15) + ((plan >> 12) & 15) + ((plan >> 16) & 15) +
                              ((plan >> 20) & 15) + ((plan >> 24) & 15) + ((plan >> 28) & 15) == 16,
                          "the eight places must issue exactly the sixteen rows of a segment");
            constexpr int cnt = (plan >> (4 * grp)) & 15;
            constexpr int first = ((grp > 0 ? (plan >> 0) & 15 : 0) + (grp > 1 ? (plan >> 4) & 15 : 0) + (grp > 2 ? (plan >> 8) & 15 : 0) +
                                   (grp > 3 ? (plan >> 12) & 15 : 0) + (grp > 4 ? (plan >> 16) & 15 : 0) + (grp > 5 ? (plan >> 20) & 15 : 0) +
                                   (grp > 6 ? (plan >> 24) & 15 : 0));
            if constexpr (cnt > 0) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = first; r < first + cnt; ++r) load_row_nt(pfr[r], voff, xn + 512 * NW * r);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        spread(integral_constant<int, 0>{});
        prio_compute();
        tail([&] { spread(integral_constant<int, 1>{}); }, [&] { spread(integral_constant<int, 2>{}); });      // T3
        spread(integral_constant<int, 3>{});
        prio_latency();
        X1_STAMP(4);
        lds_barrier();      // 2
        X1_STAMP(5);
        spread(integral_constant<int, 4>{});
        {                                                      // RA
            float2 v[16];
            TwBatch ta;
            pass2_from_lds<NW>(v, ra, [] { prio_compute(); }, [&] {
                spread(integral_constant<int, 5>{});
                tw_read_a<64>(ta, tabB + l);
            });
            prio_latency();
            spread(integral_constant<int, 6>{});
            wave_lds_sync();
            twiddle_table16<64, XROW, true, NW>(v, wb, tabB + l, ta);
            spread(integral_constant<int, 7>{});
            wave_lds_sync();
            lds_issue16<4>(rB, rb);
        }
        X1_STAMP(6);
        have_prev = true;
        live = more;
        sp = s;
        s = ns;
        vm_arrived16(pfr);      // the sixteen pinned loads of this step: the next pass 1 reads them
    }
    if (have_prev) {
        lds_arrived16(rB);
        tail([] {}, [] {});
    }
#if OTH_X1_DIAG
    if (l == 0 && diag_out) {
#pragma unroll
        for (int i = 0; i < 8; ++i) diag_out[8 * wv + i] = phase[i];
    }
#endif
}

// Welch average: sum of |X|^2 per bin; partial rows in finalize layout 4
struct X1WelchEpi {
    float acc[16];
    __device__ __forceinline__ void take(const float2 (&v)[16], int) {      // (the priming call adds zeros)
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const float2 X = v[r16(k2)];
            acc[k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[k2]));
        }
    }
};

template <int NW, bool WINDOW>      // (the scanner passes `()` as its window; windowed Welch plans take the plain kernel)
__global__ __launch_bounds__(64 * NW, 4) void welch16k1x_pipe_kernel(WelchArgs p) {
    constexpr int N = 1024 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    X1WelchEpi epi;
#pragma unroll
    for (int k = 0; k < 16; ++k) epi.acc[k] = 0.f;
    const X1Sched sc{(int)p.nseg, (int)p.nbig, p.chunk, p.tail_chunk, p.sched, p.queue ? p.queue + stream : nullptr};
    unsigned long long *diag = nullptr;
#if OTH_X1_DIAG
    diag = reinterpret_cast<unsigned long long *>(p.partial + (size_t)p.nstreams * W * N) + 128 * ((size_t)stream * W + wg);
#endif
    x1_pipe_body<NW, WINDOW>(p.x + (size_t)stream * p.stream_stride, p.step, sc, wg, W, p.win, p.tw, lds, epi, diag);
    float *dst = p.partial + ((size_t)stream * W + wg) * N + threadIdx.x;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) dst[64 * NW * k2] = epi.acc[k2];      // finalize layout 4 (16384) / 5 (8192)
}

// The fused periodogram chain (stream_to_vector -> keep_one_in_n -> fft_vcc(window, shift) -> |X| or |X|^2 [x 1/N^2] -> IIR
// weights / peak maximum / rows, python/spectrum_sensor_v2.py:85-93, psd_logger.py:43-53, local_worker.py:58-69,
// multichannel_scanner.py:78-86 at fft_len 16384) on the same loop: what chain16k_kernel<4, ...> (welch16k.hip) does,
// same SegArgs, partial rows in layout 4 for chain_reduce / chain_state.
struct X1ChainEpi {
    float acc[16];
    const SegArgs *p;
    int kb;              // bin of register k2: kb + kstride k2
    int kstride, n;      // 256 / 128 (N / 64); N
    size_t row_base;     // stream offset into p->rows, in rows
    bool halves;         // the epilogue eight bins at a time (the 8-wave build: see take_halves)
    // Eight bins at a time - value, row store, accumulation - with the lane's place in the row formed at the store: the
    // 8-wave loop has no registers for sixteen values and a 64-bit lane address next to the transform (84 B of scratch in
    // the form below; 8 B in this one).  The 16-wave build is 5 % FASTER with the form below and its one spilled
    // address (profiles/r05_ab_x1_idx32.txt), so each takes its own.
    __device__ __forceinline__ void take_halves(const float2 (&v)[16], int s) {
        const SegArgs &a = *p;
        const int store_from = (int)a.store_from, acc_end = (int)a.acc_end;       // both within [0, nseg]
        const bool store = s >= store_from, accumulate = s < acc_end;
        float *row = a.rows + (row_base + (size_t)(s - store_from)) * n;           // launch-uniform
        float w = 1.0f;
        if (accumulate && a.acc_mode == 1) {
            const int kk = acc_end - 1 - s;
            w = kk == 0 ? 1.0f : exp2f(a.l2 * (float)kk);
        }
#pragma unroll
        for (int h = 0; h < 16; h += 8) {
            float val[8];
            if (a.epilogue == 0) {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const float2 X = v[r16(h + k2)];
                    val[k2] = __builtin_amdgcn_sqrtf(fmaf(X.x, X.x, X.y * X.y));
                }
            } else {
                const float sc = a.scale;
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const float2 X = v[r16(h + k2)];
                    val[k2] = fmaf(X.x, X.x, X.y * X.y) * sc;
                }
            }
            if (store) {
                unsigned k = (unsigned)((kb + (a.fftshift ? n / 2 : 0)) & (n - 1));      // (the shift moves whole quarter rows; a
                asm volatile("" : "+v"(k));                                             // thread's bins lie inside one)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) row[k + (unsigned)(kstride * (h + k2))] = val[k2];
            }
            if (accumulate) {
                if (a.acc_mode == 1) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[h + k] = fmaf(w, val[k], acc[h + k]);
                } else if (a.acc_mode == 2) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[h + k] = fmaxf(acc[h + k], val[k]);
                }
            }
        }
    }
    __device__ __forceinline__ void take(const float2 (&v)[16], int s) {
        if (s < 0) return;      // the priming call
        if (halves) return take_halves(v, s);
        const SegArgs &a = *p;
        float val[16];
        if (a.epilogue == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                const float2 X = v[r16(k2)];
                val[k2] = __builtin_amdgcn_sqrtf(fmaf(X.x, X.x, X.y * X.y));
            }
        } else {
            const float sc = a.scale;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                const float2 X = v[r16(k2)];
                val[k2] = fmaf(X.x, X.x, X.y * X.y) * sc;
            }
        }
        if (s >= a.store_from) {
            float *row = a.rows + (row_base + (size_t)(s - a.store_from)) * n;
            const int sh = a.fftshift ? n / 2 : 0;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) row[(kb + kstride * k2 + sh) & (n - 1)] = val[k2];
        }
        if (s < a.acc_end) {
            if (a.acc_mode == 1) {
                const long long kk = a.acc_end - 1 - s;
                const float w = kk == 0 ? 1.0f : exp2f(a.l2 * (float)kk);
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = fmaf(w, val[k], acc[k]);
            } else if (a.acc_mode == 2) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = fmaxf(acc[k], val[k]);
            }
        }
    }
};

template <int NW, bool WINDOW>
__global__ __launch_bounds__(64 * NW, 4) void chain16k1x_kernel(SegArgs p) {
    constexpr int N = 1024 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x, l = tid & 63, q = l & 3;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    X1ChainEpi epi;
#pragma unroll
    for (int k = 0; k < 16; ++k) epi.acc[k] = 0.f;
    epi.p = &p;
    // thread (wave k0', lane 4 c + q): 16 waves: k0 = k0', k1 = c; 8 waves: k0 = 2 k0' + (c >> 3), k1 = c & 7
    const int c = l >> 2, k3 = ((q & 1) << 1) | (q >> 1);
    epi.kb = NW == 16 ? (tid >> 6) + 16 * c + 4096 * k3 : 2 * (tid >> 6) + (c >> 3) + 16 * (c & 7) + 2048 * k3;
    epi.kstride = N / 64;
    epi.n = N;
    epi.halves = NW == 8;
    epi.row_base = (size_t)stream * (size_t)(p.nseg - p.store_from);
    const X1Sched sc{(int)p.nseg, (int)p.nbig, p.chunk, p.tail_chunk, p.sched, p.queue ? p.queue + stream : nullptr};
    x1_pipe_body<NW, WINDOW>(p.x + (size_t)stream * p.stream_stride + p.first, p.step, sc, wg, W, p.win, p.tw, lds, epi, nullptr);
    if (p.partial) {
        float *dst = p.partial + ((size_t)stream * W + wg) * N + tid;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) dst[64 * NW * k2] = epi.acc[k2];
    }
}

template <int NW, bool WINDOW> static hipError_t launch1x_pipe(const WelchArgs &a, hipStream_t s) {
    if (a.nseg > 0x7fffffffLL) return hipErrorInvalidValue;      // x1_pipe_body counts segments in 32 bits
    const dim3 grid(a.wg_per_stream, a.nstreams);
    constexpr size_t lds = x1p_lds_bytes<NW>();
    const void *fn = reinterpret_cast<const void *>(welch16k1x_pipe_kernel<NW, WINDOW>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((welch16k1x_pipe_kernel<NW, WINDOW>), grid, dim3(64 * NW), lds, s, a);
    return hipGetLastError();
}

// 50 % overlap (scipy.signal.welch's default noverlap at nperseg = nfft = 16384, ofdm_cr_tools.py:214,322,342 with
// nFFT = 16384): the second half of a segment (rows r = 8..15 of this thread) is the first half of the next one, so it
// stays in registers (raw) and every sample is read once; two transforms per new sample.  DET = 2: constant detrend
// after the transform, X[k] -= mean FFT(w)[k], for windows whose spectrum is confined to |k| < 16 (every periodic
// cosine-sum window): those bins are register k2 = 0 of lanes (k1 = 0, q = 0) and register k2 = 15 of lanes (k1 = 15,
// q = 3) of each wave k0, so every thread corrects v[0] and v[15] with its own pair of table values (WelchArgs.fd,
// zero for all but 32 threads; the q = 3 entries carry the factor i the quad butterfly leaves on that lane).  The
// per-wave sums of the new half go to LDS by step parity, wave 0 adds them up behind barrier 1 and every thread picks
// the segment total up at the end of the step (as welch16k_kernel<2, ...>).  Twiddles of passes 2 and 3 from LDS tables.
constexpr int XH_RED = 48;      // 16 wave sums x 2 parities, tickets [32..33], segment totals [40..41]
template <int NW = 16> constexpr size_t x1h_lds_bytes() {
    return (NW * XREG + XH_RED + 16 * 64 + 16 * 4) * sizeof(float2) + 4 * sizeof(float4);
}

// NW = 8: the same at 8192 points (round 5; two workgroups per CU).  The bins |k| < 16 the detrend corrects are then
// register k2 = 0 of lanes 0 and 32 (k1 = 0, q = 0 of k0 = 2 k0' and 2 k0' + 1) and register k2 = 15 of lanes 31 and 63
// (k1 = 7, q = 3) of each wave (window_spectrum_table_1x in api.hip).
template <int NW, int DET, bool PILOT = false>
__global__ __launch_bounds__(64 * NW, 4) void welch16k1x_half_kernel(WelchArgs p) {
    static_assert(DET == 2 || !PILOT, "the pilot belongs to the detrend");
    constexpr int N = 1024 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    float2 *red = lds + NW * XREG;
    int *lnext = reinterpret_cast<int *>(red + 32);
    float2 *tabB = red + XH_RED;                                                // [16][64]: W_1024^(k1 l)
    float2 *tabC = tabB + 16 * 64;                                              // [16][4]:  W_64^(k2 q)
    float4 *quadK = reinterpret_cast<float4 *>(tabC + 16 * 4);

    const int tid = threadIdx.x;
    // the wave index as a scalar: the LDS slots of the per-wave sums and the region bases then need no vector registers
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, g = l >> 2, q = l & 3;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const long long s0 = (p.nseg * wg) / W, s1 = (p.nseg * (wg + 1)) / W;
    const float2 *xb = p.x + (size_t)stream * p.stream_stride;

    {
        for (int i = tid; i < 1024; i += 64 * NW) tabB[i] = p.tw[16 * (((i >> 6) & (NW - 1)) * (i & 63))];      // as x1_pipe_body
        if (tid < 64) tabC[tid] = p.tw[16 * NW * ((tid >> 2) * (tid & 3))];
        if (tid < 4) {
            const float be = tid >= 2 ? 1.0f : 0.0f;
            quadK[tid] = make_float4(tid < 2 ? 1.0f : -1.0f, tid == 0 ? 1.0f : (tid == 1 ? -1.0f : 0.0f), be, -be);
        }
    }
    const float2 a1 = p.tw[tid], a4 = p.tw[4 * tid];
    __syncthreads();

    float2 *wa = lds + tid;
    const float2 *ra = lds + XREG * wv + l;
    float2 *wb = lds + XREG * wv + l;
    const float2 *rb = lds + XREG * wv + XROW * g + q;

    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;

    const int sched = p.sched;
    const long long nchunks = sched ? chunk_count(p) : 1;
    int cpar = 0;
    // Registers (round 5; round 4 kept the sixteen window values in registers, loaded each new half at the top of its
    // step and spilled 25 registers around the transform - 104 B of scratch, reloaded between the step's barriers):
    //   keep[8]  the raw second half of the segment before (= this segment's first half), pilot already off
    //   nxt[8]   this segment's new half, PREFETCHED during the step before (pinned non-temporal loads, two at each of
    //            four places of the step as in the scanner kernel: a burst stalls the issuing wave at its loads)
    //   the sixteen window values come from L2 where they are used (pinned loads, see load_win): 64 KiB shared by every
    //   workgroup of the XCD, against 64 KiB of new samples per step
    float2 keep[8];
    f2v nxt[8];
    const unsigned voff = 8u * tid;
    // PILOT (every detrending plan but OTH_DETREND_CONSTANT_FAST): WelchArgs.pilot comes off every sample as it arrives.
    // pilot_inline (late round 5): formed here from the eight 2 KiB probes of fft4096.hip.h by the first four waves - every
    // workgroup for itself, 16 KiB of L2 traffic each - instead of by pilot_mean_kernel in front of the launch
    float2 pv = make_float2(0.f, 0.f);
    if (PILOT && !p.pilot_inline) pv = load_pilot(p.pilot, blockIdx.y);
    if (PILOT && p.pilot_inline) {
        float2 *slot = red;      // [8][4] per-wave probe totals; the sums of the first step are written behind barrier below
        if (wv < 4) {
            const PilotProbes probes = inline_pilot_load(xb, p.nseg, (int)p.step, tid);
            inline_pilot_store(probes, tid, slot);
        }
        lds_barrier();
        pv = inline_pilot_value(slot);
        lds_barrier();           // every wave has read the slots before the first step's sums overwrite them
    }
    for (long long cur = sched ? wg : 0; cur < nchunks;) {
        long long sb = s0, se = s1;
        if (sched) chunk_range(p, cur, sb, se);
        if (sb < se) {      // the chunk's first segment: both halves now
            const char *x0 = reinterpret_cast<const char *>(xb + sb * p.step);
            f2v first[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) load_row_nt(first[r], voff, x0 + 512 * NW * r);
#pragma unroll
            for (int r = 0; r < 8; ++r) load_row_nt(nxt[r], voff, x0 + 512 * NW * (8 + r));
            vm_arrive8(keep, first);      // (nxt has landed as well; the step below says so where it takes it)
        }
#if OTH_X1H_WIN_EARLY      // A/B (no gain, see the macro): the window values of a step requested at the END of the step before
        float wv16[16];
        auto request_window = [&]() {
            const char *wbase = reinterpret_cast<const char *>(p.win);
#pragma unroll
            for (int r = 0; r < 16; ++r) load_win(wv16[r], 4u * tid, wbase + 256 * NW * r);
        };
        request_window();
#endif
        for (long long s = sb; s < se; ++s) {
            float2 v[16];
            prio_latency();
#if !OTH_X1H_WIN_EARLY
            float wv16[16];      // (declared per step: held across the loop edge they cost the PILOT builds 17 spilled registers)
#if OTH_X1H_FAKEWIN      // timing experiment only (wrong values): what the window loads and their wait cost
            float ax = a1.x, ay = a1.y;
            asm volatile("" : "+v"(ax), "+v"(ay));      // (not loop-invariant to the compiler: no sixteen hoisted registers)
#pragma unroll
            for (int r = 0; r < 16; ++r) wv16[r] = fmaf(ax, 0.03f * (float)r, 0.5f - ay * 0.01f * (float)r);
#else
            {
                const char *wbase = reinterpret_cast<const char *>(p.win);
#pragma unroll
                for (int r = 0; r < 16; ++r) load_win(wv16[r], 4u * tid, wbase + 256 * NW * r);
            }
#endif
#endif
            // the new half (requested during the step before, or above) and the window values: the new half comes out of
            // its loading registers in the same statement that waits for it (vm_arrive8: it outlives the next loads into
            // them as the kept half, and a copy the compiler makes for that may not stand in front of the wait)
            float2 fresh[8];
            vm_arrive8(fresh, nxt);
#if !OTH_X1H_FAKEWIN
            vm_arrived_win16(wv16);
#endif
            float2 sum = make_float2(0.f, 0.f), sumf = make_float2(0.f, 0.f);
            if (s == sb) {      // the chunk's first half arrives raw
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (PILOT) keep[r] = csub(keep[r], pv);
                    sumf = cadd(sumf, keep[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float2 nw = fresh[r];
                if (PILOT) nw = csub(nw, pv);
                v[r] = make_float2(keep[r].x * wv16[r], keep[r].y * wv16[r]);
                v[8 + r] = make_float2(nw.x * wv16[8 + r], nw.y * wv16[8 + r]);
                keep[r] = nw;
                sum = cadd(sum, nw);
            }
            // the new half of the segment after this one (behind the chunk's last segment the same rows once more: an
            // unconditional definition keeps the registers free - a conditional one keeps their old values alive)
            const char *xn = reinterpret_cast<const char *>(xb + (s + 1 < se ? s + 1 : s) * p.step) + 512 * NW * 8;
            auto prefetch = [&](int r0) {
                __builtin_amdgcn_sched_barrier(0);
                load_row_nt(nxt[r0], voff, xn + 512 * NW * r0);
                load_row_nt(nxt[r0 + 1], voff, xn + 512 * NW * (r0 + 1));
                __builtin_amdgcn_sched_barrier(0);
            };
            const int par = (int)(s & 1);
            if (DET == 2) {
                sum.x = wave_total_lane63(sum.x);
                sum.y = wave_total_lane63(sum.y);
                if (s == sb) {
                    sumf.x = wave_total_lane63(sumf.x);
                    sumf.y = wave_total_lane63(sumf.y);
                }
                if (l == 63) {
                    red[16 * par + wv] = sum;
                    if (s == sb) red[16 * (par ^ 1) + wv] = sumf;
                }
            }
            prio_compute();
            dft16(v);                                              // pass 1: r -> k0
            prio_latency();
            lds_barrier();      // 1
            if (sched == 2 && s == sb && tid == 0) lnext[cpar] = (int)atomicAdd(p.queue + stream, 1u);
            if (DET == 2 && tid < 64) {      // wave 0: the 32 per-wave sums of this segment's two halves -> its total
                float2 part = make_float2(0.f, 0.f);
                int slot = l;      // opaque: formed here, not kept (and spilled) across the step
                asm volatile("" : "+v"(slot));
                if (slot < 32 && (slot & 15) < NW) part = red[slot];
                part.x = wave_total_lane63(part.x);
                part.y = wave_total_lane63(part.y);
                if (tid == 63) red[40 + par] = part;      // read behind barrier 2, at the end of the step
            }
            scatter_pow16_exa<NW>(v, wa, a1, a4);                  // x W_N^(k0 tid) -> [k0][w][l]
            prefetch(0);
            lds_barrier();      // 2
            TwBatch ta;
            pass2_from_lds<NW>(v, ra, [] { prio_compute(); }, [&] { tw_read_a<64>(ta, tabB + l); });     // pass 2
            prio_latency();
            prefetch(2);
            wave_lds_sync();
            twiddle_table16<64, XROW, true, NW>(v, wb, tabB + l, ta);
            prefetch(4);
            wave_lds_sync();
            TwBatch tc;
            dft16_from_lds<4>(v, rb, [] { prio_compute(); }, [&] { tw_read_a<4>(tc, tabC + q); });        // pass 3
            const float4 qk = quadK[q];
            twiddle_table16<4, 1, false>(v, nullptr, tabC + q, tc);
            prefetch(6);
            quad_dft4_dpp(v, qk.x, qk.y, qk.z, qk.w);              // pass 4
#if OTH_X1H_WIN_EARLY
            request_window();      // (in front of the quad butterfly the PILOT builds spilled the kept half: 20 registers)
#endif
            if (DET == 2) {
                // X[k] -= mean FFT(w)[k] where FFT(w) is not negligible: register k2 = 0 of lanes (k1 = 0, q = 0) and k2 = 15
                // of lanes (k1 = 15, q = 3) - lanes 0 and 63 of every wave; their table entries come from L2 when they are
                // used (held across the step they were four registers of all 1024 threads)
                if (NW == 16 ? (l == 0 || l == 63) : ((l & 31) == 0 || (l & 31) == 31)) {
                    int tfd = tid;      // opaque, as above: the table address is formed inside the branch
                    asm volatile("" : "+v"(tfd));
                    const float4 fd = p.fd[tfd];
                    const float2 tot = red[40 + par];
                    const float2 mean = make_float2(tot.x * (1.0f / N), tot.y * (1.0f / N));
                    v[0] = make_float2(v[0].x - (mean.x * fd.x - mean.y * fd.y), v[0].y - (mean.x * fd.y + mean.y * fd.x));
                    v[15] = make_float2(v[15].x - (mean.x * fd.z - mean.y * fd.w), v[15].y - (mean.x * fd.w + mean.y * fd.z));
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                const float2 X = v[r16(k2)];
                acc[k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[k2]));
            }
        }
        if (sched == 0) break;
        cur = (sched == 1) ? cur + W : (long long)W + lnext[cpar];
        cpar ^= 1;
    }

    float *dst = p.partial + ((size_t)stream * W + wg) * N + tid;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) dst[64 * NW * k2] = acc[k2];
}

template <int NW, int DET, bool PILOT = false> static hipError_t launch1x_half(const WelchArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    constexpr size_t lds = x1h_lds_bytes<NW>();
    const void *fn = reinterpret_cast<const void *>(welch16k1x_half_kernel<NW, DET, PILOT>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((welch16k1x_half_kernel<NW, DET, PILOT>), grid, dim3(64 * NW), lds, s, a);
    return hipGetLastError();
}
template <int NW> static hipError_t launch1x_half_n(const WelchArgs &a, hipStream_t s) {
    if (a.detrend && a.fd) return (a.pilot || a.pilot_inline) ? launch1x_half<NW, 2, true>(a, s) : launch1x_half<NW, 2>(a, s);
    return launch1x_half<NW, 0>(a, s);
}

// step = N / 2; detrend none, or constant through the frequency-domain form (a.fd = the table described above)
hipError_t launch_welch_tuned16k1x_half(int nfft, const WelchArgs &a, hipStream_t s) {
    return nfft == 8192 ? launch1x_half_n<8>(a, s) : launch1x_half_n<16>(a, s);
}

// ---- 8192 points, 50 % overlap, ROLE-SPLIT (late round 5) ------------------------------------------------------------
// The 8-wave loop above runs every phase of a segment on all of its waves, two barriers per step; two workgroups per CU
// overlap each other's phases only by chance.  Here one 1024-thread workgroup per CU is split as in welch4096ws.hip:
//   producers (threads 0..511)     loads, pilot, window, per-wave sums, pass 1, its twiddles, exchange-A writes of
//                                  segment s into image s & 1
//   consumers (threads 512..1023)  exchange-A reads of segment s - 1, pass 2 (two radix-8), exchange B inside the wave,
//                                  passes 3 and 4, the frequency-domain detrend, the accumulation
// one LDS-only barrier per step; the loads of one role overlap the butterflies of the other by construction.  Two
// images of 68 KiB + tables: 145 of the CU's 160 KiB.  Contiguous runs of segments only (the schedule this shape takes
// by default); anything else stays on the one-role kernel.
constexpr int X8W_RED = 48;      // [2][8] per-wave sums of a segment (both halves), by image parity; [16..47]: the pilot's probe totals
constexpr size_t x8ws_lds_bytes() {
    return (2 * 8 * XREG + X8W_RED + 16 * 64 + 16 * 4) * sizeof(float2) + 4 * sizeof(float4);
}

template <int DET, bool PILOT = false>
__global__ __launch_bounds__(1024, 4) void welch8kws_kernel(WelchArgs p) {
    static_assert(DET == 2 || !PILOT, "the pilot belongs to the detrend");
    constexpr int NW = 8, N = 8192;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    float2 *red = lds + 2 * NW * XREG;
    float2 *tabB = red + X8W_RED;                                               // [16][64]
    float2 *tabC = tabB + 16 * 64;                                              // [16][4]
    float4 *quadK = reinterpret_cast<float4 *>(tabC + 16 * 4);

    const int tid = threadIdx.x;
    const int role = __builtin_amdgcn_readfirstlane(tid >> 9);
    const int t = tid & 511;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63, g = l >> 2, q = l & 3;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const int s0 = (int)((p.nseg * wg) / W), s1 = (int)((p.nseg * (wg + 1)) / W);      // (the launcher checks nseg < 2^31)
    const float2 *xb = p.x + (size_t)stream * p.stream_stride;
    {
        for (int i = tid; i < 1024; i += 1024) tabB[i] = p.tw[16 * (((i >> 6) & (NW - 1)) * (i & 63))];      // as x1_pipe_body
        if (tid < 64) tabC[tid] = p.tw[16 * NW * ((tid >> 2) * (tid & 3))];
        if (tid < 4) {
            const float be = tid >= 2 ? 1.0f : 0.0f;
            quadK[tid] = make_float4(tid < 2 ? 1.0f : -1.0f, tid == 0 ? 1.0f : (tid == 1 ? -1.0f : 0.0f), be, -be);
        }
    }
    __syncthreads();

    if (role == 0) {
        // ---------------------------------------------------------------------------------------------- producer
        const Pow6x a6 = pow6_load(p.tw, t);      // W^1,2,3,4,8,12 of this thread: nine products per segment instead of thirteen
        float2 keep[8];
        f2v nxt[8];
        const unsigned voff = 8u * t;
        float2 pv = make_float2(0.f, 0.f);
        if (PILOT && !p.pilot_inline) pv = load_pilot(p.pilot, blockIdx.y);
        if (PILOT && p.pilot_inline) {      // the pilot from eight probes, by the first four producer waves (as in the one-role kernel)
            float2 *slot = red + 16;
            if (wv < 4) {
                const PilotProbes probes = inline_pilot_load(xb, p.nseg, (int)p.step, t);
                inline_pilot_store(probes, t, slot);
            }
            lds_barrier();      // (the consumers stand at the same barrier)
            pv = inline_pilot_value(slot);
        }
        float2 prev_new = make_float2(0.f, 0.f);      // lane 63: this wave's sum of the previous segment's new half
        // the sixteen window values of this thread stay in registers for the whole run: this role has them (the one-role
        // kernel reloads them from L2 with every step, and waits for them at the top of it)
        float wv16[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) wv16[r] = p.win[64 * NW * r + t];
        if (s0 < s1) {      // the run's first segment: both halves now
            const char *x0 = reinterpret_cast<const char *>(xb + (long long)s0 * p.step);
            f2v first[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) load_row_nt(first[r], voff, x0 + 512 * NW * r);
#pragma unroll
            for (int r = 0; r < 8; ++r) load_row_nt(nxt[r], voff, x0 + 512 * NW * (8 + r));
            vm_arrive8(keep, first);
        }
        for (int s = s0; s < s1; ++s) {
            const int par = (s - s0) & 1;
            float2 *wa = lds + par * NW * XREG + t;
            float2 v[16];
            prio_latency();
            float2 fresh[8];
            vm_arrive8(fresh, nxt);      // wait + copy out of the loading registers in one statement (see vm_arrive8)
            float2 sum = make_float2(0.f, 0.f), sumf = make_float2(0.f, 0.f);
            if (s == s0) {      // the run's first half arrives raw
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (PILOT) keep[r] = csub(keep[r], pv);
                    sumf = cadd(sumf, keep[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float2 nw = fresh[r];
                if (PILOT) nw = csub(nw, pv);
                v[r] = make_float2(keep[r].x * wv16[r], keep[r].y * wv16[r]);
                v[8 + r] = make_float2(nw.x * wv16[8 + r], nw.y * wv16[8 + r]);
                keep[r] = nw;
                sum = cadd(sum, nw);
            }
            const char *xn = reinterpret_cast<const char *>(xb + (long long)(s + 1 < s1 ? s + 1 : s) * p.step) + 512 * NW * 8;
            auto prefetch = [&](int r0) {
                __builtin_amdgcn_sched_barrier(0);
                load_row_nt(nxt[r0], voff, xn + 512 * NW * r0);
                load_row_nt(nxt[r0 + 1], voff, xn + 512 * NW * (r0 + 1));
                __builtin_amdgcn_sched_barrier(0);
            };
            prefetch(0);
            if (DET == 2) {      // per-wave sums of both halves of this segment, for the consumers
                sum.x = wave_total_lane63(sum.x);
                sum.y = wave_total_lane63(sum.y);
                float2 other = prev_new;
                if (s == s0) other = make_float2(wave_total_lane63(sumf.x), wave_total_lane63(sumf.y));
                if (l == 63) red[8 * par + wv] = cadd(sum, other);
                prev_new = sum;
            }
            prefetch(2);
            prio_compute();
            dft16(v);                                              // pass 1: r -> k0
            prio_latency();
            prefetch(4);
            twiddle6_exa<NW>(v, wa, a6);                           // x W_N^(k0 t) -> [k0][w][l] of this image
            prefetch(6);
            lds_barrier();
        }
    } else {
        // ---------------------------------------------------------------------------------------------- consumer
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        // the pass-2 twiddles W_1024^(k1 l) of this lane stay in registers (the one-role kernel reads them from the LDS
        // table every step: it has no registers left, this role has)
        float2 tw2[16];
#pragma unroll
        for (int k = 1; k < 16; ++k) tw2[k] = tabB[64 * k + l];
        if (PILOT && p.pilot_inline) lds_barrier();      // the producers' pilot barrier
        for (int s = s0; s < s1; ++s) {
            lds_barrier();      // image (s - s0) & 1 and its sums are complete
            const int par = (s - s0) & 1;
            float2 *im = lds + par * NW * XREG;
            const float2 *ra = im + XREG * wv + l;
            float2 *wb = im + XREG * wv + l;
            const float2 *rb = im + XREG * wv + XROW * g + q;
            float2 v[16];
            prio_latency();
            pass2_from_lds<NW>(v, ra, [] { __builtin_amdgcn_s_setprio(1); }, [] {});     // pass 2 (above the producers' butterflies)
            prio_latency();
            wave_lds_sync();
            wb[0] = v[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) wb[XROW * k] = cmul(v[p2reg<NW>(k)], tw2[k]);
            wave_lds_sync();
            TwBatch tc;
            dft16_from_lds<4>(v, rb, [] { __builtin_amdgcn_s_setprio(1); }, [&] { tw_read_a<4>(tc, tabC + q); });        // pass 3
            const float4 qk = quadK[q];
            twiddle_table16<4, 1, false>(v, nullptr, tabC + q, tc);
            quad_dft4_dpp(v, qk.x, qk.y, qk.z, qk.w);              // pass 4
            if (DET == 2) {
                if ((l & 31) == 0 || (l & 31) == 31) {      // the lanes whose registers 0 / 15 lie where FFT(w) is not negligible
                    int tfd = t;      // opaque: the table address is formed inside the branch
                    asm volatile("" : "+v"(tfd));
                    const float4 fd = p.fd[tfd];
                    float2 tot = red[8 * par];
#pragma unroll
                    for (int w = 1; w < NW; ++w) tot = cadd(tot, red[8 * par + w]);
                    const float2 mean = make_float2(tot.x * (1.0f / N), tot.y * (1.0f / N));
                    v[0] = make_float2(v[0].x - (mean.x * fd.x - mean.y * fd.y), v[0].y - (mean.x * fd.y + mean.y * fd.x));
                    v[15] = make_float2(v[15].x - (mean.x * fd.z - mean.y * fd.w), v[15].y - (mean.x * fd.w + mean.y * fd.z));
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                const float2 X = v[r16(k2)];
                acc[k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[k2]));
            }
        }
        float *dst = p.partial + ((size_t)stream * W + wg) * N + t;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) dst[64 * NW * k2] = acc[k2];
    }
}

template <int DET, bool PILOT = false> static hipError_t launch8kws(const WelchArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    constexpr size_t lds = x8ws_lds_bytes();
    const void *fn = reinterpret_cast<const void *>(welch8kws_kernel<DET, PILOT>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((welch8kws_kernel<DET, PILOT>), grid, dim3(1024), lds, s, a);
    return hipGetLastError();
}
// 8192 points, step = N / 2, contiguous runs; detrend none, or constant through the frequency-domain form (a.fd)
hipError_t launch_welch_tuned8kws(const WelchArgs &a, hipStream_t s) {
    if (a.sched != 0 || a.nseg > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.detrend && a.fd) return (a.pilot || a.pilot_inline) ? launch8kws<2, true>(a, s) : launch8kws<2>(a, s);
    return a.detrend ? hipErrorInvalidValue : launch8kws<0>(a, s);
}

template <bool WINDOW> static hipError_t launch1x(const WelchArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    constexpr size_t lds = x1_lds_bytes();
    const void *fn = reinterpret_cast<const void *>(welch16k1x_kernel<WINDOW>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((welch16k1x_kernel<WINDOW>), grid, dim3(1024), lds, s, a);
    return hipGetLastError();
}

template <int NW, bool WINDOW> static hipError_t launch_chain1x(const SegArgs &a, hipStream_t s) {
    if (a.nseg > 0x7fffffffLL) return hipErrorInvalidValue;      // x1_pipe_body counts segments in 32 bits
    const dim3 grid(a.wg_per_stream, a.nstreams);
    constexpr size_t lds = x1p_lds_bytes<NW>();
    const void *fn = reinterpret_cast<const void *>(chain16k1x_kernel<NW, WINDOW>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((chain16k1x_kernel<NW, WINDOW>), grid, dim3(64 * NW), lds, s, a);
    return hipGetLastError();
}
hipError_t launch_chain16k1x(int nfft, const SegArgs &a, bool rect, hipStream_t s) {
    if (nfft == 16384) return rect ? launch_chain1x<16, false>(a, s) : launch_chain1x<16, true>(a, s);
    if (nfft == 8192) return rect ? launch_chain1x<8, false>(a, s) : launch_chain1x<8, true>(a, s);
    return hipErrorInvalidValue;
}

// plain: the A/B switch (tuning variant "16kplain" / OTH_16K1X_MODE=plain); windowed plans always take the plain kernel
hipError_t launch_welch_tuned16k1x(int nfft, const WelchArgs &a, bool window, bool plain, hipStream_t s) {
    if (nfft == 8192) return (window || plain) ? hipErrorInvalidValue : launch1x_pipe<8, false>(a, s);      // the pipelined rectangular build only
    static const char *mode = getenv("OTH_16K1X_MODE");
    plain = plain || (mode && !strcmp(mode, "plain"));
    if (window) return launch1x<true>(a, s);
    return plain ? launch1x<false>(a, s) : launch1x_pipe<16, false>(a, s);
}

}  // namespace oth
