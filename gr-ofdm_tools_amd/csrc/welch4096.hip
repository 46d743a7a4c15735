// welch4096: the headline kernel.  Welch-averaged |FFT_4096(detrend(x) * w)|^2
// over a run of overlapping segments per workgroup, for nperseg = nfft = 4096
// (scipy.signal.welch as called at ofdm_cr_tools.py:322,342 and :214).
//
// Decomposition 4096 = 16 x 16 x 16, decimation in frequency, 256 threads, 16
// points per thread, three register-resident radix-16 butterflies:
//
//   sample index  n = 256 a + 16 b + c        bin index  k = k0 + 16 k1 + 256 k2
//
//   pass 1  thread (b,c) = n mod 256 holds a = 0..15   -> k0, times W4096^(k0 (16b+c))
//   pass 2  thread (k0,c)            holds b = 0..15   -> k1, times W256^(k1 c)
//   pass 3  thread (k0,k1)           holds c = 0..15   -> k2, accumulate |X|^2
//
// Exchange 1 (b <-> k0) crosses waves: one LDS round trip bracketed by the two
// workgroup barriers of a segment.  Exchange 2 (c <-> k1) stays inside the 16
// lanes that share k0, so it reuses the same 2176-byte LDS region with
// wave-level ordering only.  LDS image: region k0 = 16 rows x 17 float2 (one
// float2 of padding per row): every ds_write_b64 / ds_read_b64 of both
// exchanges is bank-conflict free ("LDS image").
//
// HBM: lane t reads x[s*step + 256 a + t], 512 contiguous bytes per wave
// instruction.  The window (16 values per thread) stays in registers for the
// whole launch.
//
// Two builds of this file ship (Makefile):
//   dpp   any step / zero-padded nperseg: 16 loads per segment (the overlapped half comes back from
//         L2), all 15 pass-1 twiddles in registers.
//   pipe  step = 2048 (the 50 % overlap of the reference's welch() calls): the overlapped half of a
//         segment stays in registers, the half the NEXT segment adds is prefetched while this one
//         is transformed (LDS-only barriers keep it in flight), and only the new half is summed
//         for the detrend.  The 32 extra data registers are paid for by rebuilding the pass-1
//         twiddles W^(k0 t) from W^t and W^(4t) each segment (scatter_pow16, 13 complex products),
//         which keeps the kernel at 128 VGPRs = 4 workgroups per CU.
// Both builds rebuild the pass-2 twiddles W256^(k1 c) the same way (reading them from an LDS table
// costs fifteen dependent LDS round trips) and raise the wave priority outside the butterflies.
// Segments reach workgroups in chunks through WelchArgs.sched (contiguous / interleaved /
// atomic ticket); see the comment at the chunk loop.
#include "fft_lds.hip.h"
#include "oth_internal.h"

// Build-time switches (the Makefile compiles this file once per shipped build; api.hip picks one):
//   OTH_W4096_TAG       suffix of the exported launcher
#ifndef OTH_W4096_TAG
#define OTH_W4096_TAG dpp
#endif
#ifndef OTH_W4096_PIPE
#define OTH_W4096_PIPE 0     // 1: 50 %-overlap pipeline - the overlapped half stays in registers, the next
#endif                       //    half is prefetched, pass-1 twiddles are rebuilt from two powers (needs step 2048)
#ifndef OTH_W4096_DIAG
#define OTH_W4096_DIAG 0     // 1: diagnostic build, every workgroup stamps start/end time + XCC id
#endif
#define OTH_CAT2(a, b) a##b
#define OTH_CAT(a, b) OTH_CAT2(a, b)

#include "fft4096.hip.h"

namespace oth {

namespace {

#if OTH_W4096_DIAG
#define OTH_STAMP(i)                                                     \
    do {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): s_memtime */  \
        phase[i] += now_ - last_;                                        \
        last_ = now_;                                                    \
        __builtin_amdgcn_sched_barrier(0);                               \
    } while (0)
#else
#define OTH_STAMP(i)
#endif

// NA = nperseg / 256: rows a < NA of a segment hold samples, the rest is the zero padding up to 4096
// (NA = 16: nperseg = nfft; NA = 4: the sweeper's nperseg = nfft / 4, spectrum_sweeper.py:263).
template <bool DETREND, int NA, bool PILOT = false>
__global__ __launch_bounds__(T4, OTH_W4096_PIPE ? 4 : 1) void welch4096_kernel(WelchArgs p) {
    static_assert(DETREND || !PILOT, "the pilot belongs to the detrend");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lx = reinterpret_cast<float2 *>(smem);
    float2 *red = lx + LDS_X;

    const int t = threadIdx.x;
    const int hi = t >> 4, lo = t & 15;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
#if OTH_W4096_DIAG
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long phase[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
    const long long s0 = (p.nseg * wg) / W, s1 = (p.nseg * (wg + 1)) / W;
    const float2 *xb = p.x + (size_t)stream * p.stream_stride;

    // thread-constant tables
#if OTH_W4096_PIPE
    // window rows 0..7 live in registers; rows 8..15 come from L2 with every segment (pinned loads, see load_win):
    // with the kept half and the prefetch next to the data the sixteen values did not fit into 128 VGPRs (round 4's
    // build spilled 6-12 registers inside the segment loop)
    float win[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) win[a] = p.win[256 * a + t];
    const char *const win_hi = reinterpret_cast<const char *>(p.win + 2048);
#else
    float win[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) win[a] = p.win[256 * a + t];
#endif
#if OTH_W4096_PIPE
    const float2 b1 = p.tw[t], b4 = p.tw[4 * t];      // W4096^t, W4096^(4t)
    float2 keep[8], nxt[8];
    float2 prev_tot = make_float2(0.f, 0.f);
#else
    float2 tw1[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) tw1[k] = p.tw[t * k];
#endif
    const float2 c1 = p.tw[16 * lo], c4 = p.tw[64 * lo];   // pass-2 twiddle seeds W256^c, W256^(4c)

    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    // PILOT (every detrending plan but OTH_DETREND_CONSTANT_FAST): WelchArgs.pilot comes off every sample as it arrives
    const float2 pv = load_pilot(PILOT ? p.pilot : nullptr, stream);

    // LDS addresses (in float2): exchange-1 write/read, exchange-2 write/read
    const int w1 = hi * 17 + lo;            // + k0 * RS      (thread is (b,c))
    const int r1 = hi * RS + lo;            // + b * 17       (thread is (k0,c))
    const int w2 = hi * RS + lo;            // + k1 * 17
    const int r2 = hi * RS + lo * 17;       // + c            (thread is (k0,k1))

    // Segment schedule (WelchArgs.sched):
    //   0  contiguous   workgroup w owns segments [s0, s1)                      (deterministic)
    //   1  interleaved  chunk c = w, w + W, w + 2W, ... of `chunk` segments       (deterministic)
    //   2  dynamic      first chunk = w, then chunks drawn from an atomic ticket  (load-balanced;
    //                   the set of segments a workgroup sums depends on timing)
    const int sched = p.sched;
    const long long nchunks = sched ? chunk_count(p) : 1;
    int *lnext = reinterpret_cast<int *>(red + 8);
    unsigned ticket = 0;
    for (long long cur = sched ? wg : 0; cur < nchunks;) {
      long long sb = s0, se = s1;
      if (sched) chunk_range(p, cur, sb, se);
#if OTH_W4096_PIPE
      {   // chunk prologue: both halves of its first segment (half-block h = samples [2048 h, 2048 h + 2048))
          const float2 *xs = xb + sb * 2048 + t;
#pragma unroll
          for (int a = 0; a < 8; ++a) keep[a] = PILOT ? csub(xs[256 * a], pv) : xs[256 * a];
#pragma unroll
          for (int a = 0; a < 8; ++a) nxt[a] = xs[2048 + 256 * a];
      }
#endif
      for (long long s = sb; s < se; ++s) {
        float2 v[16];
        prio_latency();
        OTH_STAMP(5);       // loop overhead / chunk prologue
#if OTH_W4096_DIAG && OTH_W4096_PIPE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // diagnostic only: isolate the wait for the prefetch
        OTH_STAMP(6);
#endif
#if OTH_W4096_PIPE
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            v[a] = keep[a];
            v[8 + a] = PILOT ? csub(nxt[a], pv) : nxt[a];
            keep[a] = v[8 + a];
        }
        float wu[8];        // window rows 8..15: issued in front of the prefetch, awaited where they are used
#pragma unroll
        for (int a = 0; a < 8; ++a) load_win(wu[a], 4u * t, win_hi + 1024 * a);
        {   // the half the NEXT segment adds; lands while this segment is transformed.  Behind the chunk's last segment
            // the same loads run once more on the half just taken: always eight loads younger than the window's, so
            // ONE counted wait serves every step (two waits in two branches made the compiler merge their operands
            // with copies in FRONT of the waits - tools/isa_async_hazard.py)
            const float2 *xn = xb + (s + 1 < se ? s + 2 : s + 1) * 2048 + t;
#pragma unroll
            for (int a = 0; a < 8; ++a) nxt[a] = load_once(xn + 256 * a);      // read once: non-temporal
        }
#else
        const float2 *xs = xb + s * p.step + t;
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = (a < NA) ? (PILOT ? csub(xs[256 * a], pv) : xs[256 * a]) : make_float2(0.f, 0.f);
#endif

        float2 mean = make_float2(0.f, 0.f);
#if OTH_W4096_PIPE
        if (DETREND) {      // only the new half is summed; the other half's total is carried over
            float2 sum = v[8];
#pragma unroll
            for (int a = 9; a < 16; ++a) sum = cadd(sum, v[a]);
            sum.x = wave_total(sum.x);
            sum.y = wave_total(sum.y);
            if ((t & 63) == 0) red[t >> 6] = sum;
            if (s == sb) {
                float2 s0h = v[0];
#pragma unroll
                for (int a = 1; a < 8; ++a) s0h = cadd(s0h, v[a]);
                s0h.x = wave_total(s0h.x);
                s0h.y = wave_total(s0h.y);
                if ((t & 63) == 0) red[4 + (t >> 6)] = s0h;
            }
        }
#else
        if (DETREND) {
            float2 sum = v[0];
#pragma unroll
            for (int a = 1; a < NA; ++a) sum = cadd(sum, v[a]);
            sum.x = wave_total(sum.x);
            sum.y = wave_total(sum.y);
            if ((t & 63) == 0) red[t >> 6] = sum;
        }
#endif
        OTH_STAMP(0);       // loads issued, sums reduced
        lds_barrier();     // A: previous segment's LDS reads are done; red[] visible
        OTH_STAMP(1);       // wait at barrier A
        prio_compute();
        if (sched == 2 && t == 0) {
            // draw the next chunk while this one is being transformed; publish it in the last segment
            if (s == sb) ticket = atomicAdd(p.queue + stream, 1u);
            if (s == se - 1) *lnext = (int)ticket;
        }
        if (DETREND) {
            const float2 s01 = cadd(red[0], red[1]), s23 = cadd(red[2], red[3]);
#if OTH_W4096_PIPE
            if (s == sb) prev_tot = cadd(cadd(red[4], red[5]), cadd(red[6], red[7]));
            const float2 new_tot = cadd(s01, s23);
            mean = make_float2((prev_tot.x + new_tot.x) * (1.0f / 4096.0f), (prev_tot.y + new_tot.y) * (1.0f / 4096.0f));
            prev_tot = new_tot;
#else
            mean = make_float2((s01.x + s23.x) * (1.0f / (256.0f * NA)), (s01.y + s23.y) * (1.0f / (256.0f * NA)));
#endif
        }
#if OTH_W4096_PIPE
        // the eight window loads are the oldest vector-memory operations in flight: at most the eight prefetch loads
        // (and the ticket draw) are younger, so vmcnt(8) has them home without draining the prefetch
        asm volatile("s_waitcnt vmcnt(8)"
                     : "+v"(wu[0]), "+v"(wu[1]), "+v"(wu[2]), "+v"(wu[3]), "+v"(wu[4]), "+v"(wu[5]), "+v"(wu[6]), "+v"(wu[7]));
#pragma unroll
        for (int a = 0; a < 8; ++a) v[a] = make_float2((v[a].x - mean.x) * win[a], (v[a].y - mean.y) * win[a]);
#pragma unroll
        for (int a = 0; a < 8; ++a) v[8 + a] = make_float2((v[8 + a].x - mean.x) * wu[a], (v[8 + a].y - mean.y) * wu[a]);
#else
#pragma unroll
        for (int a = 0; a < NA; ++a) v[a] = make_float2((v[a].x - mean.x) * win[a], (v[a].y - mean.y) * win[a]);
#endif

        // pass 1: DFT over a, twiddle W4096^(k0 t), scatter to region k0
        dft16(v);
        prio_latency();
#if OTH_W4096_PIPE
        scatter_pow16<RS>(v, lx + w1, b1, b4);
#else
        lx[w1] = v[r16(0)];
#pragma unroll
        for (int k0 = 1; k0 < 16; ++k0) lx[k0 * RS + w1] = cmul(v[r16(k0)], tw1[k0]);
#endif
        OTH_STAMP(2);       // detrend, window, pass 1, exchange-1 writes
        lds_barrier();     // B
        OTH_STAMP(3);       // wait at barrier B

        // pass 2: thread (k0,c) gathers b, DFT over b, twiddle W256^(k1 c)
#if OTH_W4096_DIAG
#pragma unroll
        for (int b = 0; b < 16; ++b) v[b] = lx[r1 + b * 17];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        OTH_STAMP(7);       // exchange-1 reads landed
        prio_compute();
        dft16(v);
#else
        dft16_from_lds<17>(v, lx + r1, [] { prio_compute(); });      // ordered reads, counted waits (fft4096.hip.h)
#endif
        prio_latency();
        wave_lds_sync();   // the 16 lanes of this k0 have all read region k0
        scatter_pow16<17>(v, lx + w2, c1, c4);
        wave_lds_sync();
        OTH_STAMP(8);       // second butterfly, W256 twiddles, exchange-2 writes issued

        // pass 3: thread (k0,k1) gathers c, DFT over c, accumulate |X|^2
#if OTH_W4096_DIAG
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = lx[r2 + c];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        OTH_STAMP(9);       // exchange-2 writes + reads landed
        prio_compute();
        dft16(v);
#else
        dft16_from_lds<1>(v, lx + r2, [] { prio_compute(); });
#endif
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const float2 X = v[r16(k2)];
            acc[k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[k2]));
        }
        OTH_STAMP(4);       // passes 2 and 3
      }
      if (sched == 0) break;
      cur = (sched == 1) ? cur + W : (long long)W + *lnext;   // *lnext was written before barrier B
    }

    // bin k0 + 16 k1 + 256 k2 of this workgroup sits at t + 256 k2 (finalize_kernel layout 1)
    float *dst = p.partial + ((size_t)stream * W + wg) * 4096;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) dst[256 * k2 + t] = acc[k2];
#if OTH_W4096_DIAG
    if (t == 0) {   // stamps live behind the partial sums, in memory nothing else reads
        unsigned long long *dbg =
            reinterpret_cast<unsigned long long *>(p.partial + (size_t)p.nstreams * W * 4096) + 4 * ((size_t)stream * W + wg);
        dbg[0] = t_start;
        dbg[1] = __builtin_amdgcn_s_memrealtime();
        dbg[2] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID, bits [3:0]
        dbg[3] = (unsigned long long)(s1 - s0);
    }
    if ((t & 63) == 0) {   // per-wave phase cycle sums, 12 x u64 per wave behind the 32-byte records
        unsigned long long *ph = reinterpret_cast<unsigned long long *>(p.partial + (size_t)p.nstreams * W * 4096) +
                                 4 * (size_t)p.nstreams * W + 12 * (((size_t)stream * W + wg) * 4 + (t >> 6));
#pragma unroll
        for (int i = 0; i < 12; ++i) ph[i] = phase[i];
    }
#endif
}

}  // namespace

// resident 256-thread workgroups per CU for this build of the kernel (VGPR / LDS limited)
int OTH_CAT(tuned4096_blocks_per_cu_, OTH_W4096_TAG)() {
    static int cached = 0;      // same answer for every device of the node (all gfx950)
    if (cached) return cached;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, welch4096_kernel<true, 16>, T4, LDS_BYTES) != hipSuccess || n < 1)
        n = 2;
    return cached = n;
}

hipError_t OTH_CAT(launch_welch_tuned4096_, OTH_W4096_TAG)(const WelchArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
#define OTH_W4096_LAUNCH(NA)                                                                         \
    case 256 * NA:                                                                                   \
        if (a.detrend && a.pilot)                                                                    \
            hipLaunchKernelGGL((welch4096_kernel<true, NA, true>), grid, dim3(T4), LDS_BYTES, s, a); \
        else if (a.detrend)                                                                          \
            hipLaunchKernelGGL((welch4096_kernel<true, NA>), grid, dim3(T4), LDS_BYTES, s, a);      \
        else                                                                                         \
            hipLaunchKernelGGL((welch4096_kernel<false, NA>), grid, dim3(T4), LDS_BYTES, s, a);     \
        break;
    switch (a.nperseg) {
        OTH_W4096_LAUNCH(16)
#if !OTH_W4096_PIPE
        OTH_W4096_LAUNCH(8)
        OTH_W4096_LAUNCH(4)
        OTH_W4096_LAUNCH(2)
        OTH_W4096_LAUNCH(1)
#endif
        default:
            return hipErrorInvalidValue;
    }
#undef OTH_W4096_LAUNCH
    return hipGetLastError();
}

}  // namespace oth
