// welch4096ws: wave-specialised form of the headline kernel (nperseg = nfft = 4096, 50 % overlap).
//
// Same arithmetic as welch4096.hip (radix-16 x 16 x 16, same LDS image), but a 512-thread workgroup is
// split into a PRODUCER half (threads 0..255: loads, window, pass 1, exchange-1 writes) and a CONSUMER
// half (threads 256..511: pass 2, exchange 2, pass 3, |X|^2 accumulation) that work one segment apart on
// two LDS images.  Why: in the one-role kernel every thread carries the constants of all three passes
// (window, both twiddle sets, accumulators, the kept half and the prefetch) and has to rebuild both
// twiddle sets every segment to stay at 128 VGPRs.  Split by role, the producer holds window + data
// pipeline and rebuilds only its twiddles, the consumer holds its 15 twiddles and the accumulators in
// registers: 9 % fewer VALU instructions per segment, one workgroup barrier per segment instead of two,
// and the loads of the producer overlap the butterflies of the consumer by construction.
//
// The constant detrend runs in the frequency domain: FFT((x - m) w) = FFT(x w) - m FFT(w).  For every
// window whose spectrum is confined to bins [0,256) U [3840,4096) (host check at plan time; exact for the
// periodic cosine-sum windows scipy.signal.welch builds) those are the k2 = 0 and k2 = 15 outputs of
// pass 3, so each consumer thread fixes two of its sixteen bins with its own pair of FFT(w) values
// (WelchArgs.fd).  The producer therefore never needs the mean: it publishes per-wave sums of the raw
// samples next to the image, and windows samples as soon as they arrive.
//
// Per step `it` (one LDS-only barrier):  producer: segment it -> image[it & 1];  consumer: right behind the
// barrier pass 2 of image[it & 1] (its LDS reads in one batch with the item word), then, once the item word
// says it is a segment, exchange 2, pass 3 and the accumulation.  The consumer is the critical path (443
// VALU instructions per step against the producer's 336): its butterflies run at a higher wave priority.
#ifndef OTH_WS_TAG
#define OTH_WS_TAG ws
#endif
// The samples are read once (a chunk's first half twice, by two workgroups far apart in time): non-temporal
// loads keep them from displacing the twiddle / window tables and the partial sums in L2 (-1.4 % kernel time).
#ifndef OTH_WS_NT_LOADS
#define OTH_WS_NT_LOADS 1
#endif
#if OTH_WS_NT_LOADS
#define OTH_WS_LOAD(p) load_once(p)
#else
#define OTH_WS_LOAD(p) (*(p))
#endif
#if OTH_WS_NT_LOADS > 1
#define OTH_WS_LOAD_HEAD(p) load_once(p)
#else
#define OTH_WS_LOAD_HEAD(p) (*(p))
#endif
// 1: the producer keeps six pass-1 twiddle powers (W^1,2,3,4,8,12: nine products per segment instead of thirteen) and
// pays for their eight registers by reading the second half of its window values from an 8 KiB LDS table per step
// (same-box A/B, five interleaved runs each: 0.5855 against 0.5906 ms = -0.9 %, profiles/r03_ab_headline_pow6.txt)
#ifndef OTH_WS_POW6
#define OTH_WS_POW6 1
#endif
#ifndef OTH_WS_SPREAD
#define OTH_WS_SPREAD 0      // A/B: 2 = four loads before and four after the producer's butterfly; 3 = 2 + 4 + 2 (after the exchange writes)
#endif
#ifndef OTH_WS_DIAG
#define OTH_WS_DIAG 0        // 1: per-wave phase cycle counters behind the partial sums (tools/archive/diag_ws.py)
#endif
// wave priorities: producer latency sections / butterflies, consumer latency sections / butterflies
#ifndef OTH_WS_PAL
#define OTH_WS_PAL 2
#endif
#ifndef OTH_WS_PAC
#define OTH_WS_PAC 0
#endif
#ifndef OTH_WS_PAS           // producer: twiddles + exchange-1 writes
#define OTH_WS_PAS OTH_WS_PAL
#endif
#ifndef OTH_WS_PBL
#define OTH_WS_PBL 2
#endif
#ifndef OTH_WS_PBC
#define OTH_WS_PBC 1         // the consumer is the critical path: its butterflies go ahead of the producer's (-4.5 %)
#endif
#define OTH_CAT2(a, b) a##b
#define OTH_CAT(a, b) OTH_CAT2(a, b)

#include <type_traits>
#include "fft4096.hip.h"

namespace oth {
namespace {

#if OTH_WS_DIAG
#define WS_STAMP(i)                                                      \
    do {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        __builtin_amdgcn_s_waitcnt(0xC07F);                              \
        phase[i] += now_ - last_;                                        \
        last_ = now_;                                                    \
        __builtin_amdgcn_sched_barrier(0);                               \
    } while (0)
#else
#define WS_STAMP(i)
#endif

constexpr int TWS = 512;
constexpr int WS_RED = 32;                 // float2: per image the four producer waves' segment sums (8 slots each)
constexpr int WS_CTRL = 16;                // ints: item kind per image [0..1], next-chunk ticket [4]
constexpr size_t WS_WIN_BYTES = OTH_WS_POW6 ? 256 * 2 * sizeof(float4) : 0;      // window values 8..15 of every producer thread
constexpr size_t WS_LDS_BYTES = (2 * LDS_X + WS_RED) * sizeof(float2) + WS_CTRL * sizeof(int) + WS_WIN_BYTES;

enum { ITEM_STOP = 0, ITEM_DATA = 1, ITEM_BUBBLE = 2 };

template <bool DETREND, bool PILOT = false>
__global__ __launch_bounds__(TWS, 4) void welch4096ws_kernel(WelchArgs p) {
    static_assert(DETREND || !PILOT, "the pilot belongs to the detrend");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *img = reinterpret_cast<float2 *>(smem);             // two images of LDS_X float2
    float2 *red = img + 2 * LDS_X;                              // [2][8]
    int *ctrl = reinterpret_cast<int *>(red + WS_RED);          // item kind [0..1], next ticket [4]; re-read after every
                                                                // lds_barrier() (it is a compiler memory barrier)

    const int tid = threadIdx.x;
    const bool producer = tid < 256;
    const int t = tid & 255;
    const int hi = t >> 4, lo = t & 15;
    const int wave = t >> 6;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const float2 *xb = p.x + (size_t)stream * p.stream_stride;
    const int sched = p.sched;
    const long long nchunks = sched ? chunk_count(p) : 1;

    // LDS addresses inside an image (float2): exchange-1 write/read, exchange-2 write/read
    const int w1 = hi * 17 + lo, r1 = hi * RS + lo, w2 = hi * RS + lo, r2 = hi * RS + lo * 17;
#if OTH_WS_DIAG
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif

    if (producer) {
        // ------------------------------------------------------------------ producer
        float win[OTH_WS_POW6 ? 8 : 16];
#if OTH_WS_POW6
        float4 *wl = reinterpret_cast<float4 *>(ctrl + WS_CTRL) + t;      // [2][256] float4: win[8..11], win[12..15]
#pragma unroll
        for (int a = 0; a < 8; ++a) win[a] = p.win[256 * a + t];
        wl[0] = make_float4(p.win[256 * 8 + t], p.win[256 * 9 + t], p.win[256 * 10 + t], p.win[256 * 11 + t]);
        wl[256] = make_float4(p.win[256 * 12 + t], p.win[256 * 13 + t], p.win[256 * 14 + t], p.win[256 * 15 + t]);
        const float2 b1 = p.tw[t], b2 = p.tw[2 * t], b3 = p.tw[3 * t], b4 = p.tw[4 * t], b8 = p.tw[8 * t],
                     b12 = p.tw[(12 * t) & 4095];
#else
#pragma unroll
        for (int a = 0; a < 16; ++a) win[a] = p.win[256 * a + t];
        // pass-1 twiddle seeds W4096^t, W4096^(4t): the fifteen twiddles are multiplied out per segment (keeping
        // even W^2, W^3, W^8, W^12 in registers as well spills at the 128-VGPR cap)
        const float2 b1 = p.tw[t], b4 = p.tw[4 * t];
#endif
        float2 kw[8], nxt[8];
        float2 prev_new = make_float2(0.f, 0.f);     // this wave's sum of the previous segment's new half
        // PILOT (every detrending plan but OTH_DETREND_CONSTANT_FAST): WelchArgs.pilot comes off every sample as it arrives, so the transform and
        // the sums see x - pilot (two scalar registers, two subtractions per sample: +1 % on the launch)
        // pilot_inline (round 5): formed in the launch from eight 2 KiB probes (below, behind the first sample loads)
        // instead of by a launch in front of this one
        float2 pv = make_float2(0.f, 0.f);
        if (PILOT && !p.pilot_inline) pv = load_pilot(p.pilot, stream);
        int it = 0;
        unsigned ticket = 0;
        using std::false_type;
        using std::true_type;
        using mid = std::integral_constant<int, 0>;     // prefetch the half after next of this chunk
        using head = std::integral_constant<int, 1>;    // last segment of the chunk: prefetch the next chunk's first segment
        using none = std::integral_constant<int, 2>;    // nothing to prefetch

        // segment indices are wave-uniform ints pinned to scalar registers (the launcher checks nseg < 2^30), so
        // the loads use scalar base + per-thread offset and cost no vector address arithmetic
        auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        auto load_chunk_head = [&](int first_seg) {
            const float2 *xs = xb + (size_t)uni(first_seg) * 2048;      // scalar base, unsigned 32-bit lane offset
#pragma unroll
            for (int j = 0; j < 4; ++j) {      // one scalar base per pair of rows: offsets t and t + 256 (immediate)
                const float2 *xj = xs + 512 * j;
                kw[2 * j] = OTH_WS_LOAD_HEAD(xj + (unsigned)t);      // raw: the chunk's first item windows them in place
                kw[2 * j + 1] = OTH_WS_LOAD_HEAD(xj + ((unsigned)t + 256u));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 *xj = xs + 2048 + 512 * j;
                nxt[2 * j] = OTH_WS_LOAD(xj + (unsigned)t);
                nxt[2 * j + 1] = OTH_WS_LOAD(xj + ((unsigned)t + 256u));
            }
        };
        auto step_end = [&](int item) {
            if (t == 0) ctrl[it & 1] = item;
            lds_barrier();
            WS_STAMP(4);
            ++it;
        };
        // One segment: image (it & 1).  Every flavour leaves the same state behind - kw = windowed first half of the
        // following segment, nxt = its second half in flight - so the hot (false, mid) flavour is straight-line code
        // with one group of eight loads per step.
        auto item = [&](auto first_, auto mode_, int s, int nsb, bool publish) {
            constexpr bool FIRST = decltype(first_)::value;
            constexpr int MODE = decltype(mode_)::value;
            const int q = it & 1;
            float2 *lx = img + q * LDS_X;
            __builtin_amdgcn_s_setprio(OTH_WS_PAL);
            float2 v[16];
#if OTH_WS_DIAG
            WS_STAMP(5);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            WS_STAMP(0);
#endif
            float2 sumf = make_float2(0.f, 0.f), sum = make_float2(0.f, 0.f);
#if OTH_WS_POW6
            const float4 wa = wl[0], wb = wl[256];      // own slots: no barrier needed
            const float wh[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};      // window values 8..15, this step only
#else
            const float *wh = win + 8;
#endif
            if (FIRST) {
#pragma unroll
                for (int a = 0; a < 8; ++a) {      // kw still holds the raw first half of the chunk's first segment
                    if (PILOT) kw[a] = csub(kw[a], pv);
                    sumf = cadd(sumf, kw[a]);
                    kw[a] = make_float2(kw[a].x * win[a], kw[a].y * win[a]);
                }
            }
#pragma unroll
            for (int a = 0; a < 8; ++a) {      // the new half is windowed for both of its roles as it arrives
                const float2 r = PILOT ? csub(nxt[a], pv) : nxt[a];
                v[a] = kw[a];
                v[8 + a] = make_float2(r.x * wh[a], r.y * wh[a]);
                if (MODE == 0) kw[a] = make_float2(r.x * win[a], r.y * win[a]);      // (else kw is reloaded below)
                sum = cadd(sum, r);
            }
            // dynamic schedule: the ticket of the chunk after this one is drawn with the chunk's first segment and
            // published (a wait for the atomic's return) before this step's loads go out, one step later if possible
            if (sched == 2 && t == 0) {
                if (FIRST) ticket = atomicAdd(p.queue + stream, 1u);
                if (publish) ctrl[4] = (int)ticket;
            }
            const float2 *xn = xb + (size_t)uni(s + 2) * 2048;      // (MODE 0 only)
            auto load_pairs = [&](int j0, int j1) {
#pragma unroll
                for (int j = j0; j < j1; ++j) {
                    const float2 *xj = xn + 512 * j;
                    nxt[2 * j] = OTH_WS_LOAD(xj + (unsigned)t);
                    nxt[2 * j + 1] = OTH_WS_LOAD(xj + ((unsigned)t + 256u));
                }
            };
            if (MODE == 0) {
#if OTH_WS_SPREAD      // A/B (round 5): the eight loads of a step at two / three places instead of one burst
                __builtin_amdgcn_sched_barrier(0);
                load_pairs(0, OTH_WS_SPREAD == 2 ? 2 : 1);
                __builtin_amdgcn_sched_barrier(0);
#else
                load_pairs(0, 4);
#endif
            } else if (MODE == 1) {
                load_chunk_head(nsb);
            }
            if (DETREND) {      // per-wave sums of both halves of this segment, side by side for the consumer
                sum.x = wave_total_lane63(sum.x);      // (these live in lane 63 of each wave only)
                sum.y = wave_total_lane63(sum.y);
                float2 other = prev_new;
                if (FIRST) other = make_float2(wave_total_lane63(sumf.x), wave_total_lane63(sumf.y));
                if ((t & 63) == 63) red[q * 8 + wave] = cadd(sum, other);
                prev_new = sum;
            }
            WS_STAMP(1);
            __builtin_amdgcn_s_setprio(OTH_WS_PAC);
            dft16(v);
            WS_STAMP(2);
            __builtin_amdgcn_s_setprio(OTH_WS_PAS);
#if OTH_WS_SPREAD
            if (MODE == 0) {
                __builtin_amdgcn_sched_barrier(0);
                load_pairs(OTH_WS_SPREAD == 2 ? 2 : 1, OTH_WS_SPREAD == 2 ? 4 : 3);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
#if OTH_WS_POW6
            scatter_pow16_six<RS>(v, lx + w1, b1, b2, b3, b4, b8, b12);
#else
            scatter_pow16<RS>(v, lx + w1, b1, b4);
#endif
#if OTH_WS_DIAG
            __builtin_amdgcn_s_waitcnt(0xC07F);
            WS_STAMP(3);
#endif
#if OTH_WS_SPREAD == 3
            if (MODE == 0) {
                __builtin_amdgcn_sched_barrier(0);
                load_pairs(3, 4);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            step_end(ITEM_DATA);
        };

        int cur = 0, sb = 0, se = 0;
        auto range = [&](int c, int &b, int &e) {
            long long lb, le;
            chunk_range(p, c, lb, le);
            b = uni((int)lb);
            e = uni((int)le);
        };
        auto open_chunk = [&](int c) -> bool {
            cur = c;
            if (sched) {
                if (cur >= nchunks) return false;
                range(cur, sb, se);
                return true;
            }
            sb = uni((int)((p.nseg * wg) / W));
            se = uni((int)((p.nseg * (wg + 1)) / W));
            return sb < se;
        };
        bool have = open_chunk(sched ? wg : 0);
        if (have) load_chunk_head(sb);
        if (PILOT && p.pilot_inline) {
            // the probe loads go out behind the chunk head's (one memory round trip for both); per-wave totals into
            // image 1, which no step writes before the barrier of step 0; one extra workgroup barrier (the consumers
            // take it in front of their loop)
            // Contiguous runs (short launches - fewer than 32 segments per workgroup - and OTH_SCHED_CONTIGUOUS): the eight
            // probes are spread over THIS workgroup's run, so the pilot follows an offset that moves through the launch
            // (round 6: a drift of 1200 sigma over 2047 segments read 1.05e-4 with one pilot per launch; every segment is
            // one workgroup's, so the pilots need not agree between workgroups).  Chunked schedules keep the launch-wide
            // probes: a workgroup's chunks lie anywhere in the stream.
            const bool own = sched == 0 && have;
            const PilotProbes probes = inline_pilot_load(own ? xb + (size_t)sb * 2048 : xb, own ? (long long)(se - sb) : p.nseg, 2048, t);
            inline_pilot_store(probes, t, img + LDS_X);
            lds_barrier();
            pv = inline_pilot_value(img + LDS_X);
        }
        while (have) {
            const int n = se - sb;
            int ncur = 0;
            if (n >= 2) {
                item(true_type{}, mid{}, sb, 0, n == 2);
                int s = sb + 1;
                if (s < se - 1) item(false_type{}, mid{}, s++, 0, true);
                for (; s + 1 < se - 1; s += 2) {      // two per trip: kw's registers swap roles instead of being copied
                    item(false_type{}, mid{}, s, 0, false);
                    item(false_type{}, mid{}, s + 1, 0, false);
                }
                if (s < se - 1) item(false_type{}, mid{}, s, 0, false);
                // last segment: the next chunk's ticket was published at least one barrier ago
                ncur = (sched == 1) ? cur + W : W + uni(ctrl[4]);
                int nsb = 0, nse = 0;
                const bool have_next = sched && ncur < nchunks;
                if (have_next) {
                    range(ncur, nsb, nse);
                    item(false_type{}, head{}, se - 1, nsb, false);
                    cur = ncur;
                    sb = nsb;
                    se = nse;
                    continue;
                }
                item(false_type{}, none{}, se - 1, 0, false);
                break;
            }
            // one-segment chunk: no prefetch across the chunk boundary; the ticket needs a barrier to become visible
            item(true_type{}, none{}, sb, 0, true);
            if (sched == 0) break;
            if (sched == 2) {
                step_end(ITEM_BUBBLE);
                ncur = W + uni(ctrl[4]);
            } else {
                ncur = cur + W;
            }
            have = open_chunk(ncur);
            if (have) load_chunk_head(sb);
        }
        step_end(ITEM_STOP);
    } else {
        // ------------------------------------------------------------------ consumer
        float2 tw2[16];
#pragma unroll
        for (int k = 1; k < 16; ++k) tw2[k] = p.tw[16 * lo * k];      // W256^(k1 c), c = lo
        float4 fw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (DETREND) fw = p.fd[t];      // FFT(w) at this thread's k2 = 0 and k2 = 15 bins
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        float2 v[16];
        float2 mean = make_float2(0.f, 0.f);
        int it = 0;
        // barrier of step `it`, then what the producer left in image it & 1: the item kind and, in the same batch
        // of LDS reads (harmless when it is not a segment), the exchange-1 reads and the per-wave sums
        auto next_item = [&]() -> int {
            __builtin_amdgcn_s_setprio(OTH_WS_PBL);
            lds_barrier();
            WS_STAMP(4);
            const int q = it & 1;
            const float2 *lq = img + q * LDS_X;
            // one batch of LDS reads: the item word and the four per-wave sums (used only after the butterfly, so
            // their wait falls behind it), then pass 2's sixteen reads as full-rate ds_read_b64 on counted waits.
            // Pass 2 therefore runs before the item word is looked at - on junk in an idle step and in the last.
            const int kind = ctrl[q];
            const float2 h0 = red[q * 8], h1 = red[q * 8 + 1], h2 = red[q * 8 + 2], h3 = red[q * 8 + 3];
            dft16_from_lds<17>(v, lq + r1, [] { __builtin_amdgcn_s_setprio(OTH_WS_PBC); });
            if (DETREND) {
                const float2 tot = cadd(cadd(h0, h1), cadd(h2, h3));
                mean = make_float2(tot.x * (1.0f / 4096.0f), tot.y * (1.0f / 4096.0f));
            }
            ++it;
            return kind;
        };
        if (PILOT && p.pilot_inline) lds_barrier();      // the producers' pilot barrier
        int item = next_item();           // nothing to consume in step 0
        for (;;) {
            // idle steps stay out of the path that updates the accumulators (with both in one conditional the
            // sixteen accumulators were copied twice per step)
            while (item == ITEM_BUBBLE) item = next_item();
            if (item == ITEM_STOP) break; // the producer left after the barrier of the step that published it
            float2 *lx = img + ((it & 1) ^ 1) * LDS_X;      // v holds pass 2 of image (it - 1) & 1
#if OTH_WS_DIAG
            WS_STAMP(5);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            WS_STAMP(0);
#endif
            WS_STAMP(1);
            __builtin_amdgcn_s_setprio(OTH_WS_PBL);
            lx[w2] = v[r16(0)];           // in place: each thread rewrites exactly the sixteen elements it read
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) lx[w2 + k1 * 17] = cmul(v[r16(k1)], tw2[k1]);
            wave_lds_sync();              // exchange 2 stays inside the wave: program order is enough
            WS_STAMP(2);
            // exchange-2 reads as ordered ds_read_b64, the first butterfly layer on counted waits (-1.7 % kernel
            // time against the sixteen plain reads, which hipcc pairs into ds_read2_b64 behind one lgkmcnt(0);
            // the same treatment of the exchange-1 reads, which needs the butterfly before the item word is
            // looked at, gave 1.3 % back)
            dft16_from_lds<1>(v, lx + r2, [] { __builtin_amdgcn_s_setprio(OTH_WS_PBC); });
            if (DETREND) {                // X[k] -= mean * FFT(w)[k] where FFT(w) is not negligible
                v[r16(0)] = make_float2(v[r16(0)].x - (mean.x * fw.x - mean.y * fw.y),
                                        v[r16(0)].y - (mean.x * fw.y + mean.y * fw.x));
                v[r16(15)] = make_float2(v[r16(15)].x - (mean.x * fw.z - mean.y * fw.w),
                                         v[r16(15)].y - (mean.x * fw.w + mean.y * fw.z));
            }
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                const float2 X = v[r16(k2)];
                acc[k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[k2]));
            }
            WS_STAMP(3);
            item = next_item();
        }
        // bin k0 + 16 k1 + 256 k2 of this workgroup sits at t + 256 k2 (finalize_kernel layout 1)
        float *dst = p.partial + ((size_t)stream * W + wg) * 4096;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) dst[256 * k2 + t] = acc[k2];
    }
#if OTH_WS_DIAG
    if ((tid & 63) == 0) {   // 8 waves x 8 phase counters per workgroup, behind the partial sums
        unsigned long long *ph = reinterpret_cast<unsigned long long *>(p.partial + (size_t)p.nstreams * W * 4096) +
                                 64 * ((size_t)stream * W + wg) + 8 * (tid >> 6);
#pragma unroll
        for (int i = 0; i < 8; ++i) ph[i] = phase[i];
    }
#endif
}

}  // namespace

int OTH_CAT(tuned4096_blocks_per_cu_, OTH_WS_TAG)() {
    static int cached = 0;
    if (cached) return cached;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, welch4096ws_kernel<true, false>, TWS, WS_LDS_BYTES) != hipSuccess || n < 1)
        n = 1;
    return cached = n;
}

hipError_t OTH_CAT(launch_welch_tuned4096_, OTH_WS_TAG)(const WelchArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    static bool armed[64] = {};        // 70 KiB of dynamic LDS needs the opt-in, once per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63, armed[63] = false;
    bool &big_lds = armed[dev];
    if (!big_lds) {
        hipError_t e = hipSuccess;
        for (const void *fn : {reinterpret_cast<const void *>(welch4096ws_kernel<true, true>),
                               reinterpret_cast<const void *>(welch4096ws_kernel<true, false>),
                               reinterpret_cast<const void *>(welch4096ws_kernel<false, false>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WS_LDS_BYTES);
        if (e != hipSuccess) return e;
        big_lds = true;
    }
    if (a.detrend && (a.pilot || a.pilot_inline))
        hipLaunchKernelGGL((welch4096ws_kernel<true, true>), grid, dim3(TWS), WS_LDS_BYTES, s, a);
    else if (a.detrend)
        hipLaunchKernelGGL((welch4096ws_kernel<true, false>), grid, dim3(TWS), WS_LDS_BYTES, s, a);
    else
        hipLaunchKernelGGL((welch4096ws_kernel<false, false>), grid, dim3(TWS), WS_LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace oth
