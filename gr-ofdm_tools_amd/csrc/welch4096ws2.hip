// welch4096ws2: the headline Welch average (nperseg = nfft = 4096, 50 % overlap) in the two-pair shape of
// csd4096ws.hip - ONE 1024-thread workgroup per CU instead of two 512-thread ones.
//
// The stream is cut in two runs of segments, A = [0, nseg / 2) and B = [nseg / 2, nseg) (the launcher passes B's first
// sample as WelchArgs.y and nseg / 2 as WelchArgs.nseg).  Threads 0..255 (PA) and 256..511 (PB) are the producers of the
// two runs (loads, window, pass 1, exchange-1 writes - welch4096ws.hip's producer on their own pair of LDS images);
// 512..1023 are eight consumer waves in which lanes 0..31 consume run A and lanes 32..63 run B for the same bins
// (pass 2, exchange 2, pass 3, |X|^2 accumulation: the headline consumer, sixteen accumulators and fifteen stored
// twiddles per thread).  Both runs follow the same chunk schedule, one LDS-only barrier per step for all sixteen waves.
// Each workgroup writes two partial rows (finalize layout 1).  Built as an A/B against welch4096ws.hip
// (OTH_W4096_VARIANT=ws2 / oth_plan_set_tuning("ws2")): DESIGN.md 4.1.
#include <type_traits>
#include "fft4096.hip.h"

namespace oth {
namespace {

constexpr int T2 = 1024;
constexpr int CS_RED = 32;                 // float2 per pair: per image the four producer waves' segment sums
constexpr int CS_CTRL = 16;                // ints per pair: item kind per image [0..1], next-chunk ticket [4] (pair 0)
constexpr size_t CS_PAIR_BYTES = (2 * LDS_X + CS_RED) * sizeof(float2) + CS_CTRL * sizeof(int);
constexpr size_t CS_FW_BYTES = 256 * sizeof(float4);      // window-spectrum entries of the detrend, one per t
constexpr size_t W2_LDS_BYTES = 2 * CS_PAIR_BYTES + CS_FW_BYTES;

enum { W2_STOP = 0, W2_DATA = 1, W2_BUBBLE = 2 };

template <bool DETREND>
__global__ __launch_bounds__(T2, 4) void welch4096ws2_kernel(WelchArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const bool producer = tid < 512;
    const int cidx = tid - 512;                                   // consumers: run A in lanes 0..31, run B in lanes 32..63
    const int pair = producer ? __builtin_amdgcn_readfirstlane(tid >> 8) : ((cidx >> 5) & 1);
    unsigned char *base = smem + pair * CS_PAIR_BYTES;
    float2 *img = reinterpret_cast<float2 *>(base);             // two images of LDS_X float2
    float2 *red = img + 2 * LDS_X;
    int *ctrl = reinterpret_cast<int *>(red + CS_RED);
    int *ctrl0 = reinterpret_cast<int *>(reinterpret_cast<float2 *>(smem) + 2 * LDS_X + CS_RED);   // pair 0's: the ticket

    const int t = producer ? (tid & 255) : (((cidx >> 6) << 5) | (cidx & 31));
    const int hi = t >> 4, lo = t & 15;
    const int wave = t >> 6;
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const float2 *xb = (pair ? p.y : p.x) + (size_t)stream * p.stream_stride;
    const int sched = p.sched;
    const long long nchunks = sched ? chunk_count(p) : 1;
    const int w1 = hi * 17 + lo, r1 = hi * RS + lo, w2 = hi * RS + lo, r2 = hi * RS + lo * 17;

    if (producer) {
        // ------------------------------------------------------------------ producer (welch4096ws.hip)
        float win[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) win[a] = p.win[256 * a + t];
        const float2 b1 = p.tw[t], b4 = p.tw[4 * t];
        float2 kw[8], nxt[8];
        float2 prev_new = make_float2(0.f, 0.f);
        int it = 0;
        unsigned ticket = 0;
        using std::false_type;
        using std::true_type;
        using mid = std::integral_constant<int, 0>;
        using head = std::integral_constant<int, 1>;
        using none = std::integral_constant<int, 2>;
        auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        auto load_chunk_head = [&](int first_seg) {
            const float2 *xs = xb + (size_t)uni(first_seg) * 2048;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 *xj = xs + 512 * j;
                kw[2 * j] = xj[(unsigned)t];
                kw[2 * j + 1] = xj[(unsigned)t + 256u];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 *xj = xs + 2048 + 512 * j;
                nxt[2 * j] = load_once(xj + (unsigned)t);
                nxt[2 * j + 1] = load_once(xj + ((unsigned)t + 256u));
            }
        };
        auto step_end = [&](int item) {
            if (t == 0) ctrl[it & 1] = item;
                        lds_barrier();
                        ++it;
        };
        auto item = [&](auto first_, auto mode_, int s, int nsb, bool publish) {
            constexpr bool FIRST = decltype(first_)::value;
            constexpr int MODE = decltype(mode_)::value;
            const int q = it & 1;
            float2 *lx = img + q * LDS_X;
            __builtin_amdgcn_s_setprio(2);
            float2 v[16];
            float2 sumf = make_float2(0.f, 0.f), sum = make_float2(0.f, 0.f);
            if (FIRST) {
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    sumf = cadd(sumf, kw[a]);
                    kw[a] = make_float2(kw[a].x * win[a], kw[a].y * win[a]);
                }
            }
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const float2 r = nxt[a];
                v[a] = kw[a];
                v[8 + a] = make_float2(r.x * win[8 + a], r.y * win[8 + a]);
                if (MODE == 0) kw[a] = make_float2(r.x * win[a], r.y * win[a]);
                sum = cadd(sum, r);
            }
            if (sched == 2 && t == 0 && pair == 0) {      // one ticket stream for both pairs
                if (FIRST) ticket = atomicAdd(p.queue + stream, 1u);
                if (publish) ctrl0[4] = (int)ticket;
            }
            if (MODE == 0) {
                const float2 *xn = xb + (size_t)uni(s + 2) * 2048;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float2 *xj = xn + 512 * j;
                    nxt[2 * j] = load_once(xj + (unsigned)t);
                    nxt[2 * j + 1] = load_once(xj + ((unsigned)t + 256u));
                }
            } else if (MODE == 1) {
                load_chunk_head(nsb);
            }
            if (DETREND) {
                sum.x = wave_total_lane63(sum.x);
                sum.y = wave_total_lane63(sum.y);
                float2 other = prev_new;
                if (FIRST) other = make_float2(wave_total_lane63(sumf.x), wave_total_lane63(sumf.y));
                if ((t & 63) == 63) red[q * 8 + wave] = cadd(sum, other);
                prev_new = sum;
            }
            __builtin_amdgcn_s_setprio(0);
            dft16(v);
            __builtin_amdgcn_s_setprio(2);
            scatter_pow16<RS>(v, lx + w1, b1, b4);
            step_end(W2_DATA);
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
        while (have) {
            const int n = se - sb;
            int ncur = 0;
            if (n >= 2) {
                item(true_type{}, mid{}, sb, 0, n == 2);
                int s = sb + 1;
                if (s < se - 1) item(false_type{}, mid{}, s++, 0, true);
                for (; s + 1 < se - 1; s += 2) {
                    item(false_type{}, mid{}, s, 0, false);
                    item(false_type{}, mid{}, s + 1, 0, false);
                }
                if (s < se - 1) item(false_type{}, mid{}, s, 0, false);
                ncur = (sched == 1) ? cur + W : W + uni(ctrl0[4]);
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
            item(true_type{}, none{}, sb, 0, true);
            if (sched == 0) break;
            if (sched == 2) {
                step_end(W2_BUBBLE);
                ncur = W + uni(ctrl0[4]);
            } else {
                ncur = cur + W;
            }
            have = open_chunk(ncur);
            if (have) load_chunk_head(sb);
        }
        step_end(W2_STOP);
    } else {
        // ------------------------------------------------------------------ consumer (welch4096ws.hip's, per lane half)
        float2 tw2[16];
#pragma unroll
        for (int k = 1; k < 16; ++k) tw2[k] = p.tw[16 * lo * k];      // W256^(k1 c), c = lo
        float4 *fwl = reinterpret_cast<float4 *>(smem + 2 * CS_PAIR_BYTES);
        if (DETREND && pair == 0) fwl[t] = p.fd[t];      // lanes l and l + 32 (same t) are in one wave: ordered
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        float2 v[16];
        int it = 0;
        auto next_item = [&]() -> int {
            __builtin_amdgcn_s_setprio(2);
            lds_barrier();
            const int q = it & 1;
            const float2 *lq = img + q * LDS_X;
            const int kind = __builtin_amdgcn_readfirstlane(ctrl0[q]);      // both runs follow the same schedule
            dft16_from_lds<17>(v, lq + r1, [] { __builtin_amdgcn_s_setprio(1); });
            ++it;
            return kind;
        };
        int item = next_item();
        for (;;) {
            while (item == W2_BUBBLE) item = next_item();
            if (item == W2_STOP) break;
            const int q = (it & 1) ^ 1;   // the image whose pass 2 sits in v
            float2 *lx = img + q * LDS_X;
            __builtin_amdgcn_s_setprio(2);
            lx[w2] = v[r16(0)];
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) lx[w2 + k1 * 17] = cmul(v[r16(k1)], tw2[k1]);
            wave_lds_sync();
            float4 fw = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 h0 = make_float2(0.f, 0.f), h1 = h0, h2 = h0, h3 = h0;
            if (DETREND) {      // image q's sums stay valid until the barrier of the next step
                fw = fwl[t];
                h0 = red[q * 8], h1 = red[q * 8 + 1], h2 = red[q * 8 + 2], h3 = red[q * 8 + 3];
            }
            dft16_from_lds<1>(v, lx + r2, [] { __builtin_amdgcn_s_setprio(1); });
            if (DETREND) {
                const float2 tot = cadd(cadd(h0, h1), cadd(h2, h3));
                const float2 mean = make_float2(tot.x * (1.0f / 4096.0f), tot.y * (1.0f / 4096.0f));
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
            item = next_item();
        }
        // two rows per workgroup (run A, run B); bin k0 + 16 k1 + 256 k2 at t + 256 k2 (finalize layout 1)
        float *dst = p.partial + (((size_t)stream * W + wg) * 2 + pair) * 4096;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) dst[256 * k2 + t] = acc[k2];
    }
}

}  // namespace

int tuned4096_blocks_per_cu_ws2() {
    static int cached = 0;
    if (cached) return cached;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, welch4096ws2_kernel<true>, T2, W2_LDS_BYTES) != hipSuccess || n < 1)
        n = 1;
    return cached = n;
}

// a.y = first sample of run B, a.nseg = segments per run
hipError_t launch_welch_tuned4096_ws2(const WelchArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    static bool armed[64] = {};        // 143 KiB of dynamic LDS needs the opt-in, once per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63, armed[63] = false;
    bool &big_lds = armed[dev];
    if (!big_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(welch4096ws2_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(welch4096ws2_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES);
        if (e != hipSuccess) return e;
        big_lds = true;
    }
    if (a.detrend)
        hipLaunchKernelGGL((welch4096ws2_kernel<true>), grid, dim3(T2), W2_LDS_BYTES, s, a);
    else
        hipLaunchKernelGGL((welch4096ws2_kernel<false>), grid, dim3(T2), W2_LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace oth
