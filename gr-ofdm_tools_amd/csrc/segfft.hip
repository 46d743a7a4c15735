// segfft: segment transforms of N = 256 R points (R = 1, 2, 4, 8, 16 -> N = 256 ... 4096) by one TEAM of
// 16 R threads, 16 points per thread, with everything that follows the transform fused into the launch:
//
//   Welch average          sum over segments of |X|^2, any step (scipy.signal.welch, ofdm_cr_tools.py:214,322,342)
//   periodogram chain      stream_to_vector -> keep_one_in_n -> fft_vcc -> |.| or |.|^2 [x 1/N^2]
//                          [-> single_pole_iir -> nlog10] [peak hold]  (spectrum_sensor_v2.py:85-93,
//                          psd_logger.py:43-53, local_worker.py:58-69, multichannel_scanner.py:78-86):
//                          the kept vectors are segments with step = keep_n * N; IIR and peak hold become
//                          weighted-sum / max accumulations over the launch (y_n = (1-a)^n y_0 + sum_s a (1-a)^(n-1-s) x_s),
//                          only the rows the caller asked for are written.
//
// Decomposition N = 16 x 16 x R, decimation in frequency, thread t = R b + c:
//   sample index  n = 16R a + R b + c                bin index  k = k0 + 16 k1 + 256 k2
//   pass 1  thread (b, c)  holds a  = 0..15  -> k0, times W_N^(k0 t)
//   pass 2  thread (k0, c) holds b  = 0..15  -> k1, times W_N^(16 k1 c)
//   pass 3  thread (k0, j) holds c  = 0..R-1 for k1 = j + R m, m < 16/R  -> k2
// Exchange 1 (b <-> k0) crosses the team; exchange 2 (c <-> k1) stays inside the R lanes that share k0.  At
// R = 4 the team is ONE wave: no workgroup barrier anywhere, sixteen independent waves per CU.  At R = 8 (two
// waves) and R = 16 (four) exchange 1 sits between two LDS-only barriers.  At R = 2 the team is HALF a wave:
// a 64-thread workgroup carries two teams, each on its own segments and its own image (a ds_read_b64 is served
// 32 lanes at a time and a ds_write_b64 16 at a time, so the two teams never meet in a bank).  At R = 1 (N = 256,
// four teams per wave) pass 3 and exchange 2 do not exist and pairs of teams interleave their rows in one image.
//
// LDS image: N float2, NO padding, XOR-swizzled so that every ds_write_b64 (16-lane groups, 32 banks) and
// ds_read_b64 (32-lane groups, 64 banks) of both exchanges is conflict-free:
//   idx(k0, row, c) = 512 (k0 >> P) + 32 row + R ((k0 & KP) ^ (row & KM)) + (c ^ (row & (R-1)))
//   P = 5 - log2 R, KP = 2^P - 1, KM = 16/R - 1 = {R=2: 7, R=4: 3, R=8: 1, R=16: 0};  row = b (exchange 1) or k1 (exchange 2).
// (GF(2) argument: the low address bits are a bijection of the index bits that vary inside a lane group for each
// of the four access shapes - DESIGN.md "segfft LDS image".)  Per access the address is (lane base ^ constant) +
// immediate; the constants take 4 / 8 / 16 values, so a segment spends 16 / 26 / 49 v_xor on addressing.
#include <type_traits>
#include "fft4096.hip.h"

namespace oth {
namespace {

template <int R> struct Geo {
    static constexpr int T = 16 * R, N = 256 * R, Q = 16 / R;
    static constexpr int LR = (R == 1) ? 0 : ((R == 2) ? 1 : ((R == 4) ? 2 : (R == 8 ? 3 : 4)));
    static constexpr int P = 5 - LR;
    static constexpr int KP = (1 << P) - 1;
    static constexpr int KM = 16 / R - 1;
    static constexpr int WAVES = T < 64 ? 1 : T / 64;      // waves per team
    static constexpr int TPB = T < 64 ? 64 / T : 1;        // teams per workgroup (sub-wave teams share a wave)
    static constexpr int BLOCK = T * TPB;
    // 256 B of slack (the images are aligned to 256 B: the swizzle XORs address bits 3..7) + one image per team +
    // per-wave half-segment sums (2 x 4 float2) + chunk tickets
    static constexpr size_t LDS_BYTES = 256 + (size_t)TPB * N * sizeof(float2) + 16 * sizeof(float2);
};

enum { LOAD_HALF = 0, LOAD_FULL = 1 };
#ifndef OTH_SEG_R2_DPP
#define OTH_SEG_R2_DPP 1      // 512 points: last radix-2 stage through DPP instead of an LDS exchange
#endif
#ifndef OTH_CHAIN_WIN_LDS
#define OTH_CHAIN_WIN_LDS 1
#endif
#ifndef OTH_CHAIN_PLAIN_MASK
#define OTH_CHAIN_PLAIN_MASK 7       // chain builds: passes (bit 0: pass 1, bit 1: pass 2, bit 2: pass 3) that take the
#endif                               // multiply-then-add butterflies - their single rows are compared bin by bin
#ifndef OTH_CHAIN_WPS
#define OTH_CHAIN_WPS 3      // waves per SIMD of the chain build (2: no spills at all, but half the speed)
#endif
enum { ACC_SUM = 0, ACC_WSUM = 1, ACC_MAX = 2, ACC_NONE = 3 };

// Image accesses are explicit ds_read_b64 / ds_write_b64 at (swizzled lane base) + immediate.  A wave's LDS
// operations execute in program order, so inside one wave (exchange 2 always, exchange 1 at R = 4) a write
// followed by another lane's read needs no fence; across waves exchange 1 sits between lds_barrier()s.

// the immediate of an asm operand must be a constant expression: template parameter + compile-time loop
// A register PAIR as one asm operand: a `double`, converted to and from float2 by the by-value helpers below.  Rounds 2-4
// wrote `v[i] = __builtin_bit_cast(float2, r[i])` on the array element itself; in the 256-point whole-segment-load and
// chain builds two of those pairs then lived in 8-byte STACK SLOTS - float halves stored, the double loaded back: 32 B of
// scratch that is not a spill, four scratch accesses per segment - which cost the 256-point chain 17 % (53.6 -> 63 % whole
// push once they were gone).  Two forms remove the slots: converting through a by-value helper (Pair<false>, shipped), or
// a 2-float vector type as the asm operand (Pair<true>).  The vector type is SLOWER wherever there was no slot to remove
// (interleaved same-box A/B: w1024 +8 %, w256 +5.5 %, chain4096 +5.7 %, w2048 +5 %, w512 +4 %; and +1.4 / +4.4 / +0.8 % on
// C2 / C3 / C5 when tried in dft16_from_lds), the helper form is not: profiles/r05_ab_pair_type_segfft.txt.
template <bool VEC> struct Pair;
template <> struct Pair<false> {
    typedef double type;
    static __device__ __forceinline__ float2 to_f2(type r) { return __builtin_bit_cast(float2, r); }
    static __device__ __forceinline__ type from_f2(float2 v) { return __builtin_bit_cast(double, v); }
};
template <> struct Pair<true> {
    typedef float type __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float2 to_f2(type r) { return make_float2(r.x, r.y); }
    static __device__ __forceinline__ type from_f2(float2 v) { return type{v.x, v.y}; }
};
template <int IMM, class PT> __device__ __forceinline__ void lds_read_imm(PT &dst, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}
template <int IMM, bool VEC = false> __device__ __forceinline__ void lds_write_imm(unsigned addr, float2 val) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(Pair<VEC>::from_f2(val)), "n"(IMM) : "memory");
}
template <int I, int N, class F> __device__ __forceinline__ void static_for(F f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// v[r16(k)] * W^k for k = 0..15 handed to put(k, value).  W^k = W^(4i) W^j (k = 4i + j) from the six stored
// powers W, W^2, W^3, W^4, W^8, W^12 (correctly rounded table values): at most ONE complex product per twiddle.
// Rebuilding all fifteen from W and W^4 (scatter_pow16 of the 4096 kernels) chains up to three products; with a
// strong off-bin tone in a rectangular-window periodogram that rounding shows in every bin (the tone's bins carry
// 2 sqrt(N) times the noise amplitude), enough to break 1e-4 on single rows.
struct Pow6 {
    float2 j1, j2, j3, i1, i2, i3;      // W^1, W^2, W^3, W^4, W^8, W^12
};
template <class F> __device__ __forceinline__ void twiddle_pow16(const float2 (&v)[16], const Pow6 &w6, F put) {
    float2 wj[4], wi[4];
    wj[1] = w6.j1;
    wj[2] = w6.j2;
    wj[3] = w6.j3;
    wi[1] = w6.i1;
    wi[2] = w6.i2;
    wi[3] = w6.i3;
    // the products below are loop-invariant: without this the compiler hoists all nine out of the segment loop
    // and keeps them in registers (+36 VGPRs per pass)
    // (every product has a wj factor: making those three opaque is enough, and costs six copies instead of twelve)
    asm volatile("" : "+v"(wj[1].x), "+v"(wj[1].y), "+v"(wj[2].x), "+v"(wj[2].y), "+v"(wj[3].x), "+v"(wj[3].y));
    put(std::integral_constant<int, 0>{}, v[0]);
    static_for<1, 16>([&](auto kc) {
        constexpr int k = decltype(kc)::value, i = k >> 2, j = k & 3;
        const float2 w = (i == 0) ? wj[j] : ((j == 0) ? wi[i] : cmul(wi[i], wj[j]));
        put(kc, cmul(v[r16(k)], w));
    });
}

// Sum over the T lanes of a team that is a wave or part of one; every lane of the team gets the total.
template <int T> __device__ __forceinline__ float team_total(float v) {
    if constexpr (T >= 64) {
        return wave_total(v);
    } else {
        v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
        v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
        v = dpp_add<0x141>(v);   // row_half_mirror
        v = dpp_add<0x140>(v);   // row_mirror: the row-of-16 sum in every lane
        if constexpr (T == 32)   // the other row of the pair, through the LDS crossbar (no memory)
            v += __int_as_float(__builtin_amdgcn_ds_bpermute((int)((__lane_id() ^ 16u) << 2), __float_as_int(v)));
        return v;
    }
}

// WPS = waves per SIMD the register allocation is held to (4 -> 128 VGPRs, 3 -> 168)
// NA: rows of the 16 x T sample matrix that hold samples, nperseg = NA T = NA N / 16; the rest of the segment is the
// zero padding of scipy.signal.welch(nperseg < nfft) - the sweeper's call is nperseg = nfft / 4 for whatever fft_len
// the flowgraph passes (spectrum_sweeper.py:263): NA = 4.  Rows a >= NA are compile-time zeros, so loads, window
// products and the first butterfly layer of pass 1 shrink with NA; LOAD_HALF keeps NA / 2 rows (step = nperseg / 2).
template <int R, int LOAD, bool DETREND, bool CHAIN, int WPS, int NA = 16, bool PILOT = false>
__global__ __launch_bounds__(Geo<R>::BLOCK, WPS) void seg_kernel(SegArgs p) {
    static_assert(DETREND || !PILOT, "the pilot belongs to the detrend");
    static_assert(NA == 16 || (!CHAIN && R >= 4 && (NA == 4 || NA == 8)), "zero-padded builds: Welch at 1024 / 2048 points");
    constexpr int NHALF = NA / 2;
    using G = Geo<R>;
    constexpr int T = G::T, N = G::N, Q = G::Q, LR = G::LR, P = G::P, KP = G::KP, KM = G::KM, TPB = G::TPB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned img0 = ((unsigned)(unsigned long long)smem + 255u) & ~255u;     // LDS byte address of the first image
    float2 *red = reinterpret_cast<float2 *>(smem + (img0 - (unsigned)(unsigned long long)smem)) + TPB * N;   // [2][4] half sums
    const int team = TPB > 1 ? (int)threadIdx.x / T : 0;
    int *lnext = reinterpret_cast<int *>(red + 12) + team;                              // chunk ticket
    // R = 1 (N = 256: two passes, no exchange 2): a row holds 16 entries, so teams 2i and 2i+1 interleave their
    // rows in one 32-entry-per-row image - together they fill the 64 banks of a 32-lane read
    const unsigned img = R == 1 ? img0 + (unsigned)(team >> 1) * 4096u + (unsigned)(team & 1) * 128u
                                : img0 + (unsigned)team * (unsigned)(N * sizeof(float2));

    const int t = TPB > 1 ? (int)threadIdx.x % T : (int)threadIdx.x;
    const int hi = t >> LR, lo = t & (R - 1);
    const int wg = blockIdx.x * TPB + team, W = p.wg_per_stream, stream = blockIdx.y;
    // (the fused chain keeps the LDS exchange: measured 71 % of the roofline with it against 54 % through DPP - the
    // exchange's wait is where its three waves per SIMD take turns at the memory pipe)
    constexpr bool R2DPP = R == 2 && !CHAIN && (OTH_SEG_R2_DPP != 0);
    // butterfly form per pass.  256 points (two passes): the multiply-then-add form in pass 2 costs 19 % of the launch
    // (four teams per wave: that build is issue-bound) and measures the same accuracy (tools/acc_rows.py), so only
    // pass 1 takes it there
    constexpr int PLM = kDft16Plain ? 7 : (CHAIN ? ((OTH_CHAIN_PLAIN_MASK) & (R == 1 ? 1 : 7)) : 0);
    constexpr bool PL1 = (PLM & 1) != 0, PL2 = (PLM & 2) != 0, PL3 = (PLM & 4) != 0;
    // asm register pairs as 2-float vectors only where the `double` form left stack slots behind (struct Pair above)
    constexpr bool kPairVec = false;      // (see struct Pair: the by-value converters alone remove the stack slots)
    using PairT = typename Pair<kPairVec>::type;
    // bin held in slot m R + k2 of the per-thread results
    auto bin_of = [&](int m, int k2) { return R2DPP ? hi + 16 * (m * R + k2) + 256 * lo : hi + 16 * (lo + R * m) + 256 * k2; };
    // WIN_LDS (the chain build): the sixteen window values of a thread live in LDS as four float4 and are read
    // per segment - sixteen registers less, which is what keeps the build's spills out of the segment loop (a
    // spill reload waits on vmcnt and with it on the prefetch in flight)
    constexpr bool WIN_LDS = CHAIN && (OTH_CHAIN_WIN_LDS != 0);
    float4 *wl = reinterpret_cast<float4 *>(red + 16) + t;      // [4][T] float4, one table per workgroup
    // the six stored twiddle powers of pass 1 (sub-wave teams) / pass 2 (4096 points) go the same way where the
    // build would otherwise still spill: [3][T] and [3][R] float4 behind the window table
    constexpr bool TW1_LDS = WIN_LDS && R <= 2, TW2_LDS = WIN_LDS && R == 16;
    float4 *tl1 = reinterpret_cast<float4 *>(red + 16) + 4 * T + t;
    float4 *tl2 = reinterpret_cast<float4 *>(red + 16) + 4 * T + (TW1_LDS ? 3 * T : 0) + lo;
    if (WIN_LDS) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            wl[qd * T] = make_float4(p.win[T * (4 * qd) + t], p.win[T * (4 * qd + 1) + t], p.win[T * (4 * qd + 2) + t],
                                     p.win[T * (4 * qd + 3) + t]);
        auto pair = [&](int i, int j) {
            const float2 u = p.tw[i & (N - 1)], w = p.tw[j & (N - 1)];
            return make_float4(u.x, u.y, w.x, w.y);
        };
        if (TW1_LDS) {
            tl1[0] = pair(t, 2 * t);
            tl1[T] = pair(3 * t, 4 * t);
            tl1[2 * T] = pair(8 * t, 12 * t);
        }
        if (TW2_LDS && hi == 0) {
            tl2[0] = pair(16 * lo, 32 * lo);
            tl2[R] = pair(48 * lo, 64 * lo);
            tl2[2 * R] = pair(128 * lo, 192 * lo);
        }
        __syncthreads();      // (teams of one workgroup write identical tables)
    }
    if (TPB > 1 && wg >= W) return;      // the odd team of the last workgroup (no workgroup barrier below when TPB > 1)
    const float2 *xb = p.x + (size_t)stream * p.stream_stride + p.first;

    // lane parts of the four LDS access shapes (byte addresses)
    const unsigned b_rw = img + 8u * (512u * (hi >> P) + R * (hi & KP) + lo);                                    // (k0, c)
    const unsigned b_w1 = img + 8u * (32u * hi + R * (hi & KM) + (lo ^ (hi & (R - 1))));                        // (b, c)
    const unsigned b_r2 = img + 8u * (512u * (hi >> P) + 32u * lo + R * ((hi & KP) ^ (lo & KM)) + lo);          // (k0, j)

    float win[16];
    if (!WIN_LDS) {
#pragma unroll
        for (int a = 0; a < NA; ++a) win[a] = p.win[T * a + t];
    }
    // W_N^(k t) for pass 1 and W_N^(16 k c) for pass 2, k = 1, 2, 3, 4, 8, 12 (table index mod N)
    const Pow6 tw1 = {p.tw[t], p.tw[(2 * t) & (N - 1)], p.tw[(3 * t) & (N - 1)], p.tw[(4 * t) & (N - 1)],
                      p.tw[(8 * t) & (N - 1)], p.tw[(12 * t) & (N - 1)]};
    const Pow6 tw2 = {p.tw[16 * lo], p.tw[(32 * lo) & (N - 1)], p.tw[(48 * lo) & (N - 1)], p.tw[(64 * lo) & (N - 1)],
                      p.tw[(128 * lo) & (N - 1)], p.tw[(192 * lo) & (N - 1)]};
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;

    constexpr int NH = (LOAD == LOAD_HALF) ? NHALF : NA;
    float2 kw[LOAD == LOAD_HALF ? NHALF : 1], nxt[NH];
    float2 prev_tot = make_float2(0.f, 0.f);
    // PILOT (every detrending plan but OTH_DETREND_CONSTANT_FAST): SegArgs.pilot comes off every sample as it arrives
    const float2 pv = load_pilot(PILOT ? p.pilot : nullptr, stream);

    const int sched = p.sched;
    const long long nchunks = sched ? chunk_count_of(p.nseg, p.nbig, p.chunk, p.tail_chunk) : 1;
    const long long s0 = (p.nseg * wg) / W, s1 = (p.nseg * (wg + 1)) / W;
    unsigned ticket = 0;
    bool primed = false;
    for (long long cur = sched ? wg : 0; cur < nchunks;) {
        long long sb = s0, se = s1;
        if (sched) chunk_range_of(p.nseg, p.nbig, p.chunk, p.tail_chunk, cur, sb, se);
        // interleaved static schedule: the chunk after this one is known, so its first segment is prefetched with this
        // chunk's last one (whole-segment loads only) and the prologue below runs once per team
        long long sb_next = -1;
        if (LOAD == LOAD_FULL && sched == 1 && cur + W < nchunks) {
            long long se_next;
            chunk_range_of(p.nseg, p.nbig, p.chunk, p.tail_chunk, cur + W, sb_next, se_next);
        }
        if (sb < se && !primed) {      // chunk prologue
            const float2 *xs = xb + sb * p.step + t;
            if (LOAD == LOAD_HALF) {
#pragma unroll
                for (int a = 0; a < NHALF; ++a) kw[a] = xs[T * a];
#pragma unroll
                for (int a = 0; a < NHALF; ++a) nxt[a] = xs[NHALF * T + T * a];
            } else {
#pragma unroll
                for (int a = 0; a < NA; ++a) nxt[a] = load_once(xs + T * a);
            }
        }
        primed = sb_next >= 0;
        for (long long s = sb; s < se; ++s) {
            float2 v[16];
            // R <= 2 (and the 4096-point chain build, which is short of registers): the swizzle constants take 8 / 16 values per access shape; hoisted out of this loop the forty
            // (base ^ constant) addresses would live in registers - keep the bases opaque and pay the v_xor instead
            unsigned a_w1 = b_w1, a_rw = b_rw, a_r2 = b_r2;
            if (R <= 2 || (CHAIN && R == 16)) asm volatile("" : "+v"(a_w1), "+v"(a_rw), "+v"(a_r2));
            prio_latency();
            // ---- samples, window, raw sums ------------------------------------------------------------
            float2 sum = make_float2(0.f, 0.f), sumf = make_float2(0.f, 0.f);
            if (WIN_LDS) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float4 w4 = wl[qd * T];
                    win[4 * qd] = w4.x, win[4 * qd + 1] = w4.y, win[4 * qd + 2] = w4.z, win[4 * qd + 3] = w4.w;
                }
            }
            if (LOAD == LOAD_HALF) {
                if (s == sb) {
#pragma unroll
                    for (int a = 0; a < NHALF; ++a) {      // kw holds the raw first half of the chunk's first segment
                        if (PILOT) kw[a] = csub(kw[a], pv);
                        sumf = cadd(sumf, kw[a]);
                        kw[a] = make_float2(kw[a].x * win[a], kw[a].y * win[a]);
                    }
                }
#pragma unroll
                for (int a = 0; a < NHALF; ++a) {
                    const float2 r = PILOT ? csub(nxt[a], pv) : nxt[a];
                    v[a] = kw[a];
                    v[NHALF + a] = make_float2(r.x * win[NHALF + a], r.y * win[NHALF + a]);
                    kw[a] = make_float2(r.x * win[a], r.y * win[a]);
                    sum = cadd(sum, r);
                }
                {   // unconditional: a prefetch under `if (s + 1 < se)` makes nxt a phi and costs 32 register copies per
                    // segment; the chunk's last segment re-reads the half it has just consumed (valid, L2-resident)
                    const float2 *xn = xb + (s + (s + 1 < se ? 2 : 1)) * (long long)(NHALF * T) + t;
#pragma unroll
                    for (int a = 0; a < NHALF; ++a) nxt[a] = load_once(xn + T * a);
                }
            } else {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const float2 r = PILOT ? csub(nxt[a], pv) : nxt[a];
                    v[a] = make_float2(r.x * win[a], r.y * win[a]);
                    sum = cadd(sum, r);
                }
                {   // unconditional (see above): the chunk's last segment fetches the next chunk's first one, or itself
                    const float2 *xn = xb + (s + 1 < se ? s + 1 : (sb_next >= 0 ? sb_next : s)) * p.step + t;
#pragma unroll
                    for (int a = 0; a < NA; ++a) nxt[a] = load_once(xn + T * a);
                }
            }
#pragma unroll
            for (int a = NA; a < 16; ++a) v[a] = make_float2(0.f, 0.f);      // the zero padding
            float2 tot = make_float2(0.f, 0.f);
            if (DETREND) {
                sum.x = team_total<T>(sum.x);
                sum.y = team_total<T>(sum.y);
                if (LOAD == LOAD_HALF && s == sb) {
                    sumf.x = team_total<T>(sumf.x);
                    sumf.y = team_total<T>(sumf.y);
                }
                if (G::WAVES > 1 && (t & 63) == 0) {
                    red[t >> 6] = sum;
                    if (LOAD == LOAD_HALF && s == sb) red[4 + (t >> 6)] = sumf;
                }
            }
            if (G::WAVES > 1) lds_barrier();     // A: the previous segment's exchange-2 reads are done everywhere; red[] visible
            if (sched == 2 && t == 0) {
                if (s == sb) ticket = atomicAdd(p.queue + stream, 1u);
                if (s == se - 1) *lnext = (int)ticket;
            }
            if (DETREND) {       // totals are read here: barrier B separates them from the next segment's red[] writes;
                                 // time-domain detrend on the windowed samples: (x - m) w = x w - m w
                if (G::WAVES > 1) {
                    float2 nt = red[0];
#pragma unroll
                    for (int w = 1; w < G::WAVES; ++w) nt = cadd(nt, red[w]);
                    if (LOAD == LOAD_HALF && s == sb) {
                        float2 ft = red[4];
#pragma unroll
                        for (int w = 1; w < G::WAVES; ++w) ft = cadd(ft, red[4 + w]);
                        prev_tot = ft;
                    }
                    sum = nt;
                } else if (LOAD == LOAD_HALF && s == sb) {
                    prev_tot = sumf;
                }
                if (LOAD == LOAD_HALF) {
                    tot = cadd(prev_tot, sum);
                    prev_tot = sum;
                } else {
                    tot = sum;
                }
                const float2 nm = make_float2(tot.x * (-1.0f / (NA * T)), tot.y * (-1.0f / (NA * T)));      // mean over nperseg
#pragma unroll
                for (int a = 0; a < NA; ++a) v[a] = make_float2(fmaf(nm.x, win[a], v[a].x), fmaf(nm.y, win[a], v[a].y));
            }
            // ---- pass 1 -----------------------------------------------------------------------------------
            auto pow6_from = [](const float4 *tb, int stride) {
                const float4 q0 = tb[0], q1 = tb[stride], q2 = tb[2 * stride];
                return Pow6{make_float2(q0.x, q0.y), make_float2(q0.z, q0.w), make_float2(q1.x, q1.y),
                            make_float2(q1.z, q1.w), make_float2(q2.x, q2.y), make_float2(q2.z, q2.w)};
            };
            Pow6 w1 = tw1;
            if (TW1_LDS) w1 = pow6_from(tl1, T);      // issued before the butterflies that hide them
            prio_compute();
            dft16<PL1>(v);
            prio_latency();
            twiddle_pow16(v, w1, [&](auto kc, float2 val) {
                constexpr int k0 = decltype(kc)::value;
                lds_write_imm<8 * (512 * (k0 >> P) + R * (k0 & KP & ~KM)), kPairVec>(a_w1 ^ (8u * R * (k0 & KM)), val);
            });
            if (G::WAVES > 1) lds_barrier();     // B
            // ---- pass 2: thread (k0, c) gathers b -----------------------------------------------------------
            Pow6 w2 = tw2;
            if (TW2_LDS) w2 = pow6_from(tl2, R);      // older than the sixteen reads below: their counted waits still hold
            {
                PairT r[16];
                // rows b with equal (b & KM, b & (R-1)) share one swizzled base
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, b = (i >> 2) + 4 * (i & 3);      // issue order 0,4,8,12, 1,5,9,13, ...
                    lds_read_imm<256 * b>(r[b], a_rw ^ (8u * (R * (b & KM) + (b & (R - 1)))));
                });
                prio_compute();
                float dep = 0.f;
#define SEG_WAIT(n, a, d) \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r[a]), "+v"(r[a + 4]), "+v"(r[a + 8]), "+v"(r[a + 12]), "+v"(d))
#pragma unroll
                for (int a0 = 0; a0 < 4; ++a0) {
                    if (a0 == 0) SEG_WAIT(12, 0, dep);
                    else if (a0 == 1) SEG_WAIT(8, 1, v[0].x);
                    else if (a0 == 2) SEG_WAIT(4, 2, v[1].x);
                    else SEG_WAIT(0, 3, v[2].x);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[a0 + 4 * j] = Pair<kPairVec>::to_f2(r[a0 + 4 * j]);
                    dft4<false>(v[a0], v[a0 + 4], v[a0 + 8], v[a0 + 12]);
                }
                dft16_layer2<PL2>(v);
            }
            if constexpr (R2DPP) {
                // R = 2 without exchange 2: the two lanes that share k0 are neighbours, so the last radix-2 stage
                // takes the partner's value through DPP inside the multiply-add: lane c keeps y_c[k1] + s y_(1-c)[k1],
                // s = +1 / -1 for c = 0 / 1, i.e. X[k0 + 16 k1] and -X[k0 + 16 k1 + 256] (only |X|^2 is used).  No LDS
                // traffic, no extra instruction; lane c ends up with k2 = c for all sixteen k1.
                prio_compute();
                float2 y[16];
                twiddle_pow16(v, w2, [&](auto kc, float2 val) { y[decltype(kc)::value] = val; });
                const float sgn = lo ? -1.0f : 1.0f;
                asm volatile("s_nop 1");      // VALU write -> DPP read of the same register needs two wait states
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) {
                    asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(y[k1].x) : "v"(sgn));
                    asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(y[k1].y) : "v"(sgn));
                }
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) v[k1] = y[k1];
            } else if constexpr (R > 1) {
                prio_latency();
                // in place: the R lanes that share k0 sit in one wave and have issued their reads of region k0 above
                twiddle_pow16(v, w2, [&](auto kc, float2 val) {
                    constexpr int k1 = decltype(kc)::value;
                    lds_write_imm<256 * k1, kPairVec>(a_rw ^ (8u * (R * (k1 & KM) + (k1 & (R - 1)))), val);
                });
                // ---- pass 3: thread (k0, j) gathers c for k1 = j + R m ---------------------------------------------
                {
                    PairT r[16];
                    static_for<0, 16>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, c = i / Q, m = i % Q;
                        // row k1 = lo + R m: its (k1 & KM) part beyond lo is (R m) & KM - zero unless R = 2
                        lds_read_imm<256 * R * m>(r[m * R + c], a_r2 ^ (8u * (c + R * ((R * m) & KM))));
                    });
                    prio_compute();
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]),
                                   "+v"(r[15]));
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = Pair<kPairVec>::to_f2(r[i]);
                }
            }
            // X[k0 + 16 (lo + R m) + 256 k2] lands in v[m R + k2] (R = 16: v[r16(k2)]; R = 1: X[k0 + 16 m] in v[r16(m)])
            if (R == 1 || R2DPP) {
            } else if (R == 2) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const float2 e = v[2 * m], o = v[2 * m + 1];
                    v[2 * m] = cadd(e, o);
                    v[2 * m + 1] = csub(e, o);
                }
            } else if (R == 4) {
#pragma unroll
                for (int m = 0; m < 4; ++m) dft4<false>(v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]);
            } else if (R == 8) {
                float2 h0[8], h1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    h0[i] = v[i];
                    h1[i] = v[8 + i];
                }
                dft8<PL3>(h0);
                dft8<PL3>(h1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = h0[i];
                    v[8 + i] = h1[i];
                }
            } else {
                dft16<PL3>(v);
            }
            auto at = [&](int m, int k2) -> float2 & { return v[R == 16 ? r16(k2) : (R == 1 ? r16(m) : m * R + k2)]; };
            if (!CHAIN) {
#pragma unroll
                for (int m = 0; m < Q; ++m)
#pragma unroll
                    for (int k2 = 0; k2 < R; ++k2) {
                        const float2 X = at(m, k2);
                        acc[m * R + k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[m * R + k2]));
                    }
            } else {
                // every choice below is uniform over the launch or the segment: one scalar branch per segment each, no
                // per-bin selects
                const bool st = s >= p.store_from, ac = s < p.acc_end;
                float val[16];
                if (p.epilogue == 0) {
#pragma unroll
                    for (int m = 0; m < Q; ++m)
#pragma unroll
                        for (int k2 = 0; k2 < R; ++k2) {
                            const float2 X = at(m, k2);
                            // v_sqrt_f32 (1 ulp) without sqrtf's denormal rescaling: |X|^2 of a transform is either 0 or far
                            // above 2^-96
                            val[m * R + k2] = __builtin_amdgcn_sqrtf(fmaf(X.x, X.x, X.y * X.y));
                        }
                } else {
                    const float sc = p.scale;
#pragma unroll
                    for (int m = 0; m < Q; ++m)
#pragma unroll
                        for (int k2 = 0; k2 < R; ++k2) {
                            const float2 X = at(m, k2);
                            val[m * R + k2] = fmaf(X.x, X.x, X.y * X.y) * sc;
                        }
                }
                if (st) {
                    float *row = p.rows + ((size_t)stream * (p.nseg - p.store_from) + (size_t)(s - p.store_from)) * N;
                    const int sh = p.fftshift ? N / 2 : 0;
#pragma unroll
                    for (int m = 0; m < Q; ++m)
#pragma unroll
                        for (int k2 = 0; k2 < R; ++k2) row[(bin_of(m, k2) + sh) & (N - 1)] = val[m * R + k2];
                }
                if (ac) {
                    if (p.acc_mode == ACC_WSUM) {
                        const long long kk = p.acc_end - 1 - s;
                        const float w = kk == 0 ? 1.0f : exp2f(p.l2 * (float)kk);
#pragma unroll
                        for (int k = 0; k < 16; ++k) acc[k] = fmaf(w, val[k], acc[k]);
                    } else if (p.acc_mode == ACC_MAX) {
#pragma unroll
                        for (int k = 0; k < 16; ++k)      // plain v_max_f32: no NaN canonicalisation (finite by construction)
                            asm("v_max_f32 %0, %1, %2" : "=v"(acc[k]) : "v"(acc[k]), "v"(val[k]));
                    }
                }
            }
        }
        if (sched == 0) break;
        if (sched == 1) {
            cur += W;
        } else {
            if (G::WAVES == 1) wave_lds_sync();   // (WAVES > 1: *lnext was written before barrier B of the last segment)
            cur = (long long)W + *lnext;
        }
    }

    if (p.partial) {      // natural bin order: k = k0 + 16 k1 + 256 k2
        float *dst = p.partial + ((size_t)stream * W + wg) * N;
#pragma unroll
        for (int m = 0; m < Q; ++m)
#pragma unroll
            for (int k2 = 0; k2 < R; ++k2) dst[bin_of(m, k2)] = acc[m * R + k2];
    }
}

// ---------------------------------------------------------------------------------------------------------
// segws: the Welch average at 50 % overlap with the team split by role, as welch4096ws.hip does for 4096 points:
// a PRODUCER team of 16 R threads (loads, window, pass 1, exchange-1 writes) and a CONSUMER team of 16 R threads
// (pass 2, exchange 2, pass 3, |X|^2 accumulation) one segment apart on two LDS images, one LDS-only barrier per
// segment.  At R = 4 that is one producer wave and one consumer wave per 128-thread workgroup.  Why: the one-role
// kernel above needs ~155 VGPRs (window, kept half, prefetch, data, accumulators, two sets of twiddle powers), i.e.
// three waves per SIMD; split by role both halves stay under 128 (four waves per SIMD), the consumer keeps all
// fifteen pass-2 twiddles in registers, and the producer's loads overlap the consumer's butterflies by construction.
//
// DET: 0 none; 1 time domain in the producer (R = 4: the producer is one wave, the sum needs no LDS);
//      2 frequency domain in the consumer, X -= mean FFT(w) on the bins |k| < 256 (SegArgs.fd; needs a window whose
//        spectrum is confined to those bins, as welch4096ws.hip).
enum { WS_STOP = 0, WS_DATA = 1 };
#ifndef OTH_SEGWS_SPREAD
#define OTH_SEGWS_SPREAD 0      // A/B (round 4): two loads at four places of the producer step: +3.5 % time at 1024, +0.8 % at 2048 - not the scanner kernel (whose 16 waves x 16 loads per step fill the queue)
#endif
#ifndef OTH_SEGWS_DEEP
#define OTH_SEGWS_DEEP 0
#endif
#ifndef OTH_SEGWS_STORED_TW1
#define OTH_SEGWS_STORED_TW1 (!OTH_SEGWS_DEEP)
#endif

template <int R, int DET, bool PILOT = false>
__global__ __launch_bounds__(32 * R, 4) void segws_kernel(SegArgs p) {
    static_assert(DET != 0 || !PILOT, "the pilot belongs to the detrend");
    using G = Geo<R>;
    constexpr int T = G::T, N = G::N, Q = G::Q, LR = G::LR, P = G::P, KP = G::KP, KM = G::KM, WV = G::WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned img0 = ((unsigned)(unsigned long long)smem + 255u) & ~255u;      // two images, 8 N bytes each
    unsigned char *tail = smem + (img0 - (unsigned)(unsigned long long)smem) + 2 * N * sizeof(float2);
    float2 *red = reinterpret_cast<float2 *>(tail);             // [2][4] per-wave sums of the segment in each image
    int *ctrl = reinterpret_cast<int *>(red + 8);               // item kind per image [0..1], next-chunk ticket [2]

    const int tid = threadIdx.x;
    const bool producer = tid < T;
    const int t = producer ? tid : tid - T;
    const int hi = t >> LR, lo = t & (R - 1);
    const int wg = blockIdx.x, W = p.wg_per_stream, stream = blockIdx.y;
    const unsigned b_rw = 8u * (512u * (hi >> P) + R * (hi & KP) + lo);
    const unsigned b_w1 = 8u * (32u * hi + R * (hi & KM) + (lo ^ (hi & (R - 1))));
    const unsigned b_r2 = 8u * (512u * (hi >> P) + 32u * lo + R * ((hi & KP) ^ (lo & KM)) + lo);

    if (producer) {
        const float2 *xb = p.x + (size_t)stream * p.stream_stride + p.first;
        float win[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) win[a] = p.win[T * a + t];
#if OTH_SEGWS_STORED_TW1
        float2 tw1[16];      // all fifteen pass-1 twiddles W_N^(k0 t) in registers: the producer has the room (no accumulators)
#pragma unroll
        for (int k = 1; k < 16; ++k) tw1[k] = p.tw[(k * t) & (N - 1)];
#else
        const Pow6 tw1 = {p.tw[t], p.tw[(2 * t) & (N - 1)], p.tw[(3 * t) & (N - 1)], p.tw[(4 * t) & (N - 1)],
                          p.tw[(8 * t) & (N - 1)], p.tw[(12 * t) & (N - 1)]};
#endif
        // The new half of segment s is half-block s + 1, requested one segment before it is consumed.  OTH_SEGWS_DEEP:
        // two buffers alternate and a half-block is requested TWO segments ahead - measured no faster (42-43 % against
        // 43-45 %: it costs the registers of the stored pass-1 twiddles), although a build without the loads runs 19 %
        // faster; the loads cost through the memory system, not through their latency.
        float2 kw[8], nA[8];
#if OTH_SEGWS_DEEP
        float2 nB[8];
#endif
        float2 prev_new = make_float2(0.f, 0.f);
        const float2 pv = load_pilot(PILOT ? p.pilot : nullptr, stream);      // PILOT: off every sample as it arrives
        const int sched = p.sched;
        const long long nchunks = sched ? chunk_count_of(p.nseg, p.nbig, p.chunk, p.tail_chunk) : 1;
        const long long s0 = (p.nseg * wg) / W, s1 = (p.nseg * (wg + 1)) / W;
        unsigned ticket = 0;
        int it = 0;
        long long sb = 0, se = 0;
        auto half = [&](long long h) { return xb + (h < se ? h : se) * (long long)(N / 2) + t; };      // clamped: valid memory
        auto segment = [&](long long s, float2(&nxt)[8]) {
            const int q = it & 1;
            const unsigned img = img0 + q * (unsigned)(N * sizeof(float2));
            float2 v[16];
            prio_latency();
            float2 sum = make_float2(0.f, 0.f), sumf = make_float2(0.f, 0.f);
            if (s == sb) {
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    if (PILOT) kw[a] = csub(kw[a], pv);
                    sumf = cadd(sumf, kw[a]);
                    kw[a] = make_float2(kw[a].x * win[a], kw[a].y * win[a]);
                }
            }
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const float2 r = PILOT ? csub(nxt[a], pv) : nxt[a];
                v[a] = kw[a];
                v[8 + a] = make_float2(r.x * win[8 + a], r.y * win[8 + a]);
                kw[a] = make_float2(r.x * win[a], r.y * win[a]);
                sum = cadd(sum, r);
            }
#ifndef OTH_SEGWS_NOLOAD      // (timing experiment: -DOTH_SEGWS_NOLOAD keeps re-using the first loaded halves)
            const float2 *xn = half(s + (OTH_SEGWS_DEEP ? 3 : 2));
            // the next half's eight loads, two at each of four places of the step instead of one burst (round 4: the
            // scanner kernel's lesson - a burst fills the memory pipeline's queue and the issuing wave stands still)
            auto spread = [&](int grp) {
#if OTH_SEGWS_SPREAD
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 2 * grp; a < 2 * grp + 2; ++a) nxt[a] = load_once(xn + T * a);
                __builtin_amdgcn_sched_barrier(0);
#else
                if (grp == 0) {
#pragma unroll
                    for (int a = 0; a < 8; ++a) nxt[a] = load_once(xn + T * a);
                }
#endif
            };
            spread(0);
#else
            auto spread = [&](int) {};
#pragma unroll
            for (int a = 0; a < 8; ++a) asm volatile("" : "+v"(nxt[a].x), "+v"(nxt[a].y));
#endif
            if (sched == 2 && t == 0) {
                if (s == sb) ticket = atomicAdd(p.queue + stream, 1u);
                if (s == se - 1) ctrl[2] = (int)ticket;
            }
            if (DET == 1) {          // one producer wave: the segment total without LDS
                sum.x = wave_total(sum.x);
                sum.y = wave_total(sum.y);
                if (s == sb) prev_new = make_float2(wave_total(sumf.x), wave_total(sumf.y));
                const float2 nm = make_float2((prev_new.x + sum.x) * (-1.0f / N), (prev_new.y + sum.y) * (-1.0f / N));
                prev_new = sum;
#pragma unroll
                for (int a = 0; a < 16; ++a) v[a] = make_float2(fmaf(nm.x, win[a], v[a].x), fmaf(nm.y, win[a], v[a].y));
            } else if (DET == 2) {   // per-wave sums of both halves, side by side for the consumer
                sum.x = wave_total_lane63(sum.x);
                sum.y = wave_total_lane63(sum.y);
                float2 other = prev_new;
                if (s == sb) other = make_float2(wave_total_lane63(sumf.x), wave_total_lane63(sumf.y));
                if ((t & 63) == 63) red[q * 4 + (t >> 6)] = cadd(sum, other);
                prev_new = sum;
            }
            spread(1);
            prio_compute();
            dft16(v);
            prio_latency();
            spread(2);
#if OTH_SEGWS_STORED_TW1
            static_for<0, 16>([&](auto kc) {
                constexpr int k0 = decltype(kc)::value;
                const float2 val = k0 ? cmul(v[r16(k0)], tw1[k0]) : v[0];
                lds_write_imm<8 * (512 * (k0 >> P) + R * (k0 & KP & ~KM))>((img + b_w1) ^ (8u * R * (k0 & KM)), val);
            });
#else
            twiddle_pow16(v, tw1, [&](auto kc, float2 val) {
                constexpr int k0 = decltype(kc)::value;
                lds_write_imm<8 * (512 * (k0 >> P) + R * (k0 & KP & ~KM))>((img + b_w1) ^ (8u * R * (k0 & KM)), val);
            });
#endif
            spread(3);
            if (t == 0) ctrl[q] = WS_DATA;
            lds_barrier();
            ++it;
        };
        for (long long cur = sched ? wg : 0; cur < nchunks;) {
            sb = s0;
            se = s1;
            if (sched) chunk_range_of(p.nseg, p.nbig, p.chunk, p.tail_chunk, cur, sb, se);
            if (sb < se) {
                const float2 *x0 = half(sb), *x1 = half(sb + 1), *x2 = half(sb + 2);
#pragma unroll
                for (int a = 0; a < 8; ++a) kw[a] = x0[T * a];
#pragma unroll
                for (int a = 0; a < 8; ++a) nA[a] = load_once(x1 + T * a);
#if OTH_SEGWS_DEEP
#pragma unroll
                for (int a = 0; a < 8; ++a) nB[a] = load_once(x2 + T * a);
#else
                (void)x2;
#endif
            }
#if OTH_SEGWS_DEEP
            for (long long s = sb; s < se; s += 2) {
                segment(s, nA);
                if (s + 1 < se) segment(s + 1, nB);
            }
#else
            for (long long s = sb; s < se; ++s) segment(s, nA);
#endif
            if (sched == 0) break;
            // ctrl[2] was written before the barrier of the chunk's last segment
            cur = (sched == 1) ? cur + W : (long long)W + __builtin_amdgcn_readfirstlane(ctrl[2]);
        }
        if (t == 0) ctrl[it & 1] = WS_STOP;
        lds_barrier();
    } else {
        float2 tw2[16];
#pragma unroll
        for (int k = 1; k < 16; ++k) tw2[k] = p.tw[(16 * lo * k) & (N - 1)];      // W_N^(16 k1 c)
        float4 fw[Q];
#pragma unroll
        for (int m = 0; m < Q; ++m) fw[m] = DET == 2 ? p.fd[Q * t + m] : make_float4(0.f, 0.f, 0.f, 0.f);
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        float2 v[16];
        int it = 0;
        for (;;) {
            prio_latency();
            lds_barrier();
            const int q = it & 1;
            ++it;
            const int kind = __builtin_amdgcn_readfirstlane(ctrl[q]);
            if (kind == WS_STOP) break;
            const unsigned img = img0 + q * (unsigned)(N * sizeof(float2));
            float2 mean = make_float2(0.f, 0.f);
            if (DET == 2) {
                float2 tot = red[q * 4];
#pragma unroll
                for (int w = 1; w < WV; ++w) tot = cadd(tot, red[q * 4 + w]);
                mean = make_float2(tot.x * (1.0f / N), tot.y * (1.0f / N));
            }
            {
                double r[16];
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, b = (i >> 2) + 4 * (i & 3);
                    lds_read_imm<256 * b>(r[b], (img + b_rw) ^ (8u * (R * (b & KM) + (b & (R - 1)))));
                });
                prio_compute();
                float dep = 0.f;
#pragma unroll
                for (int a0 = 0; a0 < 4; ++a0) {
                    if (a0 == 0) SEG_WAIT(12, 0, dep);
                    else if (a0 == 1) SEG_WAIT(8, 1, v[0].x);
                    else if (a0 == 2) SEG_WAIT(4, 2, v[1].x);
                    else SEG_WAIT(0, 3, v[2].x);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[a0 + 4 * j] = Pair<false>::to_f2(r[a0 + 4 * j]);
                    dft4<false>(v[a0], v[a0 + 4], v[a0 + 8], v[a0 + 12]);
                }
                dft16_layer2(v);
            }
            prio_latency();
            lds_write_imm<0>((img + b_rw), v[0]);
            static_for<1, 16>([&](auto kc) {
                constexpr int k1 = decltype(kc)::value;
                lds_write_imm<256 * k1>((img + b_rw) ^ (8u * (R * (k1 & KM) + (k1 & (R - 1)))), cmul(v[r16(k1)], tw2[k1]));
            });
            {
                double r[16];
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, c = i / Q, m = i % Q;
                    lds_read_imm<256 * R * m>(r[m * R + c], (img + b_r2) ^ (8u * c));
                });
                prio_compute();
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                               "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]),
                               "+v"(r[15]));
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = Pair<false>::to_f2(r[i]);
            }
            if (R == 4) {
#pragma unroll
                for (int m = 0; m < 4; ++m) dft4<false>(v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]);
            } else if (R == 8) {
                float2 h0[8], h1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    h0[i] = v[i];
                    h1[i] = v[8 + i];
                }
                dft8(h0);
                dft8(h1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = h0[i];
                    v[8 + i] = h1[i];
                }
            } else {
                dft16(v);
            }
            auto at = [&](int m, int k2) -> float2 & { return v[R == 16 ? r16(k2) : m * R + k2]; };
            if (DET == 2) {
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    float2 &lo_ = at(m, 0), &hi_ = at(m, R - 1);
                    lo_ = make_float2(lo_.x - (mean.x * fw[m].x - mean.y * fw[m].y), lo_.y - (mean.x * fw[m].y + mean.y * fw[m].x));
                    hi_ = make_float2(hi_.x - (mean.x * fw[m].z - mean.y * fw[m].w), hi_.y - (mean.x * fw[m].w + mean.y * fw[m].z));
                }
            }
#pragma unroll
            for (int m = 0; m < Q; ++m)
#pragma unroll
                for (int k2 = 0; k2 < R; ++k2) {
                    const float2 X = at(m, k2);
                    acc[m * R + k2] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[m * R + k2]));
                }
        }
        float *dst = p.partial + ((size_t)stream * W + wg) * N;
#pragma unroll
        for (int m = 0; m < Q; ++m)
#pragma unroll
            for (int k2 = 0; k2 < R; ++k2) dst[hi + 16 * (lo + R * m) + 256 * k2] = acc[m * R + k2];
    }
}

template <int R> constexpr size_t segws_lds_bytes() { return 256 + 2 * (size_t)Geo<R>::N * sizeof(float2) + 8 * sizeof(float2) + 16; }

template <int R, int DET> hipError_t launch_ws_one(const SegArgs &a, hipStream_t s) {
    const dim3 grid(a.wg_per_stream, a.nstreams);
    if constexpr (DET != 0) {
        if (a.pilot) {
            hipLaunchKernelGGL((segws_kernel<R, DET, true>), grid, dim3(2 * Geo<R>::T), segws_lds_bytes<R>(), s, a);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((segws_kernel<R, DET>), grid, dim3(2 * Geo<R>::T), segws_lds_bytes<R>(), s, a);
    return hipGetLastError();
}
template <int R> int occupancy_ws() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, segws_kernel<R, 2>, 2 * Geo<R>::T, segws_lds_bytes<R>()) != hipSuccess || n < 1)
        n = 1;
    return n;
}

// image(s) + sums + tickets, and for the chain build the window table (4 x T float4) and twiddle tables
template <int R, bool CHAIN> constexpr size_t seg_lds_bytes() {
    return Geo<R>::LDS_BYTES +
           ((CHAIN && OTH_CHAIN_WIN_LDS) ? (4 * Geo<R>::T + (R <= 2 ? 3 * Geo<R>::T : 0) + (R == 16 ? 3 * R : 0)) * sizeof(float4) : 0);
}

template <int R, int LOAD, bool DETREND, bool CHAIN, int WPS, int NA = 16> hipError_t launch_one(const SegArgs &a, hipStream_t s) {
    const dim3 grid((a.wg_per_stream + Geo<R>::TPB - 1) / Geo<R>::TPB, a.nstreams);
    constexpr size_t lds = seg_lds_bytes<R, CHAIN>();
    if constexpr (DETREND) {
        if (a.pilot) {
            hipLaunchKernelGGL((seg_kernel<R, LOAD, DETREND, CHAIN, WPS, NA, true>), grid, dim3(Geo<R>::BLOCK), lds, s, a);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((seg_kernel<R, LOAD, DETREND, CHAIN, WPS, NA>), grid, dim3(Geo<R>::BLOCK), lds, s, a);
    return hipGetLastError();
}

template <int R, int LOAD, bool DETREND, bool CHAIN, int WPS, int NA = 16> int occupancy_one() {
    int n = 0;
    constexpr size_t lds = seg_lds_bytes<R, CHAIN>();
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, seg_kernel<R, LOAD, DETREND, CHAIN, WPS, NA>, Geo<R>::BLOCK, lds) != hipSuccess ||
        n < 1)
        n = 1;
    return n * Geo<R>::TPB;
}

// zero-padded Welch builds (nperseg = nfft / 4 or nfft / 2): kind 0 step = nperseg / 2, kind 1 any step
template <int R, int NA> hipError_t launch_pad(const SegArgs &a, int kind, hipStream_t s) {
    constexpr int WPS = NA == 4 ? 4 : 3;      // (nperseg = nfft / 2 spills a few registers at 128)
    if (kind == 0) return a.detrend ? launch_one<R, LOAD_HALF, true, false, WPS, NA>(a, s) : launch_one<R, LOAD_HALF, false, false, WPS, NA>(a, s);
    return a.detrend ? launch_one<R, LOAD_FULL, true, false, WPS, NA>(a, s) : launch_one<R, LOAD_FULL, false, false, WPS, NA>(a, s);
}
template <int R, int LOAD, bool DETREND, bool CHAIN, int WPS, int NA> int occupancy_one_pilot() {
    int n = 0;
    constexpr size_t lds = seg_lds_bytes<R, CHAIN>();
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, seg_kernel<R, LOAD, DETREND, CHAIN, WPS, NA, true>, Geo<R>::BLOCK, lds) !=
            hipSuccess || n < 1)
        n = 1;
    return n * Geo<R>::TPB;
}
// The grid is sized once per plan shape, the launch then takes the build with or without the pilot: the smaller of the
// two occupancies counts (at three waves per SIMD the NA = 8 builds sit at 126-130 VGPRs, i.e. on both sides of the
// step between eight and six resident workgroups per CU).
template <int R, int NA> int occupancy_pad(int kind) {
    constexpr int WPS = NA == 4 ? 4 : 3;
    const int plain = kind == 0 ? occupancy_one<R, LOAD_HALF, true, false, WPS, NA>() : occupancy_one<R, LOAD_FULL, true, false, WPS, NA>();
    const int pilot = kind == 0 ? occupancy_one_pilot<R, LOAD_HALF, true, false, WPS, NA>()
                                : occupancy_one_pilot<R, LOAD_FULL, true, false, WPS, NA>();
    return plain < pilot ? plain : pilot;
}

// kind: 0 Welch step = N/2 (half kept in registers), 1 Welch any step, 2 chain.  wps4: the 128-VGPR build of kind 0.
template <int R> hipError_t launch_r(const SegArgs &a, int kind, bool wps4, hipStream_t s) {
    if (kind == 2) return launch_one<R, LOAD_FULL, false, true, OTH_CHAIN_WPS>(a, s);
    if constexpr (R == 16) {
        return hipErrorInvalidValue;      // the Welch average at 4096 has its own kernels (welch4096*.hip)
    } else {
        if (kind == 0) {
            if (wps4) return a.detrend ? launch_one<R, LOAD_HALF, true, false, 4>(a, s) : launch_one<R, LOAD_HALF, false, false, 4>(a, s);
            return a.detrend ? launch_one<R, LOAD_HALF, true, false, 3>(a, s) : launch_one<R, LOAD_HALF, false, false, 3>(a, s);
        }
        return a.detrend ? launch_one<R, LOAD_FULL, true, false, 3>(a, s) : launch_one<R, LOAD_FULL, false, false, 3>(a, s);
    }
}

template <int R> int occupancy_r(int kind, bool wps4) {
    if (kind == 2) return occupancy_one<R, LOAD_FULL, false, true, OTH_CHAIN_WPS>();
    if constexpr (R == 16) {
        return 1;
    } else {
        if (kind == 0) return wps4 ? occupancy_one<R, LOAD_HALF, true, false, 4>() : occupancy_one<R, LOAD_HALF, true, false, 3>();
        return occupancy_one<R, LOAD_FULL, true, false, 3>();
    }
}

}  // namespace

bool seg_supported(int nfft) { return nfft == 256 || nfft == 512 || nfft == 1024 || nfft == 2048 || nfft == 4096; }

// resident teams per CU (VGPR / LDS / wave-slot limited)
int seg_teams_per_cu(int nfft, int kind, bool wps4) {
    static int cache[5][3][2] = {};
    const int ri = nfft == 1024 ? 0 : (nfft == 2048 ? 1 : (nfft == 4096 ? 2 : (nfft == 512 ? 3 : 4)));
    int &c = cache[ri][kind][wps4 ? 1 : 0];
    if (c) return c;
    return c = nfft == 256 ? occupancy_r<1>(kind, wps4) : nfft == 512 ? occupancy_r<2>(kind, wps4)
                           : (nfft == 1024 ? occupancy_r<4>(kind, wps4)
                                           : (nfft == 2048 ? occupancy_r<8>(kind, wps4) : occupancy_r<16>(kind, wps4)));
}

// the role-split build: Welch, step = nfft / 2; det: 0 none, 1 time domain (1024 only), 2 frequency domain (SegArgs.fd)
int segws_teams_per_cu(int nfft) {
    static int cache[2] = {};
    int &c = cache[nfft == 1024 ? 0 : 1];
    if (c) return c;
    return c = nfft == 1024 ? occupancy_ws<4>() : occupancy_ws<8>();
}

hipError_t launch_segws(int nfft, const SegArgs &a, int det, hipStream_t s) {
    if (nfft == 1024) return det == 0 ? launch_ws_one<4, 0>(a, s) : (det == 1 ? launch_ws_one<4, 1>(a, s) : launch_ws_one<4, 2>(a, s));
    if (nfft == 2048) return det == 0 ? launch_ws_one<8, 0>(a, s) : (det == 2 ? launch_ws_one<8, 2>(a, s) : hipErrorInvalidValue);
    return hipErrorInvalidValue;
}

bool seg_padded_supported(int nfft, int nperseg) {
    return (nfft == 1024 || nfft == 2048) && (nperseg * 4 == nfft || nperseg * 2 == nfft);
}

int seg_padded_teams_per_cu(int nfft, int nperseg, int kind) {
    static int cache[2][2][2] = {};
    int &c = cache[nfft == 1024 ? 0 : 1][nperseg * 4 == nfft ? 0 : 1][kind ? 1 : 0];
    if (c) return c;
    if (nfft == 1024) return c = nperseg * 4 == nfft ? occupancy_pad<4, 4>(kind) : occupancy_pad<4, 8>(kind);
    return c = nperseg * 4 == nfft ? occupancy_pad<8, 4>(kind) : occupancy_pad<8, 8>(kind);
}

hipError_t launch_seg_padded(int nfft, int nperseg, const SegArgs &a, int kind, hipStream_t s) {
    if (!seg_padded_supported(nfft, nperseg)) return hipErrorInvalidValue;
    if (nfft == 1024) return nperseg * 4 == nfft ? launch_pad<4, 4>(a, kind, s) : launch_pad<4, 8>(a, kind, s);
    return nperseg * 4 == nfft ? launch_pad<8, 4>(a, kind, s) : launch_pad<8, 8>(a, kind, s);
}

hipError_t launch_seg(int nfft, const SegArgs &a, int kind, bool wps4, hipStream_t s) {
    switch (nfft) {
        case 256: return launch_r<1>(a, kind, wps4, s);
        case 512: return launch_r<2>(a, kind, wps4, s);
        case 1024: return launch_r<4>(a, kind, wps4, s);
        case 2048: return launch_r<8>(a, kind, wps4, s);
        case 4096: return launch_r<16>(a, kind, wps4, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace oth
